"""Dev tool: static instruction counts of a kernel's ISA between AHIP_ASM_NOTE markers ("; <prefix> <name>" comments).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k.s archive_amd/csrc/archive_hip.hip
    python tools/analysis/isa_regions.py /tmp/k.s _Z25inflate_resolve_wg_kernelILb0E WGN

Loops count once, so this says what the code of a region IS (VALU / SALU / branches / waits / LDS / memory), not how often it runs."""
import collections
import re
import sys

path, kernel, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(kernel) and ':' in l)
region = 'start'
stats = collections.OrderedDict()


def cls(op):
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'):
        return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_'):
        return 'vmem'
    return 'other'


for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith('s_endpgm'):
        break
    if not t:
        continue
    m = re.match(r';\s*%s (.*)' % re.escape(prefix), t)
    if m:
        region = m.group(1)
        continue
    if t.startswith(';') or t.startswith('.') or t.endswith(':'):
        continue
    stats.setdefault(region, collections.Counter())[cls(t.split()[0])] += 1
for r, c in stats.items():
    print('%-18s' % r, ' '.join('%s %d' % kv for kv in sorted(c.items())))
