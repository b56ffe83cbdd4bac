#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Rpass-analysis=kernel-resource-usage \\
          -o /tmp/lib.so archive_amd/csrc/archive_hip.hip 2> usage.txt
    python tools/resources_table.py usage.txt > profiles/r05_resources.md
"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [o if o else n for o, n in zip(out, names)]
    except Exception:
        return names


def main(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: .*?(Function Name|SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|TgSplit|Dynamic Stack): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    names = demangle([r["name"] for r in rows])
    print("| kernel | VGPRs | AGPRs | SGPRs | SGPR spills | VGPR spills | scratch B/lane | LDS B/workgroup | waves/SIMD |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*$", "", n).replace("ahip::", "")
        print("| `%s` | %s | %s | %s | %s | %s | %s | %s | %s |" % (n, r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
                                                               r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main(sys.argv[1])
