"""The committed per-kernel resource table (profiles/r06_resources.md: registers, spills, scratch, LDS, waves per SIMD from
hipcc's -Rpass-analysis=kernel-resource-usage) must be the table of the code as it is: a fresh device-only compile is held
against it for the kernels the headline rests on.  (Round 5's table had gone stale after a late kernel rewrite.)"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "profiles", "r06_resources.md")
HEADLINE = ["inflate_tokenize_kernel<false>", "inflate_resolve_kernel<false>", "inflate_resolve_wg_kernel<false>", "inflate_late_kernel<true>",
            "sm_find_kernel", "deflate_parse_kernel"]


def _rows(text):
    rows = {}
    for line in text.splitlines():
        m = re.match(r"\| `([^`]+)` \|(.*)\|\s*$", line)
        if m:
            rows[m.group(1)] = [c.strip() for c in m.group(2).split("|")]
    return rows


def test_committed_resource_table_matches_a_fresh_compile(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    usage = tmp_path / "usage.txt"
    with open(usage, "w") as f:
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage",
                        "-o", str(tmp_path / "k.o"), os.path.join(ROOT, "archive_amd", "csrc", "archive_hip.hip")], stderr=f, check=True, timeout=900)
    fresh = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "resources_table.py"), str(usage)], capture_output=True, text=True, check=True).stdout
    have, want = _rows(open(TABLE).read()), _rows(fresh)
    for k in HEADLINE:
        names = [n for n in want if n == k or n.startswith(k + "<") or n.startswith(k)]
        assert names, "kernel %s not in the fresh compile" % k
        for n in names:
            assert n in have, "%s is missing from %s: python tools/resources_table.py" % (n, TABLE)
            assert have[n] == want[n], "%s: committed %s, compiled %s -- regenerate profiles/r06_resources.md" % (n, have[n], want[n])
