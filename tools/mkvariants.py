"""Dev tool: build tuning variants of libarchive_hip.so side by side (archive_amd/lib/var_<name>.so).

    python tools/mkvariants.py name1="-DA=1 -DB=2" name2="..."

Each is the production library compiled with extra defines; tools/exp.sh runs kstats.py over them on the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "archive_amd", "csrc", "archive_hip.hip")
OUT = os.path.join(ROOT, "archive_amd", "lib")


def one(arg):
    name, _, flags = arg.partition("=")
    lib = os.path.join(OUT, "var_%s.so" % name)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", lib] + flags.split() + [CSRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, r.stderr[-2000:]


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(8) as ex:
        for name, rc, err in ex.map(one, sys.argv[1:]):
            print(name, "ok" if rc == 0 else "FAILED\n" + err)
