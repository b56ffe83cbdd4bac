"""Builds libarchive_hip.so (gfx950) in-tree with hipcc.  `python -m archive_amd.build`."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "lib", "libarchive_hip.so")
SOURCES = ["archive_hip.hip"]


def _newest_source():
    t = 0.0
    for root, _, files in os.walk(CSRC):
        for f in files:
            t = max(t, os.path.getmtime(os.path.join(root, f)))
    t = max(t, os.path.getmtime(os.path.join(_HERE, "..", "include", "archive_hip.h")))
    return t


PROF_LIB = os.path.join(_HERE, "lib", "libarchive_hip_prof.so")


def build(force=False, verbose=False, profile=False, defines=()):
    """profile=True builds the instrumented variant (per-phase cycle counters) next to the
    production library; it is only ever loaded by tools/kstats.py via AHIP_LIB."""
    lib = PROF_LIB if profile else LIB
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= _newest_source():
        return lib
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", lib]
    if profile:
        cmd.append("-DAHIP_PROFILE=1")
    cmd += ["-D" + d for d in defines]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, profile="--profile" in sys.argv))
