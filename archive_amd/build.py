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


def _compiler_id(hipcc):
    """first lines of `hipcc --version` (HIP version + clang version): what the library was built with"""
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout
        return " | ".join(l.strip() for l in out.splitlines()[:2])
    except Exception:
        return "unknown"


def compiler_changed(lib=None):
    """True when the production library exists and was built by another compiler than the hipcc at hand"""
    lib = lib or LIB
    info = lib + ".buildinfo"
    if not (os.path.exists(lib) and os.path.exists(info)):
        return False
    # (no compiler at hand, or a library from before the compiler was recorded: the prebuilt library is what there is)
    have, cur = open(info).read().strip(), _compiler_id(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
    if cur == "unknown" or have.startswith("unknown"):
        return False
    return have != cur


def build(force=False, verbose=False, profile=False, defines=()):
    """profile=True builds the instrumented variant (per-phase cycle counters) next to the
    production library; it is only ever loaded by tools/kstats.py via AHIP_LIB.

    The library is rebuilt when a source is newer than it.  What compiler built it is recorded next to it
    (<lib>.buildinfo, compiler_changed()): __graft_entry__.build() forces a rebuild when that is not the compiler at hand;
    the GPU tests, smoke() and bench.py load the prebuilt library as it travelled (what was validated is what runs)."""
    lib = PROF_LIB if profile else LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    info = lib + ".buildinfo"
    cid = _compiler_id(hipcc)
    if not force and not defines and os.path.exists(lib) and os.path.getmtime(lib) >= _newest_source():
        if not os.path.exists(info):
            with open(info, "w") as f:
                f.write("unknown (built before the compiler was recorded)\n")
        return lib
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", lib]
    if profile:
        cmd.append("-DAHIP_PROFILE=1")
    cmd += ["-D" + d for d in defines]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    with open(info, "w") as f:
        f.write(cid + "\n")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, profile="--profile" in sys.argv))
