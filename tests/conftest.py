import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    gdir = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gdir, "manifest.json")) as f:
        man = json.load(f)
    vecs = []
    for v in man["vectors"]:
        with open(os.path.join(gdir, v["name"] + ".in"), "rb") as f:
            inp = f.read()
        with open(os.path.join(gdir, v["name"] + ".out"), "rb") as f:
            exp = f.read()
        vecs.append(dict(v, input=inp, expected=exp))
    return dict(vectors=vecs, checksum_kat=man["checksum_kat"])


@pytest.fixture(scope="session")
def native_built():
    """Builds libarchive_hip.so if hipcc is around (it cross-compiles without a GPU)."""
    from archive_amd import build
    return build.build()
