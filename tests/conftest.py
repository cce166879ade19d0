import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "python-bls_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def seeded_pairs():
    with open(os.path.join(GOLDEN, "pairs_seed1_g1.bin"), "rb") as f:
        g1 = f.read()
    with open(os.path.join(GOLDEN, "pairs_seed1_g2.bin"), "rb") as f:
        g2 = f.read()
    assert len(g1) == 96 * 1025 and len(g2) == 192 * 1025
    return g1, g2


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def engine():
    """The HIP engine on device 0.  No fallback: fails if the GPU or the
    library is missing."""
    from bls_py import _native
    return _native.engine(0)


def cat(hexes):
    return b"".join(bytes.fromhex(x) for x in hexes)
