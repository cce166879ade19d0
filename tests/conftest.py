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


# ---- the two-rank rehearsal of the multi-GPU path (tests/test_gpu_two_ranks.py) ------------------------------------------
# bench.py --gpus 2 --backend gloo starts two ranks that BOTH use GPU 0 (RCCL refuses two ranks on one device; gloo carries
# the all-gather through host memory): the real GpuShardBackend and the real engine in a process group of two.  The child
# is started HERE, before this process has made any GPU call (a process that has initialised the GPU must not start
# programs on the GPU boxes of this pool other than as plain children -- and the safest child is one started before that),
# runs beside the first tests and is collected by the test.
REHEARSAL = {}


def pytest_sessionstart(session):
    import subprocess
    import tempfile
    expr = session.config.getoption("markexpr", "") or ""
    if "gpu" not in expr or "not gpu" in expr or os.environ.get("BLSGPU_NO_REHEARSAL"):
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:           # (counting devices does not initialise the GPU)
            return
    except Exception:
        return
    out = tempfile.mkdtemp(prefix="blsgpu_two_ranks_")
    log = open(os.path.join(out, "log.txt"), "w")
    REHEARSAL["dir"] = out
    REHEARSAL["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_rehearsal.py"), out],
                                         stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)


def pytest_sessionfinish(session, exitstatus):
    p = REHEARSAL.get("proc")
    if p is not None and p.poll() is None:
        p.terminate()                                # (the exact child started above)


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


def engine_with_env(env, device=0):
    """a fresh engine (context) created while the given environment knobs are set: csrc/blsgpu_api.hip reads them at
    context creation only"""
    from bls_py import _native
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _native.Engine(device)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def cat(hexes):
    return b"".join(bytes.fromhex(x) for x in hexes)
