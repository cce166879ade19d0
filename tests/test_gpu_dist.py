"""The product's own multi-GPU class (bls_py.dist.GpuShardBackend) on the real collective library: a ONE-rank RCCL
process group on the GPU box (RCCL refuses two ranks on one device; more ranks are the driver's 8-GPU node).  Every call
below goes Engine -> all_gather_into_tensor (RCCL) -> Engine, exactly the N > 1 path with world = 1; results against the
reference's golden vectors."""
import os
import socket

import pytest

from conftest import cat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group():
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        yield dist
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_sharded_pairing_through_rccl(engine, golden, seeded_pairs, rccl_group):
    import torch
    from bls_py.dist import GpuShardBackend, pairing_multi_sharded, pairing_multi_batch_sharded
    be = GpuShardBackend(engine, torch.device("cuda", 0))
    g1, g2 = seeded_pairs
    gj = golden("pairing.json")
    assert pairing_multi_sharded(be, g1, g2, 1025, 0, 1).hex() == gj["seeded"]["1025"]["out"]
    s4 = gj["small4"]
    a, b = cat(s4["g1"]), cat(s4["g2"])
    assert pairing_multi_sharded(be, a, b, 4, 0, 1).hex() == s4["out"]
    # the batch form: three verifications of four pairs -- the golden one, a rotation of it, and pairs mismatched on purpose
    rot1, rot2 = a[96:] + a[:96], b[192:] + b[:192]
    got = pairing_multi_batch_sharded(be, [a, rot1, a], [b, rot2, rot2], 0, 1)
    assert got[0].hex() == got[1].hex() == s4["out"] and got[2] == engine.pairing_multi(a, rot2, 4) != got[0]


def test_version_check_refuses_a_different_build(engine, rccl_group):
    import torch
    from bls_py.dist import GpuShardBackend

    class Other:
        def __init__(self, e):
            self.e = e

        def version(self):
            return "blsgpu/0 some other build"

        def __getattr__(self, k):
            return getattr(self.e, k)
    be = GpuShardBackend(engine, torch.device("cuda", 0))
    be.check_versions()                                       # one rank: trivially equal
    # two "ranks" with different builds: emulate the gathered list
    import torch.distributed as dist
    real = dist.all_gather_object

    def fake(out, obj, group=None):
        out[0] = "blsgpu/1 gfx950 vm-tables deadbeef"
    be2 = GpuShardBackend(Other(engine), torch.device("cuda", 0))
    dist.all_gather_object = fake
    try:
        with pytest.raises(RuntimeError):
            be2.check_versions()
    finally:
        dist.all_gather_object = real


def test_sharded_sums_through_rccl(engine, golden, seeded_pairs, oracle, rccl_group):
    import torch
    from bls_py.dist import GpuShardBackend, msm_sharded, msm_groups_sharded
    be = GpuShardBackend(engine, torch.device("cuda", 0))
    g1, g2 = seeded_pairs
    sc = [3, 0x1234567, 5, (1 << 255) - 19, 7, 0, 11]
    assert msm_sharded(be, 1, g1[:96 * 7], sc, 7, 0, 1) == oracle.g1_msm(g1[:96 * 7], sc, 7)
    assert msm_sharded(be, 2, g2[:192 * 7], None, 7, 0, 1) == oracle.g2_msm(g2[:192 * 7], None, 7)
    assert msm_sharded(be, 1, g1[:96], [0], 1, 0, 1) == (bytes(96), True)
    th = golden("threshold.json")["67_of_100"]
    pts = cat(th["unit_sigs_affine"])
    lam = [int(x, 16) for x in th["lambdas"]]
    out, inf = msm_groups_sharded(be, 2, pts * 3, lam * 3, 67, 3, 0, 1)
    assert out == bytes.fromhex(th["combined_affine"]) * 3 and inf == [False] * 3
