"""Sharded multi-pairing: one rank per GPU, one tiny exchange.

The N pairs are split contiguously; every rank folds its Miller-loop values into
ONE Fq12 partial, the partials are all-gathered (RCCL on GPUs: 144 x int32 per
rank) and every rank finishes with the product + final exponentiation.  The
product is associative and commutative, so the bytes do not depend on the world
size (fields_t.py:1114-1121 computes the same product serially).
"""


def shard_bounds(n, rank, world):
    """Contiguous split [lo, hi) of n pairs for `rank` of `world`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return n * rank // world, n * (rank + 1) // world


class GpuShardBackend:
    """Device-resident implementation on top of bls_py._native.Engine."""

    def __init__(self, engine, device):
        import torch
        self.torch = torch
        self.eng = engine
        self.device = device

    def miller_partial(self, g1: bytes, g2: bytes, n: int):
        torch = self.torch
        t1 = torch.frombuffer(bytearray(g1 or b"\0"), dtype=torch.uint8).to(self.device)
        t2 = torch.frombuffer(bytearray(g2 or b"\0"), dtype=torch.uint8).to(self.device)
        part = torch.zeros(144, dtype=torch.int32, device=self.device)
        self.eng.miller_product_dev(t1.data_ptr(), t2.data_ptr(), n, part.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
        return part

    def all_gather(self, part, group=None):
        import torch.distributed as dist
        world = dist.get_world_size(group)
        out = self.torch.zeros(world * 144, dtype=self.torch.int32, device=self.device)
        dist.all_gather_into_tensor(out, part, group=group)
        return out, world

    def final(self, gathered, world) -> bytes:
        torch = self.torch
        out = torch.zeros(576, dtype=torch.uint8, device=self.device)
        self.eng.final_exp_product_dev(gathered.data_ptr(), world, out.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return bytes(out.cpu().numpy())


def pairing_multi_sharded(backend, g1: bytes, g2: bytes, n: int, rank: int, world: int, group=None) -> bytes:
    """Every rank passes the FULL batch and gets the full result."""
    lo, hi = shard_bounds(n, rank, world)
    part = backend.miller_partial(g1[96 * lo:96 * hi], g2[192 * lo:192 * hi], hi - lo)
    gathered, w = backend.all_gather(part, group)
    if w != world:
        raise RuntimeError("world size mismatch")
    return backend.final(gathered, world)
