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

    # ---- B independent verifications at once (what bench.py runs) ----
    def miller_partials_batch(self, g1: bytes, g2: bytes, gsz: int, groups: int):
        torch = self.torch
        t1 = torch.frombuffer(bytearray(g1 or b"\0"), dtype=torch.uint8).to(self.device)
        t2 = torch.frombuffer(bytearray(g2 or b"\0"), dtype=torch.uint8).to(self.device)
        parts = torch.zeros(groups * 144, dtype=torch.int32, device=self.device)
        self.eng.miller_product_batch_dev(t1.data_ptr(), t2.data_ptr(), gsz, groups, parts.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream)
        return parts

    def all_gather_batch(self, parts, groups, group=None):
        import torch.distributed as dist
        world = dist.get_world_size(group)
        out = self.torch.zeros(world * groups * 144, dtype=self.torch.int32, device=self.device)
        dist.all_gather_into_tensor(out, parts, group=group)        # layout [rank][group]
        return out, world

    def final_batch(self, gathered, world, groups):
        torch = self.torch
        out = torch.zeros(groups * 576, dtype=torch.uint8, device=self.device)
        self.eng.final_exp_product_batch_dev(gathered.data_ptr(), world, groups, out.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        raw = bytes(out.cpu().numpy())
        return [raw[576 * g:576 * (g + 1)] for g in range(groups)]

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


def pairing_multi_batch_sharded(backend, g1_list, g2_list, rank: int, world: int, group=None):
    """`len(g1_list)` independent multi-pairings of the same size; every rank passes the FULL
    inputs and gets all results.  Rank r computes the Miller product of ITS slice of every
    verification, one all-gather moves groups x 576 bytes per rank, every rank finishes."""
    groups = len(g1_list)
    if groups == 0:
        return []
    n = len(g1_list[0]) // 96
    if any(len(a) != 96 * n or len(b) != 192 * n for a, b in zip(g1_list, g2_list)):
        raise ValueError("verifications of a batch must have the same number of pairs")
    lo, hi = shard_bounds(n, rank, world)
    a = b"".join(x[96 * lo:96 * hi] for x in g1_list)
    b = b"".join(x[192 * lo:192 * hi] for x in g2_list)
    parts = backend.miller_partials_batch(a, b, hi - lo, groups)
    gathered, w = backend.all_gather_batch(parts, groups, group)
    if w != world:
        raise RuntimeError("world size mismatch")
    return backend.final_batch(gathered, world, groups)
