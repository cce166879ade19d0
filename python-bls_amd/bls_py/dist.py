"""Sharded multi-pairing: one rank per GPU, one tiny exchange.

The N pairs are split contiguously; every rank folds its Miller-loop values into
ONE Fq12 partial, the partials are all-gathered (RCCL on GPUs: 144 x int32 per
rank) and every rank finishes with the product + final exponentiation.  The
product is associative and commutative, so the bytes do not depend on the world
size (fields_t.py:1114-1121 computes the same product serially).

Group sums: ONE large sum (BLS.aggregate_pub_keys, bls.py:203-223 -- C5) splits its
points contiguously, every rank sums its slice and the `world` affine partials (96 / 192 bytes +
an infinity flag each) are all-gathered and added on every rank; a BATCH of sums
(Threshold.aggregate_unit_sigs per group, threshold.py:127-136 -- C4) splits its groups, nothing
is exchanged but the results.

The Fq12 partials cross the wire as the engine's Montgomery limbs (576 bytes; canonical form would
cost a conversion on both sides): every rank must therefore run the same build of libblsgpu --
`GpuShardBackend.check_versions` compares `blsgpu_version()` (it names the table hash) over the
group before the first exchange and refuses a mismatch.
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
        self._versions_ok = set()

    def check_versions(self, group=None):
        """Montgomery limbs are exchanged raw: all ranks must run the same libblsgpu build."""
        import torch.distributed as dist
        key = id(group)
        if key in self._versions_ok:
            return
        mine = self.eng.version()
        seen = [None] * dist.get_world_size(group)
        dist.all_gather_object(seen, mine, group=group)
        if any(v != mine for v in seen):
            raise RuntimeError("libblsgpu builds differ across ranks: %r" % (sorted(set(seen)),))
        self._versions_ok.add(key)

    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    # ---- device-resident steps (what bench.py times: inputs already in HBM, results left there) ----
    def miller_partials_batch_dev(self, t1, t2, gsz, groups, parts, stream=None):
        self.eng.miller_product_batch_dev(t1.data_ptr(), t2.data_ptr(), gsz, groups, parts.data_ptr(),
                                          self.stream() if stream is None else stream)

    def all_gather_into(self, dst, src, group=None):
        """one RCCL all-gather on the current stream: dst = [rank][...] of every rank's src"""
        import torch.distributed as dist
        self.check_versions(group)
        dist.all_gather_into_tensor(dst, src, group=group)

    def final_batch_dev(self, gathered, world, groups, out, stream=None):
        self.eng.final_exp_product_batch_dev(gathered.data_ptr(), world, groups, out.data_ptr(),
                                             self.stream() if stream is None else stream)

    # ---- group sums ----
    def _msm_dev(self, deg, t_pts, t_sc, k, groups, t_out, t_inf):
        fn = self.eng.lib.blsgpu_g1_msm_dev if deg == 1 else self.eng.lib.blsgpu_g2_msm_dev
        self.eng._check(fn(self.eng.h, t_pts.data_ptr(), t_sc.data_ptr() if t_sc is not None else None, k, groups,
                           t_out.data_ptr(), t_inf.data_ptr(), self.stream()), "msm_dev")

    def msm_partial_dev(self, deg, t_pts, t_sc, k, rec):
        """rec (uint8[96 deg + 4], device) <- affine sum of the k points | infinity flag | 3 pad bytes"""
        self._msm_dev(deg, t_pts, t_sc, k, 1, rec[:96 * deg], rec[96 * deg:96 * deg + 1])

    def msm_finish_dev(self, deg, gathered, world, out, inf):
        """sum of the `world` gathered partial records: strip the flags ((0,0) encodes infinity), plain sum"""
        pts = gathered.view(world, 96 * deg + 4)[:, :96 * deg].contiguous()
        self._msm_dev(deg, pts, None, world, 1, out, inf)

    def up(self, b):
        return self.torch.frombuffer(bytearray(b or b"\0"), dtype=self.torch.uint8).to(self.device)

    def msm_partial(self, deg, pts: bytes, scalars, k):
        torch = self.torch
        rec = torch.zeros(96 * deg + 4, dtype=torch.uint8, device=self.device)
        if k:
            sc = None if scalars is None else self.up(b"".join(int(x).to_bytes(32, "big") for x in scalars))
            self.msm_partial_dev(deg, self.up(pts), sc, k, rec)
        else:
            rec[96 * deg] = 1
        return rec

    def all_gather_records(self, rec, group=None):
        import torch.distributed as dist
        world = dist.get_world_size(group)
        out = self.torch.zeros(world * rec.numel(), dtype=self.torch.uint8, device=self.device)
        self.all_gather_into(out, rec, group)
        return out, world

    def msm_finish(self, deg, gathered, world):
        torch = self.torch
        out = torch.zeros(96 * deg, dtype=torch.uint8, device=self.device)
        inf = torch.zeros(4, dtype=torch.uint8, device=self.device)
        self.msm_finish_dev(deg, gathered, world, out, inf)
        torch.cuda.synchronize()
        return bytes(out.cpu().numpy()), bool(inf[0].item())

    def msm_groups(self, deg, pts: bytes, scalars, k, groups):
        """`groups` independent sums of k points each -> uint8[groups x (96 deg + 4)] records on the device"""
        torch = self.torch
        out = torch.zeros(max(1, groups) * 96 * deg, dtype=torch.uint8, device=self.device)
        inf = torch.zeros(max(1, groups), dtype=torch.uint8, device=self.device)
        if groups:
            sc = None if scalars is None else self.up(b"".join(int(x).to_bytes(32, "big") for x in scalars))
            self._msm_dev(deg, self.up(pts), sc, k, groups, out, inf)
        rec = torch.zeros(groups, 96 * deg + 4, dtype=torch.uint8, device=self.device)
        if groups:
            rec[:, :96 * deg] = out.view(groups, 96 * deg)
            rec[:, 96 * deg] = inf[:groups]
        return rec.reshape(-1)

    def all_gather_ragged(self, rec, per, counts, group=None):
        """all-gather of records whose number differs by rank (counts[r] records of `per` bytes): padded to the longest"""
        torch = self.torch
        mx = max(counts)
        pad = torch.zeros(mx * per, dtype=torch.uint8, device=self.device)
        pad[:rec.numel()] = rec
        out = torch.zeros(len(counts) * mx * per, dtype=torch.uint8, device=self.device)
        self.all_gather_into(out, pad, group)
        torch.cuda.synchronize()
        raw = bytes(out.cpu().numpy())
        return b"".join(raw[r * mx * per:r * mx * per + counts[r] * per] for r in range(len(counts)))

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
        self.all_gather_into(out, part, group)
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
        self.all_gather_into(out, parts, group)                     # layout [rank][group]
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


def msm_sharded(backend, deg: int, pts: bytes, scalars, n: int, rank: int, world: int, group=None):
    """ONE sum of n points (scalars: list of ints, or None for a plain sum) split over the ranks
    (bls.py:203-223 at scale): every rank passes the FULL inputs and gets (affine bytes, is_infinity).
    Rank r sums points [lo, hi); one all-gather of a 96 deg + 4 byte record per rank; every rank adds
    the `world` partials."""
    lo, hi = shard_bounds(n, rank, world)
    sz = 96 * deg
    rec = backend.msm_partial(deg, pts[sz * lo:sz * hi], None if scalars is None else scalars[lo:hi], hi - lo)
    gathered, w = backend.all_gather_records(rec, group)
    if w != world:
        raise RuntimeError("world size mismatch")
    return backend.msm_finish(deg, gathered, world)


def msm_groups_sharded(backend, deg: int, pts: bytes, scalars, k: int, groups: int, rank: int, world: int, group=None):
    """`groups` independent sums of k points each (threshold.py:127-136 per group): rank r computes groups
    [lo, hi), the results are all-gathered.  Returns (affine bytes of all groups, [is_infinity])."""
    lo, hi = shard_bounds(groups, rank, world)
    sz = 96 * deg
    rec = backend.msm_groups(deg, pts[sz * k * lo:sz * k * hi], None if scalars is None else scalars[k * lo:k * hi], k, hi - lo)
    counts = [shard_bounds(groups, r, world)[1] - shard_bounds(groups, r, world)[0] for r in range(world)]
    raw = backend.all_gather_ragged(rec, sz + 4, counts, group)
    out = b"".join(raw[(sz + 4) * g:(sz + 4) * g + sz] for g in range(groups))
    return out, [raw[(sz + 4) * g + sz] != 0 for g in range(groups)]
