"""World-size-2 check of the sharding logic (bls_py/dist.py) on CPU with gloo.
The GPU engine is replaced by an oracle-backed stand-in with the same three
methods, so only the rank/partition/exchange logic is under test here."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class OracleShardBackend:
    """partial = canonical 576-byte product of the shard's Miller values."""
    ONE = (1).to_bytes(48, "big") + bytes(48 * 11)

    def __init__(self, oracle):
        self.o = oracle

    def miller_partial(self, g1, g2, n):
        acc = self.ONE
        for i in range(n):
            ml = self.o.miller_loop(g1[96 * i:96 * (i + 1)], g2[192 * i:192 * (i + 1)])
            acc = self.o.field_op(12, "mul", acc, ml)
        return torch.frombuffer(bytearray(acc), dtype=torch.uint8).clone()

    def all_gather(self, part, group=None):
        world = dist.get_world_size(group)
        outs = [torch.zeros(576, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(outs, part, group=group)
        return outs, world

    def final(self, gathered, world):
        acc = self.ONE
        for t in gathered:
            acc = self.o.field_op(12, "mul", acc, bytes(t.numpy()))
        return self.o.final_exp(acc)

    def miller_partials_batch(self, g1, g2, gsz, groups):
        parts = [self.miller_partial(g1[96 * gsz * g:96 * gsz * (g + 1)], g2[192 * gsz * g:192 * gsz * (g + 1)], gsz)
                 for g in range(groups)]
        return torch.cat(parts)

    def all_gather_batch(self, parts, groups, group=None):
        world = dist.get_world_size(group)
        outs = [torch.zeros(576 * groups, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(outs, parts, group=group)
        return outs, world                                            # [rank][group]

    def final_batch(self, gathered, world, groups):
        res = []
        for g in range(groups):
            acc = self.ONE
            for r in range(world):
                acc = self.o.field_op(12, "mul", acc, bytes(gathered[r][576 * g:576 * (g + 1)].numpy()))
            res.append(self.o.final_exp(acc))
        return res


    # ---- group sums: the same method names and arguments as GpuShardBackend, host tensors ----
    def _sum(self, deg, pts, scalars, k):
        if k == 0:
            return bytes(96 * deg), True
        fn = self.o.g1_msm if deg == 1 else self.o.g2_msm
        return fn(pts, None if scalars is None else list(scalars), k)

    def msm_partial(self, deg, pts, scalars, k):
        out, inf = self._sum(deg, pts, scalars, k)
        return torch.frombuffer(bytearray(out + bytes([1 if inf else 0, 0, 0, 0])), dtype=torch.uint8).clone()

    def all_gather_records(self, rec, group=None):
        world = dist.get_world_size(group)
        outs = [torch.zeros(rec.numel(), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(outs, rec, group=group)
        return torch.cat(outs), world

    def msm_finish(self, deg, gathered, world):
        raw = bytes(gathered.numpy())
        per = 96 * deg + 4
        return self._sum(deg, b"".join(raw[per * r:per * r + 96 * deg] for r in range(world)), None, world)

    def msm_groups(self, deg, pts, scalars, k, groups):
        rec = b""
        for g in range(groups):
            out, inf = self._sum(deg, pts[96 * deg * k * g:96 * deg * k * (g + 1)], None if scalars is None else scalars[k * g:k * (g + 1)], k)
            rec += out + bytes([1 if inf else 0, 0, 0, 0])
        return torch.frombuffer(bytearray(rec or b"\0"), dtype=torch.uint8).clone()[:len(rec)]

    def all_gather_ragged(self, rec, per, counts, group=None):
        mx = max(counts)
        pad = torch.zeros(mx * per, dtype=torch.uint8)
        pad[:rec.numel()] = rec
        outs = [torch.zeros(mx * per, dtype=torch.uint8) for _ in counts]
        dist.all_gather(outs, pad, group=group)
        return b"".join(bytes(outs[r].numpy())[:counts[r] * per] for r in range(len(counts)))


def _worker(rank, world, port, n, q):
    for p in (os.path.join(ROOT, "python-bls_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from bls_py.dist import pairing_multi_sharded
    g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()[:96 * n]
    g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()[:192 * n]
    out = pairing_multi_sharded(OracleShardBackend(O), g1, g2, n, rank, world)
    # three verifications at once: rotations of the batch (same product) and a shorter-by-swap variant
    from bls_py.dist import pairing_multi_batch_sharded
    rot = lambda b, sz, k: b[sz * k:] + b[:sz * k]                   # noqa: E731
    g1s = [g1, rot(g1, 96, 3), g1[96:] + g1[:96]]
    g2s = [g2, rot(g2, 192, 3), g2[:192 * (n - 1)] + g2[192 * (n - 1):]]          # last one: pairs mismatched on purpose
    batch = pairing_multi_batch_sharded(OracleShardBackend(O), g1s, g2s, rank, world)
    want = [O.pairing_multi(a, b, n) for a, b in zip(g1s, g2s)]
    assert batch == want and batch[0] == batch[1] == out and batch[2] != out
    # group sums: ONE sum split by points (C5), a batch of sums split by groups (C4) -- uneven splits,
    # a rank with no group at all, infinity among the partials
    from bls_py.dist import msm_sharded, msm_groups_sharded
    be = OracleShardBackend(O)
    sc = [3, 0x1234567, 5, (1 << 255) - 19, 7, 0, 11][:min(n, 7)]
    k = len(sc)
    assert msm_sharded(be, 1, g1[:96 * k], sc, k, rank, world) == (O.g1_msm(g1[:96 * k], sc, k)[0], False)
    assert msm_sharded(be, 2, g2[:192 * k], None, k, rank, world) == (O.g2_msm(g2[:192 * k], [1] * k, k)[0], False)
    assert msm_sharded(be, 1, g1[:96], [0], 1, rank, world) == (bytes(96), True)          # one rank idle, the other sums to infinity
    groups, kk = 3, 2
    gs = [2, 3, 5, 7, 0, 0]
    got, inf = msm_groups_sharded(be, 2, g2[:192 * kk * groups], gs, kk, groups, rank, world)
    want = [O.g2_msm(g2[192 * kk * g:192 * kk * (g + 1)], gs[kk * g:kk * (g + 1)], kk)[0] for g in range(groups)]
    assert got == b"".join(want) and inf == [False, False, True]
    assert msm_groups_sharded(be, 1, g1[:96], [9], 1, 1, rank, world) == (O.g1_msm(g1[:96], [9], 1)[0], [False])
    q.put((rank, out.hex()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from bls_py.dist import shard_bounds
    for n in (0, 1, 7, 1025, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_two_ranks_match_reference(golden, oracle):
    n, world = 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = golden("pairing.json")["seeded"]["8"]["out"]
    assert res[0] == res[1] == want


def test_bench_launches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus 2` with no launcher around it must start its own ranks (a child
    torch.distributed.run) -- the way the driver's scaling sweep calls it.  --dry-run stops before
    any GPU work: rendezvous on 127.0.0.1, one 576-byte partial per rank all-gathered over gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_in_process_group"] == 2 and line["partials_seen_from"] == [0, 1]
