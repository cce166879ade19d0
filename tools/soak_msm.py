#!/usr/bin/env python3
"""Randomised parity soak of the multi-scalar sums (GPU box): random point lists (repeats, negatives, the point at
infinity) and scalar mixes (uniform, short, zero, n - 1, 2^256 - 1, few distinct values) through every kernel family --
sorted buckets (k_srt_*), one window per lane (k_msm_lane, signed nibbles), LDS buckets (k_msm_pip), double-and-add
(k_msm), batch Horner (k_msm_horner_np) -- every result against the CPU oracle.  Not part of the test-suite; prints one
line per trial and a summary."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "python-bls_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
N = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def engine_with(env):
    from bls_py import _native
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _native.Engine(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    import oracle as O
    O.build()
    gold = os.path.join(ROOT, "tests", "golden")
    g1 = open(os.path.join(gold, "pairs_seed1_g1.bin"), "rb").read()
    g2 = open(os.path.join(gold, "pairs_seed1_g2.bin"), "rb").read()
    engines = {
        "default": engine_with({}),
        "sorted": engine_with({"BLSGPU_MSM_SORT_THRESHOLD": "1"}),
        "sorted8": engine_with({"BLSGPU_MSM_SORT_THRESHOLD": "1", "BLSGPU_MSM_SORT_BITS": "8"}),
        "sorted_vm_tail": engine_with({"BLSGPU_MSM_SORT_THRESHOLD": "1", "BLSGPU_MSM_WIDE_TAIL": "0"}),    # (round 5: the default tail is k_msm_horner_wide)
        "lane": engine_with({"BLSGPU_PIP_THRESHOLD": "1", "BLSGPU_PIP_GROUP_THRESHOLD": "1", "BLSGPU_MSM_LANE_THRESHOLD": "1",
                             "BLSGPU_MSM_SORT_THRESHOLD": str(1 << 40), "BLSGPU_MSM_SORT2_THRESHOLD": str(1 << 40),
                             "BLSGPU_MSM_PLAIN_THRESHOLD": str(1 << 40), "BLSGPU_SMUL_MIN_GROUPS": str(1 << 40)}),
        "sorted2_5": engine_with({"BLSGPU_MSM_SORT2_BITS": "5"}),                 # (round 5: ONE G2 sum with scalars on the sorted buckets, the default)
        "sorted2_13": engine_with({"BLSGPU_MSM_SORT2_BITS": "13"}),
        "smul": engine_with({"BLSGPU_SMUL_MIN_GROUPS": "1"}),                     # (round 5: batches of small sums with scalars, one group per lane / lane pair)
        "lds": engine_with({"BLSGPU_PIP_THRESHOLD": "1", "BLSGPU_PIP_GROUP_THRESHOLD": "1", "BLSGPU_MSM_SORT_THRESHOLD": str(1 << 40),
                            "BLSGPU_MSM_SORT2_THRESHOLD": str(1 << 40), "BLSGPU_HORNER_NP_THRESHOLD": "1"}),
    }
    rng = random.Random(20261004)
    bad, t0 = 0, time.time()

    def neg(p, deg):
        h = 48 * deg
        return p[:h] + b"".join(((Q - int.from_bytes(p[h + 48 * j:h + 48 * j + 48], "big")) % Q).to_bytes(48, "big") for j in range(deg))

    def points(n, deg):
        src, sz = (g1, 96) if deg == 1 else (g2, 192)
        out = []
        for _ in range(n):
            r = rng.random()
            p = src[sz * (i := rng.randrange(1025)):sz * (i + 1)]
            if r < 0.03:
                p = bytes(sz)
            elif r < 0.10 and out:
                p = rng.choice(out)
            elif r < 0.15 and out:
                p = neg(rng.choice(out), deg)
            out.append(p)
        return b"".join(out)

    def scalars(n):
        mode = rng.randrange(5)
        few = [rng.randrange(N) for _ in range(3)]
        sc = []
        for _ in range(n):
            if mode == 0:
                sc.append(rng.randrange(N))
            elif mode == 1:
                sc.append(rng.choice([rng.randrange(N), rng.randrange(1 << 40), 0, N - 1, 1, (1 << 256) - 1, 1 << 255]))
            elif mode == 2:
                sc.append(rng.choice(few))
            elif mode == 3:
                sc.append(rng.randrange(1 << rng.choice([8, 16, 64, 128])))
            else:
                sc.append(rng.randrange(1 << 256))
        return sc

    for t in range(trials):
        if t % 7 == 6:                                  # a batch of small sums with scalars: k_smul against the wavefront VM's double-and-add and the oracle
            deg = rng.choice([1, 2])
            k, groups = rng.choice([(1, 1), (1, 33), (1, 200), (2, 65), (3, 20), (8, 9)])
            n = k * groups
            pts, sc = points(n, deg), scalars(n)
            sz = 96 * deg
            om = O.g1_msm if deg == 1 else O.g2_msm
            want = [om(pts[sz * k * g:sz * k * (g + 1)], sc[k * g:k * (g + 1)], k)[0] for g in range(groups)]
            ok = True
            for nm in ("smul", "lane"):
                out, inf = (engines[nm].g1_msm if deg == 1 else engines[nm].g2_msm)(pts, sc, k, groups)
                ok = ok and all(out[sz * g:sz * (g + 1)] == want[g] and inf[g] == (want[g] == bytes(sz)) for g in range(groups))
            what = "%d G%d sums of %d with scalars" % (groups, deg, k)
        elif t % 5 == 4:                                # one PLAIN sum (no scalars): the register kernels (default) and the wavefront VM's k_msm
            deg = rng.choice([1, 2])
            k = rng.choice([2, 3, 5, 9, 64, 65, 257, 700, 1500, 4000])
            pts = points(k, deg)
            want, winf = (O.g1_msm if deg == 1 else O.g2_msm)(pts, None, k)
            res = [(e.g1_msm if deg == 1 else e.g2_msm)(pts, None, k, 1) for e in (engines["default"], engines["lane"])]
            ok = all(r[0] == want and r[1][0] == (want == bytes(96 * deg)) for r in res)
            what = "plain G%d sum of %d" % (deg, k)
        elif t % 4 == 3:                                # one G2 sum with scalars: sorted buckets (default widths, 5 and 13 bits) and the fixed windows
            k = rng.choice([1, 2, 3, 64, 65, 257, 700, 1500])
            pts, sc = points(k, 2), scalars(k)
            want, winf = O.g2_msm(pts, sc, k)
            res = {nm: engines[nm].g2_msm(pts, sc, k, 1) for nm in ("default", "sorted2_5", "sorted2_13", "lane", "lds")}
            ok = all(r[0] == want and r[1][0] == (want == bytes(192)) for r in res.values())
            what = "G2 sum of %d" % k
        elif t % 3 < 2:                                 # one G1 sum: every kernel family
            k = rng.choice([1, 2, 3, 64, 65, 257, 700, 1500, 3000, 5000])
            pts, sc = points(k, 1), scalars(k)
            want, winf = O.g1_msm(pts, sc, k)
            res = {nm: e.g1_msm(pts, sc, k, 1) for nm, e in engines.items()}
            ok = all(r[0] == want and r[1][0] == (want == bytes(96)) for r in res.values())
            what = "G1 sum of %d" % k
        else:                                           # batch of G2 sums (lane kernel / LDS buckets + batch Horner / default)
            k, groups = rng.choice([(3, 7), (5, 11), (67, 6), (20, 16), (1, 9)])
            n = k * groups
            pts, sc = points(n, 2), scalars(n)
            want = [O.g2_msm(pts[192 * k * g:192 * k * (g + 1)], sc[k * g:k * (g + 1)], k)[0] for g in range(groups)]
            ok = True
            for nm in ("default", "lane", "lds"):
                out, inf = engines[nm].g2_msm(pts, sc, k, groups)
                ok = ok and all(out[192 * g:192 * (g + 1)] == want[g] and inf[g] == (want[g] == bytes(192)) for g in range(groups))
            what = "%d G2 sums of %d" % (groups, k)
        bad += 0 if ok else 1
        print("trial %d %s %s  (%.0f s)" % (t, what, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    print("soak: %d trials, %d mismatches" % (trials, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
