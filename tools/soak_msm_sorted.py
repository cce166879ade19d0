#!/usr/bin/env python3
"""Randomised parity soak of the sorted-bucket G1 sum (GPU box): random counts, window widths (BLSGPU_MSM_SORT_BITS 5 .. 13, i.e.
every fold / window-sum shape of the tail) and both tails (k_msm_horner_wide / the wavefront VM's), scalars drawn from {random,
short, 0, 1, r - 1, 2^256 - 1}, with points at infinity, repeated points with equal scalars (a doubling inside an addition), P and
-P with equal scalars and runs of one scalar for many points sprinkled in -- every result against the CPU oracle's
double-and-add sum (oracle/, the reference's fields_t.py:705-740).  Not part of the test-suite; prints one line per trial and a
summary.  usage: soak_msm_sorted.py [trials]  (tools/soak_msm.py covers every kernel family at the default window widths)"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "python-bls_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
N = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def engine_with(env):
    from bls_py import _native
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _native.Engine(0)
    finally:
        for k in env:
            if old[k] is None:
                del os.environ[k]
            else:
                os.environ[k] = old[k]


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    import oracle as O
    O.build()
    g1 = open(os.path.join(ROOT, "tests", "golden", "pairs_seed1_g1.bin"), "rb").read()
    npts = len(g1) // 96
    engines = {}
    rng = random.Random(20261005)
    bad = 0
    t0 = time.time()
    for t in range(trials):
        bits, tail = rng.choice([5, 6, 7, 8, 9, 10, 11, 12, 13]), rng.choice(["wide", "vm"])
        key = (bits, tail)
        if key not in engines:
            engines[key] = engine_with({"BLSGPU_MSM_SORT_THRESHOLD": "1", "BLSGPU_MSM_SORT_BITS": str(bits),
                                        "BLSGPU_MSM_WIDE_TAIL": "1" if tail == "wide" else "0"})
        k = rng.choice([1, 2, 3, rng.randrange(4, 64), rng.randrange(64, 600), rng.randrange(600, 2500)])
        idx = [rng.randrange(npts) for _ in range(k)]
        pts = bytearray(b"".join(g1[96 * i:96 * (i + 1)] for i in idx))
        sc = [rng.choice([rng.randrange(N), rng.randrange(N), rng.randrange(1 << 40), 0, N - 1, 1, (1 << 256) - 1]) for _ in range(k)]
        for _ in range(rng.randrange(0, 4)):                      # sprinkle the cases the complete formulas exist for
            j = rng.randrange(k)
            what = rng.randrange(4)
            if what == 0:
                pts[96 * j:96 * (j + 1)] = bytes(96)              # infinity in the list
            elif what == 1 and k > 1:
                i = rng.randrange(k)
                pts[96 * j:96 * (j + 1)] = pts[96 * i:96 * (i + 1)]
                sc[j] = sc[i]                                      # the same point in the same buckets
            elif what == 2 and k > 1:
                i = rng.randrange(k)
                P = bytes(pts[96 * i:96 * (i + 1)])
                pts[96 * j:96 * (j + 1)] = P[:48] + ((Q - int.from_bytes(P[48:], "big")) % Q).to_bytes(48, "big")
                sc[j] = sc[i]                                      # P and -P in the same buckets
            elif k > 8:
                lo = rng.randrange(k - 8)
                hi = rng.randrange(lo + 1, k)
                for i in range(lo, hi):
                    sc[i] = sc[lo]                                 # one scalar for a run of points: long runs of one key
        got, inf = engines[key].g1_msm(bytes(pts), sc, k, 1)
        want, winf = O.g1_msm(bytes(pts), sc, k)
        ok = got == want and inf[0] == (want == bytes(96))
        bad += not ok
        print("trial %d  %d points  %d-bit windows  tail %s  %s  (%d s)" % (t, k, bits, tail, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    print("sorted-bucket soak: %d trials, %d mismatches" % (trials, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
