#!/usr/bin/env python3
"""Randomised parity soak (GPU box): random subsets of the seeded pairs, random sizes and
group shapes, through both Miller kernels and the batch entry point, every result against the
CPU oracle.  Sprinkled in: zero coordinates, points off the twist (random, (x, 0), (0, y)), the
low-order points of tests/golden/pairing_degenerate.json (order 13 on the twist, orders 3 and
11 of E(Fq) as Fq2 coordinates), their negatives, and infinity flags on valid points.  Not part of the test-suite (minutes of
oracle time); prints one line per trial and a summary."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "python-bls_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    import oracle as O
    from bls_py import _native
    O.build()
    eng = _native.Engine(0)
    os.environ["BLSGPU_MILLER_WIDE3_MAX"] = "0"        # (read at context creation) the wide Miller loop on two wavefronts at every size
    eng2 = _native.Engine(0)
    del os.environ["BLSGPU_MILLER_WIDE3_MAX"]
    os.environ["BLSGPU_MILLER_EXACT_LANES"] = "0"      # blsgpu_miller_loop_batch on the wavefront VM's reference-faithful program
    eng_vm_exact = _native.Engine(0)
    del os.environ["BLSGPU_MILLER_EXACT_LANES"]
    os.environ["BLSGPU_MILLER_EXACT_FAST"] = "0"       # ... on k_ml_lines_exact for every pair (the default: fast lines + one factor per pair)
    eng_ref_lines = _native.Engine(0)
    del os.environ["BLSGPU_MILLER_EXACT_FAST"]
    gold = os.path.join(ROOT, "tests", "golden")
    g1 = open(os.path.join(gold, "pairs_seed1_g1.bin"), "rb").read()
    g2 = open(os.path.join(gold, "pairs_seed1_g2.bin"), "rb").read()
    rng = random.Random(20261003)
    bad = 0
    t0 = time.time()
    import json
    Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    dc = json.load(open(os.path.join(gold, "pairing_degenerate.json")))["cases"]
    low = [bytes.fromhex(dc[k]["g2"][0]) for k in ("ord13", "ord13_neg", "ord3_embedded", "ord11_embedded")]

    def fq():
        return rng.randrange(Q).to_bytes(48, "big")

    def pick(n, degenerate):
        a, b, f = [], [], bytearray()
        for _ in range(n):
            i = rng.randrange(1025)
            p, q, fl = g1[96 * i:96 * (i + 1)], g2[192 * i:192 * (i + 1)], (0, 0)
            if degenerate and rng.random() < 0.08:
                kind = rng.randrange(11)
                if kind == 0:
                    p = bytes(96)                       # P = (0, 0)
                elif kind == 1:
                    q = bytes(192)                      # Q = (0, 0)
                elif kind == 2:
                    p, q = bytes(96), bytes(192)
                elif kind == 3:
                    q = fq() + fq() + fq() + fq()       # off the twist
                elif kind == 4:
                    q = fq() + fq() + bytes(96)         # (x, 0)
                elif kind == 5:
                    q = bytes(96) + fq() + fq()         # (0, y)
                elif kind == 6:
                    q = rng.choice(low)                 # low order: a step of the loop degenerates
                elif kind == 7:
                    fl = (rng.randrange(2), 1)          # flagged valid Q
                elif kind == 8:
                    q, fl = rng.choice(low), (0, 1)
                elif kind == 10:
                    p = fq() + bytes(48)                # P.y = 0
                else:
                    p = bytes(48) + fq()                # P.x = 0
            a.append(p)
            b.append(q)
            f += bytes(fl)
        return b"".join(a), b"".join(b), bytes(f)

    for t in range(trials):
        mode = 3 if t % 7 == 6 else t % 3
        if mode == 3:                                   # blsgpu_miller_loop_batch: the reference's Miller value of every pair (lane kernels; the VM's program)
            n = rng.choice([1, 2, 31, 32, 33, 70, 129])
            a, b, f = pick(n, True)
            got = rng.choice([eng, eng, eng_ref_lines, eng_vm_exact]).miller_loop_batch(a, b, n, f)
            ok = all(got[576 * i:576 * (i + 1)] == O.miller_loop(a[96 * i:96 * (i + 1)], b[192 * i:192 * (i + 1)], bool(f[2 * i + 1])) for i in range(n))
        elif mode == 0:                                   # one multi-pairing, one pair per wavefront / workgroup: k_miller (VM), k_miller_wide<3>, <2>
            n = rng.choice([1, 2, 3, 5, 63, 64, 65, 127, 200, 257, 600])
            a, b, f = pick(n, t % 2 == 1)
            e = rng.choice([eng, eng2])
            e.set_mp_threshold(1 << 40)
            e.set_miller_wide_max(rng.choice([0, 1 << 30, 1 << 30]))
            got = e.pairing_multi(a, b, n, f)
            e.set_miller_wide_max(1536)
            want = O.pairing_multi(a, b, n, threads=16, inf=f)
            ok = got == want
        elif mode == 1:                                 # one multi-pairing, three-pair kernel
            n = rng.choice([3, 4, 7, 100, 191, 192, 193, 500, 1025])
            a, b, f = pick(n, t % 2 == 0)
            eng.set_mp_threshold(0)
            eng.set_mp3_threshold(rng.choice([0, 1 << 40]))              # three / two pairs per wavefront
            got = eng.pairing_multi(a, b, n, f)
            want = O.pairing_multi(a, b, n, threads=16, inf=f)
            ok = got == want
        else:                                           # batch of equal-sized groups
            gsz, groups = rng.choice([(1, 40), (2, 33), (5, 17), (23, 9), (24, 9), (25, 8), (67, 5), (130, 3)])
            a, b, f = pick(gsz * groups, t % 2 == 0)
            e = rng.choice([eng, eng2])
            e.set_mp_threshold(rng.choice([0, 4096, 1 << 40]))
            e.set_mp3_threshold(rng.choice([0, 2 ** 64 - 1, 1 << 40]))
            e.set_miller_wide_max(rng.choice([0, 1536]))
            got = e.pairing_multi_batch(a, b, gsz, groups, f)
            e.set_miller_wide_max(1536)
            want = b"".join(O.pairing_multi(a[96 * gsz * g:96 * gsz * (g + 1)], b[192 * gsz * g:192 * gsz * (g + 1)], gsz, threads=16,
                                            inf=f[2 * gsz * g:2 * gsz * (g + 1)])
                            for g in range(groups))
            ok = got == want
            n = gsz * groups
        bad += 0 if ok else 1
        print("trial %d mode %d pairs %d %s  (%.0f s)" % (t, mode, n, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    print("soak: %d trials, %d mismatches" % (trials, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
