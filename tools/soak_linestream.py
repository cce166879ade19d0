#!/usr/bin/env python3
"""Randomised parity soak of the round-3 register kernels (GPU box): the line-stream multi-pairing forced for every size
(point chains: k_ml_lines4 on lane quads -- what calls of these sizes take by default -- or, on the QUAD_MAX = 0 axis,
k_ml_lines2 on lane pairs, the kernel of bench.py's step; then k_ml_lines_exact -> k_ml_accum -> k_ml_merge / k_ml_merge_wide -> k_ml_horner_fexp), the small-group form
(k_ml_small) and the final exponentiation six lanes per result (k_fexp_team) or one per wavefront (k_fexp_wide, fused
with the Horner kernel where the call ends in it), on random subsets of the seeded pairs with the degenerate inputs of
tools/soak_parity.py sprinkled in (zero coordinates, off-twist points, low-order points, flags), random chunking, every
result against the CPU oracle.  usage: python tools/soak_linestream.py [trials] [pairs|quads|wide]  (default: the three forms of the point chains at random)"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "python-bls_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    only_chains = {"pairs": 0, "quads": None, "wide": "wide"}.get(sys.argv[2], "bad") if len(sys.argv) > 2 else None
    if only_chains == "bad":
        sys.exit("second argument: pairs | quads | wide")
    if len(sys.argv) > 2 and sys.argv[2] == "quads":
        only_chains = "default"
    import oracle as O
    from bls_py import _native
    O.build()
    gold = os.path.join(ROOT, "tests", "golden")
    g1 = open(os.path.join(gold, "pairs_seed1_g1.bin"), "rb").read()
    g2 = open(os.path.join(gold, "pairs_seed1_g2.bin"), "rb").read()
    rng = random.Random(20261004)
    Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    dc = json.load(open(os.path.join(gold, "pairing_degenerate.json")))["cases"]
    low = [bytes.fromhex(dc[k]["g2"][0]) for k in ("ord13", "ord13_neg", "ord3_embedded", "ord11_embedded")]

    def fq():
        return rng.randrange(Q).to_bytes(48, "big")

    def pick(n, rate):
        a, b, f = [], [], bytearray()
        for _ in range(n):
            i = rng.randrange(1025)
            p, q, fl = g1[96 * i:96 * (i + 1)], g2[192 * i:192 * (i + 1)], (0, 0)
            if rng.random() < rate:
                kind = rng.randrange(10)
                if kind == 0:
                    p = bytes(96)
                elif kind == 1:
                    q = bytes(192)
                elif kind == 2:
                    p, q = bytes(96), bytes(192)
                elif kind == 3:
                    q = fq() + fq() + fq() + fq()
                elif kind == 4:
                    q = fq() + fq() + bytes(96)
                elif kind == 5:
                    q = bytes(96) + fq() + fq()
                elif kind == 6:
                    q = rng.choice(low)
                elif kind == 7:
                    fl = (rng.randrange(2), 1)
                elif kind == 8:
                    q, fl = rng.choice(low), (0, 1)
                else:
                    p = bytes(48) + fq()
            a.append(p)
            b.append(q)
            f += bytes(fl)
        return b"".join(a), b"".join(b), bytes(f)

    engines = {}

    def engine(min_group, teams, fexp, quad_max):
        key = (min_group, teams, fexp, quad_max)
        if key not in engines:
            # (read at context creation, csrc/blsgpu_api.hip) quad_max 0: lane pairs; None: lane quads; "wide": sixteen lanes per pair
            os.environ.pop("BLSGPU_LS_QUAD_MAX", None)
            os.environ.pop("BLSGPU_LS_WIDE_MAX", None)
            if quad_max != "wide":
                os.environ["BLSGPU_LS_WIDE_MAX"] = "0"
                if quad_max is not None:
                    os.environ["BLSGPU_LS_QUAD_MAX"] = str(quad_max)
            e = _native.Engine(0)
            os.environ.pop("BLSGPU_LS_QUAD_MAX", None)
            os.environ.pop("BLSGPU_LS_WIDE_MAX", None)
            e.set_ls_threshold(1, min_group)
            e.set_ls_teams(teams)
            e.set_fexp_team_threshold(fexp)
            engines[key] = e
        return engines[key]

    bad, t0 = 0, time.time()
    for t in range(trials):
        rate = rng.choice([0.0, 0.05, 0.3, 1.0])
        fexp = rng.choice([1, None])
        quad_max = rng.choice([0, None, "wide"]) if only_chains is None else (None if only_chains == "default" else only_chains)
        form = "lane pairs" if quad_max == 0 else ("sixteen lanes" if quad_max == "wide" else "lane quads")
        if t % 2 == 0:                                  # one multi-pairing through the per-line products
            n = rng.choice([1, 2, 5, 16, 17, 63, 64, 65, 200, 257, 600, 1025])
            a, b, f = pick(n, rate)
            e = engine(1, rng.choice([1, 300, 5000, 10 ** 9]), fexp, quad_max)
            got = e.pairing_multi(a, b, n, f)
            ok = got == O.pairing_multi(a, b, n, threads=16, inf=f)
            what = "one call"
        else:                                           # a batch of groups: small-group form or per-line products
            gsz, groups = rng.choice([(1, 40), (2, 33), (3, 21), (5, 17), (23, 9), (40, 7), (67, 5), (130, 3)])
            n = gsz * groups
            a, b, f = pick(n, rate)
            e = engine(rng.choice([1, 1 << 30]), rng.choice([1, 2000, 10 ** 9]), fexp, quad_max)
            got = e.pairing_multi_batch(a, b, gsz, groups, f)
            want = b"".join(O.pairing_multi(a[96 * gsz * g:96 * gsz * (g + 1)], b[192 * gsz * g:192 * gsz * (g + 1)], gsz, threads=16,
                                            inf=f[2 * gsz * g:2 * gsz * (g + 1)]) for g in range(groups))
            ok = got == want
            what = "%d x %d" % (groups, gsz)
        bad += 0 if ok else 1
        print("trial %d %s pairs %d degenerate %.2f fexp-team %s chains on %s %s  (%.0f s)" % (t, what, n, rate, fexp, form, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    print("line-stream soak: %d trials, %d mismatches" % (trials, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
