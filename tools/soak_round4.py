#!/usr/bin/env python3
"""Randomised parity soak of the round-4 kernels (GPU box), every result against the CPU oracle / the host integer code:
  * final exponentiations one per wavefront (k_fexp_wide; default for a few results) on random, sparse, subfield and
    special Fq12 values, batches of every raggedness, with 1 .. 8 partials per result multiplied by the kernel itself;
  * hash to G2 through every cofactor-clearing form (VM, lane pairs, lane QUADS with Jacobian runs) and both symbol
    routines, on random message hashes and on crafted t values (zero halves, t1 = -t0, t1 = t0, tiny values).
usage: python tools/soak_round4.py [trials]"""
import hashlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "python-bls_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    import oracle as O
    from bls_py import _native, hostmath as H
    from bls_py.util import hash512
    O.build()
    rng = random.Random(20261004)

    def engine(env):
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
    wide = engine({})
    big = str(1 << 40)
    h2c_engines = {"default": wide,
                   "quads+jacobi": engine({"BLSGPU_H2C_REG_THRESHOLD": "1", "BLSGPU_H2C_QUAD_MAX": big, "BLSGPU_H2C_LANE_THRESHOLD": "1", "BLSGPU_H2C_JACOBI_THRESHOLD": "1"}),
                   "pairs+lane": engine({"BLSGPU_H2C_REG_THRESHOLD": "1", "BLSGPU_H2C_QUAD_MAX": "0", "BLSGPU_H2C_LANE_THRESHOLD": "1", "BLSGPU_H2C_JACOBI_THRESHOLD": big}),
                   "vm": engine({"BLSGPU_H2C_REG_THRESHOLD": big, "BLSGPU_H2C_LANE_THRESHOLD": big})}

    def fq12():
        kind = rng.randrange(8)
        c = [rng.randrange(Q) for _ in range(12)]
        if kind == 0:
            c = [x if rng.random() < 0.3 else 0 for x in c]                 # sparse
        elif kind == 1:
            c = [c[0], c[1]] + [0] * 10                                     # in Fq2
        elif kind == 2:
            c = [c[0]] + [0] * 11                                           # in Fq
        elif kind == 3:
            c = [rng.choice((0, 1, Q - 1, 2, Q - 2)) for _ in range(12)]    # tiny / huge residues
        return b"".join(x.to_bytes(48, "big") for x in c)

    bad, t0 = 0, time.time()
    for t in range(trials):
        if t % 2 == 0:
            m = rng.choice([1, 2, 3, 5, 9, 17, 33])
            vals = [fq12() for _ in range(m)]
            got = wide.final_exp_batch(b"".join(vals))
            ok = got == b"".join(O.final_exp(v) for v in vals)
            what = "final_exp x %d" % m
        else:
            n = rng.choice([1, 3, 17, 64, 100])
            ts = []
            for _ in range(n):
                kind = rng.randrange(8)
                t0_ = (rng.randrange(Q), rng.randrange(Q))
                t1_ = (rng.randrange(Q), rng.randrange(Q))
                if kind == 0:
                    t1_ = ((Q - t0_[0]) % Q, (Q - t0_[1]) % Q)              # the encodings cancel
                elif kind == 1:
                    t1_ = t0_                                               # ... or double
                elif kind == 2:
                    t0_ = (0, 0)
                elif kind == 3:
                    t0_ = (t0_[0], 0)
                elif kind == 4:
                    t1_ = (0, t1_[1])
                elif kind == 5:
                    t0_ = (rng.randrange(4), rng.randrange(4))
                ts.append(b"".join(x.to_bytes(48, "big") for x in (*t0_, *t1_)))
            name = rng.choice(sorted(h2c_engines))
            got = h2c_engines[name].map_to_g2(b"".join(ts))
            want = b""
            for tb in ts:
                v = [int.from_bytes(tb[48 * j:48 * (j + 1)], "big") for j in range(4)]
                S = [H.aff_to_jac(H.F2, H.sw_encode(H.F2, (v[2 * j], v[2 * j + 1]))) for j in range(2)]
                want += H.g2_affine_bytes(H.clear_cofactor_g2(H.jac_add(H.F2, S[0], S[1])))
            ok = got == want
            what = "map_to_g2 x %d (%s)" % (n, name)
        bad += 0 if ok else 1
        print("trial %d %s %s  (%.0f s)" % (t, what, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    print("round-4 soak: %d trials, %d mismatches" % (trials, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
