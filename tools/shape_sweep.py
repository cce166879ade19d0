#!/usr/bin/env python3
"""Times the C-ABI entry points in the call SHAPES the bench does not cover (GPU box only): batches of one-point sums (scalar
multiplications), batches of small plain sums, per-pair Miller loops, batches of final exponentiations, decompression and the
hash for a few inputs, many small pairing groups.  Device buffers, wall time around back-to-back calls; one line per shape with the
rate -- a shape that runs far below its neighbours is a path nobody measured (round 5 found the plain sums that way).
usage: shape_sweep.py"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))


def main():
    import torch
    from bls_py import _native
    e = _native.Engine(0)
    L = e.lib
    dev = torch.device("cuda", 0)
    gold = os.path.join(ROOT, "tests", "golden")
    g1 = open(os.path.join(gold, "pairs_seed1_g1.bin"), "rb").read()
    g2 = open(os.path.join(gold, "pairs_seed1_g2.bin"), "rb").read()
    np1, np2 = len(g1) // 96, len(g2) // 192
    up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    zeros = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)

    def timed(what, units, fn, reps=3):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-64s %10.3f ms  %12.0f /s" % (what, dt * 1e3, units / dt), flush=True)

    def rep(src, sz, npts, n):
        return (src * (n // npts + 1))[:sz * n]
    scal = lambda n: b"".join(hashlib.sha256(b"sweep" + i.to_bytes(4, "big")).digest() for i in range(n))

    for G in (1024, 65536):
        p1, p2, sc = up(rep(g1, 96, np1, G)), up(rep(g2, 192, np2, G)), up(scal(G))
        o1, o2, fl = zeros(96 * G), zeros(192 * G), zeros(G)
        timed("G1 scalar multiplications, batch of %d (groups x 1 point)" % G, G,
              lambda: L.blsgpu_g1_msm_dev(e.h, p1.data_ptr(), sc.data_ptr(), 1, G, o1.data_ptr(), fl.data_ptr(), 0))
        timed("G2 scalar multiplications, batch of %d (groups x 1 point)" % G, G,
              lambda: L.blsgpu_g2_msm_dev(e.h, p2.data_ptr(), sc.data_ptr(), 1, G, o2.data_ptr(), fl.data_ptr(), 0))
        k = 4
        Gk = G // k
        timed("G1 plain sums, %d groups x %d points" % (Gk, k), G,
              lambda: L.blsgpu_g1_msm_dev(e.h, p1.data_ptr(), None, k, Gk, o1.data_ptr(), fl.data_ptr(), 0))
        timed("G2 plain sums, %d groups x %d points" % (Gk, k), G,
              lambda: L.blsgpu_g2_msm_dev(e.h, p2.data_ptr(), None, k, Gk, o2.data_ptr(), fl.data_ptr(), 0))
        timed("G1 sums with scalars, %d groups x %d points" % (Gk, k), G,
              lambda: L.blsgpu_g1_msm_dev(e.h, p1.data_ptr(), sc.data_ptr(), k, Gk, o1.data_ptr(), fl.data_ptr(), 0))
        of = zeros(576 * G)
        timed("Miller loops, one Fq12 per pair, %d pairs" % G, G,
              lambda: e.miller_loop_batch_dev(p1.data_ptr(), p2.data_ptr(), G, of.data_ptr(), 0))
        timed("pairings in groups of 2 pairs, %d groups" % (G // 2), G,
              lambda: e.pairing_multi_batch_dev(p1.data_ptr(), p2.data_ptr(), 2, G // 2, of.data_ptr(), 0))
    # host-buffer entries (PCIe included): decompression, final exponentiations, map / hash of a few inputs
    from bls_py import hostmath as H
    c1 = b"".join(H.g1_compress(H.g1_from_abi(g1[96 * i:96 * (i + 1)])) for i in range(256))
    c2 = b"".join(H.g2_compress(H.g2_from_abi(g2[192 * i:192 * (i + 1)])) for i in range(256))
    for n in (1, 256, 65536):
        a, b = (c1 * (n // 256 + 1))[:48 * n], (c2 * (n // 256 + 1))[:96 * n]
        timed("G1 decompression (host buffers), %d points" % n, n, lambda: e.g1_decompress(a))
        timed("G2 decompression (host buffers), %d points" % n, n, lambda: e.g2_decompress(b))
    ml = e.miller_loop_batch(g1[:96 * 64], g2[:192 * 64], 64)
    for m in (1, 64, 4096):
        xs = (ml * (m // 64 + 1))[:576 * m]
        timed("final exponentiations (host buffers), batch of %d" % m, m, lambda: e.final_exp_batch(xs))
    for n in (1, 64, 4096):
        hs = b"".join(hashlib.sha256(b"m%d" % i).digest() for i in range(n))
        timed("hash to G2 (host buffers), %d messages" % n, n, lambda: e.hash_to_g2(hs))


if __name__ == "__main__":
    main()
