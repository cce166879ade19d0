#!/usr/bin/env python3
"""Times ONE G2 sum with scalars (BLS.aggregate_sigs(secure), bls.py:225-261, as a multi-scalar sum) by size, device buffers, HIP
events around the call; GPU box only.  usage: g2_single_sum_probe.py [sizes ...]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))


def main():
    import torch
    from bls_py import _native
    sizes = [int(x) for x in sys.argv[1:]] or [1, 64, 1024, 16384, 65536]
    e = _native.Engine(0)
    g2 = open(os.path.join(ROOT, "tests", "golden", "pairs_seed1_g2.bin"), "rb").read()
    npts = len(g2) // 192
    dev = torch.device("cuda", 0)
    for n in sizes:
        pts = b"".join(g2[192 * (i % npts):192 * (i % npts + 1)] for i in range(n))
        t = b"".join(hashlib.sha256(b"probe/t" + i.to_bytes(4, "big")).digest() for i in range(n))
        dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
        ds = torch.frombuffer(bytearray(t), dtype=torch.uint8).to(dev)
        out = torch.zeros(192, dtype=torch.uint8, device=dev)
        inf = torch.zeros(1, dtype=torch.uint8, device=dev)

        def call():
            rc = e.lib.blsgpu_g2_msm_dev(e.h, dp.data_ptr(), ds.data_ptr(), n, 1, out.data_ptr(), inf.data_ptr(), 0)
            assert rc == 0, rc
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        print("%8d points  %.3f ms per G2 sum  digest %s" % (n, (time.perf_counter() - t0) / reps * 1e3, hashlib.sha256(bytes(out.cpu().numpy())).hexdigest()[:16]), flush=True)


if __name__ == "__main__":
    main()
