#!/usr/bin/env python3
"""Times ONE plain sum (no scalars: BLS.aggregate_pub_keys / aggregate_sigs without exponents, bls.py:203-261) of n G1 / G2 points on
device buffers by size; BLSGPU_MSM_PLAIN_THRESHOLD=1000000000000 gives the wavefront VM's k_msm.  GPU box only.
usage: plain_sum_probe.py [sizes ...]"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
e = _native.Engine(0)
dev = torch.device("cuda", 0)
for deg, name in ((1, "g1"), (2, "g2")):
    src = open(os.path.join(ROOT, "tests", "golden", "pairs_seed1_%s.bin" % name), "rb").read()
    sz = 96 * deg
    npts = len(src) // sz
    for n in ([int(x) for x in sys.argv[1:]] or [1024, 65536, 1 << 20]):
        pts = (src * (n // npts + 1))[:sz * n]
        dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
        out = torch.zeros(sz, dtype=torch.uint8, device=dev)
        inf = torch.zeros(1, dtype=torch.uint8, device=dev)
        fn = e.lib.blsgpu_g1_msm_dev if deg == 1 else e.lib.blsgpu_g2_msm_dev
        def call():
            rc = fn(e.h, dp.data_ptr(), None, n, 1, out.data_ptr(), inf.data_ptr(), 0)
            assert rc == 0, rc
        call(); call(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): call()
        torch.cuda.synchronize()
        print("%s plain sum of %8d points: %.3f ms" % (name, n, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
