#!/usr/bin/env python3
"""Times blsgpu_miller_loop_batch_dev (the reference's fq_miller_loop value of every pair) by pair count; BLSGPU_MILLER_EXACT_LANES=0 gives the
wavefront VM's program.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
e = _native.Engine(0)
dev = torch.device("cuda", 0)
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()
for n in (1, 32, 1024, 65536):
    a = torch.frombuffer(bytearray((g1 * (n // 1025 + 1))[:96 * n]), dtype=torch.uint8).to(dev)
    b = torch.frombuffer(bytearray((g2 * (n // 1025 + 1))[:192 * n]), dtype=torch.uint8).to(dev)
    o = torch.zeros(576 * n, dtype=torch.uint8, device=dev)
    f = lambda: e.miller_loop_batch_dev(a.data_ptr(), b.data_ptr(), n, o.data_ptr(), 0)
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    print("miller_loop_batch %6d pairs: %.3f ms" % (n, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
