"""Diagnostic: throughput of B independent 1025-pair verifications per launch sequence."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
dev = torch.device("cuda", 0)
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()
n = 1025
S = 4
engs = [_native.Engine(0) for _ in range(S)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
for B in (1, 4, 8, 16, 32, 64):
    t1 = torch.frombuffer(bytearray(g1 * B), dtype=torch.uint8).to(dev)
    t2 = torch.frombuffer(bytearray(g2 * B), dtype=torch.uint8).to(dev)
    outs = [torch.zeros(B * 576, dtype=torch.uint8, device=dev) for _ in range(S)]
    for e in engs:
        e.reserve((n + 3) * B); e.set_mp_threshold(0)
    for s_used in (1, 2, 4):
        steps = max(8, 256 // B)
        for i in range(s_used):
            engs[i].pairing_multi_batch_dev(t1.data_ptr(), t2.data_ptr(), n, B, outs[i].data_ptr(), streams[i].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i % s_used
            engs[k].pairing_multi_batch_dev(t1.data_ptr(), t2.data_ptr(), n, B, outs[k].data_ptr(), streams[k].cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = bytes(outs[0][:576].cpu().numpy()) == bytes(outs[0][576 * (B - 1):].cpu().numpy())
        print("B=%2d streams=%d: %.3f ms/launch  %.2f M pairs/s  same=%s" % (B, s_used, dt / steps * 1e3, n * B * steps / dt / 1e6, ok), flush=True)
