"""Diagnostic: throughput of 1025-pair verifications vs. number of streams in flight,
with and without per-kernel timing events; host enqueue time vs. total."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
dev = torch.device("cuda", 0)
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()
n = 1025
t1 = torch.frombuffer(bytearray(g1), dtype=torch.uint8).to(dev)
t2 = torch.frombuffer(bytearray(g2), dtype=torch.uint8).to(dev)
SMAX = 32
engs = [_native.Engine(0) for _ in range(SMAX)]
outs = [torch.zeros(576, dtype=torch.uint8, device=dev) for _ in range(SMAX)]
streams = [torch.cuda.Stream(device=dev) for _ in range(SMAX)]
for e in engs:
    e.reserve(n); e.set_mp_threshold(0)
print("GPU_MAX_HW_QUEUES", os.environ["GPU_MAX_HW_QUEUES"])
for timing in (False, True):
    for e in engs: e.timing_enable(timing)
    for S in (1, 2, 4, 8, 16, 32):
        steps = 40 * S if S < 8 else 320
        for i in range(S): engs[i].pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, outs[i].data_ptr(), streams[i].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i % S
            engs[k].pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, outs[k].data_ptr(), streams[k].cuda_stream)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for e in engs: e.timing_read()
        print("timing=%d S=%2d: %.3f ms/step (enqueue %.3f ms/step)  %.2f M pairs/s" % (timing, S, dt / steps * 1e3, t_enq / steps * 1e3, n * steps / dt / 1e6), flush=True)
