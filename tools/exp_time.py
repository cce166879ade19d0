"""Timing experiments: run the Miller kernel of variant libraries (csrc/exp_<mask>.so, results
wrong by construction) to see where its time goes under full occupancy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
lib = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
os.environ["BLSGPU_LIBRARY"] = os.path.join(ROOT, "python-bls_amd", "csrc", lib)
import torch
from bls_py import _native
e = _native.Engine(0)
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()
reps = (n + 1024) // 1025
dev = torch.device("cuda", 0)
t1 = torch.frombuffer(bytearray((g1 * reps)[:96 * n]), dtype=torch.uint8).to(dev)
t2 = torch.frombuffer(bytearray((g2 * reps)[:192 * n]), dtype=torch.uint8).to(dev)
part = torch.zeros(144, dtype=torch.int32, device=dev)
e.reserve(n)
for mp in (0, 1 << 30):
    e.set_mp_threshold(mp)
    f = lambda: e.miller_product_dev(t1.data_ptr(), t2.data_ptr(), n, part.data_ptr(), 0)
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print("%s n=%d %s: %.3f ms  %.2f M pairs/s" % (lib, n, "multi-pair teams" if mp == 0 else "single-pair teams", dt * 1e3, n / dt / 1e6))
