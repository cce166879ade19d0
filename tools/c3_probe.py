"""One multi-pairing call of n pairs (C3 and its per-GPU shards): wall time per call, inputs resident.  Run under
rocprofv3 --kernel-trace for the per-kernel timeline (tools/kernel_timeline.py reads the database)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
e = _native.Engine(0)
dev = torch.device("cuda", 0)
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()
out = torch.zeros(576, dtype=torch.uint8, device=dev)
part = torch.zeros(144, dtype=torch.int32, device=dev)
sizes = [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768, 65536]
for n in sizes:
    reps = (n + 1024) // 1025
    t1 = torch.frombuffer(bytearray((g1 * reps)[:96 * n]), dtype=torch.uint8).to(dev)
    t2 = torch.frombuffer(bytearray((g2 * reps)[:192 * n]), dtype=torch.uint8).to(dev)
    e.reserve(n)
    rec = {"pairs": n}
    for name, f in (("pairing_multi", lambda: e.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, out.data_ptr(), 0)),
                    ("miller_product_only", lambda: e.miller_product_batch_dev(t1.data_ptr(), t2.data_ptr(), n, 1, part.data_ptr()))):
        f(); torch.cuda.synchronize()
        k = 10
        t = time.perf_counter()
        for _ in range(k):
            f()
        torch.cuda.synchronize()
        rec["ms_" + name] = (time.perf_counter() - t) / k * 1e3
    print(json.dumps(rec), flush=True)
