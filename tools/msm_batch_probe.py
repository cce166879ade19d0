"""Times of the kernels of a batch of G2 sums (BASELINE configs[3] shape: groups x 67 points) run back to back:
usage: python3 tools/msm_batch_probe.py [groups] [repeats]   (run under rocprofv3 --kernel-trace for per-dispatch times)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eng = _native.Engine(0)
th = json.load(open(os.path.join(ROOT, "tests", "golden", "threshold.json")))["67_of_100"]
pts = b"".join(bytes.fromhex(s) for s in th["unit_sigs_affine"])
lam = [int(x, 16) for x in th["lambdas"]]
sc = b"".join(l.to_bytes(32, "big") for l in lam)
dev = torch.device("cuda:0")
tp = torch.frombuffer(bytearray(pts * groups), dtype=torch.uint8).to(dev)
ts = torch.frombuffer(bytearray(sc * groups), dtype=torch.uint8).to(dev)
out = torch.zeros(groups * 192, dtype=torch.uint8, device=dev)
inf = torch.zeros(groups, dtype=torch.uint8, device=dev)
for i in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng._check(eng.lib.blsgpu_g2_msm_dev(eng.h, tp.data_ptr(), ts.data_ptr(), 67, groups, out.data_ptr(), inf.data_ptr(), 0), "g2_msm_dev")
    torch.cuda.synchronize()
    print("call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3))
    if i == 2:
        time.sleep(0.2)
got = bytes(out.cpu().numpy())
print("all groups equal the reference's combined signature:", got == bytes.fromhex(th["combined_affine"]) * groups)
