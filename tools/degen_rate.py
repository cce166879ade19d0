"""Rate of a batch made ONLY of degenerate pairs (low-order / off-curve Q) through the default selection, against the
oracle on a sample.  usage: python tools/degen_rate.py [pairs_per_group] [groups]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "python-bls_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
from bls_py import _native
import oracle as O
e = _native.Engine(0)
d = json.load(open(os.path.join(ROOT, "tests/golden/pairing_degenerate.json")))["cases"]
names = ["ord13", "ord11_embedded", "off_curve", "qy_zero", "ord3_embedded", "ord13_neg", "qx_zero", "ord13_px_zero"]
a = b"".join(bytes.fromhex(d[k]["g1"][0]) for k in names)
b = b"".join(bytes.fromhex(d[k]["g2"][0]) for k in names)
gsz = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 64
g1, g2 = (a * (gsz // 8 + 1))[:96 * gsz], (b * (gsz // 8 + 1))[:192 * gsz]
want = O.pairing_multi(g1, g2, gsz, threads=8)
t1 = torch.frombuffer(bytearray(g1 * groups), dtype=torch.uint8).cuda()
t2 = torch.frombuffer(bytearray(g2 * groups), dtype=torch.uint8).cuda()
out = torch.zeros(576 * groups, dtype=torch.uint8, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    e.pairing_multi_batch_dev(t1.data_ptr(), t2.data_ptr(), gsz, groups, out.data_ptr())
    torch.cuda.synchronize(); dt = time.time() - t
res = bytes(out.cpu().numpy())
print(json.dumps({"pairs": gsz * groups, "all_degenerate": True, "ms": dt * 1e3, "pairs_per_s": gsz * groups / dt,
                  "equals_oracle": all(res[576 * g:576 * (g + 1)] == want for g in range(groups))}))
