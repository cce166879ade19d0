"""Trace the batched final exponentiation (k_fexp_team) operation by operation against the integer model
(vmgen/fexp_model.py): prints the first operation of the script after which result 0 differs."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
from vmgen import fexp_model as F

e = _native.Engine(0)
e.set_fexp_team_threshold(1)
S = F.script()
buf = torch.zeros(len(S) * 576, dtype=torch.uint8, device="cuda")
e._check(e.lib.blsgpu_ctx_set_fexp_trace(e.h, buf.data_ptr()), "trace")
g = json.load(open(os.path.join(ROOT, "tests/golden/pairing.json")))
x = bytes.fromhex(g["gen"]["miller"])
out = e.final_exp(x)
torch.cuda.synchronize()
tr = bytes(buf.cpu().numpy())
ints = lambda b: [int.from_bytes(b[48 * i:48 * (i + 1)], "big") for i in range(12)]
# model trace
M = [None] * F.NSLOTS
acc = F.from_flat12(ints(x))
names = ["END", "MUL", "CSQ", "ST", "LD", "CONJ", "FROB", "TINV"]
bad = None
for pc, (op, a) in enumerate(S):
    if op == F.MUL: acc = F.mul_dense(acc, M[a])
    elif op == F.CSQ:
        for _ in range(a): acc = F.cyc_sqr_lane_forms(acc)
    elif op == F.ST: M[a] = list(acc)
    elif op == F.LD: acc = list(M[a])
    elif op == F.CONJ: acc = F.conj6(acc)
    elif op == F.FROB: acc = F.frob(acc, F.FROB_POW[a])
    elif op == F.TINV:
        t = acc[0]; ni = F.fq_inv((t[0] * t[0] + t[1] * t[1]) % F.Q)
        acc = [(t[0] * ni % F.Q, (-t[1]) * ni % F.Q)] + [(0, 0)] * 5
    else: break
    got = F.from_flat12(ints(tr[576 * pc:576 * (pc + 1)]))
    if op == F.TINV:
        ok = got[0] == acc[0]
    else:
        ok = got == acc
    print(pc, names[op], a, "ok" if ok else "DIFF " + str([i for i in range(6) if got[i] != acc[i]]))
    if not ok and bad is None:
        bad = pc
        if len(sys.argv) < 2:
            break
print("first difference after op", bad, "| final", out.hex() == g["gen"]["final_exp"])
