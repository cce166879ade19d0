"""Where the one-result-per-wavefront final exponentiation (k_fexp_wide) spends its cycles: the kernel's own cycle
counter after every operation of the script, summed per operation kind."""
import collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
from vmgen import fexp_model as F

e = _native.Engine(0)
e.set_fexp_team_threshold(None)
S = F.script()
buf = torch.zeros(len(S) * 576, dtype=torch.uint8, device="cuda")
e._check(e.lib.blsgpu_ctx_set_fexpw_stamps(e.h, buf.data_ptr()), "stamps")
g = json.load(open(os.path.join(ROOT, "tests/golden/pairing.json")))
x = bytes.fromhex(g["gen"]["miller"])
for _ in range(3):
    out = e.final_exp(x)
torch.cuda.synchronize()
assert out.hex() == g["gen"]["final_exp"]
st = buf.cpu().numpy().view("uint64")[:len(S) + 1]
names = ["END", "MUL", "CSQ", "ST", "LD", "CONJ", "FROB", "TINV"]
tot = collections.Counter(); cnt = collections.Counter()
for pc, (op, a) in enumerate(S[:-1]):
    d = int(st[pc + 1]) - int(st[pc])
    tot[names[op]] += d
    cnt[names[op]] += a if op == F.CSQ else 1
all_ = int(st[len(S) - 1]) - int(st[0])
print(json.dumps({"cycles_total": all_, "per_kind": {k: {"cycles": tot[k], "count": cnt[k], "cycles_each": tot[k] / cnt[k]} for k in tot}}))
