"""Compare the line records k_ml_lines_exact leaves for ONE degenerate pair with the integer model
(vmgen/linestream_model.exact_pair_lines): prints the first line / coefficient that differs."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
from bls_py import _native
from vmgen import linestream_model as M
from vmgen.gen_fp28 import R, Q

e = _native.Engine(0)
e.set_ls_threshold(1, 1)
name = sys.argv[1] if len(sys.argv) > 1 else "flag_on_valid"
d = json.load(open(os.path.join(ROOT, "tests/golden/pairing_degenerate.json")))["cases"]
c = d[name]
a, b = bytes.fromhex(c["g1"][0]), bytes.fromhex(c["g2"][0])
inf = bytes(int(x) for x in c["inf"][0])
out = e.pairing_multi(a, b, 1, inf)
buf = (ctypes.c_int32 * (68 * 84))()
e._check(e.lib.blsgpu_debug_read_lines(e.h, buf, 68 * 84 * 4), "read lines")
I = lambda x: int.from_bytes(x, "big")
P, Qa = (I(a[:48]), I(a[48:])), ((I(b[:48]), I(b[48:96])), (I(b[96:144]), I(b[144:])))
want = M.exact_pair_lines(P, Qa, bool(inf[1]))
Ri = pow(R, -1, Q)
bad = 0
for L, (pos, cf) in enumerate(want):
    got = []
    for k in range(6):
        limbs = buf[L * 84 + k * 14:L * 84 + (k + 1) * 14]
        got.append(sum(int(v) << (28 * i) for i, v in enumerate(limbs)) * Ri % Q)
    g3 = [(got[0], got[1]), (got[2], got[3]), (got[4], got[5])]
    ok = [g3[i] == (cf[i][0] % Q, cf[i][1] % Q) for i in range(3)]
    if not all(ok):
        bad += 1
        if bad <= 6:
            print("line", L, M.line_schedule()[L], "positions", pos, "coefficients ok:", ok)
print(name, "lines differing:", bad, "| result", out.hex() == c["out"])
