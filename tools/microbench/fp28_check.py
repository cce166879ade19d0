"""Check tools/microbench/fp28_bench (generated carry-free products) against Python integers.
usage (on the GPU box):  python3 tools/microbench/fp28_check.py <path to fp28_bench>"""
import os, random, struct, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "python-bls_amd"))
from vmgen import gen_fp28 as G

Q, R = G.Q, G.R
random.seed(28)
exe = sys.argv[1]
vals = {}     # (variant, lane) -> list of 8 operand integers
blob = []
for v in range(1, 7):
    for lane in range(64):
        ops = []
        for t in range(8):
            kind = (lane + t) % 4
            if kind == 0: x = random.randint(0, Q - 1)
            elif kind == 1: x = random.randint(-Q, 2 * Q)            # the range a product returns
            elif kind == 2: x = random.randint(0, Q - 1) - random.randint(0, Q - 1)   # a difference of reduced values
            else: x = random.choice([0, 1, -1, Q - 1, Q, 2 * Q - 1, -Q + 1])
            ops.append(x)
        vals[(v, lane)] = ops
        for x in ops:
            blob += G.to_limbs(x)
with tempfile.TemporaryDirectory() as td:
    fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<%di" % len(blob), *blob))
    subprocess.check_call([exe, "check", fin, fout])
    out = struct.unpack("<%di" % (6 * 64 * 14), open(fout, "rb").read())
bad = 0
for v in range(1, 7):
    for lane in range(64):
        x = vals[(v, lane)]
        T = {1: x[0] * x[1], 2: x[0] * x[1] + x[2] * x[3], 3: x[0] * x[1] + x[2] * x[3] + x[4] * x[5],
             4: x[0] * x[1] + x[2] * x[3] + x[4] * x[5] + x[6] * x[7], 5: x[0] * x[0], 6: x[0] * x[0] + x[2] * x[3]}[v]
        d = out[((v - 1) * 64 + lane) * 14:((v - 1) * 64 + lane + 1) * 14]
        r = G.from_limbs(d)
        terms = {1: [(0, 1)], 2: [(0, 1), (2, 3)], 3: [(0, 1), (2, 3), (4, 5)], 4: [(0, 1), (2, 3), (4, 5), (6, 7)], 5: [(0, 0)], 6: [(0, 0), (2, 3)]}[v]
        want = G.model_dot([(G.to_limbs(x[i]), G.to_limbs(x[j])) for i, j in terms])
        ok = (r * R - T) % Q == 0 and all(0 <= d[j] < (1 << 28) for j in range(13)) and list(d) == want
        bad += not ok
print("fp28 products vs Python integers: %d cases, %d mismatches" % (6 * 64, bad))
sys.exit(1 if bad else 0)
