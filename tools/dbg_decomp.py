import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
from bls_py import _native, hostmath as H
e = _native.Engine(0)
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
for n in (1, 2, 31, 32, 33, 40):
    pts = [g1[96 * i:96 * (i + 1)] for i in range(n)]
    enc = b"".join(H.g1_compress(H.g1_from_abi(p)) for p in pts)
    out, ok = e.g1_decompress(enc)
    bad = [i for i in range(n) if out[96 * i:96 * (i + 1)] != pts[i] or not ok[i]]
    print("n=%d bad=%s" % (n, bad[:10]))
    for i in bad[:2]:
        print("  want x", pts[i][:48].hex()[:24], "y", pts[i][48:].hex()[:24]); print("  got  x", out[96*i:96*i+48].hex()[:24], "y", out[96*i+48:96*i+96].hex()[:24], "ok", ok[i])
