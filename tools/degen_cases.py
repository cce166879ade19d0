import json, os, sys
ROOT = "/root/repo"
for p in (os.path.join(ROOT, "python-bls_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from bls_py import _native
e = _native.Engine(0)
e.set_ls_threshold(1, 1)
cat = lambda hs: b"".join(bytes.fromhex(x) for x in hs)
d = json.load(open(os.path.join(ROOT, "tests/golden/pairing_degenerate.json")))["cases"]
g = json.load(open(os.path.join(ROOT, "tests/golden/pairing.json")))["edge"]
for src, name in [(g, k) for k in g] + [(d, k) for k in d]:
    c = src[name]
    n = len(c["g1"])
    inf = bytes(int(b) for pr in c["inf"] for b in pr) if "inf" in c else None
    got = e.pairing_multi(cat(c["g1"]), cat(c["g2"]), n, inf).hex() if n else None
    print("%-28s n=%d %s" % (name, n, "ok" if got == c["out"] or n == 0 else "FAIL"), c.get("inf"))
