"""First-light check of the line-stream kernels (csrc/blsgpu_ml.hip) on the GPU box: parity on small and seeded
batches with the path forced, then timing per kernel at the bench shape.  usage: python tools/ml_check.py [pairs]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "python-bls_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from bls_py import _native
import oracle as O

G = os.path.join(ROOT, "tests", "golden")
e = _native.Engine(0)
cat = lambda hs: b"".join(bytes.fromhex(x) for x in hs)
v = json.load(open(os.path.join(G, "pairing.json")))["small4"]
g1s, g2s = cat(v["g1"]), cat(v["g2"])
ok = True


def check(name, got, want):
    global ok
    good = got == want
    ok = ok and good
    print("%-50s %s" % (name, "ok" if good else "MISMATCH"), flush=True)


TIME_ONLY = "--time-only" in sys.argv
if TIME_ONLY:
    sys.argv.remove("--time-only")
e.set_ls_threshold(1, 1)
for n in (() if TIME_ONLY else (1, 2, 3, 4)):
    check("small4[:%d] forced line-stream" % n, e.pairing_multi(g1s[:96 * n], g2s[:192 * n], n), O.pairing_multi(g1s[:96 * n], g2s[:192 * n], n))
g1 = open(os.path.join(G, "pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(G, "pairs_seed1_g2.bin"), "rb").read()
gold = json.load(open(os.path.join(G, "pairing.json")))
for n in (() if TIME_ONLY else (7, 64, 65, 200)):
    check("seeded[:%d]" % n, e.pairing_multi(g1[:96 * n], g2[:192 * n], n), O.pairing_multi(g1[:96 * n], g2[:192 * n], n, threads=8))
e.set_ls_threshold(None)
want1025 = e.pairing_multi(g1, g2, 1025)
e.set_ls_threshold(1, 1)
check("seeded 1025 vs VM kernels", e.pairing_multi(g1, g2, 1025), want1025)
# batch form: 5 groups of 205 pairs
want = b"".join(O.pairing_multi(g1[96 * 205 * i:96 * 205 * (i + 1)], g2[192 * 205 * i:192 * 205 * (i + 1)], 205, threads=8) for i in range(5))
check("batch 5 x 205", e.pairing_multi_batch(g1, g2, 205, 5), want)
# degenerate fixtures
d = {} if TIME_ONLY else json.load(open(os.path.join(G, "pairing_degenerate.json")))["cases"]
for name, c in d.items():
    n = len(c["g1"])
    inf = bytes(int(b) for pr in c["inf"] for b in pr) if "inf" in c else None
    check("degenerate " + name, e.pairing_multi(cat(c["g1"]), cat(c["g2"]), n, inf).hex(), c["out"])
# a degenerate pair hidden in the seeded batch
c = json.load(open(os.path.join(G, "pairing_degenerate.json")))["cases"]["ord13"]
mix1 = g1[:96 * 100] + cat(c["g1"]) + g1[96 * 100:96 * 300]
mix2 = g2[:192 * 100] + cat(c["g2"]) + g2[192 * 100:192 * 300]
nm = 300 + len(c["g1"])
check("ord13 inside 300 seeded", e.pairing_multi(mix1, mix2, nm), O.pairing_multi(mix1, mix2, nm, threads=8))

# timing
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dg1 = torch.frombuffer(bytearray(g1 * B), dtype=torch.uint8).cuda()
dg2 = torch.frombuffer(bytearray(g2 * B), dtype=torch.uint8).cuda()
out = torch.empty(576 * B, dtype=torch.uint8, device="cuda")
for mode in (("ls",) if TIME_ONLY else ("vm", "ls")):
    e.set_ls_threshold(None if mode == "vm" else 1, 1)
    for rep in range(3):
        e.timing_enable(True)
        torch.cuda.synchronize()
        t = time.time()
        e.pairing_multi_batch_dev(dg1.data_ptr(), dg2.data_ptr(), 1025, B, out.data_ptr())
        torch.cuda.synchronize()
        dt = time.time() - t
        per = {}
        for k, m in e.timing_read():
            per[k] = per.get(k, 0.0) + m
        print(mode, "B=%d  %.2f ms  %.2f M pairings/s  kernels by kind: %s" % (B, dt * 1e3, 1025 * B / dt / 1e6, {k: round(x, 3) for k, x in sorted(per.items())}), flush=True)
    res = bytes(out.cpu().numpy())
    if mode == "vm":
        ref = res
    elif not TIME_ONLY:
        check("batch %d x 1025 line-stream == VM kernels" % B, res, ref)
        check("group 0 == the reference's seeded 1025-pair vector", res[:576].hex(), gold["seeded"]["1025"]["out"] if isinstance(gold["seeded"]["1025"], dict) else gold["seeded"]["1025"])
print("ALL OK" if ok else "FAILURES")
