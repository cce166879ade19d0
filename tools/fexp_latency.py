"""Latency of the final exponentiation forms on the GPU (one call, inputs resident): m results per call through
  vm    the wavefront VM's program (one wavefront per result, a field product per lane; rounds 1 - 3)
  team  six lanes per result, ten results per wavefront (blsgpu_fexp.hip)
  wide  one result per wavefront, a product per lane (blsgpu_fexpw.hip; round 4)
and of single multi-pairing calls end to end with the VM and the wide form.  Prints JSON lines."""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
dev = torch.device("cuda", 0)


def engine(wide, team):
    os.environ["BLSGPU_FEXP_WIDE"] = "1" if wide else "0"
    e = _native.Engine(0)
    e.set_fexp_team_threshold(1 if team else None)
    return e


E = {"vm": engine(False, False), "team": engine(False, True), "wide": engine(True, False)}
rnd = random.Random(1)
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()


def timed(f, k):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e3


# (a) m independent results: Miller values of single pairs as inputs (m groups of one pair: the Miller kernel's time is
# the same for the three engines, so differences are the final exponentiation's)
for m in (1, 4, 16, 64, 256, 1024, 2048, 4096, 6144, 8192, 10000):
    reps = (m + 1024) // 1025
    t1 = torch.frombuffer(bytearray((g1 * reps)[:96 * m]), dtype=torch.uint8).to(dev)
    t2 = torch.frombuffer(bytearray((g2 * reps)[:192 * m]), dtype=torch.uint8).to(dev)
    out = {k: torch.zeros(576 * m, dtype=torch.uint8, device=dev) for k in E}
    rec = {"results_per_call": m}
    for k, e in E.items():
        rec["ms_" + k] = timed(lambda: e.pairing_multi_batch_dev(t1.data_ptr(), t2.data_ptr(), 1, m, out[k].data_ptr(), 0), 10)
    rec["same"] = bool((out["vm"] == out["wide"]).all()) and bool((out["vm"] == out["team"]).all())
    print(json.dumps(rec), flush=True)

# (b) one multi-pairing of n pairs
gold = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests/golden/pairing.json")))["seeded"]["1025"]["out"])
for n in (1, 64, 1025, 4096, 8192, 16384, 65536):
    reps = (n + 1024) // 1025
    t1 = torch.frombuffer(bytearray((g1 * reps)[:96 * n]), dtype=torch.uint8).to(dev)
    t2 = torch.frombuffer(bytearray((g2 * reps)[:192 * n]), dtype=torch.uint8).to(dev)
    rec = {"pairs": n}
    outs = {}
    for k in ("vm", "wide"):
        e = E[k]
        e.reserve(n)
        o = torch.zeros(576, dtype=torch.uint8, device=dev)
        rec["ms_" + k] = timed(lambda: e.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, o.data_ptr(), 0), 20 if n <= 8192 else 5)
        outs[k] = bytes(o.cpu().numpy())
    rec["same"] = outs["vm"] == outs["wide"]
    if n == 1025:
        rec["golden_ok"] = outs["wide"] == gold
    print(json.dumps(rec), flush=True)
