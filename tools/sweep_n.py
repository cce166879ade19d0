"""Scaling curve of one multi-pairing call: N pairs, one final
exponentiation, inputs resident in HBM.  Prints JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
sys.path.insert(0, os.path.join(ROOT, "oracle"))
e = _native.Engine(0)
dev = torch.device("cuda", 0)
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()
gold = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests/golden/pairing.json")))["seeded"]["1025"]["out"])
out = torch.zeros(576, dtype=torch.uint8, device=dev)
SIZES = (1, 64, 1024, 1025, 4096, 8192, 12288, 16384, 24576, 32768, 65536, 1 << 18, 1 << 20)
for n in SIZES:
    reps = (n + 1024) // 1025
    t1 = torch.frombuffer(bytearray((g1 * reps)[:96 * n]), dtype=torch.uint8).to(dev)
    t2 = torch.frombuffer(bytearray((g2 * reps)[:192 * n]), dtype=torch.uint8).to(dev)
    e.reserve(n)
    f = lambda: e.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, out.data_ptr(), 0)
    rec = {"pairs": n}
    for path in ("default", "vm", "ls"):                  # default selection, wavefront-VM kernels only, line-stream forced
        if path == "ls" and n < 1024:
            continue
        e.set_ls_threshold({"default": 16384, "vm": None, "ls": 1}[path], {"default": 64, "vm": 64, "ls": 1}[path])
        f(); torch.cuda.synchronize()
        k = 20 if n <= 8192 else 3
        t = time.perf_counter()
        for _ in range(k):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / k
        res = bytes(out.cpu().numpy())
        if path == "default":
            rec.update(ms=dt * 1e3, pairings_per_s=n / dt)
            first = res
        else:
            rec["ms_" + path] = dt * 1e3
            rec["same_" + path] = res == first
    if n == 1025:
        rec["golden_ok"] = first == gold
    print(json.dumps(rec), flush=True)
