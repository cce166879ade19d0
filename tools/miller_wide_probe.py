"""Latency of small multi-pairing calls (1 .. 4096 pairs, inputs in HBM, one final exponentiation) through the wide Miller loop
(k_miller_wide, csrc/blsgpu_mlw.hip) against the wavefront VM's k_miller / k_miller_mp: whole call (HIP events around it) and the
Miller kernel alone (the engine's kernel timers, kind 0), plus the other kernels of the call.  Prints JSON lines."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
dev = torch.device("cuda", 0)
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()
out = torch.zeros(576, dtype=torch.uint8, device=dev)
NAMES = {0: "miller", 1: "reduce", 2: "final_exp", 3: "exact", 4: "ls_lines", 5: "ls_accum", 6: "ls_merge", 7: "ls_horner"}
engines = {}
for name, wide_max in (("wide", 1 << 30), ("vm", 0)):
    e = _native.Engine(0)
    e.set_ls_threshold(None)
    e.set_miller_wide_max(wide_max)
    engines[name] = e
sizes = [int(x) for x in sys.argv[1:]] or [1, 2, 3, 5, 8, 64, 256, 512, 1024, 1025, 2048, 3072, 4096]
stream = torch.cuda.Stream(device=dev)
for n in sizes:
    reps = (n + 1024) // 1025
    t1 = torch.frombuffer(bytearray((g1 * reps)[:96 * n]), dtype=torch.uint8).to(dev)
    t2 = torch.frombuffer(bytearray((g2 * reps)[:192 * n]), dtype=torch.uint8).to(dev)
    rec, res = {"pairs": n}, {}
    for name, e in engines.items():
        e.reserve(n + 8)
        f = lambda: e.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, out.data_ptr(), stream.cuda_stream)
        f(); stream.synchronize()
        best = 1e9
        for _ in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); f(); b.record(stream); stream.synchronize()
            best = min(best, a.elapsed_time(b))
        res[name] = bytes(out.cpu().numpy())
        e.timing_enable(True)
        f(); stream.synchronize()
        kt = {}
        for kind, ms in e.timing_read():
            kt[NAMES.get(kind, str(kind))] = round(kt.get(NAMES.get(kind, str(kind)), 0.0) + ms, 4)
        e.timing_enable(False)
        rec[name] = {"call_ms": round(best, 4), "kernels_ms": kt}
    rec["same"] = res["wide"] == res["vm"]
    print(json.dumps(rec), flush=True)
