"""One multi-pairing call of n pairs (inputs in HBM, one final exponentiation): the line-stream kernels with the point chains sixteen
lanes per pair (k_ml_lines_wide), on lane quads (k_ml_lines4) / lane pairs, and the kernels below the line-stream threshold
(k_miller_wide / k_miller_mp): where BLSGPU_LS_WIDE_MAX and BLSGPU_LS_THRESHOLD belong.  Prints JSON lines."""
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


def engine(env, ls):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    e = _native.Engine(0)
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    e.set_ls_threshold(1 if ls else None, 1)
    return e


engs = {"ls_wide": engine({"BLSGPU_LS_WIDE_MAX": str(1 << 40)}, True), "ls_quads_pairs": engine({"BLSGPU_LS_WIDE_MAX": "0"}, True), "no_ls": engine({}, False)}
stream = torch.cuda.Stream(device=dev)
for n in [int(x) for x in sys.argv[1:]] or [1025, 1536, 2048, 3072, 4096, 6144, 8192, 10240, 12288, 16384, 24576]:
    reps = (n + 1024) // 1025
    t1 = torch.frombuffer(bytearray((g1 * reps)[:96 * n]), dtype=torch.uint8).to(dev)
    t2 = torch.frombuffer(bytearray((g2 * reps)[:192 * n]), dtype=torch.uint8).to(dev)
    rec, res = {"pairs": n}, {}
    for name, e in engs.items():
        e.reserve(n + 8)
        f = lambda: e.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, out.data_ptr(), stream.cuda_stream)
        f(); stream.synchronize()
        best = 1e9
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); f(); b.record(stream); stream.synchronize()
            best = min(best, a.elapsed_time(b))
        res[name] = bytes(out.cpu().numpy())
        e.timing_enable(True)
        f(); stream.synchronize()
        kt = {}
        for kind, ms in e.timing_read():
            kt[NAMES.get(kind, str(kind))] = round(kt.get(NAMES.get(kind, str(kind)), 0.0) + ms, 3)
        e.timing_enable(False)
        rec[name] = {"call_ms": round(best, 3), "kernels_ms": kt}
    rec["same"] = len(set(res.values())) == 1
    print(json.dumps(rec), flush=True)
