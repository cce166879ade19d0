import json, os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
e = _native.Engine(0)
dev = torch.device("cuda", 0)
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()
out = torch.zeros(576, dtype=torch.uint8, device=dev)
part = torch.zeros(144, dtype=torch.int32, device=dev)
for n in (2048, 4096, 6144, 8192, 12288, 16384, 20480, 24576, 32768):
    reps = (n + 1024) // 1025
    t1 = torch.frombuffer(bytearray((g1 * reps)[:96 * n]), dtype=torch.uint8).to(dev)
    t2 = torch.frombuffer(bytearray((g2 * reps)[:192 * n]), dtype=torch.uint8).to(dev)
    e.reserve(n)
    rec = {"pairs": n}
    res = {}
    for path, thr in (("vm", None), ("ls", 1)):
        e.set_ls_threshold(thr, 64 if thr is None else 1)
        for name, f in (("pairing", lambda: e.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, out.data_ptr(), 0)),
                        ("product", lambda: e.miller_product_batch_dev(t1.data_ptr(), t2.data_ptr(), n, 1, part.data_ptr()))):
            f(); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(10): f()
            torch.cuda.synchronize()
            rec["%s_%s_ms" % (name, path)] = round((time.perf_counter() - t) / 10 * 1e3, 3)
        res[path] = bytes(out.cpu().numpy())
    rec["same"] = res["vm"] == res["ls"]
    print(json.dumps(rec), flush=True)
