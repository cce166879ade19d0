"""Hash-to-G2 latency against the number of messages with the cofactor clearing one message per wavefront (k_h2c_clear_wide) and
without it (the wavefront VM's k_h2c_clear / the register forms): where BLSGPU_H2C_WIDE_MAX belongs.  Prints JSON lines."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
dev = torch.device("cuda", 0)
engs = {}
for name, v in (("wide", str(1 << 40)), ("other", "0")):
    os.environ["BLSGPU_H2C_WIDE_MAX"] = v
    engs[name] = _native.Engine(0)
os.environ.pop("BLSGPU_H2C_WIDE_MAX")
stream = torch.cuda.Stream(device=dev)
for n in [int(x) for x in sys.argv[1:]] or [1, 16, 128, 256, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192]:
    mh = b"".join(hashlib.sha256(b"m%d" % i).digest() for i in range(n))
    d_in = torch.frombuffer(bytearray(mh), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(192 * n, dtype=torch.uint8, device=dev)
    rec, outs = {"messages": n}, {}
    for name, e in engs.items():
        f = lambda: e.lib.blsgpu_hash_to_g2_dev(e.h, d_in.data_ptr(), n, d_out.data_ptr(), stream.cuda_stream)
        f(); stream.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); f(); b.record(stream); stream.synchronize()
            best = min(best, a.elapsed_time(b))
        outs[name] = bytes(d_out.cpu().numpy())
        rec[name + "_ms"] = round(best, 4)
    rec["same"] = outs["wide"] == outs["other"]
    print(json.dumps(rec), flush=True)
