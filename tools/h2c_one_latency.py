"""Latency of hashing ONE (or a few) message hashes to G2 (blsgpu_hash_to_g2_dev) under the engine's selection knobs --
which forms of the encodings / cofactor clearing a single-signature verification should take.  Prints JSON lines."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native
dev = torch.device("cuda", 0)
CONFIGS = {
    "default": {},
    "clearing_on_lane_quads": {"BLSGPU_H2C_REG_THRESHOLD": "1"},
    "clearing_on_lane_pairs": {"BLSGPU_H2C_REG_THRESHOLD": "1", "BLSGPU_H2C_QUAD_MAX": "0"},
    "quads+lane_encodings": {"BLSGPU_H2C_REG_THRESHOLD": "1", "BLSGPU_H2C_LANE_THRESHOLD": "1"},
    "quads+lane_encodings+division_step_symbols": {"BLSGPU_H2C_REG_THRESHOLD": "1", "BLSGPU_H2C_LANE_THRESHOLD": "1", "BLSGPU_H2C_JACOBI_THRESHOLD": "1"},
}
ref = None
for n in (1, 2, 16):
    mh = b"".join(hashlib.sha256(b"m%d" % i).digest() for i in range(n))
    d_in = torch.frombuffer(bytearray(mh), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(192 * n, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream(device=dev)
    outs = {}
    for name, env in CONFIGS.items():
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        e = _native.Engine(0)
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        f = lambda: e.lib.blsgpu_hash_to_g2_dev(e.h, d_in.data_ptr(), n, d_out.data_ptr(), stream.cuda_stream)
        f(); stream.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); f(); b.record(stream); stream.synchronize()
            best = min(best, a.elapsed_time(b))
        outs[name] = bytes(d_out.cpu().numpy())
        print(json.dumps({"messages": n, "config": name, "ms": round(best, 4), "same_as_default": outs[name] == outs["default"]}), flush=True)
