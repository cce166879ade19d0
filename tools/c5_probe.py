#!/usr/bin/env python3
"""Times one 2^20-point G1 sum (device buffers, HIP events around the call) for the environment's
BLSGPU_MSM_* settings; prints ms per call.  GPU box only; experiment helper, not part of the tests."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))


def main():
    import hashlib
    import torch
    from bls_py import _native
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    e = _native.Engine(0)
    gen = bytes.fromhex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
                        "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")
    a = [int.from_bytes(hashlib.sha256(b"probe/a" + i.to_bytes(4, "big")).digest(), "big") for i in range(n)]
    pts = b""
    step = 1 << 18
    for lo in range(0, n, step):
        m = min(step, n - lo)
        pts += e.g1_msm(gen * m, a[lo:lo + m], 1, m)[0]
    t = b"".join(hashlib.sha256(b"probe/t" + i.to_bytes(4, "big")).digest() for i in range(n))
    dev = torch.device("cuda", 0)
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
    ds = torch.frombuffer(bytearray(t), dtype=torch.uint8).to(dev)
    out = torch.zeros(96, dtype=torch.uint8, device=dev)
    inf = torch.zeros(1, dtype=torch.uint8, device=dev)
    def call():
        rc = e.lib.blsgpu_g1_msm_dev(e.h, dp.data_ptr(), ds.data_ptr(), n, 1, out.data_ptr(), inf.data_ptr(), 0)
        assert rc == 0, rc

    for _ in range(2):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    print("%s  %.3f ms per %d-point sum  digest %s" % (" ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("BLSGPU_")),
                                                    (time.perf_counter() - t0) / reps * 1e3, n, hashlib.sha256(bytes(out.cpu().numpy())).hexdigest()[:16]))


if __name__ == "__main__":
    main()
