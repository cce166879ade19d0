#!/usr/bin/env python3
"""Hash-to-G2 (after SHA-256) rate against the number of messages; every size is checked
against the first 256 results of the smallest run (the same messages repeat)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))


def main():
    import torch
    from bls_py import _native, hostmath as H, util
    dev = torch.device("cuda", 0)
    eng = _native.Engine(0)
    uniq = b"".join(H.g2_hash_field_elements(hashlib.sha256(b"cfg-h2c-%d" % i).digest(), util.hash512) for i in range(256))
    want0 = H.g2_affine_bytes(H.hash_to_g2_prehashed(hashlib.sha256(b"cfg-h2c-0").digest(), util.hash512))
    ref = None
    for n in [int(a) for a in sys.argv[1:]] or [256, 1024, 4096, 8192, 16384, 32768, 65536, 262144]:
        tin = torch.frombuffer(bytearray(uniq * (n // 256)), dtype=torch.uint8).to(dev)
        tout = torch.zeros(n * 192, dtype=torch.uint8, device=dev)
        fn = lambda: eng.lib.blsgpu_map_to_g2_dev(eng.h, tin.data_ptr(), n, tout.data_ptr(), 0)
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        got = bytes(tout.cpu().numpy())
        ref = ref or got[:192 * 256]
        ok = got[:192] == want0 and ref * (n // 256) == got
        print(json.dumps({"messages": n, "ms": dt * 1e3, "messages_per_s": n / dt, "check": ok}), flush=True)


if __name__ == "__main__":
    main()
