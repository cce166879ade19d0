#!/usr/bin/env python3
"""Latency of ONE blsgpu_pairing_multi_dev call against the pair count, through k_miller (one pair per
wavefront) and through k_miller_mp (three pairs per wavefront): where the default mp_threshold belongs."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))


def main():
    import torch
    from bls_py import _native
    dev = torch.device("cuda", 0)
    eng = _native.Engine(0)
    gold = os.path.join(ROOT, "tests", "golden")
    g1 = open(os.path.join(gold, "pairs_seed1_g1.bin"), "rb").read()
    g2 = open(os.path.join(gold, "pairs_seed1_g2.bin"), "rb").read()
    out = torch.zeros(576, dtype=torch.uint8, device=dev)
    for n in (1025, 2048, 3072, 4096, 5120, 6144, 7168, 8192, 9216, 10240, 12288, 16384, 24576, 32768, 65536):
        reps = (n + 1024) // 1025
        t1 = torch.frombuffer(bytearray((g1 * reps)[:96 * n]), dtype=torch.uint8).to(dev)
        t2 = torch.frombuffer(bytearray((g2 * reps)[:192 * n]), dtype=torch.uint8).to(dev)
        eng.reserve(n)
        row = {"pairs": n}
        for name, thr, thr3 in (("k_miller", 1 << 40, 0), ("k_miller_mp2", 0, 1 << 40), ("k_miller_mp3", 0, 0)):
            eng.set_mp_threshold(thr)
            eng.set_mp3_threshold(thr3)
            for _ in range(2):
                eng.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, out.data_ptr(), 0)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                eng.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, out.data_ptr(), 0)
            torch.cuda.synchronize()
            row[name + "_ms"] = (time.perf_counter() - t) / 5 * 1e3
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
