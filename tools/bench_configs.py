#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs that bench.py does not
headline: C3 shard shape (8192 pairs on one GPU), C4 (threshold combine + verify,
10 000 groups, k = 67) and C5 (1 M-point G1 multi-scalar sum).  Prints JSON lines.
Inputs are synthetic but valid curve points (seeded golden pairs, repeated);
correctness is checked against size-independent identities."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
N = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def main():
    import torch
    from bls_py import _native
    dev = torch.device("cuda", 0)
    eng = _native.Engine(0)
    gold = os.path.join(ROOT, "tests", "golden")
    g1 = open(os.path.join(gold, "pairs_seed1_g1.bin"), "rb").read()
    g2 = open(os.path.join(gold, "pairs_seed1_g2.bin"), "rb").read()
    th = json.load(open(os.path.join(gold, "threshold.json")))["67_of_100"]
    which = set(sys.argv[1:]) or {"c3", "c4", "c5", "h2c", "decompress"}

    def up(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    if "c3" in which:
        n = 8192
        reps = (n + 1024) // 1025
        t1, t2 = up((g1 * reps)[:96 * n]), up((g2 * reps)[:192 * n])
        out = torch.zeros(576, dtype=torch.uint8, device=dev)
        eng.reserve(n)
        dt = timed(lambda: eng.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, out.data_ptr(), 0), 10)
        print(json.dumps({"config": "C3 shard: 8192 pairs, one GPU, one final exp", "pairings_per_s": n / dt, "ms": dt * 1e3}))
    if "c4" in which:
        groups, k = 10000, 67
        pts = b"".join(bytes.fromhex(s) for s in th["unit_sigs_affine"])
        lam = b"".join(int(x, 16).to_bytes(32, "big") for x in th["lambdas"])
        tp, ts = up(pts * groups), up(lam * groups)
        tout = torch.zeros(groups * 192, dtype=torch.uint8, device=dev)
        tinf = torch.zeros(groups, dtype=torch.uint8, device=dev)
        lib, h = eng.lib, eng.h
        dt = timed(lambda: lib.blsgpu_g2_msm_dev(h, tp.data_ptr(), ts.data_ptr(), k, groups, tout.data_ptr(), tinf.data_ptr(), 0), 2)
        got = bytes(tout.cpu().numpy())
        ok = got == bytes.fromhex(th["combined_affine"]) * groups
        # verify step: e(-G1, sig) e(pk, H(m)) per group = 2 pairs x 10 000 groups
        sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
        from bls_py import hostmath as H, util
        from bls_py.keys import PublicKey
        pk = PublicKey.from_bytes(bytes.fromhex(th["master_pk"])).value.to_affine()._aff()
        hm = H.hash_to_g2_prehashed(util.hash256(bytes.fromhex(th["msg"])), util.hash512)
        ng1 = H.jac_to_affine(H.F1, H.jac_mul(H.F1, H.aff_to_jac(H.F1, H.G1_GEN), N - 1))
        pg1 = (H.g1_affine_bytes(ng1) + H.g1_affine_bytes(pk)) * groups
        pg2 = (bytes.fromhex(th["combined_affine"]) + H.g2_affine_bytes(hm)) * groups
        v1, v2 = up(pg1), up(pg2)
        vout = torch.zeros(groups * 576, dtype=torch.uint8, device=dev)
        dtv = timed(lambda: lib.blsgpu_pairing_multi_batch_dev(h, v1.data_ptr(), v2.data_ptr(), None, 2, groups, vout.data_ptr(), 0), 2)
        one = (1).to_bytes(48, "big") + bytes(48 * 11)
        okv = bytes(vout.cpu().numpy()) == one * groups
        print(json.dumps({"config": "C4: threshold k=67 combine + verify, 10000 groups", "combine_s": dt, "verify_s": dtv,
                          "groups_per_s": groups / (dt + dtv), "combine_ok": ok, "verify_all_true": okv}))
    if "c5" in which:
        n = 1 << 20
        base = g1[:96 * 1024]
        sc = b"".join(hashlib.sha256(b"blsgpu/t" + i.to_bytes(4, "big")).digest() for i in range(1024))
        sc = b"".join((int.from_bytes(sc[32 * i:32 * (i + 1)], "big") % N).to_bytes(32, "big") for i in range(1024))
        tp, ts = up(base * (n // 1024)), up(sc * (n // 1024))
        tout = torch.zeros(96, dtype=torch.uint8, device=dev)
        tinf = torch.zeros(1, dtype=torch.uint8, device=dev)
        lib, h = eng.lib, eng.h
        dt = timed(lambda: lib.blsgpu_g1_msm_dev(h, tp.data_ptr(), ts.data_ptr(), n, 1, tout.data_ptr(), tinf.data_ptr(), 0), 1)
        got = bytes(tout.cpu().numpy())
        # identity: 1024 copies of a 1024-point sum = 1024 * (that sum)
        small, _ = eng.g1_msm(base, sc, 1024)
        want, _ = eng.g1_msm(small, [n // 1024], 1)
        print(json.dumps({"config": "C5: 1M-point G1 multi-scalar sum", "seconds": dt, "points_per_s": n / dt, "check": got == want}))
    if "h2c" in which:
        from bls_py import hostmath as H, util
        n = 16384
        uniq = [H.g2_hash_field_elements(hashlib.sha256(b"cfg-h2c-%d" % i).digest(), util.hash512) for i in range(256)]
        tin = up(b"".join(uniq) * (n // 256))
        tout = torch.zeros(n * 192, dtype=torch.uint8, device=dev)
        lib, h = eng.lib, eng.h
        dt = timed(lambda: lib.blsgpu_map_to_g2_dev(h, tin.data_ptr(), n, tout.data_ptr(), 0), 3)
        got = bytes(tout.cpu().numpy())
        t0 = time.perf_counter()
        want0 = H.g2_affine_bytes(H.hash_to_g2_prehashed(hashlib.sha256(b"cfg-h2c-0").digest(), util.hash512))
        host_dt = time.perf_counter() - t0
        ok = got[:192] == want0 and got[:192 * 256] * (n // 256) == got
        print(json.dumps({"config": "hash to G2 (after SHA-256): 16384 messages", "seconds": dt, "messages_per_s": n / dt,
                          "host_python_one_message_s": host_dt, "check": ok}))
    if "decompress" in which:
        from bls_py import hostmath as H
        lib, h = eng.lib, eng.h
        for deg, src, insz in ((1, g1, 48), (2, g2, 96)):
            pts = [src[2 * insz * i:2 * insz * (i + 1)] for i in range(1024)]
            enc = b"".join((H.g1_compress(H.g1_from_abi(p)) if deg == 1 else H.g2_compress(H.g2_from_abi(p))) for p in pts)
            n = 1 << 16
            tin = up(enc * (n // 1024))
            tout = torch.zeros(n * 2 * insz, dtype=torch.uint8, device=dev)
            tok = torch.zeros(n, dtype=torch.uint8, device=dev)
            fn = lib.blsgpu_g1_decompress_dev if deg == 1 else lib.blsgpu_g2_decompress_dev
            dt = timed(lambda: fn(h, tin.data_ptr(), n, tout.data_ptr(), tok.data_ptr(), 0), 3)
            ok = bytes(tout.cpu().numpy()) == src[:2 * insz * 1024] * (n // 1024) and bool(tok.all())
            print(json.dumps({"config": "G%d decompression, %d points" % (deg, n), "seconds": dt, "points_per_s": n / dt, "check": ok}))


if __name__ == "__main__":
    main()
