"""Device-resident aggregate-verify pipeline (the GPU part of B x BLS.verify with 1024 distinct
messages each): hash the message hashes to G2, place them behind the aggregate signature, run the
batched multi-pairing, compare with one.  Public keys are taken as already folded (one key per
message, exponent 1: the simple-aggregation case of bls.py:177-192).  Prints one JSON line."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native, hostmath as H
from bls_py.bls import BLS
from bls_py.keys import PrivateKey
from bls_py.bls12381 import n as ORDER

dev = torch.device("cuda", 0)
e = _native.Engine(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 1024
# one real aggregate of 1024 signatures, reused for the B verifications with rotated message order
sks = [PrivateKey(int.from_bytes(hashlib.sha256(b"pipe%d" % i).digest(), "big") % (ORDER - 1) + 1) for i in range(n)]
msgs = [i.to_bytes(4, "big") for i in range(n)]
sigs = PrivateKey.sign_batch(sks, msgs)
agg = BLS.aggregate_sigs(sigs)
mh = [hashlib.sha256(m).digest() for m in msgs]
pks = [H.g1_affine_bytes(s.aggregation_info.public_keys[0].value.to_affine()._aff()) for s in sigs]
neg_g1 = H.g1_affine_bytes(H.jac_to_affine(H.F1, H.jac_mul(H.F1, H.aff_to_jac(H.F1, H.G1_GEN), ORDER - 1)))
sig_b = H.g2_affine_bytes(agg.value.to_affine()._aff())
g1_one, mh_all = bytearray(), bytearray()
for v in range(B):
    order = [(i + 31 * v) % n for i in range(n)]
    g1_one += neg_g1 + b"".join(pks[i] for i in order)
    mh_all += b"".join(mh[i] for i in order)
up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
d_g1, d_mh = up(g1_one), up(mh_all)
d_g2 = torch.zeros(B * (n + 1) * 192, dtype=torch.uint8, device=dev).view(B, n + 1, 192)
d_g2[:, 0, :] = up(sig_b)
d_h = torch.zeros(B * n * 192, dtype=torch.uint8, device=dev)
d_out = torch.zeros(B * 576, dtype=torch.uint8, device=dev)
e.reserve((n + 4) * B)
e.set_mp_threshold(0)

def run():
    e.lib.blsgpu_hash_to_g2_dev(e.h, d_mh.data_ptr(), B * n, d_h.data_ptr(), 0)
    d_g2[:, 1:, :] = d_h.view(B, n, 192)
    e.pairing_multi_batch_dev(d_g1.data_ptr(), d_g2.data_ptr(), n + 1, B, d_out.data_ptr(), 0)

run(); torch.cuda.synchronize()
one = (1).to_bytes(48, "big") + bytes(48 * 11)
ok = bytes(d_out.cpu().numpy()) == one * B
t = time.perf_counter()
for _ in range(5):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(json.dumps({"config": "device pipeline: %d aggregate verifications x 1024 messages (hash to G2 + 1025-pair multi-pairing each)" % B,
                  "ms": dt * 1e3, "verifications_per_s": B / dt, "signatures_per_s": B * n / dt, "all_verify_true": ok}))
