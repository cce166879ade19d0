"""One BLS.verify-shaped call (hash one message to G2 + two-pair multi-pairing) repeated a few times, for
`rocprofv3 --kernel-trace --stats`: which kernels a single-signature verification waits for.  usage: python tools/single_verify_trace.py [reps]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
import torch
from bls_py import _native, hostmath as H
from bls_py.keys import PrivateKey
N = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
e = _native.Engine(0)
sk = PrivateKey(int.from_bytes(hashlib.sha256(b"single").digest(), "big") % (N - 1) + 1)
msg = b"one message"
sig = sk.sign(msg)
mh = hashlib.sha256(msg).digest()
pk = H.g1_affine_bytes(sig.aggregation_info.public_keys[0].value.to_affine()._aff())
neg_g1 = H.g1_affine_bytes(H.jac_to_affine(H.F1, H.jac_mul(H.F1, H.aff_to_jac(H.F1, H.G1_GEN), N - 1)))
sig_b = H.g2_affine_bytes(sig.value.to_affine()._aff())
one = (1).to_bytes(48, "big") + bytes(48 * 11)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    assert e.verify_pipeline(neg_g1, sig_b, mh, 1, keys_affine=pk) == one
print("ok")
