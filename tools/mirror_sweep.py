#!/usr/bin/env python3
"""Wall times of the host mirror's (bls_py) user-visible operations at a few sizes on the GPU box: key generation, signing,
aggregation (simple / secure), verification, (de)serialisation.  What Python spends beside the engine calls shows here.
usage: mirror_sweep.py [n]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
from bls_py.bls import BLS
from bls_py.keys import PrivateKey, PublicKey
from bls_py.signature import Signature

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
order = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def timed(what, fn, reps=2):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    dt = (time.perf_counter() - t0) / reps
    print("%-70s %9.3f ms" % (what, dt * 1e3), flush=True)
    return r


sks = [PrivateKey(int.from_bytes(hashlib.sha256(b"mirror" + i.to_bytes(4, "big")).digest(), "big") % (order - 1) + 1) for i in range(n)]
msgs = [i.to_bytes(4, "big") for i in range(n)]
PrivateKey.sign_batch(sks[:2], msgs[:2])
pks = timed("get_public_key x %d" % n, lambda: [s.get_public_key() for s in sks], 1)
sigs = timed("sign_batch of %d (distinct messages)" % n, lambda: PrivateKey.sign_batch(sks, msgs))
one = timed("sign (one message)", lambda: sks[0].sign(msgs[0]))
agg = timed("aggregate_sigs of %d (distinct messages)" % n, lambda: BLS.aggregate_sigs(sigs))
timed("verify of that aggregate", lambda: BLS.verify(agg))
timed("verify of ONE signature", lambda: BLS.verify(one))
same = timed("sign_batch of %d (ONE message)" % n, lambda: PrivateKey.sign_batch(sks, [b"same"] * n))
agg2 = timed("aggregate_sigs of %d on one message (secure exponents)" % n, lambda: BLS.aggregate_sigs(same))
timed("verify of that aggregate", lambda: BLS.verify(agg2))
timed("aggregate_pub_keys of %d, secure" % n, lambda: BLS.aggregate_pub_keys(pks, True))
timed("aggregate_pub_keys of %d, plain" % n, lambda: BLS.aggregate_pub_keys(pks, False))
ser = timed("serialize %d signatures" % n, lambda: [s.serialize() for s in sigs])
timed("Signature.from_bytes_batch of %d" % n, lambda: Signature.from_bytes_batch(ser))
pser = timed("serialize %d public keys" % n, lambda: [p.serialize() for p in pks])
timed("PublicKey.from_bytes_batch of %d" % n, lambda: PublicKey.from_bytes_batch(pser))
