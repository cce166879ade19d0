"""The reference's scheme-level scenarios with the pairing on the GPU (default
provider = HIP engine through the C ABI).  Needs an MI355X."""
import hashlib
import json
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def hip_provider(engine):
    from bls_py import backend
    backend.use(None)          # default HIP provider
    assert type(backend.get()).__name__ == "HipProvider"
    yield


def test_verify_4_signatures_C1(golden):
    from bls_py.bls import BLS
    from bls_py.keys import PrivateKey
    v = golden("verify4.json")
    sks = [PrivateKey.from_seed(bytes.fromhex(s)) for s in v["seeds"]]
    sigs = [sk.sign(bytes.fromhex(m)) for sk, m in zip(sks, v["msgs"])]
    agg = BLS.aggregate_sigs(sigs)
    assert agg.serialize().hex() == v["agg_sig"]
    assert BLS.verify(agg) is True
    bad = BLS.aggregate_sigs(sigs[:3])
    bad.set_aggregation_info(agg.aggregation_info)
    assert BLS.verify(bad) is False


def test_reference_vectors(golden):
    from bls_py.aggregation_info import AggregationInfo
    from bls_py.bls import BLS
    from bls_py.keys import PrivateKey, PublicKey
    from bls_py.signature import Signature
    v = golden("scheme.json")["vectors"]
    sk1, sk2 = [PrivateKey.from_seed(bytes.fromhex(x)) for x in v["seeds"]]
    m = bytes.fromhex(v["msg"])
    sig1, sig2 = sk1.sign(m), sk2.sign(m)
    agg = BLS.aggregate_sigs([sig1, sig2])
    assert BLS.verify(sig1) and BLS.verify(agg)
    agg_pk = BLS.aggregate_pub_keys([sk1.get_public_key(), sk2.get_public_key()], True)
    agg.set_aggregation_info(AggregationInfo.from_msg(agg_pk, m))
    assert BLS.verify(agg)
    sig1.set_aggregation_info(sig2.aggregation_info)
    assert not BLS.verify(sig1)
    # (de)serialise, re-attach the info, verify (tests.py:238-243)
    sig = Signature.from_bytes(sk1.sign(b"round trip").serialize())
    sig.set_aggregation_info(AggregationInfo.from_msg(PublicKey.from_bytes(sk1.get_public_key().serialize()), b"round trip"))
    assert BLS.verify(sig)


def test_sign_aggregate_verify_batch_C2():
    """BASELINE configs[1] end to end at a reduced count (host hashing is pure
    Python): n signatures -> aggregate -> verify = n + 1 pairings; one flipped
    message must fail."""
    import hashlib
    from bls_py.bls import BLS
    from bls_py.keys import PrivateKey
    n = 96
    sks = [PrivateKey(int.from_bytes(hashlib.sha256(b"blsgpu/a" + (1).to_bytes(4, "big") + i.to_bytes(4, "big")).digest(), "big")
                      % (0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001 - 1) + 1) for i in range(n)]
    sigs = [sk.sign(i.to_bytes(4, "big")) for i, sk in enumerate(sks)]
    agg = BLS.aggregate_sigs(sigs)
    assert len(agg.aggregation_info.public_keys) == n
    assert BLS.verify(agg) is True
    sigs[7] = sks[7].sign(b"\xff\xff\xff\xff")
    forged = BLS.aggregate_sigs_simple(sigs)
    forged.set_aggregation_info(agg.aggregation_info)
    assert BLS.verify(forged) is False


def test_sign_aggregate_verify_full_C2():
    """BASELINE configs[1] at full size: 1024 (sk, msg) signed with the batched GPU
    helper (checked against the per-key host path on a sample), aggregated, verified =
    1025 pairings + 1024 hashes to G2 + 1024 key foldings on the GPU; a flipped message fails."""
    import hashlib
    from bls_py.bls import BLS
    from bls_py.keys import PrivateKey
    n = 1024
    order = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    sks = [PrivateKey(int.from_bytes(hashlib.sha256(b"blsgpu/a" + (1).to_bytes(4, "big") + i.to_bytes(4, "big")).digest(), "big")
                      % (order - 1) + 1) for i in range(n)]
    msgs = [i.to_bytes(4, "big") for i in range(n)]
    sigs = PrivateKey.sign_batch(sks, msgs)
    for i in (0, 511, 1023):
        one = sks[i].sign(msgs[i])
        assert one.serialize() == sigs[i].serialize()
        assert one.aggregation_info.public_keys[0].serialize() == sigs[i].aggregation_info.public_keys[0].serialize()
    agg = BLS.aggregate_sigs(sigs)
    assert len(agg.aggregation_info.public_keys) == n
    assert BLS.verify(agg) is True
    sigs[7] = sks[7].sign(b"\xff\xff\xff\xff")
    forged = BLS.aggregate_sigs_simple(sigs)
    forged.set_aggregation_info(agg.aggregation_info)
    assert BLS.verify(forged) is False


def test_threshold_combine_and_verify_C4(golden):
    from bls_py.aggregation_info import AggregationInfo
    from bls_py.bls import BLS
    from bls_py.keys import PublicKey
    from bls_py.signature import Signature
    from bls_py.threshold import Threshold
    rec = golden("threshold.json")["67_of_100"]
    unit = [Signature.from_bytes(bytes.fromhex(s)) for s in rec["unit_sigs"]]
    comb = Threshold.aggregate_unit_sigs(unit, rec["players"], rec["T"])
    assert comb.serialize().hex() == rec["combined"]
    comb.set_aggregation_info(AggregationInfo.from_msg(PublicKey.from_bytes(bytes.fromhex(rec["master_pk"])),
                                                       bytes.fromhex(rec["msg"])))
    assert BLS.verify(comb) is True


def test_verify_batch_on_gpu():
    """BLS.verify_batch through the HIP engine: 6 aggregates of 40 signatures each (one forged)
    plus single signatures -> the booleans of BLS.verify one by one."""
    import hashlib
    from bls_py.bls import BLS
    from bls_py.keys import PrivateKey
    order = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    sks = [PrivateKey(int.from_bytes(hashlib.sha256(b"vb%d" % i).digest(), "big") % (order - 1) + 1) for i in range(40)]
    aggs = []
    for a in range(6):
        sigs = PrivateKey.sign_batch(sks, [b"agg%d-%d" % (a, i) for i in range(40)])
        if a == 4:
            sigs[7] = sks[7].sign(b"forged")
            forged = BLS.aggregate_sigs_simple(sigs)
            forged.set_aggregation_info(BLS.aggregate_sigs(PrivateKey.sign_batch(sks, [b"agg4-%d" % i for i in range(40)])).aggregation_info)
            aggs.append(forged)
        else:
            aggs.append(BLS.aggregate_sigs(sigs))
    singles = [sks[0].sign(b"one"), sks[1].sign(b"two")]
    batch = aggs + singles
    got = BLS.verify_batch(batch)
    assert got == [True, True, True, True, False, True, True, True]
    assert got == [BLS.verify(s) for s in batch]


def test_secure_aggregation_and_division_on_gpu(golden):
    """SURVEY 8f rank 4 on the HIP provider: the reference's hex vectors for the colliding-message
    aggregate (tests.py:147), the nested 6-signature aggregate (:172), divide_by (:177, :198) and its
    error paths (:179-189) -- the secure path's G2 scalar multiples, the divisions' G2 sums and every
    verify run on the GPU."""
    from bls_py.bls import BLS
    from bls_py.keys import PrivateKey
    s = golden("scheme.json")
    v = s["vectors"]
    sk1, sk2 = [PrivateKey.from_seed(bytes.fromhex(x)) for x in v["seeds"]]
    m = bytes.fromhex(v["msg"])
    agg = BLS.aggregate_sigs([sk1.sign(m), sk2.sign(m)])            # same message: secure path
    assert agg.serialize().hex() == v["agg_sig"] and BLS.verify(agg)
    assert BLS.aggregate_pub_keys([sk1.get_public_key(), sk2.get_public_key()], True).serialize().hex() == v["agg_pk_secure"]
    sigs = [(sk1, sk2)[k].sign(bytes.fromhex(mm)) for k, mm in zip(v["agg2_signers"], v["agg2_msgs"])]
    agg2 = BLS.aggregate_sigs(sigs)
    assert agg2.serialize().hex() == v["agg2_sig"] and BLS.verify(agg2)
    n = s["nested"]
    m1, m2, m3, m4 = [bytes.fromhex(x) for x in n["msgs"]]
    s1, s2, s3, s4, s5, s6 = sk1.sign(m1), sk2.sign(m2), sk2.sign(m1), sk1.sign(m3), sk1.sign(m1), sk1.sign(m4)
    sL = BLS.aggregate_sigs([s1, s2])
    sR = BLS.aggregate_sigs([s3, s4, s5])
    sF = BLS.aggregate_sigs([sL, sR, s6])
    assert [sL.serialize().hex(), sR.serialize().hex(), sF.serialize().hex()] == [n["sig_L"], n["sig_R"], n["sig_final"]]
    assert BLS.verify(sL) and BLS.verify(sR) and BLS.verify(sF)
    quo = sF.divide_by([s2, s5, s6])
    assert quo.serialize().hex() == n["quotient"] and BLS.verify(quo) and BLS.verify(sF)
    assert quo.divide_by([]) == quo
    with pytest.raises(Exception):
        quo.divide_by([s6])                                       # not a subset (tests.py:179-181)
    sF.divide_by([s1])                                            # a unique message may be divided
    with pytest.raises(Exception):
        sF.divide_by([sL])                                        # s3, s5 share m1 with s1 (tests.py:183-189)
    s7, s8 = sk2.sign(m3), sk2.sign(m4)
    sR2 = BLS.aggregate_sigs([s7, s8])
    sF2 = BLS.aggregate_sigs([sF, sR2])
    quo2 = sF2.divide_by([sR2])
    assert BLS.verify(quo2) and quo2.serialize().hex() == n["quotient2"]


def test_verify_pipeline_equals_the_tuple_path(golden):
    """BLS.verify's device-resident pipeline (HipProvider.verify_pipeline: hash-to-G2, key sums and the multi-pairing without a
    host round trip) against the object / tuple path of bls.py:153-201 through the same engine: plain aggregates (one key per
    message, exponent 1), a secure aggregate (several keys and large exponents per message), tampered infos, an infinity
    signature (which keeps the tuple path)."""
    from bls_py import backend
    from bls_py.bls import BLS
    from bls_py.keys import PrivateKey
    from bls_py.signature import Signature
    from bls_py.ec import JacobianPoint
    from bls_py import hostmath as H

    class TupleOnly:
        """the HIP provider without the pipeline entry: BLS.verify falls back to ate_pairing_multi"""
        def __init__(self, p):
            self._p = p

        def __getattr__(self, k):
            if k == "verify_pipeline":
                raise AttributeError(k)
            return getattr(self._p, k)
    hip = backend.get()
    sks = [PrivateKey.from_seed(bytes([i + 1] * 5)) for i in range(6)]
    plain = BLS.aggregate_sigs([sk.sign(bytes([i, 100 + i])) for i, sk in enumerate(sks)])
    same_msg = BLS.aggregate_sigs([sk.sign(b"one message") for sk in sks[:4]])                      # secure: exponents from hash_pks
    nested = BLS.aggregate_sigs([same_msg, sks[4].sign(b"one message"), sks[5].sign(b"another")])
    bad = BLS.aggregate_sigs([sk.sign(bytes([i, 100 + i])) for i, sk in enumerate(sks[:5])])
    bad.set_aggregation_info(plain.aggregation_info)
    inf_sig = Signature.from_g2(JacobianPoint._from(H.F2, None), plain.aggregation_info)
    cases = [plain, same_msg, nested, bad, inf_sig]
    fast = [BLS.verify(s) for s in cases]
    backend.use(TupleOnly(hip))
    try:
        slow = [BLS.verify(s) for s in cases]
    finally:
        backend.use(None)
    assert fast == slow == [True, True, True, False, False]


def test_verify_needs_nothing_but_the_c_abi(golden):
    """blsgpu_verify_pipeline (include/blsgpu.h; bls.py:153-201): BLS.verify of the reference's 4-signature scenario
    (tests/golden/verify4.json: True; the tampered aggregate: False) in a process where torch is never imported -- the
    host side is ctypes over libblsgpu.so and nothing else -- and the entry itself against the multi-pairing of the
    same points assembled by hand."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import json, os, sys
sys.path.insert(0, os.path.join(%r, "python-bls_amd"))
from bls_py.bls import BLS
from bls_py.keys import PrivateKey
v = json.load(open(os.path.join(%r, "tests", "golden", "verify4.json")))
sks = [PrivateKey.from_seed(bytes([i + 1] * 5)) for i in range(4)]
sigs = [sk.sign(bytes([i, 100 + i])) for i, sk in enumerate(sks)]
agg = BLS.aggregate_sigs(sigs)
assert agg.serialize().hex() == v["agg_sig"], "aggregate differs from the reference's"
ok = BLS.verify(agg)
bad = BLS.aggregate_sigs(sigs[:3]); bad.set_aggregation_info(agg.aggregation_info)
print(json.dumps({"verify": ok, "tampered": BLS.verify(bad), "torch_loaded": "torch" in sys.modules}))
''' % (ROOT, ROOT)
    env = dict(os.environ, BLSGPU_NO_TORCH="1", PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec == {"verify": True, "tampered": False, "torch_loaded": False}
    # the entry point itself: keys given / key sums with exponent 1 and padding, against blsgpu_pairing_multi on the same points
    from bls_py import _native, hostmath as H
    from bls_py.util import hash512
    e = _native.engine(0)
    n = 5
    hashes = [hashlib.sha256(b"pipeline-%d" % i).digest() for i in range(n)]
    keys = [H.g1_affine_bytes(H.jac_to_affine(H.F1, H.jac_mul(H.F1, H.aff_to_jac(H.F1, H.G1_GEN), 3 + i))) for i in range(n)]
    sig = H.g2_affine_bytes(H.jac_to_affine(H.F2, H.jac_mul(H.F2, H.aff_to_jac(H.F2, H.G2_GEN), 77)))
    neg = H.g1_affine_bytes(H.jac_to_affine(H.F1, H.jac_neg(H.F1, H.aff_to_jac(H.F1, H.G1_GEN))))
    qs = e.hash_to_g2(b"".join(hashes))
    want = e.pairing_multi(neg + b"".join(keys), sig + qs, n + 1)
    assert e.verify_pipeline(neg, sig, b"".join(hashes), n, keys_affine=b"".join(keys)) == want
    pts = b"".join(k + bytes(96) for k in keys)                                  # k = 2: the key and a padding slot
    sc = b"".join((1).to_bytes(32, "big") + bytes(32) for _ in keys)
    assert e.verify_pipeline(neg, sig, b"".join(hashes), n, key_pts=pts, key_scalars=sc, k=2) == want
    assert e.verify_pipeline(neg, sig, b"", 0) == e.pairing_multi(neg, sig, 1)
