"""Host-side scheme logic (python-bls_amd/bls_py: keys, signatures, aggregation
tree, threshold, hashing, (de)serialisation) against vectors produced by the
reference.  CPU only: the pairing provider is replaced by the CPU oracle -- the
product itself never does that (bls_py/backend.py)."""
import pytest

from conftest import cat


@pytest.fixture(scope="module", autouse=True)
def oracle_provider(oracle):
    from bls_py import backend

    class P:
        def pairing_multi(self, g1, g2, n, inf=None):
            return oracle.pairing_multi(g1, g2, n, threads=8, inf=inf)

        def miller_loop_batch(self, g1, g2, n, inf=None):
            return b"".join(oracle.miller_loop(g1[96 * i:96 * (i + 1)], g2[192 * i:192 * (i + 1)], bool(inf and inf[2 * i + 1]))
                            for i in range(n))

        def line_eval_batch(self, r, q, p, n):
            return b"".join(oracle.line_eval(r[192 * i:192 * (i + 1)], None if q is None else q[192 * i:192 * (i + 1)],
                                             p[96 * i:96 * (i + 1)]) for i in range(n))

        def final_exp(self, x):
            return oracle.final_exp(x)

        def _msm(self, fn, psz, pts, scalars, k, groups):
            out, inf = b"", []
            for g in range(groups):
                sc = None if scalars is None else scalars[k * g:k * (g + 1)]
                o, i = fn(pts[psz * k * g:psz * k * (g + 1)], sc, k)
                out += o
                inf.append(i)
            return out, inf

        def g1_msm(self, pts, scalars, k, groups=1):
            return self._msm(oracle.g1_msm, 96, pts, scalars, k, groups)

        def g2_msm(self, pts, scalars, k, groups=1):
            return self._msm(oracle.g2_msm, 192, pts, scalars, k, groups)

        def pairing_multi_batch(self, g1, g2, gsz, groups, inf=None):
            return b"".join(oracle.pairing_multi(g1[96 * gsz * g:96 * gsz * (g + 1)], g2[192 * gsz * g:192 * gsz * (g + 1)], gsz,
                                                 inf=None if inf is None else inf[2 * gsz * g:2 * gsz * (g + 1)])
                            for g in range(groups))

        def g1_decompress(self, data):
            from bls_py import hostmath as H
            out, ok = b"", []
            for i in range(len(data) // 48):
                try:
                    out += H.g1_affine_bytes(H.g1_decompress(data[48 * i:48 * (i + 1)]))
                    ok.append(True)
                except ValueError:
                    out += bytes(96)
                    ok.append(False)
            return out, ok

        def g2_decompress(self, data):
            from bls_py import hostmath as H
            out, ok = b"", []
            for i in range(len(data) // 96):
                try:
                    out += H.g2_affine_bytes(H.g2_decompress(data[96 * i:96 * (i + 1)]))
                    ok.append(True)
                except ValueError:
                    out += bytes(192)
                    ok.append(False)
            return out, ok

        def hash_to_g2(self, msg_hashes):
            from bls_py import hostmath as H, util
            return b"".join(H.g2_affine_bytes(H.hash_to_g2_prehashed(msg_hashes[32 * i:32 * (i + 1)], util.hash512))
                            for i in range(len(msg_hashes) // 32))

        def map_to_g2(self, t):
            # stand-in for the GPU map: the host integer implementation, pinned to the
            # reference by test_hash_to_curve_and_sw_encode below
            from bls_py import hostmath as H
            out = b""
            for i in range(len(t) // 192):
                v = [int.from_bytes(t[192 * i + 48 * j:192 * i + 48 * (j + 1)], "big") for j in range(4)]
                S = [H.aff_to_jac(H.F2, H.sw_encode(H.F2, (v[2 * j], v[2 * j + 1]))) for j in range(2)]
                out += H.g2_affine_bytes(H.clear_cofactor_g2(H.jac_add(H.F2, S[0], S[1])))
            return out
    backend.use(P())
    yield
    backend.use(None)


def test_hash_to_curve_and_sw_encode(golden):
    from bls_py import hostmath as H, util
    g = golden("hash_to_curve.json")
    for rec in g["hash_to_g2"]:
        assert util.hash256(bytes.fromhex(rec["msg"])).hex() == rec["msg_hash"]
        p = H.hash_to_g2_prehashed(bytes.fromhex(rec["msg_hash"]), util.hash512)
        assert H.g2_affine_bytes(p).hex() == rec["point"]
    p = H.hash_to_g1_prehashed(util.hash256(b""), util.hash512)
    assert H.g1_compress(p).hex() == g["hash_to_g1_empty"]["ser"]          # tests.py:80
    for rec in g["sw_encode_fq"]:
        assert H.g1_affine_bytes(H.sw_encode(H.F1, int(rec["t"], 16))).hex() == rec["point"]
    for rec in g["sw_encode_fq2"]:
        b = bytes.fromhex(rec["t"])
        t = (int.from_bytes(b[:48], "big"), int.from_bytes(b[48:], "big"))
        assert H.g2_affine_bytes(H.sw_encode(H.F2, t)).hex() == rec["point"]
    assert H.sw_encode(H.F1, 0) is None                                    # tests.py:104


def test_serialisation_round_trips(golden):
    from bls_py.keys import PrivateKey, PublicKey
    from bls_py.signature import Signature
    from bls_py import hostmath as H
    for rec in golden("scheme.json")["serialization"]:
        sk = PrivateKey.from_seed(bytes.fromhex(rec["seed"]))
        assert sk.serialize().hex() == rec["sk"]
        pk = sk.get_public_key()
        assert pk.serialize().hex() == rec["pk"] and pk.size() == 48
        assert PublicKey.from_bytes(bytes.fromhex(rec["pk"])) == pk
        assert H.g1_affine_bytes(PublicKey.from_bytes(bytes.fromhex(rec["pk"])).value.to_affine()._aff()).hex() == rec["pk_affine"]
        sig = sk.sign(bytes.fromhex(rec["msg"]))
        assert sig.serialize().hex() == rec["sig"] and sig.size() == 96
        back = Signature.from_bytes(bytes.fromhex(rec["sig"]))
        assert back == sig
        assert H.g2_affine_bytes(back.value.to_affine()._aff()).hex() == rec["sig_affine"]


def test_scheme_vectors(golden):
    """tests.py:110-147 scenario: keys, signatures, aggregates, verify True/False."""
    from bls_py.aggregation_info import AggregationInfo
    from bls_py.bls import BLS
    from bls_py.keys import PrivateKey
    v = golden("scheme.json")["vectors"]
    sk1, sk2 = [PrivateKey.from_seed(bytes.fromhex(x)) for x in v["seeds"]]
    pk1, pk2 = sk1.get_public_key(), sk2.get_public_key()
    m = bytes.fromhex(v["msg"])
    sig1, sig2 = sk1.sign(m), sk2.sign(m)
    assert [sk1.serialize().hex(), sk2.serialize().hex()] == v["sk"]
    assert [pk1.get_fingerprint(), pk2.get_fingerprint()] == v["fingerprint"]
    assert [sig1.serialize().hex(), sig2.serialize().hex()] == v["sig"]
    agg = BLS.aggregate_sigs([sig1, sig2])
    assert agg.serialize().hex() == v["agg_sig"]
    agg_pk = BLS.aggregate_pub_keys([pk1, pk2], True)
    assert agg_pk.serialize().hex() == v["agg_pk_secure"]
    assert BLS.aggregate_pub_keys([pk1, pk2], False).serialize().hex() == v["agg_pk_simple"]
    agg_sk = BLS.aggregate_priv_keys([sk1, sk2], [pk1, pk2], True)
    assert agg_sk.serialize().hex() == v["agg_sk"]
    assert agg_sk.sign(m).serialize() == agg.serialize()
    assert BLS.verify(sig1) is True and BLS.verify(agg) is True
    agg.set_aggregation_info(AggregationInfo.from_msg(agg_pk, m))
    assert BLS.verify(agg) is True
    sig1.set_aggregation_info(sig2.aggregation_info)
    assert BLS.verify(sig1) is False
    sigs = [(sk1, sk2)[k].sign(bytes.fromhex(mm)) for k, mm in zip(v["agg2_signers"], v["agg2_msgs"])]
    agg2 = BLS.aggregate_sigs(sigs)
    assert agg2.serialize().hex() == v["agg2_sig"] and BLS.verify(agg2)


def test_nested_aggregation_and_division(golden):
    """tests.py:150-198 scenario."""
    from bls_py.bls import BLS
    from bls_py.keys import PrivateKey
    s = golden("scheme.json")
    sk1, sk2 = [PrivateKey.from_seed(bytes.fromhex(x)) for x in s["vectors"]["seeds"]]
    n = s["nested"]
    m1, m2, m3, m4 = [bytes.fromhex(x) for x in n["msgs"]]
    s1, s2, s3, s4, s5, s6 = sk1.sign(m1), sk2.sign(m2), sk2.sign(m1), sk1.sign(m3), sk1.sign(m1), sk1.sign(m4)
    sL = BLS.aggregate_sigs([s1, s2])
    sR = BLS.aggregate_sigs([s3, s4, s5])
    sF = BLS.aggregate_sigs([sL, sR, s6])
    assert [sL.serialize().hex(), sR.serialize().hex(), sF.serialize().hex()] == [n["sig_L"], n["sig_R"], n["sig_final"]]
    info = sF.aggregation_info
    tree = [[h.hex(), pk.serialize().hex(), hex(info.tree[(h, pk)])] for h, pk in zip(info.message_hashes, info.public_keys)]
    assert tree == n["final_tree"]
    assert BLS.verify(sL) and BLS.verify(sR) and BLS.verify(sF)
    quo = sF.divide_by([s2, s5, s6])
    assert quo.serialize().hex() == n["quotient"] and BLS.verify(quo) and BLS.verify(sF)
    assert quo.divide_by([]) == quo
    with pytest.raises(Exception):
        quo.divide_by([s6])               # not a subset any more
    sF.divide_by([s1])                    # fine
    with pytest.raises(Exception):
        sF.divide_by([sL])                # not unique
    s7, s8 = sk2.sign(m3), sk2.sign(m4)         # dividing by an aggregate (tests.py:191-198)
    sR2 = BLS.aggregate_sigs([s7, s8])
    quo2 = BLS.aggregate_sigs([sF, sR2]).divide_by([sR2])
    assert BLS.verify(quo2) and quo2.serialize().hex() == n["quotient2"]


def test_verify4_inputs_match_reference(golden):
    """C1: the exact (Ps, Qs) byte strings BLS.verify sends to the pairing."""
    from bls_py import backend
    from bls_py.bls import BLS
    from bls_py.keys import PrivateKey
    v = golden("verify4.json")
    sks = [PrivateKey.from_seed(bytes.fromhex(s)) for s in v["seeds"]]
    assert [sk.serialize().hex() for sk in sks] == v["sk"]
    assert [sk.get_public_key().serialize().hex() for sk in sks] == v["pk"]
    sigs = [sk.sign(bytes.fromhex(m)) for sk, m in zip(sks, v["msgs"])]
    assert [s.serialize().hex() for s in sigs] == v["sig"]
    agg = BLS.aggregate_sigs(sigs)
    assert agg.serialize().hex() == v["agg_sig"]
    seen = {}
    inner = backend.get()

    class Spy:
        g1_msm = staticmethod(inner.g1_msm)
        g2_msm = staticmethod(inner.g2_msm)
        map_to_g2 = staticmethod(inner.map_to_g2)
        hash_to_g2 = staticmethod(inner.hash_to_g2)
        pairing_multi_batch = staticmethod(inner.pairing_multi_batch)

        def pairing_multi(self, g1, g2, n, inf=None):
            seen["g1"], seen["g2"], seen["n"], seen["inf"] = g1, g2, n, inf
            return inner.pairing_multi(g1, g2, n, inf)
    backend.use(Spy())
    try:
        assert BLS.verify(agg) is True
    finally:
        backend.use(inner)
    assert seen["n"] == 5 and seen["g1"][:96] == bytes.fromhex(v["pairing_g1"][0])
    assert seen["g2"][:192] == bytes.fromhex(v["pairing_g2"][0])
    # the order of the per-message pairs follows dict insertion = sorted (mh, pk)
    assert sorted(seen["g1"][96 * i:96 * (i + 1)].hex() for i in range(1, 5)) == sorted(v["pairing_g1"][1:])
    bad = BLS.aggregate_sigs(sigs[:3])
    bad.set_aggregation_info(agg.aggregation_info)
    assert bad.serialize().hex() == v["tampered_sig"] and BLS.verify(bad) is False


def test_threshold_vectors(golden):
    from bls_py.keys import PrivateKey
    from bls_py.signature import Signature
    from bls_py.threshold import Threshold
    rec = golden("threshold.json")["3_of_5"]
    players = rec["players"]
    lam = Threshold.lagrange_coeffs_at_zero(players)
    assert [hex(int(x)) for x in lam] == rec["lambdas"]
    shares = [int(s, 16) for s in rec["shares"]]
    assert int(Threshold.interpolate_at_zero(players, shares)) == int(rec["poly"][0], 16)
    msg = bytes.fromhex(rec["msg"])
    unit = [PrivateKey(s).sign(msg) for s in shares]
    assert [u.serialize().hex() for u in unit] == rec["unit_sigs"]
    comb = Threshold.aggregate_unit_sigs(unit, players, rec["T"])
    assert comb.serialize().hex() == rec["combined"]
    big = golden("threshold.json")["67_of_100"]
    lam = Threshold.lagrange_coeffs_at_zero(big["players"])
    assert [hex(int(x)) for x in lam] == big["lambdas"]
    sigs = [Signature.from_bytes(bytes.fromhex(s)) for s in big["unit_sigs"][:4]]
    assert [s.serialize().hex() for s in sigs] == big["unit_sigs"][:4]


def test_aggregate_pub_keys_vectors(golden):
    from bls_py.bls import BLS
    from bls_py.keys import PublicKey
    from bls_py.ec import JacobianPoint
    from bls_py import hostmath as H
    rec = golden("msm.json")["16"]
    pks = [PublicKey(JacobianPoint._from(H.F1, H.aff_to_jac(H.F1, H.g1_from_abi(bytes.fromhex(p)))))
           for p in rec["pk_affine"]]
    lst = list(pks)
    sec = BLS.aggregate_pub_keys(lst, True)
    assert [pk.serialize().hex() for pk in lst] == rec["sorted_ser"]        # sorted in place (bls.py:210)
    assert sec.serialize().hex() == rec["secure"]
    assert BLS.aggregate_pub_keys(list(pks), False).serialize().hex() == rec["simple"]
    with pytest.raises(Exception):
        BLS.aggregate_pub_keys([], True)


def test_verify_batch_matches_verify_one_by_one(golden, oracle_provider):
    """BLS.verify_batch: several aggregate signatures of different shapes, one tampered, one with
    a missing tree entry -> the same booleans as BLS.verify on each."""
    from bls_py.aggregation_info import AggregationInfo
    from bls_py.bls import BLS
    from bls_py.keys import PrivateKey
    sks = [PrivateKey.from_seed(bytes([i + 1] * 5)) for i in range(4)]
    msgs = [bytes([i, 100 + i]) for i in range(4)]
    sigs = [sk.sign(m) for sk, m in zip(sks, msgs)]
    agg4 = BLS.aggregate_sigs(sigs)
    agg2 = BLS.aggregate_sigs(sigs[:2])
    tampered = BLS.aggregate_sigs_simple(sigs[:3])
    tampered.set_aggregation_info(agg4.aggregation_info)
    same_msg = BLS.aggregate_sigs([sk.sign(b"same") for sk in sks[:3]])
    broken = sks[0].sign(b"x")
    info = broken.aggregation_info
    broken.set_aggregation_info(AggregationInfo({}, info.message_hashes, info.public_keys))     # tree entry missing
    batch = [agg4, sigs[0], tampered, agg2, same_msg, broken, sigs[3]]
    got = BLS.verify_batch(batch)
    assert got == [BLS.verify(s) for s in batch] == [True, True, False, True, True, False, True]
    assert BLS.verify_batch([]) == []


def test_pairing_wrappers_host_logic(golden):
    """bls_py.pairing (the reference's pairing.py:16-92): unwrapping, flags and type checks --
    the arithmetic here is the injected oracle's, on the GPU it is tests/test_gpu_parity.py's."""
    from bls_py import pairing as PR
    from bls_py.ec import AffinePoint, default_ec, default_ec_twist
    from bls_py.fields import Fq, Fq2
    q = default_ec.q

    def p1(h, inf=False):
        b = bytes.fromhex(h)
        return AffinePoint(Fq(q, int.from_bytes(b[:48], "big")), Fq(q, int.from_bytes(b[48:], "big")), inf, default_ec)

    def p2(h, inf=False):
        v = [int.from_bytes(bytes.fromhex(h)[48 * i:48 * (i + 1)], "big") for i in range(4)]
        return AffinePoint(Fq2(q, v[0], v[1]), Fq2(q, v[2], v[3]), inf, default_ec_twist)
    g = golden("pairing.json")["gen"]
    P, Qp = p1(g["g1"]), p2(g["g2"])
    assert PR.miller_loop(P, Qp).serialize().hex() == g["miller"]
    assert PR.ate_pairing(P, Qp).serialize().hex() == g["final_exp"]
    c = golden("lines.json")["generic"]
    assert PR.double_line_eval(p2(c["r"]), p1(c["p"])).serialize().hex() == c["dbl"]
    assert PR.add_line_eval(p2(c["r"]), p2(c["q"]), p1(c["p"])).serialize().hex() == c["add"]
    # a VALID point that carries inf=True keeps its coordinates on the way down (pairing.py:90-91)
    v = golden("pairing.json")["edge"]["flag_on_valid"]
    assert PR.ate_pairing_multi([p1(v["g1"][0], True)], [p2(v["g2"][0], True)]).serialize().hex() == v["out"]
    d = golden("pairing_degenerate.json")["cases"]["flag_in_team"]
    Ps = [p1(x, f[0]) for x, f in zip(d["g1"], d["inf"])]
    Qs = [p2(x, f[1]) for x, f in zip(d["g2"], d["inf"])]
    assert PR.ate_pairing_multi(Ps, Qs).serialize().hex() == d["out"]
    for bad in (lambda: PR.miller_loop(Qp, P), lambda: PR.ate_pairing_multi([P], [Qp, Qp]), lambda: PR.double_line_eval(P, P)):
        with pytest.raises(Exception):
            bad()
