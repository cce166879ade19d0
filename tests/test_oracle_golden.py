"""Pins the CPU oracle (oracle/bls381_oracle.c) to vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import hashlib

from conftest import cat

DEG = {"1": 1, "2": 2, "6": 6, "12": 12}


def test_field_kats(golden, oracle):
    f = golden("fields.json")
    for key, d in DEG.items():
        rec = f[key]
        ops = [bytes.fromhex(x) for x in rec["operands"]]
        for op in ("add", "mul", "sub"):
            for e in rec[op]:
                assert oracle.field_op(d, op, ops[e["i"]], ops[e["j"]]).hex() == e["r"], (d, op, e["i"], e["j"])
        for i in range(4):
            assert oracle.field_op(d, "neg", ops[i]).hex() == rec["neg"][i]
            assert oracle.field_op(d, "inv", ops[i]).hex() == rec["inv"][i]
            # inversion round trip, tests.py:49-52
            assert oracle.field_op(d, "inv", oracle.field_op(d, "inv", ops[i])) == ops[i]


def test_frobenius_and_pow(golden, oracle):
    f = golden("fields.json")
    for d, key in ((2, "fq2_qi_pow"), (6, "fq6_qi_pow"), (12, "fq12_qi_pow")):
        x = bytes.fromhex(f[str(d)]["operands"][0])
        for e in f[key]:
            assert oracle.qi_pow(d, x, e["i"]).hex() == e["r"], (d, e["i"])
    x = bytes.fromhex(f["12"]["operands"][1])
    for e in f["fq12_pow"]:
        assert oracle.fq12_pow(x, int(e["e"], 16)).hex() == e["r"]


def test_generator_pairing_anchor(golden, oracle):
    g = golden("pairing.json")["gen"]
    ml = oracle.miller_loop(bytes.fromhex(g["g1"]), bytes.fromhex(g["g2"]))
    assert ml.hex() == g["miller"]
    # anchors recorded in SURVEY.md section 8(c)
    assert hashlib.sha256(ml).hexdigest() == "8a49c80ae193b3013e50545b69d942e8a4557b02a1cc84718cb8d836fc95bd4d"
    fe = oracle.final_exp(ml)
    assert fe.hex() == g["final_exp"]
    assert hashlib.sha256(fe).hexdigest() == "70f0561453673ff155a40ba3618727f8a411c492748d845280dd71dce099905a"


def test_final_exp_kats(golden, oracle):
    for rec in golden("pairing.json")["final_exp"]:
        assert oracle.final_exp(bytes.fromhex(rec["in"])).hex() == rec["out"]


def test_small4_and_miller_values(golden, oracle):
    v = golden("pairing.json")["small4"]
    assert oracle.pairing_multi(cat(v["g1"]), cat(v["g2"]), 4).hex() == v["out"]
    for a, b, m in zip(v["g1"], v["g2"], v["miller"]):
        assert oracle.miller_loop(bytes.fromhex(a), bytes.fromhex(b)).hex() == m
    assert hashlib.sha256(bytes.fromhex(v["out"])).hexdigest() == \
        "a3eae78ea9a1be90a28dcee40d9a9e9c3bc856612474ca3536b9d9963be56e04"


def test_edge_cases(golden, oracle):
    for name, v in golden("pairing.json")["edge"].items():
        n = len(v["g1"])
        inf = bytes(int(x) for pr in v["inf"] for x in pr)
        assert oracle.pairing_multi(cat(v["g1"]), cat(v["g2"]), n, inf=inf).hex() == v["out"], name
        for a, b, pr, m in zip(v["g1"], v["g2"], v["inf"], v.get("miller", [])):
            assert oracle.miller_loop(bytes.fromhex(a), bytes.fromhex(b), pr[1]).hex() == m, name


def test_seeded_batches(golden, oracle, seeded_pairs):
    g1, g2 = seeded_pairs
    s = golden("pairing.json")["seeded"]
    for n in (8, 65, 1025):
        rec = s[str(n)]
        assert hashlib.sha256(g1[:96 * n]).hexdigest() == rec["sha256_g1"]
        assert hashlib.sha256(g2[:192 * n]).hexdigest() == rec["sha256_g2"]
        got = oracle.pairing_multi(g1[:96 * n], g2[:192 * n], n, threads=8)
        assert got.hex() == rec["out"], n
    # threaded product == the reference's serial loop
    assert oracle.pairing_multi(g1[:96 * 8], g2[:192 * 8], 8, threads=1).hex() == s["8"]["out"]


def test_verify4_pairing_inputs(golden, oracle):
    """C1: the (Ps, Qs) that BLS.verify hands to the pairing (bls.py:197-199)
    multiply to one; the tampered aggregate does not."""
    v = golden("verify4.json")
    out = oracle.pairing_multi(cat(v["pairing_g1"]), cat(v["pairing_g2"]), 5)
    assert out.hex() == v["pairing_out"]
    one = (1).to_bytes(48, "big") + bytes(48 * 11)
    assert out == one and v["verify"] is True and v["tampered_verify"] is False


def test_group_law_vectors(golden, oracle):
    p = golden("points.json")
    g1 = bytes.fromhex(p["g1"][0]["p"])
    g2 = bytes.fromhex(p["g2"][0]["p"])
    for rec in p["g1"]:
        got, inf = oracle.g1_msm(g1, [int(rec["k"], 16)], 1)
        assert got.hex() == rec["p"] and not inf
    for rec in p["g2"]:
        got, inf = oracle.g2_msm(g2, [int(rec["k"], 16)], 1)
        assert got.hex() == rec["p"] and not inf
    a = p["g1_add"]
    assert oracle.g1_msm(bytes.fromhex(a["a"] + a["b"]), None, 2)[0].hex() == a["sum"]
    assert oracle.g1_msm(bytes.fromhex(a["a"] + a["a"]), None, 2)[0].hex() == a["dbl"]
    a = p["g2_add"]
    assert oracle.g2_msm(bytes.fromhex(a["a"] + a["b"]), None, 2)[0].hex() == a["sum"]
    assert oracle.g2_msm(bytes.fromhex(a["a"] + a["a"]), None, 2)[0].hex() == a["dbl"]


def test_msm_vectors(golden, oracle):
    """aggregate_pub_keys (bls.py:203-223): sum t_i * pk_i over the sorted keys."""
    m = golden("msm.json")
    for n in ("2", "16"):
        rec = m[n]
        pts = {bytes.fromhex(x) for x in rec["pk_affine"]}
        # sort by compressed serialisation as the reference does (keys.py:63-64)
        def ser(pt):
            x, y = pt[:48], int.from_bytes(pt[48:], "big")
            q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
            b = bytearray(x)
            if y > q // 2:
                b[0] |= 0x80
            return bytes(b)
        spts = sorted(pts, key=ser)
        assert [ser(x).hex() for x in spts] == rec["sorted_ser"]
        ts = [int(t, 16) for t in rec["scalars"]]
        got, inf = oracle.g1_msm(b"".join(spts), ts, len(spts))
        assert got.hex() == rec["secure_affine"] and not inf
        got, inf = oracle.g1_msm(b"".join(spts), None, len(spts))
        assert got.hex() == rec["simple_affine"]


def test_threshold_vectors(golden, oracle):
    """Threshold.aggregate_unit_sigs (threshold.py:127-136) as a G2 MSM."""
    t = golden("threshold.json")
    for key, rec in t.items():
        pts = cat(rec["unit_sigs_affine"])
        lam = [int(x, 16) for x in rec["lambdas"]]
        got, inf = oracle.g2_msm(pts, lam, len(lam))
        assert got.hex() == rec["combined_affine"] and not inf, key


def test_fast_algorithm_flavour_equals_the_reference_flavour(oracle, seeded_pairs, golden):
    """oracle.pairing_multi_fast (bench.py's second CPU baseline: projective twist point, sparse lines, shared
    squaring) gives the reference's bytes on ordinary pairs, for every thread split"""
    g1, g2 = seeded_pairs
    for n, th in ((1, 1), (2, 2), (7, 3), (65, 8)):
        a, b = g1[:96 * n], g2[:192 * n]
        want = bytes.fromhex(golden("pairing.json")["seeded"]["65"]["out"]) if n == 65 else oracle.pairing_multi(a, b, n, threads=4)
        assert oracle.pairing_multi_fast(a, b, n, th) == want
    v = golden("pairing.json")["small4"]
    from conftest import cat
    assert oracle.pairing_multi_fast(cat(v["g1"]), cat(v["g2"]), 4, 2).hex() == v["out"]
