"""CPU: the integer model of the line-stream multi-pairing (vmgen/linestream_model.py -- the data flow of
csrc/blsgpu_ml.hip: lines per pair, per-line products in the w-power basis with the lane-wise wrap rule, merge,
Horner) against the reference's golden vectors and the oracle."""
import json
import os

from conftest import GOLDEN, cat
from vmgen import linestream_model as M


def _pairs(g1, g2, n):
    I = lambda b: int.from_bytes(b, "big")
    out = []
    for i in range(n):
        a, b = g1[96 * i:96 * (i + 1)], g2[192 * i:192 * (i + 1)]
        out.append(((I(a[:48]), I(a[48:])), ((I(b[:48]), I(b[48:96])), (I(b[96:144]), I(b[144:])))))
    return out


def _bytes(flat):
    return b"".join(x.to_bytes(48, "big") for x in flat)


def test_schedule_matches_the_kernel_constants():
    sched = M.line_schedule()
    assert len(sched) == 68 and sum(1 for _, k in sched if k == "t") == 63
    # csrc/blsgpu_ml.hip line_is_tangent(): the chord lines
    assert [i for i, (_, k) in enumerate(sched) if k == "c"] == [1, 4, 8, 18, 51]


def test_model_equals_reference_vectors(oracle):
    with open(os.path.join(GOLDEN, "pairing.json")) as f:
        v = json.load(f)["small4"]
    g1, g2 = cat(v["g1"]), cat(v["g2"])
    for chunk in (1, 3, 4):                      # chunked products + dense merges
        assert oracle.final_exp(_bytes(M.miller_product(_pairs(g1, g2, 4), chunk))).hex() == v["out"]
    for n in (1, 2, 3):
        want = oracle.pairing_multi(g1[:96 * n], g2[:192 * n], n)
        assert oracle.final_exp(_bytes(M.miller_product(_pairs(g1, g2, n), 2))) == want


def test_model_on_seeded_pairs(oracle, seeded_pairs):
    g1, g2 = seeded_pairs
    n = 9
    want = oracle.pairing_multi(g1[:96 * n], g2[:192 * n], n)
    assert oracle.final_exp(_bytes(M.miller_product(_pairs(g1, g2, n), 4))) == want


def test_degenerate_pairs_are_flagged():
    """Q off the twist or of an order that ends the chain at Z = 0: pair_lines says so (the kernels then hand the
    pair to the slow program)"""
    with open(os.path.join(GOLDEN, "pairing_degenerate.json")) as f:
        cases = json.load(f)["cases"]
    for name in ("ord13", "off_curve", "qy_zero"):
        c = cases[name]
        p = _pairs(cat(c["g1"]), cat(c["g2"]), 1)[0]
        assert M.pair_lines(*p)[1] is False, name


def test_exact_lines_reproduce_the_reference_miller_values():
    """exact_pair_lines / exact_miller (the model of k_ml_lines_exact): fq_miller_loop itself, bit for bit, for the
    generators and for EVERY pair of the reference-generated degenerate cases (flags included)"""
    with open(os.path.join(GOLDEN, "pairing.json")) as f:
        g = json.load(f)["gen"]
    p = _pairs(bytes.fromhex(g["g1"]), bytes.fromhex(g["g2"]), 1)[0]
    assert _bytes(M.to_flat12(M.exact_miller(*p))).hex() == g["miller"]
    with open(os.path.join(GOLDEN, "pairing_degenerate.json")) as f:
        cases = json.load(f)["cases"]
    for name, c in cases.items():
        for i in range(len(c["g1"])):
            p = _pairs(bytes.fromhex(c["g1"][i]), bytes.fromhex(c["g2"][i]), 1)[0]
            assert _bytes(M.to_flat12(M.exact_miller(*p, bool(c["inf"][i][1])))).hex() == c["miller"][i], (name, i)
