"""Parity of the HIP engine, called through the C ABI (ctypes -> libblsgpu.so),
against the reference's golden vectors and the CPU oracle.  Bit-exact: the
outputs are 576-byte canonical Fq12 serialisations.  Needs an MI355X."""
import hashlib

import pytest

from conftest import cat

pytestmark = pytest.mark.gpu
ONE = (1).to_bytes(48, "big") + bytes(48 * 11)


def test_generators_and_final_exp(engine, golden):
    g = golden("pairing.json")["gen"]
    out = engine.pairing_multi(bytes.fromhex(g["g1"]), bytes.fromhex(g["g2"]), 1)
    assert out.hex() == g["final_exp"]
    assert hashlib.sha256(out).hexdigest() == "70f0561453673ff155a40ba3618727f8a411c492748d845280dd71dce099905a"
    # fq12_final_exp on the REFERENCE's own Miller value and on arbitrary elements
    assert engine.final_exp(bytes.fromhex(g["miller"])).hex() == g["final_exp"]
    for rec in golden("pairing.json")["final_exp"]:
        assert engine.final_exp(bytes.fromhex(rec["in"])).hex() == rec["out"]
    assert engine.final_exp(bytes(576)) == bytes(576)


def test_small_multiples(engine, golden):
    v = golden("pairing.json")["small4"]
    assert engine.pairing_multi(cat(v["g1"]), cat(v["g2"]), 4).hex() == v["out"]
    # every prefix size, so that ragged workgroups (1..4 teams) are covered
    for n in (1, 2, 3):
        from oracle import pairing_multi as ref
        assert engine.pairing_multi(cat(v["g1"][:n]), cat(v["g2"][:n]), n) == ref(cat(v["g1"][:n]), cat(v["g2"][:n]), n)


# every edge case of the reference-generated fixture, flags included (flag_on_valid: a VALID point
# carrying inf=True -- the reference lets Q's flag skip the chord updates, fields_t.py:676-677)
EDGE = ["empty", "p_inf", "q_inf", "both_inf", "q_inf_py_zero", "both_inf_in_batch", "p_zero_noflag",
        "q_zero_noflag", "flag_on_valid", "mixed", "repeat", "q_and_negq", "p_and_negp"]


def flags(v):
    return bytes(int(b) for pr in v["inf"] for b in pr)


@pytest.mark.parametrize("name", EDGE)
def test_edge_cases(engine, golden, name):
    v = golden("pairing.json")["edge"][name]
    n = len(v["g1"])
    assert engine.pairing_multi(cat(v["g1"]), cat(v["g2"]), n, flags(v)).hex() == v["out"]
    if "miller" in v:                  # blsgpu_miller_loop_batch: the reference's Miller values themselves
        assert engine.miller_loop_batch(cat(v["g1"]), cat(v["g2"]), n, flags(v)).hex() == "".join(v["miller"])


DEGEN = ["ord13", "ord13_neg", "ord13_p_zero", "ord13_px_zero", "ord13_in_team", "ord13_first_of_4", "ord13_twice",
         "ord3_embedded", "ord11_embedded", "ord11_in_team", "off_curve", "off_curve_in_team", "qy_zero", "qx_zero",
         "q_zero_px_zero", "q_zero_p_order3", "p_zero_off_curve", "flag_on_valid", "flag_in_team", "flag_on_ord13",
         "flag_on_off_curve", "pflag_only", "all_kinds"]


# (mp threshold, mp3 threshold, wide maximum): k_miller_wide (round 5: one pair per two-wavefront workgroup, a product per lane,
# csrc/blsgpu_mlw.hip) is what calls of up to 1536 pairs take by default; k_miller is the wavefront VM's one-pair-per-wavefront form
KERNELS = {"k_miller": (1 << 30, 2 ** 64 - 1, 0), "k_miller_wide": (1 << 30, 2 ** 64 - 1, 1 << 30),
           "k_miller_mp<3>": (0, 0, 1536), "k_miller_mp<2>": (0, 1 << 30, 1536)}


class kernel_choice:
    """force one of the four Miller kernels whatever the batch size (thresholds of include/blsgpu.h)"""

    def __init__(self, engine, which):
        self.engine, self.thr = engine, KERNELS[which]

    def __enter__(self):
        self.engine.set_mp_threshold(self.thr[0])
        self.engine.set_mp3_threshold(self.thr[1])
        self.engine.set_miller_wide_max(self.thr[2])
        self.engine.set_ls_threshold(None)                      # (calls of >= 2304 pairs would take the line-stream kernels)

    def __exit__(self, *a):
        self.engine.set_ls_threshold(2304, 64)
        self.engine.set_mp_threshold(4096)
        self.engine.set_mp3_threshold(2 ** 64 - 1)              # back to the measured schedule
        self.engine.set_miller_wide_max(1536)


@pytest.mark.parametrize("kernel", list(KERNELS))
@pytest.mark.parametrize("name", DEGEN)
def test_degenerate_pairs(engine, golden, name, kernel):
    """Inputs on which the reference's special cases decide (low-order, off-curve, zero, flagged:
    tests/golden/pairing_degenerate.json, reference-generated).  The fast kernels must notice and
    k_miller_slow must reproduce the reference's bytes -- through k_miller (one pair per wavefront), k_miller_wide (one pair
    per two-wavefront workgroup) and k_miller_mp with three and with two pairs per wavefront."""
    v = golden("pairing_degenerate.json")["cases"][name]
    n = len(v["g1"])
    with kernel_choice(engine, kernel):
        assert engine.pairing_multi(cat(v["g1"]), cat(v["g2"]), n, flags(v)).hex() == v["out"]


@pytest.mark.parametrize("kernel", list(KERNELS))
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 64, 65, 1025])
def test_every_kernel_on_ordinary_batches(engine, golden, seeded_pairs, oracle, kernel, n):
    """the four Miller kernels give the same bytes on ordinary batches of every raggedness"""
    g1, g2 = seeded_pairs
    a, b = g1[:96 * n], g2[:192 * n]
    want = bytes.fromhex(golden("pairing.json")["seeded"]["1025"]["out"]) if n == 1025 else oracle.pairing_multi(a, b, n, threads=8)
    with kernel_choice(engine, kernel):
        assert engine.pairing_multi(a, b, n) == want


def test_degenerate_pairs_inside_large_batches(engine, golden, seeded_pairs, oracle):
    """the same pairs hidden in ordinary batches: 300 seeded pairs with degenerate ones spliced in
    (every third team of k_miller_mp gets one), and as groups of a batch call"""
    g1, g2 = seeded_pairs
    d = golden("pairing_degenerate.json")["cases"]
    a, b, inf = bytearray(), bytearray(), bytearray()
    picks = [d[k] for k in ("ord13", "ord11_embedded", "off_curve", "flag_on_valid", "qy_zero", "qx_zero")]
    for i in range(300):
        a += g1[96 * i:96 * (i + 1)]; b += g2[192 * i:192 * (i + 1)]; inf += bytes(2)
        if i % 9 == 4:
            c = picks[(i // 9) % len(picks)]
            a += bytes.fromhex(c["g1"][0]); b += bytes.fromhex(c["g2"][0]); inf += bytes(int(x) for x in c["inf"][0])
    n = len(a) // 96
    want = oracle.pairing_multi(bytes(a), bytes(b), n, threads=8, inf=bytes(inf))
    for kernel in KERNELS:
        with kernel_choice(engine, kernel):
            assert engine.pairing_multi(bytes(a), bytes(b), n, bytes(inf)) == want, kernel
    # batch entry: groups of 3 (one k_miller wavefront per pair) and of 27 (per-group product tree); then groups of 2
    # and 3 as the teams of k_miller_mp<2|3> (what a batch of >= 4096 pairs selects: the threshold is lowered here)
    def check_groups(gsz):
        groups = n // gsz
        m = gsz * groups
        out = engine.pairing_multi_batch(bytes(a[:96 * m]), bytes(b[:192 * m]), gsz, groups, bytes(inf[:2 * m]))
        for g in range(groups):
            sl = slice(gsz * g, gsz * (g + 1))
            want = oracle.pairing_multi(bytes(a[96 * sl.start:96 * sl.stop]), bytes(b[192 * sl.start:192 * sl.stop]), gsz,
                                        inf=bytes(inf[2 * sl.start:2 * sl.stop]))
            assert out[576 * g:576 * (g + 1)] == want, (gsz, g)

    for gsz in (3, 27):
        check_groups(gsz)
    engine.set_mp_threshold(0)
    try:
        for gsz in (2, 3):
            check_groups(gsz)
    finally:
        engine.set_mp_threshold(4096)


@pytest.mark.parametrize("form", ["fast_lines_and_one_factor_per_pair", "reference_lines_for_every_pair", "wavefront_vm_program"])
def test_miller_loop_batch_is_the_reference_value(engine, golden, seeded_pairs, oracle, form):
    """blsgpu_miller_loop_batch == fq_miller_loop bit for bit (not up to the final exponentiation): from the line-stream stage's
    fast lines with one Fq2 factor per pair (k_ml_exact_fixup, the default since round 5; degenerate pairs and py = 0 take the
    reference's own lines inside the same launches), with the reference's lines for every pair (k_ml_lines_exact + k_ml_small,
    BLSGPU_MILLER_EXACT_FAST=0), and through the wavefront VM's reference-faithful program k_miller_exact
    (BLSGPU_MILLER_EXACT_LANES=0)"""
    from conftest import engine_with_env
    if form == "wavefront_vm_program":
        engine = engine_with_env({"BLSGPU_MILLER_EXACT_LANES": "0"})
    elif form == "reference_lines_for_every_pair":
        engine = engine_with_env({"BLSGPU_MILLER_EXACT_FAST": "0"})
    g = golden("pairing.json")
    assert engine.miller_loop_batch(bytes.fromhex(g["gen"]["g1"]), bytes.fromhex(g["gen"]["g2"]), 1).hex() == g["gen"]["miller"]
    v = g["small4"]
    assert engine.miller_loop_batch(cat(v["g1"]), cat(v["g2"]), 4).hex() == "".join(v["miller"])
    for name, c in golden("pairing_degenerate.json")["cases"].items():
        n = len(c["g1"])
        assert engine.miller_loop_batch(cat(c["g1"]), cat(c["g2"]), n, flags(c)).hex() == "".join(c["miller"]), name
    g1, g2 = seeded_pairs
    n = 333                                                             # (ragged last wavefronts of both lane kernels)
    out = engine.miller_loop_batch(g1[:96 * n], g2[:192 * n], n)
    for i in (0, 1, 33, 69, 331, 332):
        assert out[576 * i:576 * (i + 1)] == oracle.miller_loop(g1[96 * i:96 * (i + 1)], g2[192 * i:192 * (i + 1)])
    # py = 0 (not a point of G1, but the reference's loop is a total function of the coordinates): the factor would divide by it
    g1z = bytearray(g1[:96 * 5])
    g1z[96 * 1 + 48:96 * 2] = bytes(48)
    g1z[96 * 3:96 * 4] = bytes(96)
    out = engine.miller_loop_batch(bytes(g1z), g2[:192 * 5], 5)
    for i in range(5):
        assert out[576 * i:576 * (i + 1)] == oracle.miller_loop(bytes(g1z[96 * i:96 * (i + 1)]), g2[192 * i:192 * (i + 1)]), i
    assert engine.miller_loop_batch(b"", b"", 0) == b""


def test_line_evaluations(engine, golden, oracle, seeded_pairs):
    """blsgpu_line_eval_batch == fq2_double_line_eval / fq2_add_line_eval (reference vectors, then the oracle)"""
    for name, c in golden("lines.json").items():
        r, q, p = bytes.fromhex(c["r"]), bytes.fromhex(c["q"]), bytes.fromhex(c["p"])
        assert engine.line_eval_batch(r, None, p, 1).hex() == c["dbl"], name
        assert engine.line_eval_batch(r, q, p, 1).hex() == c["add"], name
    g1, g2 = seeded_pairs
    n = 40
    r, q, p = g2[:192 * n], g2[192 * n:192 * 2 * n], g1[:96 * n]
    dbl, add = engine.line_eval_batch(r, None, p, n), engine.line_eval_batch(r, q, p, n)
    for i in (0, 7, 39):
        ri, qi, pi = r[192 * i:192 * (i + 1)], q[192 * i:192 * (i + 1)], p[96 * i:96 * (i + 1)]
        assert dbl[576 * i:576 * (i + 1)] == oracle.line_eval(ri, None, pi)
        assert add[576 * i:576 * (i + 1)] == oracle.line_eval(ri, qi, pi)


def test_pairing_module_wrappers(golden):
    """bls_py.pairing: the six functions of the reference's pairing.py:16-92 with its signatures"""
    from bls_py import pairing as PR
    from bls_py.ec import AffinePoint, default_ec, default_ec_twist
    from bls_py.fields import Fq, Fq2, Fq12
    q = default_ec.q
    g = golden("pairing.json")["gen"]

    def p1(h, inf=False):
        b = bytes.fromhex(h)
        return AffinePoint(Fq(q, int.from_bytes(b[:48], "big")), Fq(q, int.from_bytes(b[48:], "big")), inf, default_ec)

    def p2(h, inf=False):
        v = [int.from_bytes(bytes.fromhex(h)[48 * i:48 * (i + 1)], "big") for i in range(4)]
        return AffinePoint(Fq2(q, v[0], v[1]), Fq2(q, v[2], v[3]), inf, default_ec_twist)
    P, Qp = p1(g["g1"]), p2(g["g2"])
    ml = PR.miller_loop(P, Qp)
    assert type(ml) is Fq12 and ml.serialize().hex() == g["miller"]
    assert PR.final_exponentiation(ml, default_ec).serialize().hex() == g["final_exp"]
    assert PR.ate_pairing(P, Qp).serialize().hex() == g["final_exp"]
    c = golden("lines.json")["generic"]
    R, Q2, P5 = p2(c["r"]), p2(c["q"]), p1(c["p"])
    assert PR.double_line_eval(R, P5).serialize().hex() == c["dbl"]
    assert PR.add_line_eval(R, Q2, P5).serialize().hex() == c["add"]
    v = golden("pairing.json")["edge"]["flag_on_valid"]
    assert PR.ate_pairing_multi([p1(v["g1"][0], True)], [p2(v["g2"][0], True)]).serialize().hex() == v["out"]
    d = golden("pairing_degenerate.json")["cases"]["ord13_in_team"]
    assert PR.ate_pairing_multi([p1(x) for x in d["g1"]], [p2(x) for x in d["g2"]]).serialize().hex() == d["out"]
    with pytest.raises(Exception):
        PR.miller_loop(Qp, P)
    with pytest.raises(Exception):
        PR.ate_pairing_multi([P], [Qp, Qp])


@pytest.mark.parametrize("size", [8192, 65536])
def test_seeded_digest_vs_reference(engine, golden, size):
    """SURVEY 8c F-PAIR: the reference's multi-pairing of the first 8192 / 65 536 PRF-seeded pairs (BASELINE
    configs[2] at full size; fixtures made by make_golden.py seeded8192 / seeded65536 -- 10 and 80 minutes of
    pure Python).  The inputs are rebuilt here from the PRF scalars with the engine's own group sums and
    checked against the fixture's input digests first."""
    v = golden("pairing_seeded_%d.json" % size)
    n = v["n"]
    nord = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001

    def prf(tag, i):
        return int.from_bytes(hashlib.sha256(tag + (1).to_bytes(4, "big") + i.to_bytes(4, "big")).digest(), "big") % (nord - 1) + 1
    gen = golden("pairing.json")["gen"]
    a = [prf(b"blsgpu/a", i) for i in range(n)]
    b = [prf(b"blsgpu/b", i) for i in range(n)]
    g1, _ = engine.g1_msm(bytes.fromhex(gen["g1"]) * n, a, 1, n)
    g2, _ = engine.g2_msm(bytes.fromhex(gen["g2"]) * n, b, 1, n)
    assert hashlib.sha256(g1).hexdigest() == v["sha256_g1"] and hashlib.sha256(g2).hexdigest() == v["sha256_g2"]
    out = engine.pairing_multi(g1, g2, n)
    assert out.hex() == v["out"] and hashlib.sha256(out).hexdigest() == v["sha256_out"]


@pytest.mark.parametrize("n", [8, 65, 1025])
def test_seeded_batches_vs_reference(engine, golden, seeded_pairs, n):
    g1, g2 = seeded_pairs
    out = engine.pairing_multi(g1[:96 * n], g2[:192 * n], n)
    assert out.hex() == golden("pairing.json")["seeded"][str(n)]["out"]


@pytest.mark.parametrize("n", [5, 63, 64, 257])
def test_seeded_batches_vs_oracle(engine, oracle, seeded_pairs, n):
    g1, g2 = seeded_pairs
    off = 300
    a, b = g1[96 * off:96 * (off + n)], g2[192 * off:192 * (off + n)]
    assert engine.pairing_multi(a, b, n) == oracle.pairing_multi(a, b, n, threads=8)


def test_verify4_pairing(engine, golden):
    v = golden("verify4.json")
    assert engine.pairing_multi(cat(v["pairing_g1"]), cat(v["pairing_g2"]), 5) == ONE


def test_order_independence_and_bilinearity(engine, oracle, seeded_pairs):
    """Size-independent properties: the product is order independent, and k
    copies of a batch give the k-th power (checked with the oracle's fq12_pow)."""
    g1, g2 = seeded_pairs
    n = 96
    a, b = g1[:96 * n], g2[:192 * n]
    base = engine.pairing_multi(a, b, n)
    ra = b"".join(a[96 * i:96 * (i + 1)] for i in reversed(range(n)))
    rb = b"".join(b[192 * i:192 * (i + 1)] for i in reversed(range(n)))
    assert engine.pairing_multi(ra, rb, n) == base
    assert engine.pairing_multi(a * 3, b * 3, 3 * n) == oracle.fq12_pow(base, 3)


def test_full_size_batch_property(engine, oracle, golden, seeded_pairs):
    """C3-shape shard: 8 x 1025 = 8200 pairs on one GPU equals golden^8."""
    g1, g2 = seeded_pairs
    want = oracle.fq12_pow(bytes.fromhex(golden("pairing.json")["seeded"]["1025"]["out"]), 8)
    assert engine.pairing_multi(g1 * 8, g2 * 8, 8200) == want


def test_sharded_path_matches_single_call(engine, seeded_pairs):
    """blsgpu_miller_product_dev on shards + blsgpu_final_exp_product_dev ==
    blsgpu_pairing_multi (the multi-GPU decomposition, run on one device)."""
    import torch
    g1, g2 = seeded_pairs
    n = 1025
    dev = torch.device("cuda:0")
    t1 = torch.frombuffer(bytearray(g1), dtype=torch.uint8).to(dev)
    t2 = torch.frombuffer(bytearray(g2), dtype=torch.uint8).to(dev)
    shards = [(0, 300), (300, 301), (301, 1025), (1025, 1025)]      # ragged, one empty
    parts = torch.zeros(len(shards), 144, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for k, (lo, hi) in enumerate(shards):
        engine.miller_product_dev(t1.data_ptr() + 96 * lo, t2.data_ptr() + 192 * lo, hi - lo,
                                  parts[k].data_ptr(), st)
    out = torch.zeros(576, dtype=torch.uint8, device=dev)
    engine.final_exp_product_dev(parts.data_ptr(), len(shards), out.data_ptr(), st)
    torch.cuda.synchronize()
    whole = torch.zeros(576, dtype=torch.uint8, device=dev)
    engine.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, whole.data_ptr(), st)
    torch.cuda.synchronize()
    assert bytes(out.cpu().numpy()) == bytes(whole.cpu().numpy()) == engine.pairing_multi(g1, g2, n)


def test_reference_boundary_signature(golden):
    """fields_t_hip mirrors fq_ate_pairing_multi(Ps, Qs) of fields_t_c: tuples of
    Python ints in, a 12-tuple of ints out (fields_t_c.pyx:2333-2346)."""
    from bls_py import fields_t_hip as fh
    v = golden("pairing.json")["small4"]

    def ints(h, k):
        b = bytes.fromhex(h)
        return [int.from_bytes(b[48 * i:48 * (i + 1)], "big") for i in range(k)]
    Ps = tuple((ints(a, 2)[0], ints(a, 2)[1], False) for a in v["g1"])
    Qs = tuple(((ints(b, 4)[0], ints(b, 4)[1]), (ints(b, 4)[2], ints(b, 4)[3]), False) for b in v["g2"])
    res = fh.fq_ate_pairing_multi(Ps, Qs)
    assert isinstance(res, tuple) and len(res) == 12
    assert b"".join(x.to_bytes(48, "big") for x in res).hex() == v["out"]
    g = golden("pairing.json")["gen"]
    assert fh.fq12_final_exp(tuple(ints(g["miller"], 12))) == tuple(ints(g["final_exp"], 12))
    assert fh.fq_ate_pairing_multi((), ()) == (1,) + (0,) * 11
    # flags travel with the coordinates, as in fields_t_c.pyx:2333-2346
    e = golden("pairing.json")["edge"]["flag_on_valid"]
    Pf = ((ints(e["g1"][0], 2)[0], ints(e["g1"][0], 2)[1], True),)
    Qf = (((ints(e["g2"][0], 4)[0], ints(e["g2"][0], 4)[1]), (ints(e["g2"][0], 4)[2], ints(e["g2"][0], 4)[3]), True),)
    assert b"".join(x.to_bytes(48, "big") for x in fh.fq_ate_pairing_multi(Pf, Qf)).hex() == e["out"]
    assert fh.fq_miller_loop(*Ps[0], *Qs[0]) == tuple(ints(v["miller"][0], 12))
    c = golden("lines.json")["generic"]
    r, qq, p = ints(c["r"], 4), ints(c["q"], 4), ints(c["p"], 2)
    assert fh.fq2_double_line_eval((r[0], r[1]), (r[2], r[3]), p[0], p[1]) == tuple(ints(c["dbl"], 12))
    assert fh.fq2_add_line_eval((r[0], r[1]), (r[2], r[3]), (qq[0], qq[1]), (qq[2], qq[3]), p[0], p[1]) == tuple(ints(c["add"], 12))


def test_batched_independent_pairings(engine, golden, seeded_pairs, oracle):
    """blsgpu_pairing_multi_batch: many small fq_ate_pairing_multi calls at once
    (the verify step of C4: 2 pairs per group), and blsgpu_final_exp_batch."""
    g1, g2 = seeded_pairs
    for gsz, groups in ((2, 9), (1, 5), (3, 4), (5, 1)):
        n = gsz * groups
        out = engine.pairing_multi_batch(g1[:96 * n], g2[:192 * n], gsz, groups)
        for g in range(groups):
            want = oracle.pairing_multi(g1[96 * gsz * g:96 * gsz * (g + 1)], g2[192 * gsz * g:192 * gsz * (g + 1)], gsz)
            assert out[576 * g:576 * (g + 1)] == want, (gsz, g)
    v = golden("verify4.json")
    one = (1).to_bytes(48, "big") + bytes(48 * 11)
    assert engine.pairing_multi_batch(cat(v["pairing_g1"]) * 3, cat(v["pairing_g2"]) * 3, 5, 3) == one * 3
    recs = golden("pairing.json")["final_exp"]
    ins = b"".join(bytes.fromhex(r["in"]) for r in recs) * 4
    assert engine.final_exp_batch(ins) == b"".join(bytes.fromhex(r["out"]) for r in recs) * 4


@pytest.mark.parametrize("gsz,groups,kernel", [(25, 5, "k_miller_mp<3>"), (25, 5, "k_miller_mp<2>"), (25, 5, "k_miller"), (25, 5, "k_miller_wide"),
                                               (205, 5, "k_miller_wide"), (64, 16, "k_miller_wide"),
                                               (205, 5, "k_miller_mp<3>"), (205, 5, "k_miller_mp<2>"), (1025, 3, "k_miller_mp<3>"),
                                               (1025, 3, "k_miller_mp<2>"), (64, 16, "k_miller")])
def test_batched_large_groups(engine, seeded_pairs, gsz, groups, kernel):
    """Groups long enough for the per-group product tree (grouped k_miller / k_miller_mp +
    k_reduce over blockIdx.y): every group must equal its own blsgpu_pairing_multi.
    gsz not a multiple of 2, 3 or 4 exercises the ragged last team of every group."""
    g1, g2 = seeded_pairs
    n = gsz * groups
    reps = (n + 1024) // 1025
    a, b = (g1 * reps)[:96 * n], (g2 * reps)[:192 * n]
    with kernel_choice(engine, kernel):
        out = engine.pairing_multi_batch(a, b, gsz, groups)
        singles = [engine.pairing_multi(a[96 * gsz * g:96 * gsz * (g + 1)], b[192 * gsz * g:192 * gsz * (g + 1)], gsz)
                   for g in range(groups)]
    assert [out[576 * g:576 * (g + 1)] for g in range(groups)] == singles
    assert len(set(singles)) == groups or gsz * groups > 1025


def test_batched_sharded_form(engine, golden, seeded_pairs):
    """The multi-GPU decomposition of a batch on one device: 2 'ranks' x 3 verifications;
    rank r holds a slice of every verification.  all-gather layout [rank][group]."""
    import torch
    g1, g2 = seeded_pairs
    dev = torch.device("cuda:0")
    groups, cuts = 3, (0, 400, 1025)                          # rank 0: pairs [0,400), rank 1: [400,1025)
    rot = [0, 7, 500]                                         # every verification is a rotation of the seeded batch
    def rotated(buf, sz, k):
        return buf[sz * k:sz * 1025] + buf[:sz * k]
    parts = torch.zeros(2, groups, 144, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    keep = []
    for r in range(2):
        lo, hi = cuts[r], cuts[r + 1]
        a = b"".join(rotated(g1, 96, k)[96 * lo:96 * hi] for k in rot)
        b = b"".join(rotated(g2, 192, k)[192 * lo:192 * hi] for k in rot)
        ta = torch.frombuffer(bytearray(a), dtype=torch.uint8).to(dev)
        tb = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        keep += [ta, tb]
        engine.miller_product_batch_dev(ta.data_ptr(), tb.data_ptr(), hi - lo, groups, parts[r].data_ptr(), st)
    out = torch.zeros(groups, 576, dtype=torch.uint8, device=dev)
    engine.final_exp_product_batch_dev(parts.data_ptr(), 2, groups, out.data_ptr(), st)
    torch.cuda.synchronize()
    want = golden("pairing.json")["seeded"]["1025"]["out"]     # a rotation does not change the product
    for g in range(groups):
        assert bytes(out[g].cpu().numpy()).hex() == want


def test_batched_many_groups_sliced(engine, golden):
    """More groups than one launch carries (slices of 32768): 40000 groups of 24 pairs, all the
    same pairs, so every result must equal the first (itself checked against blsgpu_pairing_multi)."""
    v = golden("pairing.json")["small4"]
    a = cat(v["g1"]) * 6
    b = cat(v["g2"]) * 6
    groups = 40000
    out = engine.pairing_multi_batch(a * groups, b * groups, 24, groups)
    one = engine.pairing_multi(a, b, 24)
    assert out == one * groups


def test_two_streams_on_one_context_and_growing_workspace(golden, seeded_pairs):
    """One context used from two streams, with batch sizes that make the workspace grow while work is
    enqueued: calls on different streams are ordered on the device (include/blsgpu.h "Threading and
    streams"), replaced buffers stay alive until blsgpu_ctx_trim."""
    import torch
    from bls_py import _native
    eng = _native.Engine(0)                                   # a fresh context: small initial workspace
    g1, g2 = seeded_pairs
    dev = torch.device("cuda:0")
    t1 = torch.frombuffer(bytearray(g1 * 9), dtype=torch.uint8).to(dev)
    t2 = torch.frombuffer(bytearray(g2 * 9), dtype=torch.uint8).to(dev)
    s = [torch.cuda.Stream(device=dev) for _ in range(2)]
    sizes = [64, 1025, 300, 4100, 1025, 9225, 8, 1025]        # 4100 and 9225 pairs outgrow the 4096-pair default
    outs = [torch.zeros(576, dtype=torch.uint8, device=dev) for _ in sizes]
    for k, n in enumerate(sizes):
        eng.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, outs[k].data_ptr(), s[k % 2].cuda_stream)
    torch.cuda.synchronize()
    want1025 = golden("pairing.json")["seeded"]["1025"]["out"]
    ref = _native.engine(0)
    for k, n in enumerate(sizes):
        got = bytes(outs[k].cpu().numpy())
        if n == 1025:
            assert got.hex() == want1025
        else:
            assert got == ref.pairing_multi((g1 * 9)[:96 * n], (g2 * 9)[:192 * n], n), n
    eng.trim()
    eng.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), 1025, outs[0].data_ptr(), s[1].cuda_stream)
    torch.cuda.synchronize()
    assert bytes(outs[0].cpu().numpy()).hex() == want1025
    eng.close()
