"""Parity of the LINE-STREAM multi-pairing (csrc/blsgpu_ml.hip: k_ml_lines2 -> k_ml_accum -> k_ml_merge ->
k_ml_horner_wide, degenerate pairs' lines rewritten by k_ml_lines_exact) through the C ABI, with the path FORCED
for every size (blsgpu_ctx_set_ls_threshold), against the reference's golden vectors and the CPU oracle.  Bit-exact
(576-byte canonical Fq12 serialisations).  Needs an MI355X."""
import hashlib
import json
import os

import pytest

from conftest import GOLDEN, cat, engine_with_env

pytestmark = pytest.mark.gpu


def flags(v):
    return bytes(int(b) for pr in v["inf"] for b in pr)


# The point chains have two forms chosen by the size of the call (csrc/blsgpu_api.hip launch_miller_ls): up to
# BLSGPU_LS_QUAD_MAX = 20 480 pairs one pair per lane QUAD (k_ml_lines4), above it one pair per lane PAIR (k_ml_lines2 -- the
# kernel of bench.py's 524 800-pair step).  Every test below that takes `ls` runs once per form: "lane_pairs" creates the
# context with BLSGPU_LS_QUAD_MAX=0, so k_ml_lines2's own flagging path (Q flagged / off the twist / final Z = 0 behind
# sp::tangent_step) meets the edge and degenerate vectors at these sizes too.
# Round 5 added a third form for calls of up to 10 240 pairs: sixteen lanes per pair with the values in LDS (k_ml_lines_wide,
# csrc/blsgpu_lsw.hip) -- "sixteen_lanes" below is the default selection at these sizes, "lane_quads" switches it off.
CHAIN_FORMS = {"sixteen_lanes": {}, "lane_quads": {"BLSGPU_LS_WIDE_MAX": "0"}, "lane_pairs": {"BLSGPU_LS_WIDE_MAX": "0", "BLSGPU_LS_QUAD_MAX": "0"}}


@pytest.fixture(scope="module", params=list(CHAIN_FORMS))
def ls(request):
    """an engine of its own whose every multi-pairing runs the line-stream kernels"""
    e = engine_with_env(CHAIN_FORMS[request.param])
    e.set_ls_threshold(1, 1)
    return e


def test_default_selection_is_line_stream_for_large_calls(engine, ls, seeded_pairs, golden):
    """the shared engine (default thresholds) and the forced one agree on the reference's 1025-pair vector, and a
    call of 32 x 1025 pairs -- line-stream by default -- returns it for every group"""
    g1, g2 = seeded_pairs
    want = golden("pairing.json")["seeded"]["1025"]["out"]
    assert ls.pairing_multi(g1, g2, 1025).hex() == want
    out = engine.pairing_multi_batch(g1 * 32, g2 * 32, 1025, 32)
    assert all(out[576 * g:576 * (g + 1)].hex() == want for g in range(32))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 64, 65])
def test_ordinary_batches(ls, golden, seeded_pairs, oracle, n):
    g1, g2 = seeded_pairs
    a, b = g1[:96 * n], g2[:192 * n]
    g = golden("pairing.json")["seeded"]
    want = bytes.fromhex(g[str(n)]["out"]) if str(n) in g and isinstance(g[str(n)], dict) else oracle.pairing_multi(a, b, n, threads=8)
    assert ls.pairing_multi(a, b, n) == want


def test_small4_and_generators(ls, golden):
    v = golden("pairing.json")["small4"]
    assert ls.pairing_multi(cat(v["g1"]), cat(v["g2"]), 4).hex() == v["out"]
    g = golden("pairing.json")["gen"]
    assert ls.pairing_multi(bytes.fromhex(g["g1"]), bytes.fromhex(g["g2"]), 1).hex() == g["final_exp"]


EDGE = ["p_inf", "q_inf", "both_inf", "q_inf_py_zero", "both_inf_in_batch", "p_zero_noflag",
        "q_zero_noflag", "flag_on_valid", "mixed", "repeat", "q_and_negq", "p_and_negp"]


@pytest.mark.parametrize("name", EDGE)
def test_edge_cases(ls, golden, name):
    v = golden("pairing.json")["edge"][name]
    n = len(v["g1"])
    assert ls.pairing_multi(cat(v["g1"]), cat(v["g2"]), n, flags(v)).hex() == v["out"]


def test_degenerate_pairs(ls, golden):
    """every reference-generated case on which the reference's special cases decide: the lines kernel must flag the
    pair (Q flagged / off the twist / final Z = 0) and k_ml_lines_exact must write the reference's own line values"""
    for name, v in golden("pairing_degenerate.json")["cases"].items():
        n = len(v["g1"])
        assert ls.pairing_multi(cat(v["g1"]), cat(v["g2"]), n, flags(v)).hex() == v["out"], name


def _spliced(golden, seeded_pairs, count=300, every=9):
    g1, g2 = seeded_pairs
    d = golden("pairing_degenerate.json")["cases"]
    a, b, inf = bytearray(), bytearray(), bytearray()
    picks = [d[k] for k in ("ord13", "ord11_embedded", "off_curve", "flag_on_valid", "qy_zero", "qx_zero")]
    for i in range(count):
        a += g1[96 * i:96 * (i + 1)]; b += g2[192 * i:192 * (i + 1)]; inf += bytes(2)
        if i % every == 4:
            c = picks[(i // every) % len(picks)]
            a += bytes.fromhex(c["g1"][0]); b += bytes.fromhex(c["g2"][0]); inf += bytes(int(x) for x in c["inf"][0])
    return bytes(a), bytes(b), bytes(inf)


def test_degenerate_pairs_inside_batches_and_groups(ls, golden, seeded_pairs, oracle):
    a, b, inf = _spliced(golden, seeded_pairs)
    n = len(a) // 96
    assert ls.pairing_multi(a, b, n, inf) == oracle.pairing_multi(a, b, n, threads=8, inf=inf)
    for gsz in (27, 100):                       # groups of a batch call, some with and some without such pairs
        groups = n // gsz
        m = gsz * groups
        out = ls.pairing_multi_batch(a[:96 * m], b[:192 * m], gsz, groups, inf[:2 * m])
        for g in range(groups):
            lo, hi = gsz * g, gsz * (g + 1)
            want = oracle.pairing_multi(a[96 * lo:96 * hi], b[192 * lo:192 * hi], gsz, inf=inf[2 * lo:2 * hi], threads=8)
            assert out[576 * g:576 * (g + 1)] == want, (gsz, g)


def test_all_degenerate_batch(ls, golden, oracle):
    """nothing but low-order / off-curve pairs: every pair goes through the slow program and the fold"""
    d = golden("pairing_degenerate.json")["cases"]
    names = ["ord13", "ord11_embedded", "off_curve", "qy_zero", "ord3_embedded", "ord13_neg"]
    a = b"".join(bytes.fromhex(d[k]["g1"][0]) for k in names) * 8
    b = b"".join(bytes.fromhex(d[k]["g2"][0]) for k in names) * 8
    n = len(a) // 96
    assert ls.pairing_multi(a, b, n) == oracle.pairing_multi(a, b, n, threads=8)
    out = ls.pairing_multi_batch(a, b, 6, 8)
    assert out == oracle.pairing_multi(a[:96 * 6], b[:192 * 6], 6) * 8


def test_4096_low_order_pairs_against_the_oracle(ls, golden, oracle):
    """a batch of nothing but low-order / off-curve pairs: every pair's line records are rewritten by
    k_ml_lines_exact with the reference's own line values; one 4096-pair call and 32 groups of 128"""
    d = golden("pairing_degenerate.json")["cases"]
    names = ["ord13", "ord11_embedded", "off_curve", "qy_zero", "ord3_embedded", "ord13_neg", "qx_zero", "ord13_px_zero"]
    a = b"".join(bytes.fromhex(d[k]["g1"][0]) for k in names) * 512
    b = b"".join(bytes.fromhex(d[k]["g2"][0]) for k in names) * 512
    assert ls.pairing_multi(a, b, 4096) == oracle.pairing_multi(a, b, 4096, threads=8)
    assert ls.pairing_multi_batch(a, b, 128, 32) == oracle.pairing_multi(a[:96 * 128], b[:192 * 128], 128, threads=8) * 32


@pytest.mark.parametrize("teams", [1, 700, 4000, 10 ** 9])
def test_chunking_and_merge_levels(golden, seeded_pairs, teams):
    """chunks per group from 1 (no merge) to 65 (three merge levels at fan-in 8): the same bytes"""
    from bls_py import _native
    e = _native.Engine(0)
    e.set_ls_threshold(1, 1)
    e.set_ls_teams(teams)
    g1, g2 = seeded_pairs
    assert e.pairing_multi(g1, g2, 1025).hex() == golden("pairing.json")["seeded"]["1025"]["out"]
    # ragged groups: 5 groups of 205
    out = e.pairing_multi_batch(g1, g2, 205, 5)
    vm = engine_with_env({"BLSGPU_FEXP_WIDE": "0"})       # the wavefront VM end to end (its own final exponentiation too)
    vm.set_ls_threshold(None)
    vm.set_fexp_team_threshold(None)
    assert out == vm.pairing_multi_batch(g1, g2, 205, 5)


def test_full_size_reference_digests(ls, engine):
    """BASELINE configs[2]: the reference's own multi-pairing of the first 8192 and 65 536 PRF pairs
    (tests/golden/pairing_seeded_*.json, generated by importing the reference) through the line-stream kernels"""
    import bench
    with open(os.path.join(GOLDEN, "pairing.json")) as f:
        gj = json.load(f)
    gen1, gen2 = bytes.fromhex(gj["gen"]["g1"]), bytes.fromhex(gj["gen"]["g2"])
    g1, g2, _, _ = bench.seeded_points(engine, gen1, gen2, list(range(65536)))
    for n in (8192, 65536):
        with open(os.path.join(GOLDEN, "pairing_seeded_%d.json" % n)) as f:
            want = json.load(f)
        out = ls.pairing_multi(g1[:96 * n], g2[:192 * n], n)
        assert out.hex() == want["out"] and hashlib.sha256(out).hexdigest() == want["sha256_out"]


def test_miller_product_partials_compose(ls, seeded_pairs, golden):
    """the sharded form (blsgpu_miller_product_batch_dev + blsgpu_final_exp_product_batch_dev): partials of two
    halves of the batch, computed by the line-stream kernels, multiply to the reference's result"""
    import torch
    g1, g2 = seeded_pairs
    dev = torch.device("cuda", 0)
    up = lambda x: torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev)
    parts = torch.zeros(2 * 144, dtype=torch.int32, device=dev)
    out = torch.zeros(576, dtype=torch.uint8, device=dev)
    h = 512
    t1, t2 = up(g1[:96 * h]), up(g2[:192 * h])
    u1, u2 = up(g1[96 * h:]), up(g2[192 * h:])
    ls.miller_product_batch_dev(t1.data_ptr(), t2.data_ptr(), h, 1, parts[:144].data_ptr())
    ls.miller_product_batch_dev(u1.data_ptr(), u2.data_ptr(), 1025 - h, 1, parts[144:].data_ptr())
    ls.final_exp_product_batch_dev(parts.data_ptr(), 2, 1, out.data_ptr())
    torch.cuda.synchronize()
    assert bytes(out.cpu().numpy()).hex() == golden("pairing.json")["seeded"]["1025"]["out"]


@pytest.mark.parametrize("chains", list(CHAIN_FORMS))
def test_small_groups_one_accumulator_per_group(golden, seeded_pairs, oracle, chains):
    """k_ml_small: batches of groups of 1, 2, 3, 7 and 27 pairs (threshold verifies, single signatures), with degenerate
    pairs spliced in, against the oracle group by group"""
    e = engine_with_env(CHAIN_FORMS[chains])
    e.set_ls_threshold(1, 1 << 30)                     # every group is "small"
    a, b, inf = _spliced(golden, seeded_pairs, count=220, every=13)
    n = len(a) // 96
    for gsz in (1, 2, 3, 7, 27):
        groups = min(n // gsz, 40)
        m = gsz * groups
        out = e.pairing_multi_batch(a[:96 * m], b[:192 * m], gsz, groups, inf[:2 * m])
        for g in range(groups):
            lo, hi = gsz * g, gsz * (g + 1)
            want = oracle.pairing_multi(a[96 * lo:96 * hi], b[192 * lo:192 * hi], gsz, inf=inf[2 * lo:2 * hi])
            assert out[576 * g:576 * (g + 1)] == want, (gsz, g)
    # one long "small" group: the whole loop on one team
    g1, g2 = seeded_pairs
    assert e.pairing_multi(g1[:96 * 65], g2[:192 * 65], 65).hex() == golden("pairing.json")["seeded"]["65"]["out"]


def test_default_selection_above_the_quad_threshold_meets_degenerate_pairs(engine, golden, seeded_pairs, oracle):
    """DEFAULT thresholds, one call of 64 x 333 = 21 312 pairs (> BLSGPU_LS_QUAD_MAX = 20 480: the point chains run on
    k_ml_lines2) in which every 333-pair run carries the six spliced degenerate picks: 384 flagged pairs among ordinary
    ones, rewritten by k_ml_lines_exact.  The multi-pairing of 64 copies of a batch is the batch's value to the 64th
    power; the batch's value is the oracle's (pinned to the reference's vectors).  Also as 64 groups of a batch call."""
    a, b, inf = _spliced(golden, seeded_pairs)
    n = len(a) // 96
    assert n == 333
    one = oracle.pairing_multi(a, b, n, threads=8, inf=inf)
    k = 64
    assert k * n > 20480
    assert engine.pairing_multi(a * k, b * k, n * k, inf * k) == oracle.fq12_pow(one, k)
    assert engine.pairing_multi_batch(a * k, b * k, n, k, inf * k) == one * k


def test_default_selection_for_a_batch_of_two_pair_verifications(engine, seeded_pairs, oracle):
    """10 000 groups of 2 pairs (BASELINE configs[3]'s verifies) take the small-group line-stream kernels by default:
    every 97th group against the oracle, and the whole output against the VM kernels"""
    from bls_py import _native
    g1, g2 = seeded_pairs
    groups = 10000
    a = (g1 * 20)[:96 * 2 * groups]
    b = (g2 * 20)[:192 * 2 * groups]
    out = engine.pairing_multi_batch(a, b, 2, groups)
    for g in range(0, groups, 97):
        assert out[576 * g:576 * (g + 1)] == oracle.pairing_multi(a[96 * 2 * g:96 * 2 * (g + 1)], b[192 * 2 * g:192 * 2 * (g + 1)], 2), g
    vm = engine_with_env({"BLSGPU_FEXP_WIDE": "0"})       # the wavefront VM end to end: Miller kernels AND its own final exponentiation
    vm.set_ls_threshold(None)
    vm.set_fexp_team_threshold(None)
    assert vm.pairing_multi_batch(a, b, 2, groups) == out


def test_single_call_beyond_one_slice_of_line_records(engine, seeded_pairs, golden, oracle):
    """one multi-pairing of 1100 x 1025 = 1 127 500 pairs: more than the 2^20 pairs whose line records (24 GB) one launch
    sequence keeps -- the call is cut into runs that each leave a partial; the result is the reference's 1025-pair
    value to the 1100th power"""
    g1, g2 = seeded_pairs
    k = 1100
    want = oracle.fq12_pow(bytes.fromhex(golden("pairing.json")["seeded"]["1025"]["out"]), k)
    assert engine.pairing_multi(g1 * k, g2 * k, 1025 * k) == want


def test_many_single_pair_groups_by_default(engine, seeded_pairs, oracle):
    """40 000 groups of ONE pair each (single pairings in bulk): line-stream small-group form + batched final
    exponentiation by default; against the VM kernels, and a sample against the oracle"""
    from bls_py import _native
    g1, g2 = seeded_pairs
    groups = 40000
    a, b = (g1 * 40)[:96 * groups], (g2 * 40)[:192 * groups]
    out = engine.pairing_multi_batch(a, b, 1, groups)
    for g in (0, 1, 1024, 1025, 39999):
        assert out[576 * g:576 * (g + 1)] == oracle.pairing_multi(a[96 * g:96 * (g + 1)], b[192 * g:192 * (g + 1)], 1), g
    assert out[:576 * 1025] == out[576 * 1025:576 * 2050]           # the inputs repeat with period 1025
    vm = engine_with_env({"BLSGPU_FEXP_WIDE": "0"})       # (an independent path: no register kernel anywhere in it)
    vm.set_ls_threshold(None)
    vm.set_fexp_team_threshold(None)
    assert vm.pairing_multi_batch(a[:96 * 3000], b[:192 * 3000], 1, 3000) == out[:576 * 3000]
