"""Parity of the one-result-per-wavefront final exponentiation (csrc/blsgpu_fexpw.hip k_fexp_wide: every Fq product of
a step on its own lane; the default for calls that end in a few results) through the C ABI, against the reference's
final-exponentiation vectors (tests/golden/pairing.json, fq12_final_exp fields_t.py:1124-1128), the CPU oracle, the
six-lanes-per-result form and the wavefront-VM program.  Needs an MI355X."""
import os
import random

import pytest

from conftest import cat

pytestmark = pytest.mark.gpu
Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def _engine(env):
    from bls_py import _native
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _native.Engine(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def wide():
    """every call's final exponentiations one per wavefront, whatever their number"""
    e = _engine({"BLSGPU_FEXP_WIDE": "1"})
    e.set_fexp_team_threshold(None)
    return e


@pytest.fixture(scope="module")
def vm():
    """the wavefront-VM program only (rounds 1 - 3's path for a few results)"""
    e = _engine({"BLSGPU_FEXP_WIDE": "0"})
    e.set_fexp_team_threshold(None)
    return e


def test_reference_vectors_and_zero(wide, golden):
    g = golden("pairing.json")
    for rec in [{"in": g["gen"]["miller"], "out": g["gen"]["final_exp"]}] + g["final_exp"]:
        assert wide.final_exp(bytes.fromhex(rec["in"])).hex() == rec["out"]
    assert wide.final_exp(bytes(576)) == bytes(576)                 # 0 -> 0: the inversion's 0^-1 := 0 (fields_t.py:47-55)
    one = (1).to_bytes(48, "big") + bytes(48 * 11)
    assert wide.final_exp(one) == one


def test_batches_vs_oracle_team_form_and_vm(wide, vm, engine, oracle):
    rnd = random.Random(17)
    vals = [b"".join(rnd.randrange(Q).to_bytes(48, "big") for _ in range(12)) for _ in range(21)]
    vals[5] = bytes(576)
    vals[9] = (1).to_bytes(48, "big") + bytes(48 * 11)
    vals[11] = b"".join((Q - 1).to_bytes(48, "big") for _ in range(12))      # every coefficient q - 1
    want = [oracle.final_exp(v) for v in vals]
    for m in (1, 2, 7, 21):
        out = wide.final_exp_batch(b"".join(vals[:m]))
        assert [out[576 * i:576 * (i + 1)] for i in range(m)] == want[:m], m
    big = b"".join(vals) * 30                                       # 630 results: six lanes per result by default
    assert wide.final_exp_batch(big) == engine.final_exp_batch(big) == b"".join(want) * 30
    assert vm.final_exp_batch(b"".join(vals)) == b"".join(want)


def test_default_selection_is_the_wide_form_for_a_single_call(engine, wide, vm, seeded_pairs, golden):
    """a 1025-pair verification: the shared engine (defaults), the forced one and the VM program give the reference's bytes"""
    g1, g2 = seeded_pairs
    want = golden("pairing.json")["seeded"]["1025"]["out"]
    for e in (engine, wide, vm):
        assert e.pairing_multi(g1, g2, 1025).hex() == want
    v = golden("pairing.json")["small4"]
    for e in (engine, wide, vm):
        assert e.pairing_multi(cat(v["g1"]), cat(v["g2"]), 4).hex() == v["out"]


def test_pairing_batches_with_several_partials_per_result(wide, seeded_pairs, oracle, golden):
    g1, g2 = seeded_pairs
    # groups whose partials the kernel multiplies itself (one partial per pair up to 8; beyond, k_reduce folds first)
    for gsz, groups in ((1, 12), (2, 25), (3, 11), (7, 9), (8, 3), (9, 3), (27, 5), (205, 5)):
        m = gsz * groups
        out = wide.pairing_multi_batch(g1[:96 * m], g2[:192 * m], gsz, groups)
        for g in range(groups):
            assert out[576 * g:576 * (g + 1)] == oracle.pairing_multi(g1[96 * gsz * g:96 * gsz * (g + 1)], g2[192 * gsz * g:192 * gsz * (g + 1)], gsz, threads=8), (gsz, g)
    d = golden("pairing_degenerate.json")["cases"]
    for name in ("all_kinds", "ord13", "q_zero_p_order3"):
        n = len(d[name]["g1"])
        inf = bytes(int(b) for pr in d[name]["inf"] for b in pr)
        assert wide.pairing_multi(cat(d[name]["g1"]), cat(d[name]["g2"]), n, inf).hex() == d[name]["out"]
    # the line-stream path in front of it (one partial per group from k_ml_horner_wide)
    wide.set_ls_threshold(1, 1)
    try:
        assert wide.pairing_multi(g1, g2, 1025).hex() == golden("pairing.json")["seeded"]["1025"]["out"]
        out = wide.pairing_multi_batch(g1[:96 * 1000], g2[:192 * 1000], 100, 10)
        for g in range(10):
            assert out[576 * g:576 * (g + 1)] == oracle.pairing_multi(g1[9600 * g:9600 * (g + 1)], g2[19200 * g:19200 * (g + 1)], 100, threads=8)
    finally:
        wide.set_ls_threshold(16384, 64)


def test_sharded_form_with_several_partials_per_result(wide, seeded_pairs, golden):
    """blsgpu_final_exp_product_batch_dev with 4 partials per result (what an all-gather over 4 ranks hands over)"""
    import torch
    g1, g2 = seeded_pairs
    dev = torch.device("cuda", 0)
    up = lambda x: torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev)
    B, W = 3, 4
    parts = torch.zeros(W * B * 144, dtype=torch.int32, device=dev)
    out = torch.zeros(576 * B, dtype=torch.uint8, device=dev)
    cuts = [0, 200, 513, 800, 1025]
    keep = []
    for r in range(W):
        lo, hi = cuts[r], cuts[r + 1]
        t1, t2 = up(g1[96 * lo:96 * hi] * B), up(g2[192 * lo:192 * hi] * B)
        keep += [t1, t2]
        wide.miller_product_batch_dev(t1.data_ptr(), t2.data_ptr(), hi - lo, B, parts[r * B * 144:(r + 1) * B * 144].data_ptr())
    wide.final_exp_product_batch_dev(parts.data_ptr(), W, B, out.data_ptr())
    torch.cuda.synchronize()
    res = bytes(out.cpu().numpy())
    assert all(res[576 * b:576 * (b + 1)].hex() == golden("pairing.json")["seeded"]["1025"]["out"] for b in range(B))
