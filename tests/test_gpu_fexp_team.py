"""Parity of the batched final exponentiation (csrc/blsgpu_fexp.hip k_fexp_team: six lanes per result on the register
arithmetic) through the C ABI with the path FORCED for every batch size, against the reference's vectors, the oracle
and the wavefront-VM program.  Needs an MI355X."""
import json
import os
import random

import pytest

from conftest import GOLDEN, cat

pytestmark = pytest.mark.gpu
Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


@pytest.fixture(scope="module")
def fx():
    from bls_py import _native
    e = _native.Engine(0)
    e.set_fexp_team_threshold(1)
    return e


def test_reference_vectors_and_zero(fx, golden):
    g = golden("pairing.json")
    recs = [{"in": g["gen"]["miller"], "out": g["gen"]["final_exp"]}] + g["final_exp"]
    assert fx.final_exp(bytes.fromhex(g["gen"]["miller"])).hex() == g["gen"]["final_exp"]
    for rec in recs:
        assert fx.final_exp(bytes.fromhex(rec["in"])).hex() == rec["out"]
    assert fx.final_exp(bytes(576)) == bytes(576)                   # 0 -> 0: the inversion's 0^-1 := 0 (fields_t.py:47-55)


def test_batches_of_every_raggedness_vs_oracle_and_vm(fx, engine, oracle):
    rnd = random.Random(11)
    vals = [b"".join(rnd.randrange(Q).to_bytes(48, "big") for _ in range(12)) for _ in range(23)]
    vals[5] = bytes(576)                                            # a zero among them
    vals[9] = (1).to_bytes(48, "big") + bytes(48 * 11)              # and a one
    want = [oracle.final_exp(v) for v in vals]
    for m in (1, 2, 9, 10, 11, 20, 23):                             # ten results per wavefront
        out = fx.final_exp_batch(b"".join(vals[:m]))
        assert [out[576 * i:576 * (i + 1)] for i in range(m)] == want[:m], m
    big = b"".join(vals) * 30                                       # 690 results: the default selection of `engine` too
    assert fx.final_exp_batch(big) == engine.final_exp_batch(big) == b"".join(want) * 30


def test_pairing_batches_through_the_team_final_exponentiation(fx, seeded_pairs, oracle, golden):
    g1, g2 = seeded_pairs
    # groups whose partials are multiplied by the kernel first (one partial per pair, then per team of the VM kernels)
    for gsz, groups in ((1, 12), (2, 25), (3, 11), (7, 9), (27, 5), (205, 5)):
        m = gsz * groups
        out = fx.pairing_multi_batch(g1[:96 * m], g2[:192 * m], gsz, groups)
        for g in range(groups):
            assert out[576 * g:576 * (g + 1)] == oracle.pairing_multi(g1[96 * gsz * g:96 * gsz * (g + 1)], g2[192 * gsz * g:192 * gsz * (g + 1)], gsz, threads=8), (gsz, g)
    assert fx.pairing_multi(g1, g2, 1025).hex() == golden("pairing.json")["seeded"]["1025"]["out"]
    d = golden("pairing_degenerate.json")["cases"]["all_kinds"]
    n = len(d["g1"])
    inf = bytes(int(b) for pr in d["inf"] for b in pr)
    assert fx.pairing_multi(cat(d["g1"]), cat(d["g2"]), n, inf).hex() == d["out"]


def test_sharded_form_with_several_partials_per_result(fx, seeded_pairs, golden):
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
        fx.miller_product_batch_dev(t1.data_ptr(), t2.data_ptr(), hi - lo, B, parts[r * B * 144:(r + 1) * B * 144].data_ptr())
    fx.final_exp_product_batch_dev(parts.data_ptr(), W, B, out.data_ptr())
    torch.cuda.synchronize()
    res = bytes(out.cpu().numpy())
    assert all(res[576 * b:576 * (b + 1)].hex() == golden("pairing.json")["seeded"]["1025"]["out"] for b in range(B))
