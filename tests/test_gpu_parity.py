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


# flag_on_valid: a VALID point carrying inf=True.  The reference lets Q's flag
# skip the chord updates (fields_t.py:676-677); the C ABI carries coordinates
# only (infinity = (0,0), fields_t.py:609-622), so that input is not expressible.
EDGE = ["empty", "p_inf", "q_inf", "both_inf", "q_inf_py_zero", "both_inf_in_batch", "p_zero_noflag",
        "q_zero_noflag", "mixed", "repeat", "q_and_negq", "p_and_negp"]


@pytest.mark.parametrize("name", EDGE)
def test_edge_cases(engine, golden, name):
    v = golden("pairing.json")["edge"][name]
    n = len(v["g1"])
    assert engine.pairing_multi(cat(v["g1"]), cat(v["g2"]), n).hex() == v["out"]


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
    with pytest.raises(ValueError):
        fh.fq_ate_pairing_multi((Ps[0],), ((Qs[0][0], Qs[0][1], True),))


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


@pytest.mark.parametrize("gsz,groups,mp", [(25, 5, 0), (25, 5, 1 << 30), (205, 5, 0), (1025, 3, 0), (64, 16, 1 << 30)])
def test_batched_large_groups(engine, seeded_pairs, gsz, groups, mp):
    """Groups long enough for the per-group product tree (grouped k_miller / k_miller_mp +
    k_reduce over blockIdx.y): every group must equal its own blsgpu_pairing_multi.
    gsz not a multiple of 3 or 4 exercises the ragged last team of every group."""
    g1, g2 = seeded_pairs
    n = gsz * groups
    reps = (n + 1024) // 1025
    a, b = (g1 * reps)[:96 * n], (g2 * reps)[:192 * n]
    try:
        engine.set_mp_threshold(mp)
        out = engine.pairing_multi_batch(a, b, gsz, groups)
        singles = [engine.pairing_multi(a[96 * gsz * g:96 * gsz * (g + 1)], b[192 * gsz * g:192 * gsz * (g + 1)], gsz)
                   for g in range(groups)]
    finally:
        engine.set_mp_threshold(4096)
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
