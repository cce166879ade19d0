"""G1 / G2 multi-scalar sums on the GPU (blsgpu_g1_msm / blsgpu_g2_msm through the
C ABI) against reference vectors and the CPU oracle.  Bit-exact affine bytes."""
import hashlib
import random

import pytest

from conftest import cat

pytestmark = pytest.mark.gpu
N = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def prf(tag, seed, i):
    h = hashlib.sha256(tag + seed.to_bytes(4, "big") + i.to_bytes(4, "big")).digest()
    return int.from_bytes(h, "big") % (N - 1) + 1


@pytest.fixture(scope="module", params=["default_selection", "without_the_sorted_buckets"])
def either_engine(request, engine):
    """Round 5: ONE G1 sum with scalars takes the sorted buckets at every size by default (1.2 - 1.5 ms against 1.6 - 2.6 of the
    other kernel families below 16 384 points); the second engine keeps those families -- double-and-add on the wavefront VM, LDS
    buckets, one window per lane -- under the same tests."""
    if request.param == "default_selection":
        return engine
    return request.getfixturevalue("fixed_window_engine")


@pytest.fixture(scope="module")
def fixed_window_engine():
    """the default selection without the sorted buckets: the independent kernels a sorted-bucket result is compared with"""
    return _engine_with_values({"BLSGPU_MSM_SORT_THRESHOLD": str(1 << 40), "BLSGPU_MSM_SORT2_THRESHOLD": str(1 << 40),
                                "BLSGPU_MSM_PLAIN_THRESHOLD": str(1 << 40), "BLSGPU_SMUL_MIN_GROUPS": str(1 << 40)})


def test_scalar_multiples_of_generators(either_engine, golden):
    engine = either_engine
    p = golden("points.json")
    g1, g2 = bytes.fromhex(p["g1"][0]["p"]), bytes.fromhex(p["g2"][0]["p"])
    ks = [int(r["k"], 16) for r in p["g1"]]
    out, inf = engine.g1_msm(g1 * len(ks), ks, 1, len(ks))           # len(ks) groups of one point
    assert out.hex() == "".join(r["p"] for r in p["g1"]) and not any(inf)
    out, inf = engine.g2_msm(g2 * len(ks), ks, 1, len(ks))
    assert out.hex() == "".join(r["p"] for r in p["g2"]) and not any(inf)
    a = p["g1_add"]
    assert engine.g1_msm(bytes.fromhex(a["a"] + a["b"]), None, 2)[0].hex() == a["sum"]
    assert engine.g1_msm(bytes.fromhex(a["a"] + a["a"]), None, 2)[0].hex() == a["dbl"]      # doubling inside add
    a = p["g2_add"]
    assert engine.g2_msm(bytes.fromhex(a["a"] + a["b"]), None, 2)[0].hex() == a["sum"]
    assert engine.g2_msm(bytes.fromhex(a["a"] + a["a"]), None, 2)[0].hex() == a["dbl"]


def test_infinity_and_degenerate_cases(either_engine, golden):
    engine = either_engine
    p = golden("points.json")
    P = bytes.fromhex(p["g1"][3]["p"])
    negP = P[:48] + ((Q - int.from_bytes(P[48:], "big")) % Q).to_bytes(48, "big")
    out, inf = engine.g1_msm(P + negP, None, 2)
    assert out == bytes(96) and inf == [True]                         # P + (-P)
    out, inf = engine.g1_msm(P + P, [5, N - 5], 2)
    assert out == bytes(96) and inf == [True]
    out, inf = engine.g1_msm(P + bytes(96), [7, 9], 2)                # (0,0) input = infinity
    assert out == engine.g1_msm(P, [7], 1)[0] and inf == [False]
    out, inf = engine.g1_msm(P * 3, [0, 0, 0], 3)                     # c == 0 -> infinity (fields_t.py:710)
    assert out == bytes(96) and inf == [True]
    out, inf = engine.g1_msm(b"", None, 0, 2)                         # empty sums
    assert out == bytes(192) and inf == [True, True]
    Q2 = bytes.fromhex(p["g2"][2]["p"])
    out, inf = engine.g2_msm(Q2 + bytes(192), [3, 4], 2)
    assert out == engine.g2_msm(Q2, [3], 1)[0]


@pytest.mark.parametrize("k,groups", [(1, 1), (5, 3), (6, 1), (7, 2), (13, 5), (25, 2), (67, 1)])
def test_random_sums_vs_oracle(either_engine, oracle, seeded_pairs, k, groups):
    engine = either_engine
    g1, g2 = seeded_pairs
    rnd = random.Random(k * 100 + groups)
    n = k * groups
    pts1, pts2 = g1[96 * 10:96 * (10 + n)], g2[192 * 10:192 * (10 + n)]
    sc = [rnd.randrange(N) if rnd.random() < 0.8 else rnd.randrange(1 << 40) for _ in range(n)]
    out1, _ = engine.g1_msm(pts1, sc, k, groups)
    out2, _ = engine.g2_msm(pts2, sc, k, groups)
    for g in range(groups):
        w1, _ = oracle.g1_msm(pts1[96 * k * g:96 * k * (g + 1)], sc[k * g:k * (g + 1)], k)
        w2, _ = oracle.g2_msm(pts2[192 * k * g:192 * k * (g + 1)], sc[k * g:k * (g + 1)], k)
        assert out1[96 * g:96 * (g + 1)] == w1
        assert out2[192 * g:192 * (g + 1)] == w2


def _compress_g1(pt):
    b = bytearray(pt[:48])
    if int.from_bytes(pt[48:], "big") > Q // 2:
        b[0] |= 0x80
    return bytes(b)


@pytest.mark.parametrize("n", [16, 1024])
def test_aggregate_pub_keys_vectors(engine, golden, n):
    """bls.py:203-223 as a G1 MSM: keys a_i*G1 (PRF scalars), sorted by their
    compressed bytes, t_i = hash_pks; both the secure and the plain sum."""
    rec = golden("msm.json")[str(n)]
    g = bytes.fromhex(golden("points.json")["g1"][0]["p"])
    sks = [prf(b"blsgpu/a", 1, i) for i in range(n)]
    pks, inf = engine.g1_msm(g * n, sks, 1, n)                       # n scalar multiplications
    assert hashlib.sha256(pks).hexdigest() == rec["sha256_inputs"] and not any(inf)
    pts = sorted((pks[96 * i:96 * (i + 1)] for i in range(n)), key=_compress_g1)
    ser = [_compress_g1(p) for p in pts]
    assert hashlib.sha256("".join(s.hex() for s in ser).encode()).hexdigest() == rec["sha256_sorted_ser"]
    digest = hashlib.sha256(b"".join(ser)).digest()
    ts = [int.from_bytes(hashlib.sha256(i.to_bytes(4, "big") + digest).digest(), "big") % N for i in range(n)]
    assert hashlib.sha256(b"".join(t.to_bytes(32, "big") for t in ts)).hexdigest() == rec["sha256_scalars"]
    out, _ = engine.g1_msm(b"".join(pts), ts, n)
    assert out.hex() == rec["secure_affine"]
    out, _ = engine.g1_msm(b"".join(pts), None, n)
    assert out.hex() == rec["simple_affine"]


def test_threshold_combine_batched(engine, golden):
    """threshold.py:127-136: 67 unit signatures x Lagrange weights, as a batch of groups."""
    rec = golden("threshold.json")["67_of_100"]
    pts = cat(rec["unit_sigs_affine"])
    lam = [int(x, 16) for x in rec["lambdas"]]
    groups = 5
    out, inf = engine.g2_msm(pts * groups, lam * groups, 67, groups)
    assert out == bytes.fromhex(rec["combined_affine"]) * groups and not any(inf)
    small = golden("threshold.json")["3_of_5"]
    out, _ = engine.g2_msm(cat(small["unit_sigs_affine"]), [int(x, 16) for x in small["lambdas"]], 3)
    assert out.hex() == small["combined_affine"]


def _engine_with(names):
    """an engine with the named thresholds at 1; unless the sorted buckets are what is asked for they are switched off, so that the
    fixture's kernel family also serves the single G1 sums with scalars"""
    values = {k: "1" for k in names}
    values.setdefault("BLSGPU_MSM_SORT_THRESHOLD", str(1 << 40))
    values.setdefault("BLSGPU_MSM_SORT2_THRESHOLD", str(1 << 40))
    values.setdefault("BLSGPU_MSM_PLAIN_THRESHOLD", str(1 << 40))
    values.setdefault("BLSGPU_SMUL_MIN_GROUPS", str(1 << 40))
    return _engine_with_values(values)


@pytest.fixture(scope="module")
def lane_engine():
    """An engine whose sums use the bucket method with one (group, chunk, window) per lane from 1 point on."""
    return _engine_with(("BLSGPU_PIP_THRESHOLD", "BLSGPU_PIP_GROUP_THRESHOLD", "BLSGPU_MSM_LANE_THRESHOLD"))


@pytest.fixture(scope="module")
def pip_engine():
    """An engine whose single sums use the bucket method from 1 point on."""
    return _engine_with(("BLSGPU_PIP_THRESHOLD", "BLSGPU_PIP_GROUP_THRESHOLD"))


@pytest.mark.parametrize("k", [1, 5, 6, 7, 383, 384, 385, 1000])
def test_bucket_method_vs_oracle(pip_engine, oracle, seeded_pairs, k):
    """k_msm_pip (Pippenger, buckets in LDS) on ragged sizes (chunk = 384 points at these
    sizes): full-range, short, zero and maximal scalars; G1 and (up to 385) G2."""
    g1, g2 = seeded_pairs
    rnd = random.Random(k)
    pts1 = (g1 * 2)[96 * 3:96 * (3 + k)]
    sc = [rnd.choice([rnd.randrange(N), rnd.randrange(1 << 40), 0, N - 1, 1, (1 << 255) - 19]) for _ in range(k)]
    out1, inf1 = pip_engine.g1_msm(pts1, sc, k)
    w1, _ = oracle.g1_msm(pts1, sc, k)
    assert out1 == w1 and inf1 == [w1 == bytes(96)]
    if k <= 385:
        pts2 = (g2 * 2)[192 * 3:192 * (3 + k)]
        out2, _ = pip_engine.g2_msm(pts2, sc, k)
        assert out2 == oracle.g2_msm(pts2, sc, k)[0]


def test_bucket_method_degenerate(pip_engine, engine, golden):
    p = golden("points.json")
    P = bytes.fromhex(p["g1"][3]["p"])
    negP = P[:48] + ((Q - int.from_bytes(P[48:], "big")) % Q).to_bytes(48, "big")
    assert pip_engine.g1_msm(P + negP, None, 2) == (bytes(96), [True])                  # plain sum, P + (-P)
    assert pip_engine.g1_msm(P * 3, [0, 0, 0], 3) == (bytes(96), [True])                # all-zero scalars
    assert pip_engine.g1_msm(P + bytes(96), [7, 9], 2) == engine.g1_msm(P, [7], 1)      # (0,0) input = infinity
    assert pip_engine.g1_msm(P * 40, [5] * 40, 40) == engine.g1_msm(P, [200], 1)        # one bucket, many points
    assert pip_engine.g1_msm(P * 7, None, 7) == engine.g1_msm(P, [7], 1)


def test_aggregate_pub_keys_1024_bucket_method(pip_engine, golden):
    """The reference's aggregate_pub_keys vector (bls.py:203-223) through the bucket method."""
    rec = golden("msm.json")["1024"]
    g = bytes.fromhex(golden("points.json")["g1"][0]["p"])
    n = 1024
    sks = [prf(b"blsgpu/a", 1, i) for i in range(n)]
    pks, _ = pip_engine.g1_msm(g * n, sks, 1, n)
    pts = sorted((pks[96 * i:96 * (i + 1)] for i in range(n)), key=_compress_g1)
    digest = hashlib.sha256(b"".join(_compress_g1(p) for p in pts)).digest()
    ts = [int.from_bytes(hashlib.sha256(i.to_bytes(4, "big") + digest).digest(), "big") % N for i in range(n)]
    assert pip_engine.g1_msm(b"".join(pts), ts, n)[0].hex() == rec["secure_affine"]
    assert pip_engine.g1_msm(b"".join(pts), None, n)[0].hex() == rec["simple_affine"]


@pytest.mark.parametrize("k,groups", [(1, 3), (5, 4), (7, 2), (67, 3), (100, 2)])
def test_bucket_method_batches_vs_oracle(pip_engine, oracle, seeded_pairs, k, groups):
    """Batches of independent sums through the bucket kernels (one chunk per group)."""
    g1, g2 = seeded_pairs
    rnd = random.Random(k * 7 + groups)
    n = k * groups
    pts1, pts2 = g1[96 * 20:96 * (20 + n)], g2[192 * 20:192 * (20 + n)]
    sc = [rnd.choice([rnd.randrange(N), rnd.randrange(1 << 40), 0, N - 1]) for _ in range(n)]
    out1, inf1 = pip_engine.g1_msm(pts1, sc, k, groups)
    out2, inf2 = pip_engine.g2_msm(pts2, sc, k, groups)
    for g in range(groups):
        w1, _ = oracle.g1_msm(pts1[96 * k * g:96 * k * (g + 1)], sc[k * g:k * (g + 1)], k)
        w2, _ = oracle.g2_msm(pts2[192 * k * g:192 * k * (g + 1)], sc[k * g:k * (g + 1)], k)
        assert out1[96 * g:96 * (g + 1)] == w1 and inf1[g] == (w1 == bytes(96))
        assert out2[192 * g:192 * (g + 1)] == w2 and inf2[g] == (w2 == bytes(192))


@pytest.mark.parametrize("k,groups", [(1, 1), (7, 1), (200, 1), (1000, 1), (5, 4), (67, 3)])
def test_lane_bucket_method_vs_oracle(lane_engine, oracle, seeded_pairs, k, groups):
    """k_msm_lane (buckets in HBM, one window per lane; chunks of 1 point at these sizes for a
    single sum, so the chunk fold kernel runs too), G1 and G2."""
    g1, g2 = seeded_pairs
    rnd = random.Random(k * 11 + groups)
    n = k * groups
    pts1, pts2 = (g1 * 2)[96 * 5:96 * (5 + n)], (g2 * 2)[192 * 5:192 * (5 + n)]
    sc = [rnd.choice([rnd.randrange(N), rnd.randrange(1 << 40), 0, N - 1, 1]) for _ in range(n)]
    out1, inf1 = lane_engine.g1_msm(pts1, sc, k, groups)
    for g in range(groups):
        w1, _ = oracle.g1_msm(pts1[96 * k * g:96 * k * (g + 1)], sc[k * g:k * (g + 1)], k)
        assert out1[96 * g:96 * (g + 1)] == w1 and inf1[g] == (w1 == bytes(96))
    if n <= 400:
        out2, _ = lane_engine.g2_msm(pts2, sc, k, groups)
        for g in range(groups):
            assert out2[192 * g:192 * (g + 1)] == oracle.g2_msm(pts2[192 * k * g:192 * k * (g + 1)], sc[k * g:k * (g + 1)], k)[0]


def test_lane_bucket_method_scalars_beyond_the_signed_digit_range(lane_engine, oracle, seeded_pairs):
    """k_msm_lane recodes scalars below 2^256 - 0x88..8 into signed nibbles (8 buckets); larger ones keep plain
    nibbles (15 buckets), also mixed inside one chunk: the boundary values and 2^256 - 1 against the oracle."""
    g1, g2 = seeded_pairs
    edge = [(1 << 256) - 1, 0x77777777 << 224, (0x77777777 << 224) - 1, int("7" * 64, 16), int("7" * 64, 16) + 1, 1 << 255,
            N - 1, 5, 0, int("8" * 64, 16), int("f" * 63 + "8", 16)]
    for k, groups in ((len(edge), 1), (3, 3)):
        n = k * groups
        pts1, pts2 = g1[96 * 2:96 * (2 + n)], g2[192 * 2:192 * (2 + n)]
        sc = (edge * 2)[:n]
        out1, _ = lane_engine.g1_msm(pts1, sc, k, groups)
        out2, _ = lane_engine.g2_msm(pts2, sc, k, groups)
        for g in range(groups):
            assert out1[96 * g:96 * (g + 1)] == oracle.g1_msm(pts1[96 * k * g:96 * k * (g + 1)], sc[k * g:k * (g + 1)], k)[0]
            assert out2[192 * g:192 * (g + 1)] == oracle.g2_msm(pts2[192 * k * g:192 * k * (g + 1)], sc[k * g:k * (g + 1)], k)[0]


def test_lane_bucket_method_degenerate(lane_engine, engine, golden):
    p = golden("points.json")
    P = bytes.fromhex(p["g1"][3]["p"])
    negP = P[:48] + ((Q - int.from_bytes(P[48:], "big")) % Q).to_bytes(48, "big")
    assert lane_engine.g1_msm(P + negP, None, 2) == (bytes(96), [True])
    assert lane_engine.g1_msm(P * 3, [0, 0, 0], 3) == (bytes(96), [True])
    assert lane_engine.g1_msm(P + bytes(96), [7, 9], 2) == engine.g1_msm(P, [7], 1)
    assert lane_engine.g1_msm(P * 40, [5] * 40, 40) == engine.g1_msm(P, [200], 1)


@pytest.fixture(scope="module")
def horner_engine():
    """Batches of G2 sums through the bucket kernels with the window Horner of SEVERAL sums per team (k_msm_horner_np,
    what a batch of >= 1024 sums selects) from 1 sum on."""
    return _engine_with(("BLSGPU_PIP_GROUP_THRESHOLD", "BLSGPU_HORNER_NP_THRESHOLD"))


@pytest.mark.parametrize("k,groups", [(3, 2), (5, 4), (7, 5), (4, 6), (3, 11), (67, 7)])
def test_batch_horner_vs_oracle(horner_engine, oracle, seeded_pairs, k, groups):
    """ragged team counts (5 sums per team), a sum that is the point at infinity, zero and extreme scalars"""
    _, g2 = seeded_pairs
    rnd = random.Random(k * 17 + groups)
    n = k * groups
    pts = bytearray((g2 * 2)[192 * 4:192 * (4 + n)])
    sc = [rnd.choice([rnd.randrange(N), rnd.randrange(1 << 40), 0, N - 1, 1]) for _ in range(n)]
    if groups >= 4:                                             # sum number 2: P, -P and zeros -> infinity
        P = bytes(pts[192 * 2 * k:192 * 2 * k + 192])
        negP = P[:96] + b"".join(((Q - int.from_bytes(P[96 + 48 * j:144 + 48 * j], "big")) % Q).to_bytes(48, "big") for j in range(2))
        pts[192 * (2 * k + 1):192 * (2 * k + 2)] = negP
        for i in range(k):
            sc[2 * k + i] = 9 if i < 2 else 0
    out, inf = horner_engine.g2_msm(bytes(pts), sc, k, groups)
    for g in range(groups):
        want, _ = oracle.g2_msm(bytes(pts[192 * k * g:192 * k * (g + 1)]), sc[k * g:k * (g + 1)], k)
        assert out[192 * g:192 * (g + 1)] == want and inf[g] == (want == bytes(192)), g
    if groups >= 4:
        assert inf[2]


def _engine_with_values(values):
    import os
    from bls_py import _native
    old = {k: os.environ.get(k) for k in values}
    os.environ.update(values)
    try:
        e = _native.Engine(0)
    finally:
        for k in values:
            if old[k] is None:
                del os.environ[k]
            else:
                os.environ[k] = old[k]
    return e


@pytest.fixture(scope="module")
def lane_np_engine():
    """The lane-pair bucket kernel with the window Horner of round 3 (VM form, k_msm_pip_windows + k_msm_horner_np /
    k_msm_pip_horner): what the lane path ran before the Horner on lane quads, kept under test as the unselected form."""
    return _engine_with_values({"BLSGPU_PIP_THRESHOLD": "1", "BLSGPU_PIP_GROUP_THRESHOLD": "1", "BLSGPU_MSM_LANE_THRESHOLD": "1",
                                "BLSGPU_HORNER_NP_THRESHOLD": "1", "BLSGPU_HORNER_QUADS_THRESHOLD": "1000000000",
                                "BLSGPU_MSM_SORT_THRESHOLD": str(1 << 40), "BLSGPU_MSM_SORT2_THRESHOLD": str(1 << 40)})


@pytest.mark.parametrize("k,groups", [(3, 2), (5, 4), (4, 16), (3, 17), (2, 33), (67, 7)])
def test_batch_horner_on_lane_quads_vs_oracle(lane_engine, lane_np_engine, oracle, seeded_pairs, k, groups):
    """k_msm_horner_quads (a batch of G2 sums on the lane-pair bucket kernel: one sum per lane quad, window sums read in
    the L28 form): ragged quad counts (16 sums per wavefront), a sum that is the point at infinity (flag and (0, 0)),
    zero / short / extreme scalars -- against the oracle and the VM-form Horner."""
    _, g2 = seeded_pairs
    rnd = random.Random(k * 19 + groups)
    n = k * groups
    pts = bytearray((g2 * 3)[192 * 5:192 * (5 + n)])
    sc = [rnd.choice([rnd.randrange(N), rnd.randrange(1 << 40), 0, N - 1, 1, (1 << 256) - 1]) for _ in range(n)]
    if groups >= 4:                                             # sum number 2: P, -P and zeros -> infinity
        P = bytes(pts[192 * 2 * k:192 * 2 * k + 192])
        negP = P[:96] + b"".join(((Q - int.from_bytes(P[96 + 48 * j:144 + 48 * j], "big")) % Q).to_bytes(48, "big") for j in range(2))
        pts[192 * (2 * k + 1):192 * (2 * k + 2)] = negP
        for i in range(k):
            sc[2 * k + i] = 9 if i < 2 else 0
    out, inf = lane_engine.g2_msm(bytes(pts), sc, k, groups)
    assert (out, inf) == lane_np_engine.g2_msm(bytes(pts), sc, k, groups)
    for g in range(groups if groups <= 17 else 6):
        want, _ = oracle.g2_msm(bytes(pts[192 * k * g:192 * k * (g + 1)]), sc[k * g:k * (g + 1)], k)
        assert out[192 * g:192 * (g + 1)] == want and inf[g] == (want == bytes(192)), g
    if groups >= 4:
        assert inf[2] and out[192 * 2:192 * 3] == bytes(192)


def test_batch_of_plain_g2_sums_on_the_lane_kernels(lane_engine, oracle, seeded_pairs):
    """no scalars (the sums of aggregate_sigs): every window but the lowest sums to infinity, so the window Horner on
    lane quads doubles and adds infinities 63 times before the one real window; a group of P, -P sums to infinity"""
    _, g2 = seeded_pairs
    k, groups = 3, 9
    pts = bytearray((g2 * 2)[192 * 11:192 * (11 + k * groups)])
    P = bytes(pts[192 * 4 * k:192 * 4 * k + 192])
    negP = P[:96] + b"".join(((Q - int.from_bytes(P[96 + 48 * j:144 + 48 * j], "big")) % Q).to_bytes(48, "big") for j in range(2))
    pts[192 * (4 * k + 1):192 * (4 * k + 2)] = negP
    pts[192 * (4 * k + 2):192 * (4 * k + 3)] = bytes(192)           # group 4: P, -P, infinity
    out, inf = lane_engine.g2_msm(bytes(pts), None, k, groups)
    for g in range(groups):
        want, _ = oracle.g2_msm(bytes(pts[192 * k * g:192 * k * (g + 1)]), None, k)
        assert out[192 * g:192 * (g + 1)] == want and inf[g] == (want == bytes(192)), g
    assert inf[4]


def test_batch_horner_on_lane_quads_workgroup_shapes(lane_engine, seeded_pairs):
    """k_msm_horner_quads as 64-thread workgroups (BLSGPU_WG256_MAX_WAVES=0) and as the default 256-thread ones: the same
    bytes, ragged quad counts"""
    e64 = _engine_with_values({"BLSGPU_PIP_THRESHOLD": "1", "BLSGPU_PIP_GROUP_THRESHOLD": "1", "BLSGPU_MSM_LANE_THRESHOLD": "1",
                               "BLSGPU_WG256_MAX_WAVES": "0"})
    _, g2 = seeded_pairs
    rnd = random.Random(77)
    for k, groups in ((3, 17), (2, 67), (67, 7)):
        n = k * groups
        pts = (g2 * 3)[192 * 9:192 * (9 + n)]
        sc = [rnd.choice([rnd.randrange(N), rnd.randrange(1 << 40), 0, N - 1]) for _ in range(n)]
        assert e64.g2_msm(pts, sc, k, groups) == lane_engine.g2_msm(pts, sc, k, groups)


@pytest.fixture(scope="module", params=["tail_on_the_wide_machine", "tail_on_the_wavefront_vm"])
def sorted_engine(request):
    """An engine whose single G1 sums with scalars use the sorted buckets (k_srt_*) from 1 point on -- once with the default tail
    (window sums and the Horner over the windows on k_msm_horner_wide) and once with the wavefront VM's (k_srt_windows +
    k_msm_pip_horner<1>, BLSGPU_MSM_WIDE_TAIL=0)."""
    if request.param == "tail_on_the_wavefront_vm":
        return _engine_with_values({"BLSGPU_MSM_SORT_THRESHOLD": "1", "BLSGPU_MSM_WIDE_TAIL": "0"})
    return _engine_with(("BLSGPU_MSM_SORT_THRESHOLD",))


@pytest.mark.parametrize("k", [1, 2, 65, 700, 3000])
def test_sorted_buckets_vs_oracle(sorted_engine, oracle, seeded_pairs, k):
    """k_srt_count/scan/scatter/accum/fix/bits/windows (12-bit windows at these sizes: the default below 2^18 points)
    against the oracle:
    random, short, zero and extreme scalars, a point at infinity and repeated points in the list."""
    g1, _ = seeded_pairs
    rnd = random.Random(k * 13 + 1)
    pts = bytearray((g1 * 3)[96 * 3:96 * (3 + k)])
    sc = [rnd.choice([rnd.randrange(N), rnd.randrange(N), rnd.randrange(1 << 40), 0, N - 1, 1, (1 << 256) - 1]) for _ in range(k)]
    if k >= 65:
        pts[96 * 7:96 * 8] = bytes(96)                          # infinity in the list
        pts[96 * 9:96 * 10] = pts[96 * 8:96 * 9]                # the same point twice ...
        sc[9] = sc[8] = rnd.randrange(N)                        # ... in the same buckets (doubling inside the addition)
    got, inf = sorted_engine.g1_msm(bytes(pts), sc, k, 1)
    want, _ = oracle.g1_msm(bytes(pts), sc, k)
    assert got == want and inf[0] == (want == bytes(96))


def test_sorted_buckets_degenerate(sorted_engine, fixed_window_engine, golden):
    engine = fixed_window_engine
    p = golden("points.json")
    P = bytes.fromhex(p["g1"][3]["p"])
    negP = P[:48] + ((Q - int.from_bytes(P[48:], "big")) % Q).to_bytes(48, "big")
    assert sorted_engine.g1_msm(P + negP, [5, 5], 2) == (bytes(96), [True])
    assert sorted_engine.g1_msm(P * 3, [0, 0, 0], 3) == (bytes(96), [True])
    assert sorted_engine.g1_msm(P + bytes(96), [7, 9], 2) == engine.g1_msm(P, [7], 1)
    # every point in the same buckets: the list is one run per window
    assert sorted_engine.g1_msm(P * 400, [5] * 400, 400) == engine.g1_msm(P, [2000], 1)
    assert sorted_engine.g1_msm(P * 40, [N - 1] * 40, 40) == engine.g1_msm(P, [(N - 1) * 40 % N], 1)


def test_sorted_buckets_one_scalar_for_all(sorted_engine, fixed_window_engine, seeded_pairs):
    """All points share ONE scalar: every window's list is a single run that spans thousands of the equal pieces --
    k_srt_fix_long's strided sums and butterfly finish it (round 3: no host-side guard, no fall-back to the fixed
    windows, the call never synchronises).  sum s P_i = s (sum P_i), the plain sum through the fixed-window kernels."""
    g1, _ = seeded_pairs
    k = 6000
    pts = (g1 * 6)[:96 * k]
    for s in (0x1234567, N - 2):
        plain, _ = fixed_window_engine.g1_msm(pts, None, k, 1)
        want, _ = fixed_window_engine.g1_msm(plain, [s], 1, 1)
        assert sorted_engine.g1_msm(pts, [s] * k, k, 1) == (want, [False])


@pytest.mark.parametrize("n", [20000, 100000])
def test_sorted_buckets_default_selection_mid_sizes(engine, fixed_window_engine, golden, n):
    """The DEFAULT selection (sorted buckets, 13-bit windows of signed digits) at sizes between the test vectors and configs[4]:
    n different points a_i G with PRF scalars t_i, checked by  sum t_i (a_i G) = (sum t_i a_i) G  with the right-hand side from the
    fixed-window kernels."""
    gen1 = bytes.fromhex(golden("pairing.json")["gen"]["g1"])
    a = [_prf(b"blsgpu/a", 7, i) for i in range(n)]
    t = [_prf(b"blsgpu/t", 7, i) for i in range(n)]
    pts, _ = engine.g1_msm(gen1 * n, a, 1, n)
    got, inf = engine.g1_msm(pts, t, n, 1)
    want, _ = fixed_window_engine.g1_msm(gen1, [sum(x * y for x, y in zip(a, t)) % N_ORDER], 1, 1)
    assert got == want and not inf[0]


def test_reference_1024_key_aggregate_through_sorted_buckets(sorted_engine, golden):
    """BLS.aggregate_pub_keys(secure) of the reference's 1024 keys (msm.json, reference-generated) through the sorted
    buckets (the engine of C5), and 16 copies of the list (16 384 points: the size from which it is the default)."""
    import hashlib
    rec = golden("msm.json")["1024"]
    g = bytes.fromhex(golden("points.json")["g1"][0]["p"])
    n = 1024
    sks = [_prf(b"blsgpu/a", 1, i) for i in range(n)]
    pks, _ = sorted_engine.g1_msm(g * n, sks, 1, n)
    pts = sorted((pks[96 * i:96 * (i + 1)] for i in range(n)), key=_compress_g1)
    digest = hashlib.sha256(b"".join(_compress_g1(p) for p in pts)).digest()
    ts = [int.from_bytes(hashlib.sha256(i.to_bytes(4, "big") + digest).digest(), "big") % N for i in range(n)]
    assert sorted_engine.g1_msm(b"".join(pts), ts, n)[0].hex() == rec["secure_affine"]
    want, _ = sorted_engine.g1_msm(bytes.fromhex(rec["secure_affine"]), [16], 1, 1)
    assert sorted_engine.g1_msm(b"".join(pts) * 16, ts * 16, 16 * n)[0] == want


N_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def _prf(tag, seed, i):
    import hashlib
    return int.from_bytes(hashlib.sha256(tag + seed.to_bytes(4, "big") + i.to_bytes(4, "big")).digest(), "big") % (N_ORDER - 1) + 1


def test_c4_full_size_10000_groups(engine, golden):
    """BASELINE configs[3] at full size with the DEFAULT kernel selection: 10 000 threshold groups of
    k = 67 shares.  Group g's Lagrange weights are the reference's times c_g, so its combine must be
    c_g x the reference's combined signature (threshold.json, reference-generated) and
    e(-G1, sig_g) e(c_g pk, H(m)) must be one for every group."""
    from bls_py import hostmath as H, util
    from bls_py.keys import PublicKey
    th = golden("threshold.json")["67_of_100"]
    groups, k = 10000, 67
    pts = b"".join(bytes.fromhex(s) for s in th["unit_sigs_affine"])
    lam = [int(x, 16) for x in th["lambdas"]]
    cg = [1] + [_prf(b"blsgpu/c4", 1, g) for g in range(1, groups)]
    sc = b"".join(((l * c) % N_ORDER).to_bytes(32, "big") for c in cg for l in lam)
    got, inf = engine.g2_msm(pts * groups, sc, k, groups)
    assert not any(inf)
    gold_pt = bytes.fromhex(th["combined_affine"])
    assert got[:192] == gold_pt                                       # c_0 = 1: the reference's own combine
    want, _ = engine.g2_msm(gold_pt * groups, cg, 1, groups)
    assert got == want
    pk = PublicKey.from_bytes(bytes.fromhex(th["master_pk"])).value.to_affine()._aff()
    hm = H.g2_affine_bytes(H.hash_to_g2_prehashed(util.hash256(bytes.fromhex(th["msg"])), util.hash512))
    ng1 = H.g1_affine_bytes(H.jac_to_affine(H.F1, H.jac_mul(H.F1, H.aff_to_jac(H.F1, H.G1_GEN), N_ORDER - 1)))
    pks, _ = engine.g1_msm(H.g1_affine_bytes(pk) * groups, cg, 1, groups)
    pg1 = b"".join(ng1 + pks[96 * g:96 * (g + 1)] for g in range(groups))
    pg2 = b"".join(got[192 * g:192 * (g + 1)] + hm for g in range(groups))
    one = (1).to_bytes(48, "big") + bytes(48 * 11)
    assert engine.pairing_multi_batch(pg1, pg2, 2, groups) == one * groups
    # one tampered share: that group, and only that group, stops verifying
    bad = bytearray(pg2)
    bad[192 * 2 * 777:192 * 2 * 777 + 192] = got[192 * 778:192 * 779]
    res = engine.pairing_multi_batch(pg1, bytes(bad), 2, groups)
    assert [g for g in range(groups) if res[576 * g:576 * (g + 1)] != one] == [777]


def test_c5_full_size_distinct_points(engine, golden):
    """BASELINE configs[4] at full size with the DEFAULT kernel selection: one G1 multi-scalar sum over
    2^20 DIFFERENT points a_i G (PRF scalars) against the REFERENCE's own sum over the same points and scalars
    (tests/golden/msm_seeded_1048576.json: 2^21 scalar multiplications of the reference's pure Python, fields_t.py:705-741 and
    :762-797 through ec.py, generated by tests/golden/make_golden.py msm_seeded), the reference's digests of the first 16 384
    input points and six sample points; and by  sum t_i (a_i G) = (sum t_i a_i) G."""
    import hashlib
    gen1 = bytes.fromhex(golden("pairing.json")["gen"]["g1"])
    n = 1 << 20
    a = [_prf(b"blsgpu/a", 5, i) for i in range(n)]
    t = [_prf(b"blsgpu/t", 5, i) for i in range(n)]
    pts = b""
    for lo in range(0, n, 1 << 18):
        p, _ = engine.g1_msm(gen1 * (1 << 18), a[lo:lo + (1 << 18)], 1, 1 << 18)
        pts += p
    sample = [pts[96 * i:96 * (i + 1)] for i in range(0, n, 257)]
    assert len(set(sample)) == len(sample)
    fx = golden("msm_seeded_1048576.json")
    assert fx["n"] == n and fx["seed"] == 5
    for i, want_pt in fx["sample_points"].items():                 # the inputs as the reference computes them
        assert pts[96 * int(i):96 * (int(i) + 1)].hex() == want_pt, i
    step = fx["chunk"]
    for k, d in enumerate(fx["chunk_point_digests_first_4"]):
        assert hashlib.sha256(pts[96 * step * k:96 * step * (k + 1)]).hexdigest() == d, k
    got, inf = engine.g1_msm(pts, t, n, 1)
    assert got.hex() == fx["sum_affine"] and not inf[0]           # the reference's sum
    want, _ = engine.g1_msm(gen1, [sum(x * y for x, y in zip(a, t)) % N_ORDER], 1, 1)
    assert got == want
    # plain sum (scalars = NULL) of the same points
    got, _ = engine.g1_msm(pts, None, n, 1)
    want, _ = engine.g1_msm(gen1, [sum(a) % N_ORDER], 1, 1)
    assert got == want


# ---- ONE G2 sum with scalars on the sorted buckets (round 5: BLS.aggregate_sigs(secure), bls.py:225-261, as a multi-scalar sum) ----
@pytest.fixture(scope="module", params=["default_window_bits", "5_bit_windows", "11_bit_windows", "13_bit_windows"])
def sorted_g2_engine(request):
    """The default selection sends every single G2 sum with scalars to the sorted buckets (k_srt_*<2> on lane pairs, the tail on
    k_msm_horner_wide2); the window width follows the size by default, the other rows pin it (every fold shape of the tail)."""
    if request.param == "default_window_bits":
        return _engine_with_values({})
    return _engine_with_values({"BLSGPU_MSM_SORT2_BITS": request.param.split("_")[0]})


def _neg_g2(P):
    return P[:96] + b"".join(((Q - int.from_bytes(P[96 + 48 * j:144 + 48 * j], "big")) % Q).to_bytes(48, "big") for j in range(2))


@pytest.mark.parametrize("k", [1, 2, 65, 700])
def test_sorted_buckets_g2_vs_oracle(sorted_g2_engine, oracle, seeded_pairs, k):
    """random, short, zero and extreme scalars, a point at infinity, a repeated point and a point with its negative in the same
    buckets, against the oracle's double-and-add"""
    _, g2 = seeded_pairs
    rnd = random.Random(k * 29 + 3)
    pts = bytearray((g2 * 2)[192 * 5:192 * (5 + k)])
    sc = [rnd.choice([rnd.randrange(N), rnd.randrange(N), rnd.randrange(1 << 40), 0, N - 1, 1, (1 << 256) - 1]) for _ in range(k)]
    if k >= 65:
        pts[192 * 7:192 * 8] = bytes(192)                       # infinity in the list
        pts[192 * 9:192 * 10] = pts[192 * 8:192 * 9]            # the same point twice ...
        sc[9] = sc[8] = rnd.randrange(N)                        # ... in the same buckets (doubling inside the addition)
        pts[192 * 12:192 * 13] = _neg_g2(bytes(pts[192 * 11:192 * 12]))
        sc[12] = sc[11] = rnd.randrange(N)                      # P and -P in the same buckets
    got, inf = sorted_g2_engine.g2_msm(bytes(pts), sc, k, 1)
    want, _ = oracle.g2_msm(bytes(pts), sc, k)
    assert got == want and inf[0] == (want == bytes(192))


def test_sorted_buckets_g2_degenerate_and_long_runs(sorted_g2_engine, fixed_window_engine, golden, seeded_pairs):
    P = bytes.fromhex(golden("points.json")["g2"][3]["p"])
    assert sorted_g2_engine.g2_msm(P + _neg_g2(P), [5, 5], 2) == (bytes(192), [True])
    assert sorted_g2_engine.g2_msm(P * 3, [0, 0, 0], 3) == (bytes(192), [True])
    assert sorted_g2_engine.g2_msm(P + bytes(192), [7, 9], 2) == fixed_window_engine.g2_msm(P, [7], 1)
    assert sorted_g2_engine.g2_msm(P * 40, [N - 1] * 40, 40) == fixed_window_engine.g2_msm(P, [(N - 1) * 40 % N], 1)
    # ONE scalar for thousands of points: every window's list is a single run over many of the equal pieces (k_srt_fix_long<2>)
    _, g2 = seeded_pairs
    k = 3000
    pts = (g2 * 3)[:192 * k]
    plain, _ = fixed_window_engine.g2_msm(pts, None, k, 1)
    for s in (0x1234567, N - 2):
        want, _ = fixed_window_engine.g2_msm(plain, [s], 1, 1)
        assert sorted_g2_engine.g2_msm(pts, [s] * k, k, 1) == (want, [False])


# ---- ONE plain sum of many points (no scalars) on the register kernels (round 5: k_sum_chunks + folds + the wide machine's tail) ----
@pytest.mark.parametrize("deg", [1, 2])
@pytest.mark.parametrize("k", [2, 3, 9, 33, 64, 65, 700, 5000, 40000])
def test_plain_sums_vs_oracle(engine, fixed_window_engine, oracle, seeded_pairs, k, deg):
    """BLS.aggregate_pub_keys / aggregate_sigs without exponents (bls.py:203-261) as one sum: points at infinity in the list, a
    point twice in a row (a doubling inside the mixed addition), a point followed by its negative, ragged chunk and fold counts;
    against the oracle and against the wavefront VM's k_msm (the engine without the register path)."""
    src, sz = (seeded_pairs[0], 96) if deg == 1 else (seeded_pairs[1], 192)
    npts = len(src) // sz
    rnd = random.Random(k * 7 + deg)
    pts = bytearray(b"".join(src[sz * (i % npts):sz * (i % npts + 1)] for i in (rnd.randrange(npts) for _ in range(k))))
    if k >= 9:
        pts[sz * 2:sz * 3] = bytes(sz)                               # infinity
        pts[sz * 4:sz * 5] = pts[sz * 3:sz * 4]                      # the same point twice in one chunk
        P = bytes(pts[sz * 6:sz * 7])
        h = 48 * deg
        pts[sz * 7:sz * 8] = P[:h] + b"".join(((Q - int.from_bytes(P[h + 48 * j:h + 48 * j + 48], "big")) % Q).to_bytes(48, "big") for j in range(deg))
    pts = bytes(pts)
    f = (lambda e: e.g1_msm(pts, None, k, 1)) if deg == 1 else (lambda e: e.g2_msm(pts, None, k, 1))
    got = f(engine)
    assert got == f(fixed_window_engine)
    if k <= 5000:
        want, _ = (oracle.g1_msm if deg == 1 else oracle.g2_msm)(pts, None, k)
        assert got == (want, [want == bytes(sz)])


def test_plain_sums_that_are_infinity(engine, golden):
    p = golden("points.json")
    for deg, key in ((1, "g1"), (2, "g2")):
        sz, h = 96 * deg, 48 * deg
        P = bytes.fromhex(p[key][3]["p"])
        negP = P[:h] + b"".join(((Q - int.from_bytes(P[h + 48 * j:h + 48 * j + 48], "big")) % Q).to_bytes(48, "big") for j in range(deg))
        f = engine.g1_msm if deg == 1 else engine.g2_msm
        assert f((P + negP) * 40, None, 80, 1) == (bytes(sz), [True])
        assert f(bytes(sz) * 100, None, 100, 1) == (bytes(sz), [True])
        assert f(bytes(sz) * 99 + P, None, 100, 1) == (P, [False])


# ---- batches of scalar multiplications / of small sums with scalars: one group per lane or lane pair (k_smul, round 5) ----
@pytest.fixture(scope="module")
def smul_engine():
    """k_smul from one group on (the default takes it from 4096 groups of at most 8 points)"""
    return _engine_with_values({"BLSGPU_SMUL_MIN_GROUPS": "1"})


@pytest.mark.parametrize("deg", [1, 2])
@pytest.mark.parametrize("k,groups", [(1, 1), (1, 70), (3, 33), (8, 5), (1, 5000), (2, 4100)])
def test_batches_of_small_sums_with_scalars(smul_engine, engine, fixed_window_engine, oracle, seeded_pairs, k, groups, deg):
    """key generation, sk H(m) for many messages, pk_i e_i per message (bls.py:177-192): random, short, zero and extreme scalars
    (2^256 - 1 is taken as the integer it is, fields_t.py:705-740), points at infinity, a sum of P and -P with equal scalars, ragged
    last wavefronts -- against the oracle (the first groups) and against the wavefront VM's double-and-add (all of them); the last
    two rows also through the default selection."""
    src, sz = (seeded_pairs[0], 96) if deg == 1 else (seeded_pairs[1], 192)
    npts = len(src) // sz
    rnd = random.Random(k * 1000 + groups + deg)
    n = k * groups
    pts = bytearray(b"".join(src[sz * j:sz * (j + 1)] for j in (rnd.randrange(npts) for _ in range(n))))
    sc = [rnd.choice([rnd.randrange(N), rnd.randrange(N), rnd.randrange(1 << 40), 0, N - 1, 1, (1 << 256) - 1, 1 << 255, 15, 16]) for _ in range(n)]
    if n >= 40:
        pts[sz * 5:sz * 6] = bytes(sz)                                   # infinity
    if k >= 2 and groups >= 3:                                          # group 2: P, -P with one scalar -> infinity (when k = 2)
        P = bytes(pts[sz * 2 * k:sz * (2 * k + 1)])
        h = 48 * deg
        pts[sz * (2 * k + 1):sz * (2 * k + 2)] = P[:h] + b"".join(((Q - int.from_bytes(P[h + 48 * j:h + 48 * j + 48], "big")) % Q).to_bytes(48, "big") for j in range(deg))
        sc[2 * k + 1] = sc[2 * k] = rnd.randrange(N)
    pts = bytes(pts)
    f = (lambda e: e.g1_msm(pts, sc, k, groups)) if deg == 1 else (lambda e: e.g2_msm(pts, sc, k, groups))
    got, inf = f(smul_engine)
    assert (got, inf) == f(fixed_window_engine)
    if groups >= 4096:
        assert (got, inf) == f(engine)
    om = oracle.g1_msm if deg == 1 else oracle.g2_msm
    for g in range(min(groups, 12)):
        want, _ = om(pts[sz * k * g:sz * k * (g + 1)], sc[k * g:k * (g + 1)], k)
        assert got[sz * g:sz * (g + 1)] == want and inf[g] == (want == bytes(sz)), g


def test_batches_of_small_sums_in_slices(engine, fixed_window_engine, seeded_pairs):
    """more groups than one 2 GB table holds (50 100 groups of eight G2 points: 16 multiples of 336 bytes per point): the batch runs
    in slices of groups; every group against the wavefront VM's double-and-add"""
    _, g2 = seeded_pairs
    k, groups = 8, 50100
    n = k * groups
    npts = len(g2) // 192
    rnd = random.Random(50100)
    pts = b"".join(g2[192 * j:192 * (j + 1)] for j in (rnd.randrange(npts) for _ in range(n)))
    sc = [rnd.randrange(1 << 256) for _ in range(n)]
    assert engine.g2_msm(pts, sc, k, groups) == fixed_window_engine.g2_msm(pts, sc, k, groups)

