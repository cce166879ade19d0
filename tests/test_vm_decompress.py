"""Decompression VM programs (vmgen.decomp_programs) in the Python interpreter against
the host integer implementation (bls_py.hostmath.g1_/g2_decompress, pinned to the
reference by the serialisation vectors of tests/golden/scheme.json and by the reference-generated
verdicts of tests/golden/g2_real_u.json), including the inputs the reference rejects -- among them every
G2 x whose u = x^3 + b' has zero imaginary part (Fq2.modsqrt's a1 = 0 branch, fields.py:466-467).  CPU only."""
import random

import pytest

from bls_py import hostmath as H
from vmgen import decomp_programs as DP, h2c_programs as HP, programs as P, sim

Q = sim.Q
NE = 4


@pytest.fixture(scope="module")
def progs():
    return HP.h2c_scratch_consts(), DP.build_d1(NE), DP.build_d2(NE)


def run(consts, built, deg, xs, bigs):
    segs, L, script = built
    m = sim.Machine(consts, L.TEMP0 + max(s.ntemp for s in segs.values()))
    for e, (x, big) in enumerate(zip(xs, bigs)):
        for c in range(deg):
            m.team[L.X + deg * e + c] = x[c] if deg == 2 else x
        m.team[L.BIG + e] = sim.to_m(1) if big else 0
    for name in script:
        m.run(segs[name])
    n = 2 * deg + 1
    return [[m.team[L.OUT + n * e + k] % Q for k in range(n)] for e in range(NE)]


def encode(deg, x, big):
    b = bytearray(x.to_bytes(48, "big") if deg == 1 else x[0].to_bytes(48, "big") + x[1].to_bytes(48, "big"))
    if big:
        b[0] |= 0x80
    return bytes(b)


def check(progs, deg, xs, bigs):
    consts, d1, d2 = progs
    got = run(consts, d1 if deg == 1 else d2, deg, xs, bigs)
    accepted = 0
    for x, big, o in zip(xs, bigs, got):
        try:
            want = (H.g1_decompress if deg == 1 else H.g2_decompress)(encode(deg, x, big))
        except (ValueError, H.RealSquareRoot):
            assert o[-1] == 0
            continue
        accepted += 1
        flat = [want[0], want[1]] if deg == 1 else [want[0][0], want[0][1], want[1][0], want[1][1]]
        assert o == flat + [1], (x, big)
    return accepted


def test_g1_random_and_edges(progs):
    rng = random.Random(3)
    acc = 0
    for it in range(5):
        xs = [rng.randrange(1 << 381) for _ in range(NE)]           # masked x may exceed q
        if it == 0:
            xs[:2] = [0, Q + 5]
        acc += check(progs, 1, xs, [rng.random() < .5 for _ in range(NE)])
    assert 4 <= acc <= 16                                           # about half of all x are on the curve


def fq2_cube_roots_with_real_u(count):
    """x with x^3 + 4(1+i) real: the reference's `a1 == 0` branch of Fq2.modsqrt."""
    def fpow(a, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = H.f2_mul(r, a)
            a = H.f2_sqr(a)
            e >>= 1
        return r
    n = Q * Q - 1
    s, m = 0, n
    while m % 3 == 0:
        m //= 3
        s += 1
    g = (2, 1)
    while fpow(g, n // 3) == (1, 0):
        g = (g[0] + 1, g[1])
    zg = fpow(g, m)
    out, u0 = [], 1
    while len(out) < count:
        w = ((u0 - 4) % Q, (-4) % Q)
        u0 += 1
        if fpow(w, n // 3) != (1, 0):
            continue
        x0, z = fpow(w, pow(3, -1, m)), (1, 0)
        for _ in range(3 ** s):
            cand = H.f2_mul(x0, z)
            if H.f2_mul(H.f2_sqr(cand), cand) == w:
                out.append(cand)
                break
            z = H.f2_mul(z, zg)
    return out


def test_g2_random_special_and_edges(progs):
    rng = random.Random(4)
    acc = 0
    for it in range(4):
        xs = [(rng.randrange(1 << 381), rng.randrange(Q)) for _ in range(NE)]
        acc += check(progs, 2, xs, [rng.random() < .5 for _ in range(NE)])
    assert 3 <= acc <= 13
    special = fq2_cube_roots_with_real_u(8)
    ok = 0
    for i in range(0, 8, NE):
        ok += check(progs, 2, special[i:i + NE], [False, True, False, True])
        ok += check(progs, 2, special[i:i + NE], [True, False, True, False])
    assert ok == 0                         # u with zero imaginary part: the reference rejects them all (g2_real_u.json)
    check(progs, 2, [(0, 0), (1, 0), (0, 1), (Q - 1, 0)], [False, True, True, False])


def test_g2_encodings_whose_u_is_real(progs, golden):
    """Reference-generated verdicts (Signature.from_bytes raises for all of them): the host mirror raises the
    matching exception and the GPU program's validity flag is 0."""
    recs = golden("g2_real_u.json")["decompress"]
    consts, d1, d2 = progs
    for i in range(0, len(recs), NE):
        chunk = recs[i:i + NE]
        xs, bigs = [], []
        for r in chunk:
            enc = bytes.fromhex(r["encoding"])
            bigs.append(bool(enc[0] & 0x80))
            xs.append((int.from_bytes(bytes([enc[0] & 0x1f]) + enc[1:48], "big"), int.from_bytes(enc[48:], "big")))
            with pytest.raises(H.RealSquareRoot if r["reference"] == "Exception" else ValueError):
                H.g2_decompress(enc)
        while len(xs) < NE:
            xs.append(xs[0]), bigs.append(bigs[0])
        got = run(consts, d2, 2, xs, bigs)
        assert all(o[-1] == 0 for o in got)
