"""Hash-to-G2 VM programs (vmgen.h2c_programs) executed by the Python interpreter
in the GPU's Montgomery domain, against the host integer implementation
(bls_py.hostmath, itself pinned to the reference by tests/golden/hash_to_curve.json).
CPU only."""
import hashlib
import random

import pytest

from bls_py import hostmath as H
from bls_py.util import hash512
from vmgen import h2c_programs as HP, programs as P, sim

Q = sim.Q
NE, NM = 4, 2


@pytest.fixture(scope="module")
def progs():
    consts = HP.h2c_scratch_consts()
    return consts, HP.build_h1(NE), HP.build_h2(NM)


def encode(progs, ts):
    consts, (segs, L, script), _ = progs
    assert len(ts) == NE
    m = sim.Machine(consts, 700)
    for e, t in enumerate(ts):
        m.team[L.T + 2 * e], m.team[L.T + 2 * e + 1] = t
    for name in script:
        m.run(segs[name])
        if name in ("h1_a", "h1w_a"):
            HP.real_u_step(m.team, L, NE)            # the zero test the stage-0 kernel does natively
    return [[m.team[L.S + 5 * e + k] for k in range(5)] for e in range(NE)]


def affine_of(s):
    x0, x1, y0, y1, z = [sim.from_m(v) % Q for v in s]
    if z == 0:
        assert (x0, x1, y0, y1) == (0, 0, 1, 0)
        return None
    assert z == 1
    return ((x0, x1), (y0, y1))


def clear(progs, pairs):
    consts, _, (segs, L, script) = progs
    assert len(pairs) == NM
    m = sim.Machine(consts, 700)
    for i, pr in enumerate(pairs):
        for j, s in enumerate(pr):
            for k in range(5):
                m.team[L.S + 10 * i + 5 * j + k] = s[k]
    for name in script:
        m.run(segs[name])
    out = []
    for i in range(NM):
        v = [m.team[L.OUT + 4 * i + k] % Q for k in range(4)]
        out.append(None if v == [0, 0, 0, 0] else ((v[0], v[1]), (v[2], v[3])))
    return out


def t_values(msg):
    return [(int.from_bytes(hash512(msg + b"G2_%d_c0" % j), "big") % Q,
             int.from_bytes(hash512(msg + b"G2_%d_c1" % j), "big") % Q) for j in range(2)]


def test_sw_encode_random(progs):
    rng = random.Random(11)
    ts = [(rng.randrange(Q), rng.randrange(Q)) for _ in range(3)] + [(rng.randrange(Q), 0)]
    got = [affine_of(s) for s in encode(progs, ts)]
    assert got == [H.sw_encode(H.F2, t) for t in ts]


def test_sw_encode_early_exits(progs):
    """t = 0 -> infinity (ec.py:450-452).  The other exit, t^2 + b' + 1 = 0, has no
    solution over Fq2, so the programs do not carry it."""
    with pytest.raises(ValueError):
        H.f2_sqrt(((-5) % Q, (-4) % Q))              # t^2 = -(b' + 1), b' = 4 + 4i
    ts = [(0, 0), (1, 0), (0, 1), (Q - 1, Q - 1)]
    got = [affine_of(s) for s in encode(progs, ts)]
    assert got[0] is None
    assert got == [H.sw_encode(H.F2, t) for t in ts]


def test_full_hash_matches_host_and_golden(progs, golden):
    vec = golden("hash_to_curve.json")
    msgs = [hashlib.sha256(b"vm-h2c-%d" % i).digest() for i in range(NM)]
    ts = [t for m in msgs for t in t_values(m)]
    enc = encode(progs, ts)
    got = clear(progs, [(enc[2 * i], enc[2 * i + 1]) for i in range(NM)])
    assert got == [H.hash_to_g2_prehashed(m, hash512) for m in msgs]
    assert vec                                       # the pin of hostmath lives in test_scheme_host


def test_infinity_summand(progs):
    """One encoding at infinity: the complete addition in kernel H2 must return the other."""
    rng = random.Random(5)
    t1 = (rng.randrange(Q), rng.randrange(Q))
    enc = encode(progs, [(0, 0), t1, (0, 0), (0, 0)])
    got = clear(progs, [(enc[0], enc[1]), (enc[2], enc[3])])
    S1 = H.sw_encode(H.F2, t1)
    x = -H.C.x
    F2 = H.F2
    jac, aff = (lambda A: H.aff_to_jac(F2, A)), (lambda J: H.jac_to_affine(F2, J))
    Pj = jac(S1)
    psi2 = jac(H.psi(H.psi(aff(H.jac_double(F2, Pj)))))
    a0 = H.jac_mul(F2, Pj, x)
    a1 = H.jac_mul(F2, a0, x)
    a2 = H.jac_add(F2, H.jac_add(F2, a1, a0), H.jac_neg(F2, Pj))
    a3 = jac(H.psi(aff(H.jac_mul(F2, Pj, x + 1))))
    want = aff(H.jac_add(F2, H.jac_add(F2, a2, H.jac_neg(F2, a3)), psi2))
    assert got[0] == want
    assert got[1] is None                            # infinity + infinity -> (0, 0) at the ABI


def test_candidate_with_real_u_is_skipped(progs, golden):
    """Reference-generated (tests/golden/g2_real_u.json): values t whose first candidate x1 has
    u = x1^3 + b' with zero imaginary part.  The reference's sw_encode skips x1 (Fq2.modsqrt's a1 = 0 branch
    makes y_for_x fail) and takes x2 or x3; so do the host mirror and the GPU program, through to the hash."""
    recs = golden("g2_real_u.json")["sw_encode"]
    assert len(recs) == 4

    def fq2s(h, k):
        b = bytes.fromhex(h)
        v = [int.from_bytes(b[48 * i:48 * (i + 1)], "big") for i in range(2 * k)]
        return [(v[2 * i], v[2 * i + 1]) for i in range(k)]
    for lo in (0, 2):
        ts = [t for r in recs[lo:lo + 2] for t in fq2s(r["t"], 2)]
        enc = encode(progs, ts)
        for j, r in enumerate(recs[lo:lo + 2]):
            want = fq2s(r["sw_encode_t0"], 2)
            assert affine_of(enc[2 * j]) == (want[0], want[1]) == H.sw_encode(H.F2, ts[2 * j])
            assert affine_of(enc[2 * j])[0] != fq2s(r["skipped_x1"], 1)[0]
        got = clear(progs, [(enc[0], enc[1]), (enc[2], enc[3])])
        for j, r in enumerate(recs[lo:lo + 2]):
            want = fq2s(r["point"], 2)
            assert got[j] == (want[0], want[1])
