"""Multi-scalar-sum VM programs (vmgen.msm_programs: the complete projective addition and doubling, the accumulate /
fold / affine segments of the team kernels, and the batch Horner of several sums per team) executed by the Python
interpreter in the GPU's Montgomery domain against the host integer arithmetic (bls_py.hostmath, itself pinned to the
reference by tests/golden/points.json and msm.json in test_scheme_host / test_oracle_golden).  CPU only."""
import random

import pytest

from bls_py import hostmath as H
from vmgen import msm_programs as MP, programs as P, sim

Q = sim.Q


def field(deg):
    return H.F1 if deg == 1 else H.F2


def rand_point(rng, deg):
    """a point of E(Fq) / E'(Fq2) (any subgroup: the formulas are complete on the whole curve)"""
    F = field(deg)
    G = H.G1_GEN if deg == 1 else H.G2_GEN
    return H.jac_to_affine(F, H.jac_mul(F, H.aff_to_jac(F, G), rng.randrange(1, 1 << 64)))


def put_proj(m, lay, base, p, A, deg):
    """affine A (or None = infinity) as the projective Montgomery triple (x : y : 1) / (0 : 1 : 0)"""
    o = lay.pt(base, p)
    if A is None:
        coords = [(0,) * deg, (1,) + (0,) * (deg - 1), (0,) * deg]
    else:
        x, y = A
        coords = [(x,) if deg == 1 else x, (y,) if deg == 1 else y, (1,) + (0,) * (deg - 1)]
    for k in range(3):
        for i in range(deg):
            m.team[o + k * deg + i] = sim.to_m(coords[k][i])


def get_affine(m, lay, base, p, deg):
    F = field(deg)
    o = lay.pt(base, p)
    c = [[sim.from_m(m.team[o + k * deg + i]) % Q for i in range(deg)] for k in range(3)]
    X, Y, Z = (c[k][0] if deg == 1 else tuple(c[k]) for k in range(3))
    if Z == F.zero:
        return None
    zi = F.inv(Z)
    return (F.mul(X, zi), F.mul(Y, zi))


def host_add(deg, A, B):
    F = field(deg)
    return H.jac_to_affine(F, H.jac_add(F, H.aff_to_jac(F, A), H.aff_to_jac(F, B)))      # None = infinity on the host


@pytest.fixture(scope="module", params=[1, 2])
def team(request):
    deg = request.param
    NP = 3
    segs, lay = MP.build(deg, NP)
    return deg, NP, segs, lay


def machine(segs, lay):
    return sim.Machine(P.const_table(), lay.TEMP0 + max(s.ntemp for s in segs.values()))


def test_step_accumulate_and_fold(team):
    """<tag>_acc: R_p += S_p with P + Q, P + P (doubling inside the addition), P + (-P), infinity on either side;
    <tag>_step additionally doubles A_p; <tag>_fold sums the NP accumulators; <tag>_dbl, <tag>_padd on the point registers"""
    deg, NP, segs, lay = team
    F = field(deg)
    rng = random.Random(100 + deg)
    tag = "g%d" % deg
    A, B = rand_point(rng, deg), rand_point(rng, deg)
    negA = (A[0], F.neg(A[1]))
    cases = [(A, B), (A, A), (A, negA)]
    m = machine(segs, lay)
    for p, (r, s) in enumerate(cases):
        put_proj(m, lay, lay.R, p, r, deg)
        put_proj(m, lay, lay.S, p, s, deg)
    m.run(segs[tag + "_acc"])
    got = [get_affine(m, lay, lay.R, p, deg) for p in range(NP)]
    dbl = H.jac_to_affine(F, H.jac_double(F, H.aff_to_jac(F, A)))
    assert got == [host_add(deg, A, B), dbl, None]
    # infinity + P, P + infinity, infinity + infinity; the running doubles of <tag>_step
    m = machine(segs, lay)
    for p, (r, s) in enumerate([(None, B), (A, None), (None, None)]):
        put_proj(m, lay, lay.R, p, r, deg)
        put_proj(m, lay, lay.S, p, s, deg)
        put_proj(m, lay, lay.A, p, [A, B, None][p], deg)
    m.run(segs[tag + "_step"])
    assert [get_affine(m, lay, lay.R, p, deg) for p in range(NP)] == [B, A, None]
    assert [get_affine(m, lay, lay.A, p, deg) for p in range(NP)] == \
        [dbl, H.jac_to_affine(F, H.jac_double(F, H.aff_to_jac(F, B))), None]
    m.run(segs[tag + "_fold"])                                   # R_0 + R_1 + R_2 = B + A
    assert get_affine(m, lay, lay.PR0, 0, deg) == host_add(deg, A, B)
    m.run(segs[tag + "_dbl"])
    two = H.jac_to_affine(F, H.jac_double(F, H.aff_to_jac(F, host_add(deg, A, B))))
    assert get_affine(m, lay, lay.PR0, 0, deg) == two
    put_proj(m, lay, lay.PR1, 0, A, deg)
    m.run(segs[tag + "_padd"])
    assert get_affine(m, lay, lay.PR0, 0, deg) == host_add(deg, two, A)
    m.run(segs[tag + "_affine"])                                 # canonical affine output, raw (non-Montgomery) integers
    want = host_add(deg, two, A)
    flat = [want[0], want[1]] if deg == 1 else [want[0][0], want[0][1], want[1][0], want[1][1]]
    assert [m.team[lay.OUT + i] % Q for i in range(2 * deg)] == flat


def test_batch_horner_programs():
    """g2h_dbl / g2h_acc / g2h_affine (k_msm_horner_np): NP sums per team -- one Horner step 2^3 R_p + S_p for every p,
    with infinity among the running sums and the addends, then the affine outputs ((0, 0) for infinity)."""
    deg, NP = 2, 5
    segs, lay = MP.build_horner(deg, NP)
    F = H.F2
    rng = random.Random(7)
    m = sim.Machine(P.const_table(), lay.TEMP0 + max(s.ntemp for s in segs.values()))
    Rs = [rand_point(rng, deg), None, rand_point(rng, deg), rand_point(rng, deg), None]
    Ss = [rand_point(rng, deg), rand_point(rng, deg), None, None, None]
    Ss[3] = (lambda E: (E[0], F.neg(E[1])))(H.jac_to_affine(F, H.jac_mul(F, H.aff_to_jac(F, Rs[3]), 8)))   # 8 R_3 + S_3 = infinity
    for p in range(NP):
        put_proj(m, lay, lay.R, p, Rs[p], deg)
        put_proj(m, lay, lay.S, p, Ss[p], deg)
    for _ in range(3):
        m.run(segs["g2h_dbl"])
    m.run(segs["g2h_acc"])
    want = []
    for r, s in zip(Rs, Ss):
        r8 = None if r is None else H.jac_to_affine(F, H.jac_mul(F, H.aff_to_jac(F, r), 8))
        want.append(host_add(deg, r8, s))
    assert want[3] is None and want[4] is None
    assert [get_affine(m, lay, lay.R, p, deg) for p in range(NP)] == want
    m.run(segs["g2h_affine"])
    for p in range(NP):
        got = [m.team[lay.OUT + p * 4 + i] % Q for i in range(4)]
        assert got == ([0, 0, 0, 0] if want[p] is None else [want[p][0][0], want[p][0][1], want[p][1][0], want[p][1][1]])
