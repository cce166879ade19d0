"""CPU: the lane tables of the wide tail of the sorted-bucket G1 sum (vmgen/g1w_model.py -- what csrc/g1w_tables_gfx950.h holds and
csrc/blsgpu_g1w.hip k_msm_horner_wide executes) run digit by digit, the multiplier's 64-bit column bounds and the stored-value
range asserted, against the host's integer curve arithmetic (bls_py/hostmath.py): sum_i 2^(c i) P_i as the reference's
double-and-add would give it (fields_t.py:705-740), including the inputs the complete formulas are there for -- points at
infinity anywhere in the list, an addend equal to the running sum (a doubling inside the addition), an addend opposite to it."""
import os
import random

from bls_py import hostmath as H
from vmgen import g1w_model as M

F = H.F1
INF = (0, 1, 0)


def _rand_point(rng):
    return H.jac_to_affine(F, H.jac_mul(F, H.aff_to_jac(F, H.G1_GEN), rng.randrange(1, H.N)))


def _hom(A, rng=None):
    if A is None:
        return INF
    z = rng.randrange(1, H.Q) if rng else 1            # any representative of the projective point
    return (A[0] * z % H.Q, A[1] * z % H.Q, z)


def _affine(R):
    if R[2] % H.Q == 0:
        return None
    zi = H.fq_inv(R[2])
    return (R[0] * zi % H.Q, R[1] * zi % H.Q)


def _want(points, cbits):
    acc = None
    for i, A in enumerate(points):
        if A is None:
            continue
        t = H.jac_mul(F, H.aff_to_jac(F, A), 1 << (cbits * i))
        acc = t if acc is None else H.jac_add(F, acc, t)
    return None if acc is None else H.jac_to_affine(F, acc)


def test_every_output_is_one_product_per_lane():
    assert [s.name for s in M.KINDS] == ["DBL1", "DBL2", "ADD1", "ADD2"]
    assert all(s.K == 1 and len(s.outputs) <= 16 for s in M.KINDS)


def test_horner_over_bits_and_over_windows():
    rng = random.Random(20250511)
    pts = [_rand_point(rng) for _ in range(5)]
    for cbits in (1, 3):
        R, mx = M.horner([_hom(A, rng) for A in pts], cbits)
        assert mx < 1.02                               # every stored value in (-q/64, q + q/64)
        assert _affine(R) == _want(pts, cbits)


def test_infinity_equal_and_opposite_addends():
    rng = random.Random(77)
    P, S = _rand_point(rng), _rand_point(rng)
    two_p = H.jac_to_affine(F, H.jac_double(F, H.aff_to_jac(F, P)))
    neg_two_p = (two_p[0], (-two_p[1]) % H.Q)
    cases = [[None, P, None], [P, None], [None, None], [two_p, P], [neg_two_p, P], [S, neg_two_p, P], [None, None, S]]
    for pts in cases:
        R, _ = M.horner([_hom(A, rng) for A in pts], 1)
        assert _affine(R) == _want(pts, 1), pts


def test_generated_tables_are_current():
    from vmgen import gen_g1w
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        p = gen_g1w.generate(os.path.join(td, "t.h"))
        assert open(p).read() == open(os.path.join(root, "python-bls_amd", "csrc", "g1w_tables_gfx950.h")).read()
