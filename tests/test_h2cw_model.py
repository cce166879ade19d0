"""CPU: the lane tables and the step program of the wide cofactor clearing (vmgen/h2cw_model.py -- what csrc/h2cw_tables_gfx950.h
holds and csrc/blsgpu_h2cw.hip executes) run digit by digit, the multiplier's 64-bit column bounds and the stored-value range
asserted, against the host's integer hash-to-G2 (bls_py/hostmath.py, pinned to the reference's vectors of ec.py:528-550 by
tests/test_hostmath_fixtures.py), including the inputs the complete formulas exist for: an encoding at infinity, S + (-S)."""
import hashlib
import os

from bls_py import hostmath as H
from bls_py.util import hash512
from vmgen import h2cw_model as M, gen_fexp


def _affine(R):
    if R[2] == (0, 0):
        return None
    zi = H.f2_inv(R[2])
    return (H.f2_mul(R[0], zi), H.f2_mul(R[1], zi))


def _hom(A):
    return ((0, 0), (1, 0), (0, 0)) if A is None else (A[0], A[1], (1, 0))


def _encodings(m):
    t = H.g2_hash_field_elements(m, hash512)
    v = [int.from_bytes(t[48 * j:48 * (j + 1)], "big") for j in range(4)]
    return [H.sw_encode(H.F2, (v[2 * j], v[2 * j + 1])) for j in range(2)]


def test_program_is_the_compiled_script():
    prog = M.program()
    script = [x for x in gen_fexp.h2c_clear_script() if x[0] != 0]
    steps = [w for w in prog if w != M.END and not (w & M.COPY)]
    dbl = sum(1 for op, _ in script if op in (5, 8))
    add = sum(1 for op, _ in script if op in (1, 2))
    psi = sum(1 for op, _ in script if op == 6)
    assert len(steps) == 2 * dbl + 2 * add + psi and dbl == 2 * 63 + 1
    assert sum(1 for w in prog if w != M.END and (w & M.COPY)) == sum(1 for op, _ in script if op in (3, 4))
    assert all(s.K == 1 for s in M.KINDS)              # every output at most four products: one product per lane


def test_tables_give_the_hash_to_g2():
    m = hashlib.sha256(b"wide-clearing-0").digest()
    S = _encodings(m)
    R, mx = M.clear(_hom(S[0]), _hom(S[1]), H._PSI_X, H._PSI_Y)
    assert mx < 1.02                                   # every stored value in (-q/64, q + q/64)
    assert _affine(R) == tuple(H.hash_to_g2_prehashed(m, hash512))


def test_infinity_and_opposite_encodings():
    """the cases the complete formulas are there for (ec.py:450-452: t = 0 encodes to infinity; S1 = -S0 sums to infinity)"""
    S = _encodings(hashlib.sha256(b"wide-clearing-1").digest())
    neg = (S[0][0], H.f2_neg(S[0][1]))
    R, _ = M.clear(_hom(S[0]), _hom(neg), H._PSI_X, H._PSI_Y)
    assert _affine(R) is None
    R, _ = M.clear(_hom(None), _hom(S[1]), H._PSI_X, H._PSI_Y)
    want = H.clear_cofactor_g2(H.aff_to_jac(H.F2, S[1]))
    assert _affine(R) == tuple(want)


def test_window_horner_of_a_g2_sum_on_the_same_tables():
    """csrc/blsgpu_h2cw.hip k_msm_horner_wide2 (the tail of the sorted-bucket G2 sum: bls.py:225-261 as one multi-scalar sum):
    sum_i 2^(c i) P_i with infinity in the list, an addend equal to the running sum and one opposite to it"""
    import random
    rng = random.Random(5)
    F = H.F2
    G = tuple(H.hash_to_g2_prehashed(hashlib.sha256(b"g2-horner").digest(), hash512))
    mul = lambda k: H.jac_to_affine(F, H.jac_mul(F, H.aff_to_jac(F, G), k))
    P, S = mul(rng.randrange(1, H.N)), mul(rng.randrange(1, H.N))
    two_p = H.jac_to_affine(F, H.jac_double(F, H.aff_to_jac(F, P)))
    neg = lambda A: (A[0], H.f2_neg(A[1]))

    def hom(A):
        if A is None:
            return ((0, 0), (1, 0), (0, 0))
        z = (rng.randrange(1, H.Q), rng.randrange(H.Q))
        return (H.f2_mul(A[0], z), H.f2_mul(A[1], z), z)

    def want(pts, c):
        acc = None
        for i, A in enumerate(pts):
            if A is not None:
                t = H.jac_mul(F, H.aff_to_jac(F, A), 1 << (c * i))
                acc = t if acc is None else H.jac_add(F, acc, t)
        return None if acc is None else tuple(H.jac_to_affine(F, acc))
    for pts, c in (([P, S, None, P], 3), ([two_p, P], 1), ([neg(two_p), P], 1), ([None, None], 2), ([S, neg(two_p), P], 1)):
        R, mx = M.horner([hom(A) for A in pts], c)
        assert mx < 1.02
        got = _affine(R)
        assert (None if got is None else tuple(got)) == want(pts, c), (pts, c)


def test_generated_tables_are_current():
    from vmgen import gen_h2cw
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        p = gen_h2cw.generate(os.path.join(td, "t.h"))
        assert open(p).read() == open(os.path.join(root, "python-bls_amd", "csrc", "h2cw_tables_gfx950.h")).read()
