"""CPU: the identity behind k_ml_exact_fixup (csrc/blsgpu_ml.hip; vmgen/gen_mlx.py) on the integer model of the line-stream stage:
the Miller value of one pair from the FAST lines, times one Fq2 factor recovered from the lines' third coefficients and py, rotated
by w^-3, is the reference's own fq_miller_loop value (fields_t.py:1091-1111) -- checked against the model's exact Miller loop, which
tests/test_linestream_model.py pins to the reference's vectors -- and the generated constants are current."""
import os
import random

from bls_py import hostmath as H
from bls_py.util import hash512
from vmgen import gen_mlx, linestream_model as LS


def _pairs(count, seed):
    rng = random.Random(seed)
    G2 = tuple(H.hash_to_g2_prehashed(b"\x05" * 32, hash512))
    for _ in range(count):
        P = H.jac_to_affine(H.F1, H.jac_mul(H.F1, H.aff_to_jac(H.F1, H.G1_GEN), rng.randrange(1, H.N)))
        Qa = tuple(H.jac_to_affine(H.F2, H.jac_mul(H.F2, H.aff_to_jac(H.F2, G2), rng.randrange(1, H.N))))
        yield P, Qa


def test_the_schedule_constants():
    S, chords = gen_mlx.schedule_constants()
    assert chords == [1, 4, 8, 18, 51] and S % 2 == 1 and S.bit_length() == 64       # (csrc/blsgpu_ml.hip line_positions)


def test_fast_value_times_the_recovered_factor_is_the_exact_value():
    for P, Qa in _pairs(3, 11):
        lines, ok = LS.pair_lines(P, Qa)
        assert ok
        f_fast = LS.horner([LS.line_to_dense(l) for l in lines])
        assert gen_mlx.exact_from_fast(f_fast, lines, P[1]) == LS.exact_miller(P, Qa)


def test_generated_constants_are_current():
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        p = gen_mlx.generate(os.path.join(td, "t.h"))
        assert open(p).read() == open(os.path.join(root, "python-bls_amd", "csrc", "mlx_consts_gfx950.h")).read()
