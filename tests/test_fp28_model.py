"""CPU: the integer model of the generated 28-bit-limb Montgomery sums of products (vmgen/gen_fp28.model_dot: the
column arithmetic of csrc/fp28_mul_gfx950.h with the 64-bit accumulator range asserted) against Python integers, the
constants the register kernels use, and that the committed header is what the generator emits."""
import os
import random

from vmgen import gen_fp28 as G


def _val(d):
    return G.from_limbs(d)


def test_model_dot_is_a_montgomery_sum_of_products():
    rnd = random.Random(1)
    Rinv = pow(G.R, -1, G.Q)
    for K in (1, 2, 3, 4, 6):
        for _ in range(6):
            terms, want = [], 0
            for _ in range(K):
                a, b = rnd.randrange(-G.Q, 2 * G.Q), rnd.randrange(-G.Q, 2 * G.Q)     # values of products: (-q, 2q)
                terms.append((G.to_limbs(a), G.to_limbs(b)))
                want += a * b
            r = _val(G.model_dot(terms))
            assert (r - want * Rinv) % G.Q == 0 and -G.Q < r < 2 * G.Q, K


def test_column_bound_of_the_line_product():
    """k_ml_accum's fp28_dot6: an unwrapped term 1 + 1 units of 2^56 per limb product, the two wrapped ones 1 + 2 (the
    imaginary part of xi f is a sum of two digits): 8 units, the most a 64-bit column holds beside the m q part --
    model_dot asserts the range on the worst-case digits"""
    top = (1 << 28) - 1
    one = [top] * 13 + [0]
    two = [2 * top] * 13 + [0]
    neg = [-top] * 13 + [0]
    terms = [(one, one), (neg, neg), (one, one), (two, one), (one, one), (two, one)]
    G.model_dot(terms)                                              # must not trip the overflow assertion
    G.model_dot([(x, [-v for v in y]) for x, y in terms])           # nor on the negative side


def test_constants_and_committed_header(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = G.generate(os.path.join(tmp_path, "fp28.h"))
    assert open(p).read() == open(os.path.join(root, "python-bls_amd", "csrc", "fp28_mul_gfx950.h")).read()
    text = open(p).read()
    for name in ("BLS28_Q", "BLS28_ONE", "BLS28_R2", "BLS28_HALF", "BLS28_SW_S3", "BLS28_SW_HH", "BLS28_SW_SINV", "BLS28_WIDE_C2",
                 "BLS28_FROM_VM", "BLS28_TO_VM", "BLS28_VM_ONE_WORDS", "fp28_dot6", "fp28_sqr2"):
        assert name in text
    assert _val(G.to_limbs(G.R % G.Q)) == G.R % G.Q and (2 * ((G.Q + 1) // 2)) % G.Q == 1
