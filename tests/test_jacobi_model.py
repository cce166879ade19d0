"""CPU: the fixed-iteration binary Jacobi-symbol algorithm of csrc/blsgpu_h2c.hip (swl::jacobi) as plain integers,
against Euler's criterion -- the quadratic characters the hash-to-G2 stages decide without a power."""
import random

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def jacobi_fixed(a, iters=768):
    """the kernel's loop: (a, n, sign); every iteration: if a is odd, swap so that a >= n (sign flips when both are 3
    mod 4) and subtract; then halve a non-zero a (sign flips when n is 3 or 5 mod 8)"""
    n, s, last = Q, 0, 0
    for it in range(iters):
        if a & 1:
            if a < n:
                a, n = n, a
                s ^= (a & n & 2) >> 1
            a -= n
        if a:
            a >>= 1
            s ^= ((n >> 1) ^ (n >> 2)) & 1
            last = it
    return (0 if n != 1 else (-1 if s else 1)), last


def test_jacobi_equals_euler_and_fits_the_iteration_budget():
    rnd = random.Random(5)
    vals = [0, 1, 2, 3, 4, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2, 1 << 380]
    vals += [rnd.randrange(Q) for _ in range(1500)] + [rnd.randrange(1 << k) for k in (8, 64, 200, 380) for _ in range(40)]
    worst = 0
    for v in vals:
        j, last = jacobi_fixed(v)
        e = pow(v, (Q - 1) // 2, Q)
        assert j == (0 if e == 0 else (1 if e == 1 else -1)), v
        worst = max(worst, last)
    assert worst < 768 - 4                       # bits(a) + bits(n) <= 762 iterations do work
