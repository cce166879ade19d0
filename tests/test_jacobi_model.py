"""CPU: the two Jacobi-symbol routines of the hash-to-G2 stages as plain integers against Euler's criterion -- the
fixed-iteration binary algorithm of csrc/blsgpu_h2c.hip (swl::jacobi, round 3; now the fallback) and the division-step
form of csrc/fq32.h (fq_jacobi_var, round 4; the C++ itself runs on the host in tests/test_abi_and_host.py)."""
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


# ---- the division-step form (csrc/fq32.h jac_posdivsteps30_var / fq_jacobi_var) -------------------------------------------
M32 = 0xFFFFFFFF


def _ctz(x):
    return (x & -x).bit_length() - 1


def posdivsteps30(eta, f0, g0, jac):
    """30 division steps that keep f, g non-negative, on the 32 low bits; returns eta, the matrix, jac, inner iterations"""
    u, v, q, r = 1, 0, 0, 1
    f, g, i, iters = f0 & M32, g0 & M32, 30, 0
    while True:
        iters += 1
        zeros = _ctz(g | ((M32 << i) & M32))
        g >>= zeros
        u, v = (u << zeros) & M32, (v << zeros) & M32
        eta -= zeros
        i -= zeros
        jac ^= zeros & ((f >> 1) ^ (f >> 2))
        if i == 0:
            break
        if eta < 0:
            eta = -eta
            jac ^= (f & g) >> 1
            f, g, u, q, v, r = g, f, q, u, r, v
        lim = min(eta + 1, i)
        m = (M32 >> (32 - lim)) & 63
        w = (g * (f * (f * f - 2))) & m
        g, q, r = (g + f * w) & M32, (q + u * w) & M32, (r + v * w) & M32
    return eta, (u, v, q, r), jac & 1, iters


def jacobi_divsteps(x, cap=56):
    if x == 0:
        return 0, 0, 0
    f, g, eta, jac, worst_inner = Q, x, -1, 0, 0
    for it in range(cap):
        eta, (u, v, q, r), jac, iters = posdivsteps30(eta, f & M32, g & M32, jac)
        worst_inner = max(worst_inner, iters)
        nf, ng = u * f + v * g, q * f + r * g
        assert nf % (1 << 30) == 0 and ng % (1 << 30) == 0            # the matrix divides exactly
        f, g = nf >> 30, ng >> 30
        assert 0 < f < 2 * Q and 0 <= g < 2 * Q and f & 1             # non-negative, in range: 13 limbs of 30 bits hold them
        if f == 1:
            return (-1 if jac else 1), it + 1, worst_inner
    return 2, cap, worst_inner


def test_division_step_symbol_equals_euler_and_converges():
    rnd = random.Random(9)
    vals = [0, 1, 2, 3, 4, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2, 1 << 380]
    vals += [rnd.randrange(Q) for _ in range(6000)] + [rnd.randrange(1 << k) for k in (8, 64, 200, 380) for _ in range(60)]
    worst, worst_inner = 0, 0
    for v in vals:
        j, batches, inner = jacobi_divsteps(v)
        e = pow(v, (Q - 1) // 2, Q)
        assert j == (0 if e == 0 else (1 if e == 1 else -1)), v
        worst, worst_inner = max(worst, batches), max(worst_inner, inner)
    assert worst <= 46 and worst_inner <= 20      # 42 / 16 over 30 000 values; the kernel allows 56 batches, then falls back
