"""Fq2 / Fq6 / Fq12 formulas over traced expressions (core.E).

Tower and basis are the reference's (fields.py:321-329, 485-490, 624-629):
  Fq2  = Fq[u]/(u^2+1),   xi = 1+u
  Fq6  = Fq2[v]/(v^3-xi)
  Fq12 = Fq6[w]/(w^2-v)
An Fq2 is a pair (c0, c1) of E; Fq6 a triple of Fq2; Fq12 a pair of Fq6.
flat12()/unflat12() convert to the reference's flat 12-tuple order.

Every function returns *lazy* linear combinations; products materialise their
operands.  `m2` selects the Fq2 multiplier: Karatsuba (3 products, operand
pre-additions) or schoolbook (4 products, none).
"""

# ------------------------------------------------------------------ Fq2 ----


def f2_add(a, b):
    return (a[0] + b[0], a[1] + b[1])


def f2_sub(a, b):
    return (a[0] - b[0], a[1] - b[1])


def f2_neg(a):
    return (-a[0], -a[1])


def f2_scale(a, k):
    return (a[0] * k, a[1] * k)


def f2_mul_xi(a):
    """(a0 + a1 u)(1 + u) = (a0 - a1) + (a0 + a1) u   (fields_t.py:113-116)"""
    return (a[0] - a[1], a[0] + a[1])


def f2_conj(a):
    return (a[0], -a[1])


def f2_mat(a):
    return (a[0].mat(), a[1].mat())


def f2_mul_kara(a, b):
    t0 = a[0] * b[0]
    t1 = a[1] * b[1]
    t2 = (a[0] + a[1]) * (b[0] + b[1])
    return (t0 - t1, t2 - t0 - t1)


def f2_mul_school(a, b):
    return (a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0])


def f2_sqr_complex(a):
    """(a0+a1)(a0-a1), 2 a0 a1 : 2 products"""
    return ((a[0] + a[1]) * (a[0] - a[1]), (a[0] * a[1]) * 2)


def f2_sqr_3(a):
    """a0^2 - a1^2, 2 a0 a1 : 3 products, no operand pre-additions"""
    return (a[0] * a[0] - a[1] * a[1], (a[0] * a[1]) * 2)


def f2_mul_fq(a, s):
    return (a[0] * s, a[1] * s)


class Cfg:
    """Formula choices (tuned by looking at the scheduler's round statistics)."""
    def __init__(self, m2=f2_mul_kara, s2=f2_sqr_complex, mat2=True):
        self.m2, self.s2, self.mat2 = m2, s2, mat2

    def mul2(self, a, b):
        r = self.m2(a, b)
        return f2_mat(r) if self.mat2 else r

    def sqr2(self, a):
        r = self.s2(a)
        return f2_mat(r) if self.mat2 else r


# ------------------------------------------------------------------ Fq6 ----
def f6_add(a, b):
    return tuple(f2_add(x, y) for x, y in zip(a, b))


def f6_sub(a, b):
    return tuple(f2_sub(x, y) for x, y in zip(a, b))


def f6_neg(a):
    return tuple(f2_neg(x) for x in a)


def f6_mul_v(a):
    """(a0, a1, a2) * v = (xi a2, a0, a1)   (fields_t.py:215-220)"""
    return (f2_mul_xi(a[2]), a[0], a[1])


def f6_mul(cfg, a, b):
    """Karatsuba over Fq2: 6 Fq2 products."""
    v0 = cfg.mul2(a[0], b[0])
    v1 = cfg.mul2(a[1], b[1])
    v2 = cfg.mul2(a[2], b[2])
    m12 = cfg.mul2(f2_add(a[1], a[2]), f2_add(b[1], b[2]))
    m01 = cfg.mul2(f2_add(a[0], a[1]), f2_add(b[0], b[1]))
    m02 = cfg.mul2(f2_add(a[0], a[2]), f2_add(b[0], b[2]))
    c0 = f2_add(v0, f2_mul_xi(f2_sub(f2_sub(m12, v1), v2)))
    c1 = f2_add(f2_sub(f2_sub(m01, v0), v1), f2_mul_xi(v2))
    c2 = f2_add(f2_sub(f2_sub(m02, v0), v2), v1)
    return (c0, c1, c2)


def f6_mul_by_01(cfg, x, c0, c1):
    """x * (c0 + c1 v): 5 Fq2 products."""
    v0 = cfg.mul2(x[0], c0)
    v1 = cfg.mul2(x[1], c1)
    m12 = cfg.mul2(f2_add(x[1], x[2]), c1)
    m01 = cfg.mul2(f2_add(x[0], x[1]), f2_add(c0, c1))
    m02 = cfg.mul2(f2_add(x[0], x[2]), c0)
    r0 = f2_add(f2_mul_xi(f2_sub(m12, v1)), v0)
    r1 = f2_sub(f2_sub(m01, v0), v1)
    r2 = f2_add(f2_sub(m02, v0), v1)
    return (r0, r1, r2)


def f6_mul_by_1(cfg, x, c1):
    """x * (c1 v): 3 Fq2 products."""
    return (f2_mul_xi(cfg.mul2(x[2], c1)), cfg.mul2(x[0], c1), cfg.mul2(x[1], c1))


def f6_mul_fq2(cfg, a, s):
    return tuple(cfg.mul2(x, s) for x in a)


def f6_inv(cfg, x):
    """fields_t.py:170-184 (same cofactor formula), one Fq2 inversion."""
    a, b, c = x
    g0 = f2_sub(cfg.sqr2(a), cfg.mul2(b, f2_mul_xi(c)))
    g1 = f2_sub(f2_mul_xi(cfg.sqr2(c)), cfg.mul2(a, b))
    g2 = f2_sub(cfg.sqr2(b), cfg.mul2(a, c))
    g0, g1, g2 = f2_mat(g0), f2_mat(g1), f2_mat(g2)
    den = f2_add(cfg.mul2(g0, a), f2_mul_xi(f2_add(cfg.mul2(g1, c), cfg.mul2(g2, b))))
    f = f2_inv(cfg, den)
    return (cfg.mul2(g0, f), cfg.mul2(g1, f), cfg.mul2(g2, f))


def f2_inv(cfg, x):
    """fields_t.py:81-85: (a, -b) / (a^2 + b^2)"""
    n = (x[0] * x[0] + x[1] * x[1]).inv()
    return ((x[0] * n).mat(), (-(x[1] * n)).mat())


# ----------------------------------------------------------------- Fq12 ----
def f12_mul(cfg, a, b):
    """Karatsuba over Fq6: 18 Fq2 products."""
    t0 = f6_mul(cfg, a[0], b[0])
    t1 = f6_mul(cfg, a[1], b[1])
    m = f6_mul(cfg, f6_add(a[0], a[1]), f6_add(b[0], b[1]))
    return (f6_add(t0, f6_mul_v(t1)), f6_sub(f6_sub(m, t0), t1))


def f12_sqr(cfg, a):
    """Complex squaring: 12 Fq2 products.
    c0 = (a0 + a1)(a0 + v a1) - t - v t,  c1 = 2 t,  t = a0 a1"""
    t = f6_mul(cfg, a[0], a[1])
    m = f6_mul(cfg, f6_add(a[0], a[1]), f6_add(a[0], f6_mul_v(a[1])))
    c0 = f6_sub(f6_sub(m, t), f6_mul_v(t))
    c1 = f6_add(t, t)
    return (c0, c1)


def f12_mul_by_014(cfg, f, l0, l1, l4):
    """f * (l0 + l1 v + l4 v w): 13 Fq2 products (sparse line multiplication)."""
    a, b = f
    t0 = f6_mul_by_01(cfg, a, l0, l1)
    t1 = f6_mul_by_1(cfg, b, l4)
    m = f6_mul_by_01(cfg, f6_add(a, b), l0, f2_add(l1, l4))
    return (f6_add(t0, f6_mul_v(t1)), f6_sub(f6_sub(m, t0), t1))


def f12_conj(a):
    return (a[0], f6_neg(a[1]))


def f12_inv(cfg, x):
    """fields_t.py:328-337"""
    a, b = x
    d = f6_sub(f6_mul(cfg, a, a), f6_mul_v(f6_mul(cfg, b, b)))
    d = tuple(f2_mat(c) for c in d)
    f = f6_inv(cfg, d)
    return (f6_mul(cfg, a, f), f6_neg(f6_mul(cfg, b, f)))


def f12_frob(cfg, x, i, gam):
    """x -> x^(q^i) (fields_t.py:355-364); gam(j) returns the traced constant
    gamma_i^j = xi^(j (q^i - 1) / 6) as an Fq2 (or None when it is 1)."""
    def c(a):
        return f2_conj(a) if i % 2 else a

    def g(a, j):
        k = gam(j)
        return c(a) if k is None else cfg.mul2(c(a), k)
    (a0, a1, a2), (b0, b1, b2) = x
    # coefficient of v^s w^t picks up gamma^(2s + t)
    return ((c(a0), g(a1, 2), g(a2, 4)), (g(b0, 1), g(b1, 3), g(b2, 5)))


def f12_cyclo_sqr(cfg, x, ref=None):
    """Granger-Scott squaring in the cyclotomic subgroup: 9 Fq2 squarings'
    worth of products (as 3 Fq4 squarings).  `ref`: the same element again, used
    for the linear +-2x terms (chained squarings pass the materialised copy there
    and the lazy one as x, so that one LIN level serves both)."""
    (a0, a1, a2), (b0, b1, b2) = x
    (ra0, ra1, ra2), (rb0, rb1, rb2) = ref if ref is not None else x

    def fp4_sq(a, b):
        t0 = cfg.sqr2(a)
        t1 = cfg.sqr2(b)
        c0 = f2_add(f2_mul_xi(t1), t0)
        c1 = f2_sub(f2_sub(cfg.sqr2(f2_add(a, b)), t0), t1)
        return c0, c1
    t0, t1 = fp4_sq(a0, b1)
    t2, t3 = fp4_sq(b0, a2)
    t4, t5 = fp4_sq(a1, b2)
    c00 = f2_sub(f2_scale(t0, 3), f2_scale(ra0, 2))
    c01 = f2_sub(f2_scale(t2, 3), f2_scale(ra1, 2))
    c02 = f2_sub(f2_scale(t4, 3), f2_scale(ra2, 2))
    c10 = f2_add(f2_scale(f2_mul_xi(t5), 3), f2_scale(rb0, 2))
    c11 = f2_add(f2_scale(t1, 3), f2_scale(rb1, 2))
    c12 = f2_add(f2_scale(t3, 3), f2_scale(rb2, 2))
    return ((c00, c01, c02), (c10, c11, c12))


def flat12(x):
    return [e for c6 in x for c2 in c6 for e in c2]


def unflat12(l):
    return ((( l[0], l[1]), (l[2], l[3]), (l[4], l[5])),
            ((l[6], l[7]), (l[8], l[9]), (l[10], l[11])))
