"""Host-side integer arithmetic for everything that is NOT the GPU hot path:
point (de)compression, hash-to-curve, key generation, small group operations
on Python ints.  Fq2 elements are pairs (c0, c1); curve points are Jacobian
triples with None for infinity.

Semantics follow the reference where they are observable:
  * Fq2 square root: complex method with the reference's quirk that an element
    with zero imaginary part must have a root IN Fq (fields.py:463-482);
  * sw_encode / Fouque-Tibouchi hashing and the Budroni-Pintore cofactor
    clearing of ec.py:449-550;
  * scalar multiplication returns infinity when c % q == 0 (fields_t.py:710).
The pairing itself never runs here: it goes to the HIP engine.
"""
from . import bls12381 as C

Q = C.q
N = C.n

# ------------------------------------------------------------------ Fq / Fq2


def fq_inv(a):
    return pow(a, Q - 2, Q)            # 0 -> 0, like the reference's fq_invert


def fq_sqrt(a):
    """Square root in Fq (q = 3 mod 4); ValueError if none (fields.py:199-205)."""
    a %= Q
    if a == 0:
        return 0
    if pow(a, (Q - 1) // 2, Q) != 1:
        raise ValueError("No sqrt exists")
    return pow(a, (Q + 1) // 4, Q)


def f2_add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


def f2_sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def f2_neg(a):
    return (-a[0] % Q, -a[1] % Q)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def f2_sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % Q, 2 * a[0] * a[1] % Q)


def f2_muli(a, k):
    return (a[0] * k % Q, a[1] * k % Q)


def f2_inv(a):
    f = fq_inv((a[0] * a[0] + a[1] * a[1]) % Q)
    return (a[0] * f % Q, -a[1] * f % Q)


def f2_conj(a):
    return (a[0], -a[1] % Q)


def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


class RealSquareRoot(Exception):
    """What the reference's y_for_x ends in for a G2 x whose u = x^3 + b' has zero imaginary part and a
    square real part: Fq2.modsqrt returns an Fq there (fields.py:466-467) and the AffinePoint constructor
    refuses mixed coordinate types with Exception('x,y should be field elements') (ec.py:24-30)."""


def f2_sqrt(a):
    """fields.py:463-482 (complex method).  Zero imaginary part: the reference's branch returns an
    element of Fq, which no caller can use -- ValueError when a0 is no square of Fq (fields.py:199-205),
    RealSquareRoot where the reference's caller y_for_x fails on the type of the root."""
    a0, a1 = a[0] % Q, a[1] % Q
    if a1 == 0:
        if a0 == 0:
            return (0, 0)
        fq_sqrt(a0)                                  # ValueError('No sqrt exists') for a non-residue
        raise RealSquareRoot("x,y should be field elements")
    alpha = (a0 * a0 + a1 * a1) % Q
    if pow(alpha, (Q - 1) // 2, Q) == Q - 1:
        raise ValueError("No sqrt exists")
    alpha = fq_sqrt(alpha)
    inv2 = fq_inv(2)
    delta = (a0 + alpha) * inv2 % Q
    if pow(delta, (Q - 1) // 2, Q) == Q - 1:
        delta = (a0 - alpha) * inv2 % Q
    x0 = fq_sqrt(delta)
    x1 = a1 * fq_inv(2 * x0 % Q) % Q
    return (x0, x1)


# ------------------------------------------------- generic short Weierstrass
class Field:
    """Tiny vtable so that G1 (Fq) and G2 (Fq2) share the group-law code."""

    def __init__(self, zero, one, add, sub, mul, sqr, neg, inv, muli, b, sqrt, is_zero):
        (self.zero, self.one, self.add, self.sub, self.mul, self.sqr, self.neg, self.inv,
         self.muli, self.b, self.sqrt, self.is_zero) = (zero, one, add, sub, mul, sqr, neg, inv,
                                                        muli, b, sqrt, is_zero)


F1 = Field(0, 1, lambda a, b: (a + b) % Q, lambda a, b: (a - b) % Q, lambda a, b: a * b % Q,
           lambda a: a * a % Q, lambda a: -a % Q, fq_inv, lambda a, k: a * k % Q, C.b % Q, fq_sqrt,
           lambda a: a % Q == 0)
F2 = Field((0, 0), (1, 0), f2_add, f2_sub, f2_mul, f2_sqr, f2_neg, f2_inv, f2_muli, C.b_twist, f2_sqrt,
           lambda a: a[0] % Q == 0 and a[1] % Q == 0)


def jac_double(F, P):
    """a = 0 Jacobian doubling (same formulas as fields_t.py:878-933)."""
    if P is None:
        return None
    X, Y, Z = P
    if F.is_zero(Y):
        return None
    ysq = F.sqr(Y)
    S = F.muli(F.mul(X, ysq), 4)
    M = F.muli(F.sqr(X), 3)
    X3 = F.sub(F.sqr(M), F.muli(S, 2))
    Y3 = F.sub(F.mul(M, F.sub(S, X3)), F.muli(F.sqr(ysq), 8))
    Z3 = F.muli(F.mul(Y, Z), 2)
    return (X3, Y3, Z3)


def jac_add(F, P, R):
    """General Jacobian addition (fields_t.py:762-875)."""
    if P is None:
        return R
    if R is None:
        return P
    X1, Y1, Z1 = P
    X2, Y2, Z2 = R
    z1s, z2s = F.sqr(Z1), F.sqr(Z2)
    u1, u2 = F.mul(X1, z2s), F.mul(X2, z1s)
    s1, s2 = F.mul(Y1, F.mul(z2s, Z2)), F.mul(Y2, F.mul(z1s, Z1))
    if u1 == u2:
        return jac_double(F, P) if s1 == s2 else None
    h, r = F.sub(u2, u1), F.sub(s2, s1)
    hs = F.sqr(h)
    hc = F.mul(h, hs)
    v = F.mul(u1, hs)
    X3 = F.sub(F.sub(F.sqr(r), hc), F.muli(v, 2))
    Y3 = F.sub(F.mul(r, F.sub(v, X3)), F.mul(s1, hc))
    Z3 = F.mul(F.mul(Z1, Z2), h)
    return (X3, Y3, Z3)


def jac_neg(F, P):
    return None if P is None else (P[0], F.neg(P[1]), P[2])


def jac_mul(F, P, c):
    """Double-and-add; infinity when c % q == 0 (fields_t.py:705-740)."""
    if P is None or c % Q == 0:
        return None
    if c < 0:
        raise ValueError("negative scalar")
    acc, add = None, P
    while c:
        if c & 1:
            acc = jac_add(F, acc, add)
        add = jac_double(F, add)
        c >>= 1
    return acc


def jac_to_affine(F, P):
    if P is None:
        return None
    if P[2] == F.one:                      # already normalised (deserialised keys, GPU results): no inversion
        return (P[0], P[1])
    zi = F.inv(P[2])
    zi2 = F.sqr(zi)
    return (F.mul(P[0], zi2), F.mul(P[1], F.mul(zi2, zi)))


def aff_to_jac(F, A):
    return None if A is None else (A[0], A[1], F.one)


def on_curve(F, A):
    if A is None:
        return True
    x, y = A
    return F.sqr(y) == F.add(F.mul(F.sqr(x), x), F.b)


def y_for_x(F, x):
    """Both roots of y^2 = x^3 + b, [y, -y]; ValueError if x is not on the curve
    (ec.py:255-269)."""
    u = F.add(F.mul(F.sqr(x), x), F.b)
    y = F.sqrt(u)
    if F.is_zero(y) or not on_curve(F, (x, y)):
        raise ValueError("No y for point x")
    return [y, F.neg(y)]


G1_GEN = (C.gx, C.gy)
G2_GEN = (C.g2x, C.g2y)

# ------------------------------------------------------------ serialisation


def fq_bytes(v):
    return int(v % Q).to_bytes(48, "big")


def g1_affine_bytes(A):
    """96-byte ABI form x || y; infinity = zeros."""
    return bytes(96) if A is None else fq_bytes(A[0]) + fq_bytes(A[1])


def g2_affine_bytes(A):
    return bytes(192) if A is None else b"".join(fq_bytes(c) for c in (A[0][0], A[0][1], A[1][0], A[1][1]))


def g1_from_abi(b):
    x, y = int.from_bytes(b[:48], "big"), int.from_bytes(b[48:96], "big")
    return None if x == 0 and y == 0 else (x, y)


def g2_from_abi(b):
    v = [int.from_bytes(b[48 * i:48 * (i + 1)], "big") for i in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def g1_compress(A):
    """48 bytes: x with bit 0x80 = "y > q//2" (ec.py:94-111); infinity -> x = 0."""
    if A is None:
        return bytes(48)
    out = bytearray(fq_bytes(A[0]))
    if A[1] > Q // 2:
        out[0] |= 0x80
    return bytes(out)


def g2_compress(A):
    """96 bytes x.c0 || x.c1; sign bit from the IMAGINARY part of y only (ec.py:98-100)."""
    if A is None:
        return bytes(96)
    out = bytearray(fq_bytes(A[0][0]) + fq_bytes(A[0][1]))
    if A[1][1] > Q // 2:
        out[0] |= 0x80
    return bytes(out)


def g1_decompress(buf):
    """keys.py:28-40"""
    big = buf[0] & 0x80
    x = int.from_bytes(bytes([buf[0] & 0x1f]) + buf[1:], "big") % Q
    ys = sorted(y_for_x(F1, x))
    return (x, ys[1] if big else ys[0])


def g2_decompress(buf):
    """signature.py:21-38"""
    big = buf[0] & 0x80
    b = bytes([buf[0] & 0x1f]) + buf[1:]
    x = (int.from_bytes(b[:48], "big") % Q, int.from_bytes(b[48:], "big") % Q)
    ys = y_for_x(F2, x)
    y = ys[0]
    if (big and ys[1][1] > Q // 2) or (not big and ys[1][1] < Q // 2):
        y = ys[1]
    return (x, y)


# ------------------------------------------------------------ hash to curve
def _gamma(j):
    return f2_pow((1, 1), j * (Q - 1) // 6)          # xi^(j (q-1)/6)


_PSI_X = f2_inv(_gamma(2))                          # w^(2 - 2q)
_PSI_Y = f2_inv(_gamma(3))                          # w^(3 - 3q)


def psi(A):
    """untwist -> Frobenius -> twist (ec.py:440-444), in twist coordinates."""
    if A is None:
        return None
    return (f2_mul(f2_conj(A[0]), _PSI_X), f2_mul(f2_conj(A[1]), _PSI_Y))


def _lex_gt_neg(F, y):
    return (y > Q // 2) if F is F1 else (y[1] > Q // 2)


def sw_encode(F, t):
    """Shallue-van de Woestijne encoding (ec.py:449-507).  t is a field element
    (int for G1, pair for G2); returns an affine point or None for infinity."""
    if F.is_zero(t):
        return None
    nt = F.neg(t)
    parity = (t > nt) if F is F1 else (t[1] > nt[1])
    one = F.one
    w = F.add(F.add(F.sqr(t), F.b), one)
    if F.is_zero(w):
        g = G1_GEN if F is F1 else G2_GEN
        if F is F1 and parity:
            return (g[0], F.neg(g[1]))
        return g
    s3 = C.sqrt_n3 if F is F1 else (C.sqrt_n3, 0)
    s3h = C.sqrt_n3m1o2 if F is F1 else (C.sqrt_n3m1o2, 0)
    w = F.mul(F.mul(F.inv(w), s3), t)
    x1 = F.add(F.neg(F.mul(w, t)), s3h)
    x2 = F.sub(F.neg(one), x1)
    x3 = F.add(F.inv(F.sqr(w)), one)

    def ok(x):
        try:
            y_for_x(F, x)
            return 1
        except (ValueError, RealSquareRoot):          # the reference: a bare `except` (ec.py:489-498)
            return -1
    a, b = ok(x1), ok(x2)
    x = (x1, x2, x3)[((a - 1) * b) % 3]
    y = y_for_x(F, x)[0]
    if _lex_gt_neg(F, y) is not parity:
        y = F.neg(y)
    return (x, y)


def hash_to_g1_prehashed(m, hash512):
    """ec.py:511-520"""
    t0 = int.from_bytes(hash512(m + b"G1_0"), "big") % Q
    t1 = int.from_bytes(hash512(m + b"G1_1"), "big") % Q
    P = jac_add(F1, aff_to_jac(F1, sw_encode(F1, t0)), aff_to_jac(F1, sw_encode(F1, t1)))
    return jac_to_affine(F1, jac_mul(F1, P, C.h))


def g2_hash_field_elements(m, hash512):
    """The four field elements of ec.py:531-534 as 192 bytes (t0.c0, t0.c1, t1.c0, t1.c1):
    the input of blsgpu_map_to_g2."""
    return b"".join(fq_bytes(int.from_bytes(hash512(m + tag), "big") % Q)
                    for tag in (b"G2_0_c0", b"G2_0_c1", b"G2_1_c0", b"G2_1_c1"))


def hash_to_g2_prehashed(m, hash512):
    """ec.py:528-550: two SW encodings, then Budroni-Pintore cofactor clearing."""
    def t(tag):
        return int.from_bytes(hash512(m + tag), "big") % Q
    t0 = (t(b"G2_0_c0"), t(b"G2_0_c1"))
    t1 = (t(b"G2_1_c0"), t(b"G2_1_c1"))
    return clear_cofactor_g2(jac_add(F2, aff_to_jac(F2, sw_encode(F2, t0)), aff_to_jac(F2, sw_encode(F2, t1))))


def clear_cofactor_g2(P):
    """ec.py:536-550 (Budroni-Pintore); Jacobian in, affine out."""
    x = -C.x
    aff = lambda J: jac_to_affine(F2, J)            # noqa: E731
    jac = lambda A: aff_to_jac(F2, A)               # noqa: E731
    psi2 = jac(psi(psi(aff(jac_double(F2, P)))))
    a0 = jac_mul(F2, P, x)
    a1 = jac_mul(F2, a0, x)
    a2 = jac_add(F2, jac_add(F2, a1, a0), jac_neg(F2, P))
    a3 = jac(psi(aff(jac_mul(F2, P, x + 1))))
    return aff(jac_add(F2, jac_add(F2, a2, jac_neg(F2, a3)), psi2))
