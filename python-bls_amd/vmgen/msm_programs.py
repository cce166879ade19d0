"""VM programs for G1 / G2 scalar multiplication and multi-scalar sums
(fq_/fq2_scalar_mult_jacobian and the sums of
bls.py:203-223, threshold.py:127-136).

The reference's algorithm is kept (LSB-first double-and-add, fields_t.py:705-740)
but on homogeneous projective coordinates with the COMPLETE a = 0 formulas of
Renes-Costello-Batina (2015, algorithms 7 and 9): no exceptional cases, so the
point at infinity (0:1:0), doubling-inside-add and P + (-P) need no branches.
E(Fq) and E'(Fq2) have odd order, which is what completeness needs.  Parity is
defined on the affine result, as in the reference (Jacobian coordinates are
representation dependent).

One team processes NP points in lock step (their formulas are packed into the
same rounds):   R_p <- R_p + S_p ;  A_p <- 2 A_p   per scalar bit, where the
kernel writes S_p = A_p or infinity according to the bit before each step.
"""
from . import tower as tw
from .core import Builder, schedule
from .programs import C_ONE, C_R2, C_RAW1, C_ZERO, NCONST


class FA:
    """Field adapter: coordinate arithmetic over Fq (deg 1) or Fq2 (deg 2)."""

    def __init__(self, deg, cfg):
        self.deg, self.cfg = deg, cfg

    def mul(self, a, b):
        return (a[0] * b[0],) if self.deg == 1 else self.cfg.mul2(a, b)

    def sqr(self, a):
        return (a[0] * a[0],) if self.deg == 1 else self.cfg.sqr2(a)

    def add(self, a, b):
        return tuple(x + y for x, y in zip(a, b))

    def sub(self, a, b):
        return tuple(x - y for x, y in zip(a, b))

    def scale(self, a, k):
        return tuple(x * k for x in a)

    def b3(self, a):
        """3b * a: b = 4 on G1, b' = 4 xi on the twist."""
        return self.scale(a, 12) if self.deg == 1 else tw.f2_scale(tw.f2_mul_xi(a), 12)

    def mat(self, a):
        return tuple(x.mat() for x in a)


def padd(F, P1, P2):
    """Complete addition, RCB algorithm 7 (a = 0): 12 multiplications."""
    X1, Y1, Z1 = P1
    X2, Y2, Z2 = P2
    t0, t1, t2 = F.mul(X1, X2), F.mul(Y1, Y2), F.mul(Z1, Z2)
    t3 = F.sub(F.sub(F.mul(F.add(X1, Y1), F.add(X2, Y2)), t0), t1)
    t4 = F.sub(F.sub(F.mul(F.add(Y1, Z1), F.add(Y2, Z2)), t1), t2)
    t5 = F.sub(F.sub(F.mul(F.add(X1, Z1), F.add(X2, Z2)), t0), t2)
    x3 = F.scale(t0, 3)
    bz = F.b3(t2)
    z3 = F.add(t1, bz)
    t1m = F.sub(t1, bz)
    y3 = F.b3(t5)
    X3 = F.sub(F.mul(t3, t1m), F.mul(t4, y3))
    Y3 = F.add(F.mul(t1m, z3), F.mul(y3, x3))
    Z3 = F.add(F.mul(z3, t4), F.mul(x3, t3))
    return (X3, Y3, Z3)


def pdbl(F, P):
    """Complete doubling, RCB algorithm 9 (a = 0): 6M + 2S."""
    X, Y, Z = P
    t0 = F.sqr(Y)
    t1 = F.mul(Y, Z)
    t2 = F.b3(F.sqr(Z))
    txy = F.mul(X, Y)
    z8 = F.scale(t0, 8)
    d = F.sub(t0, F.scale(t2, 3))
    X3 = F.scale(F.mul(d, txy), 2)
    Y3 = F.add(F.mul(t2, z8), F.mul(d, F.add(t0, t2)))
    Z3 = F.mul(t1, z8)
    return (X3, Y3, Z3)


class Layout:
    """Slot map of an MSM team (after the shared constants)."""

    def __init__(self, deg, NP):
        self.deg, self.NP = deg, NP
        c = deg
        o = NCONST
        self.IN = o; o += NP * 2 * c          # raw affine inputs
        self.R = o; o += NP * 3 * c           # accumulators
        self.A = o; o += NP * 3 * c           # running doubles
        self.S = o; o += NP * 3 * c           # selected addends (written by the kernel)
        self.PR0 = o; o += 3 * c              # point registers for reduction trees
        self.PR1 = o; o += 3 * c
        self.OUT = o; o += 2 * c              # canonical affine output
        self.TEMP0 = o

    def pt(self, base, p=0):
        return base + p * 3 * self.deg


def _in_pt(b, lay, base, p=0):
    c = lay.deg
    o = lay.pt(base, p)
    return tuple(tuple(b.inp(o + k * c + i) for i in range(c)) for k in range(3))


def _out_pt(b, lay, P, base, p=0, zero=None):
    c = lay.deg
    o = lay.pt(base, p)
    for k in range(3):
        for i in range(c):
            e = P[k][i]
            b.out(e if not e.is_zero() else zero, o + k * c + i)


def build(deg, NP, cfg=None, verbose=False):
    cfg = cfg or tw.Cfg(mat2=False)
    F = FA(deg, cfg)
    lay = Layout(deg, NP)
    c = deg
    tag = "g%d" % deg
    builders = []
    # raw affine (x, y) -> Montgomery projective (x, y, 1) in A_p
    b = Builder(tag + "_load")
    r2, one, zero = b.inp(C_R2), b.inp(C_ONE), b.inp(C_ZERO)
    for p in range(NP):
        for k in range(2):
            for i in range(c):
                b.out(b.inp(lay.IN + (p * 2 + k) * c + i) * r2, lay.pt(lay.A, p) + k * c + i)
        b.out(one, lay.pt(lay.A, p) + 2 * c)
        for i in range(1, c):
            b.out(zero, lay.pt(lay.A, p) + 2 * c + i)
    builders.append(b)
    # one scalar bit for NP points
    b = Builder(tag + "_step")
    zero = b.inp(C_ZERO)
    for p in range(NP):
        R, S, A = _in_pt(b, lay, lay.R, p), _in_pt(b, lay, lay.S, p), _in_pt(b, lay, lay.A, p)
        _out_pt(b, lay, padd(F, R, S), lay.R, p, zero)
        _out_pt(b, lay, pdbl(F, A), lay.A, p, zero)
    builders.append(b)
    # bucket method: R_p <- R_p + S_p for the NP points (no doubling)
    b = Builder(tag + "_acc")
    zero = b.inp(C_ZERO)
    for p in range(NP):
        R, S = _in_pt(b, lay, lay.R, p), _in_pt(b, lay, lay.S, p)
        _out_pt(b, lay, padd(F, R, S), lay.R, p, zero)
    builders.append(b)
    # PR0 <- 2 PR0 (window shifts of the bucket method)
    b = Builder(tag + "_dbl")
    zero = b.inp(C_ZERO)
    _out_pt(b, lay, pdbl(F, _in_pt(b, lay, lay.PR0)), lay.PR0, 0, zero)
    builders.append(b)
    # R_0 <- R_0 + R_1 + ... (balanced tree)
    b = Builder(tag + "_fold")
    zero = b.inp(C_ZERO)
    pts = [_in_pt(b, lay, lay.R, p) for p in range(NP)]
    while len(pts) > 1:
        nxt = []
        for i in range(0, len(pts) - 1, 2):
            s = padd(F, pts[i], pts[i + 1])
            nxt.append(tuple(F.mat(cc) for cc in s))
        if len(pts) % 2:
            nxt.append(pts[-1])
        pts = nxt
    _out_pt(b, lay, pts[0], lay.PR0, 0, zero)
    builders.append(b)
    # PR0 <- PR0 + PR1
    b = Builder(tag + "_padd")
    zero = b.inp(C_ZERO)
    _out_pt(b, lay, padd(F, _in_pt(b, lay, lay.PR0), _in_pt(b, lay, lay.PR1)), lay.PR0, 0, zero)
    builders.append(b)
    # OUT <- canonical affine of PR0 (Z = 0 gives (0, 0): the reference's infinity)
    b = Builder(tag + "_affine")
    raw1, zero = b.inp(C_RAW1), b.inp(C_ZERO)
    X, Y, Z = _in_pt(b, lay, lay.PR0)
    if deg == 1:
        zi = (Z[0].inv(),)
    else:
        zi = tw.f2_inv(cfg, Z)
    xa, ya = F.mul(X, zi), F.mul(Y, zi)
    for k, v in enumerate((xa, ya)):
        for i in range(c):
            e = v[i].mat() * raw1
            b.out(e if not e.is_zero() else zero, lay.OUT + k * c + i)
    builders.append(b)
    segs = {}
    for b in builders:
        segs[b.name] = schedule(b, temp_base=lay.TEMP0, verbose=verbose)
    return segs, lay


class HornerLayout:
    """Slot map of a Horner team (k_msm_horner_np): NP running sums R_p, NP addends S_p, NP affine outputs."""

    def __init__(self, deg, NP):
        self.deg, self.NP = deg, NP
        o = NCONST
        self.R = o; o += NP * 3 * deg
        self.S = o; o += NP * 3 * deg
        self.OUT = o; o += NP * 2 * deg
        self.TEMP0 = o

    def pt(self, base, p=0):
        return base + p * 3 * self.deg


def build_horner(deg, NP, cfg=None, verbose=False):
    """The window Horner of a BATCH of sums (k_msm_pip_horner does one sum per team and leaves most lanes idle: a
    doubling is a handful of products): NP sums per team in lock step.
      <tag>_dbl:    R_p <- 2 R_p
      <tag>_acc:    R_p <- R_p + S_p
      <tag>_affine: OUT_p <- canonical affine of R_p (Z = 0 gives (0, 0))"""
    cfg = cfg or tw.Cfg(mat2=False)
    F = FA(deg, cfg)
    lay = HornerLayout(deg, NP)
    c = deg
    tag = "g%dh" % deg
    builders = []
    b = Builder(tag + "_dbl")
    zero = b.inp(C_ZERO)
    for p in range(NP):
        _out_pt(b, lay, pdbl(F, _in_pt(b, lay, lay.R, p)), lay.R, p, zero)
    builders.append(b)
    b = Builder(tag + "_acc")
    zero = b.inp(C_ZERO)
    for p in range(NP):
        _out_pt(b, lay, padd(F, _in_pt(b, lay, lay.R, p), _in_pt(b, lay, lay.S, p)), lay.R, p, zero)
    builders.append(b)
    b = Builder(tag + "_affine")
    raw1, zero = b.inp(C_RAW1), b.inp(C_ZERO)
    for p in range(NP):
        X, Y, Z = _in_pt(b, lay, lay.R, p)
        zi = (Z[0].inv(),) if deg == 1 else tw.f2_inv(cfg, Z)
        for k, v in enumerate((F.mul(X, zi), F.mul(Y, zi))):
            for i in range(c):
                e = v[i].mat() * raw1
                b.out(e if not e.is_zero() else zero, lay.OUT + (p * 2 + k) * c + i)
    builders.append(b)
    segs = {}
    for b in builders:
        segs[b.name] = schedule(b, temp_base=lay.TEMP0, verbose=verbose)
    return segs, lay
