"""The VM programs of the pairing hot path: segments, memory map, scripts.

Algorithm (what the GPU runs) versus the reference (what it must equal):

* Miller loop over |x| = 0xd201000000010000, MSB-1 .. 0, NO final conjugation
  -- exactly the loop of fields_t.py:1091-1111 -- but with the twist point T in
  homogeneous projective coordinates and the line l(P) scaled by elements that
  the final exponentiation kills (Fq2 factors and w^3):
      tangent:  l = (Y^2 - 3b'Z^2) + (-3 X^2 px) v + (2YZ py) v w
      chord  :  l = (th xq - la yq) + (-th px) v + (la py) v w,
                th = Y - yq Z, la = X - xq Z
  so that f <- f^2 * l is a 13-Fq2-product sparse multiplication.
* Final exponentiation to the SAME power (q^12-1)/n as fields_t.py:1124-1128,
  split as easy part (q^6-1)(q^2+1) then E = (q^4-q^2+1)/n with the exact
  identity  E = ((x-1)^2/3)(x+q)(x^2+q^2-1) + 1  (x = -|x|), using cyclotomic
  squarings.  The result is the identical field element, hence identical bytes.
"""
from . import tower as tw
from . import core
from .core import Builder, schedule
from .sim import Q, R, to_m

NX = 0xd201000000010000                   # |x|  (fields_t.py:25)

# ------------------------------------------------------------- memory map --
# One scratchpad per team (= wavefront = pairing), in 48-byte slots.
# constants (copied in by the kernel at start-up)
C_ZERO, C_ONE, C_R2, C_RAW1 = 0, 1, 2, 3
C_K1 = 4                                  # -(2^384 - 1) mod q  (LIN complement surplus)
C_GAM = 5                                 # gamma_i^j, i=1..3, j=1..5: Fq2 each (final exponentiation only)
C_HALF = C_GAM + 3 * 5 * 2                # 1/2 (the reference-faithful Miller program: xi^-1 = (1 - u)/2)
NCONST = C_HALF + 1                       # 36
# The multi-pair Miller programs never touch the gamma constants: their scratchpad
# (MPLayout) starts right after C_K1.
# named values
PX, PY = 36, 37
QX0, QX1, QY0, QY1 = 38, 39, 40, 41
TX, TY, TZ = 42, 44, 46                   # Fq2 each
LD = 48                                   # pending tangent line  l0,l1,l4 (6)
LA = 54                                   # pending chord line    l0,l1,l4 (6)
NPX3 = 60                                 # -3*px
REG0 = 62                                 # Fq12 registers R[k] = REG0 + 12 k
NREG = 5
F = REG0                                  # the Miller accumulator is register 0
TEMP0 = REG0 + 12 * NREG                  # 122


core.K1_SLOT = C_K1


def reg(k):
    assert 0 <= k < NREG
    return REG0 + 12 * k


def _fq2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = ((r[0] * a[0] - r[1] * a[1]) % Q, (r[0] * a[1] + r[1] * a[0]) % Q)
        a = ((a[0] * a[0] - a[1] * a[1]) % Q, (2 * a[0] * a[1]) % Q)
        e >>= 1
    return r


def gamma(i, j):
    """xi^(j (q^i - 1)/6) in Fq2 (plain domain)."""
    return _fq2_pow((1, 1), j * (Q ** i - 1) // 6)


def const_table():
    """Montgomery contents of the constant region."""
    c = [0] * NCONST
    c[C_ZERO] = 0
    c[C_ONE] = to_m(1)
    c[C_R2] = to_m(R % Q)          # content R^2: raw x -> x R
    c[C_RAW1] = 1                  # content 1: x R -> x
    c[C_K1] = (-(R - 1)) % Q       # plain integer used by the LIN accumulation (not a field element)
    for i in (1, 2, 3):
        for j in range(1, 6):
            g = gamma(i, j)
            k = C_GAM + ((i - 1) * 5 + (j - 1)) * 2
            c[k], c[k + 1] = to_m(g[0]), to_m(g[1])
    c[C_HALF] = to_m((Q + 1) // 2)
    return c


def T(off):
    return off


def C(off):
    return off


def in2(b, off, name=None):
    return (b.inp(T(off), name), b.inp(T(off + 1)))


def out2(b, e, off):
    b.out(e[0], T(off))
    b.out(e[1], T(off + 1))


def in12(b, base):
    return tw.unflat12([b.inp(base + i) for i in range(12)])


def out12(b, x, base, zero=None):
    for i, e in enumerate(tw.flat12(x)):
        b.out(e if not e.is_zero() else zero, base + i)


# ------------------------------------------------------------ Miller loop --
def xi3b_times(c):
    """3 b' c with b' = 4 xi:  12 xi c   (linear)"""
    return tw.f2_scale(tw.f2_mul_xi(c), 12)


def t_double(cfg, Tp, px3n, py):
    """Tangent step.  Tp = (X, Y, Z) Fq2 each.  Returns (T2, line).
    Homogeneous doubling on y^2 = x^3 + b' scaled by 4 to avoid halving:
      X3 = 2 XY (B - F), Y3 = (B + F)^2 - 12 E^2, Z3 = 4 B H
      B = Y^2, C = Z^2, E = 3b'C, F = 3E, H = 2YZ
    line: l0 = B - E, l1 = X^2 * (-3 px), l4 = H * py"""
    X, Y, Z = Tp
    A = cfg.mul2(X, Y)
    B = cfg.sqr2(Y)
    Cc = cfg.sqr2(Z)
    XX = cfg.sqr2(X)
    YZ = cfg.mul2(Y, Z)
    E = xi3b_times(Cc)
    Fv = tw.f2_scale(E, 3)
    H = tw.f2_scale(YZ, 2)
    X3 = tw.f2_scale(cfg.mul2(A, tw.f2_sub(B, Fv)), 2)
    G = tw.f2_add(B, Fv)
    Y3 = tw.f2_sub(cfg.sqr2(G), tw.f2_scale(cfg.sqr2(E), 12))
    Z3 = tw.f2_scale(cfg.mul2(B, H), 4)
    l0 = tw.f2_sub(B, E)
    l1 = tw.f2_mul_fq(XX, px3n)
    l4 = tw.f2_mul_fq(H, py)
    return (X3, Y3, Z3), (l0, l1, l4)


def t_add(cfg, Tp, Qa, px, py, px_is_m3=False):
    """Chord step T + Q (Q affine).  px_is_m3: `px` holds -3 px (the multi-pair layout keeps
    only that multiple); the line is then returned times 3, a factor in Fq that the final
    exponentiation removes like the other line scalings.  Mixed addition:
      th = Y - yq Z, la = X - xq Z, C = th^2, D = la^2, E = la D, Fz = Z C,
      G = X D, H = E + Fz - 2G, X3 = la H, Y3 = th (G - H) - E Y, Z3 = Z E
    line: l0 = th xq - la yq, l1 = -th px, l4 = la py"""
    X, Y, Z = Tp
    xq, yq = Qa
    th = tw.f2_mat(tw.f2_sub(Y, cfg.mul2(yq, Z)))
    la = tw.f2_mat(tw.f2_sub(X, cfg.mul2(xq, Z)))
    Cc = cfg.sqr2(th)
    D = cfg.sqr2(la)
    E = cfg.mul2(la, D)
    Fz = cfg.mul2(Z, Cc)
    G = cfg.mul2(X, D)
    H = tw.f2_mat(tw.f2_sub(tw.f2_add(E, Fz), tw.f2_scale(G, 2)))
    X3 = cfg.mul2(la, H)
    Y3 = tw.f2_sub(cfg.mul2(th, tw.f2_sub(G, H)), cfg.mul2(E, Y))
    Z3 = cfg.mul2(Z, E)
    l0 = tw.f2_sub(cfg.mul2(th, xq), cfg.mul2(la, yq))
    if px_is_m3:
        return (X3, Y3, Z3), (tw.f2_scale(l0, 3), tw.f2_mul_fq(th, px), tw.f2_scale(tw.f2_mul_fq(la, py), 3))
    l1 = tw.f2_neg(tw.f2_mul_fq(th, px))
    l4 = tw.f2_mul_fq(la, py)
    return (X3, Y3, Z3), (l0, l1, l4)


MP_G = 3                                   # pairs per team in the multi-pair programs


class PairSlots:
    """slots of the one pair of the single-pair programs (the named values above)"""
    F = F

    def __init__(self, g=0):
        assert g == 0
        o = PX
        self.PX, self.PY, self.QX0, self.QY0 = o, o + 1, o + 2, o + 4
        self.TX, self.TY, self.TZ = o + 6, o + 8, o + 10
        self.LD, self.LA, self.NPX3 = o + 12, o + 18, o + 24


assert PairSlots(0).TX == TX and PairSlots(0).LD == LD and PairSlots(0).LA == LA and PairSlots(0).NPX3 == NPX3


class MPLayout:
    """Scratchpad of the multi-pair Miller programs (G pairs per team, one accumulator), in
    the slot numbers the kernel uses.  Ordered by lifetime so that the scheduler's
    temporaries can start as low as possible -- the scratchpad size decides how many teams a
    compute unit holds, and every further team per CU is worth ~5 %:
      constants the Miller loop reads (C_ZERO .. C_K1) | F (12) | per pair PX PY T(6) LD(6) |
      the Q window (4 per pair) | temporaries.
    The chord lines never reach a slot: a chord step is fused into the body that multiplies its
    line in (seg_body_chord).  Q is an input that only those five bodies read: after the first
    segment has brought it into Montgomery form the kernel keeps it in three registers per
    lane (a SAVE round) and writes it back into the window in front of such a body (RESTORE
    round); for every other segment the window is temporaries (TEMP_LO)."""

    def __init__(self, G):
        self.G = G
        o = C_GAM
        self.F = o; o += 12
        self.CORE = o; o += 14 * G
        self.Q = o; o += 4 * G
        self.TEMP_LO, self.TEMP_HI = self.Q, o
        assert 4 * G * 12 <= 3 * 64, "the Q window must fit three registers per lane"

    def pair(self, g):
        lay = self

        class S:
            F = lay.F
            PX, PY = lay.CORE + 14 * g, lay.CORE + 14 * g + 1
            TX, TY, TZ = PX + 2, PX + 4, PX + 6
            LD = PX + 8
            QX0, QY0 = lay.Q + 4 * g, lay.Q + 4 * g + 2
            LA = None                      # chord lines live inside seg_body_chord only
            NPX3 = None                    # the PX slot itself holds -3 px here (seg_init)
        return S


def _out_t(b, ps, T2, ld, la, zero):
    _out2z(b, T2[0], ps.TX, zero), _out2z(b, T2[1], ps.TY, zero), _out2z(b, T2[2], ps.TZ, zero)
    for i, c in enumerate(ld):
        _out2z(b, c, ps.LD + 2 * i, zero)
    if la is not None:
        for i, c in enumerate(la):
            _out2z(b, c, ps.LA + 2 * i, zero)


def _out2z(b, e, off, zero):
    b.out(e[0] if not e[0].is_zero() else zero, T(off))
    b.out(e[1] if not e[1].is_zero() else zero, T(off + 1))


def seg_init(cfg, first_add, G=1, name="init", lay=None):
    """Raw inputs -> Montgomery; T = Q, F = 1; first tangent(+chord) step, for G pairs."""
    b = Builder(name)
    r2 = b.inp(C(C_R2))
    one = b.inp(C(C_ONE))
    zero = b.inp(C(C_ZERO))
    slots = lay.pair if lay else PairSlots
    Fb = slots(0).F
    assert 2 * G <= 11
    b.out(one, T(Fb))
    for i in range(1 + 2 * G, 12):
        b.out(zero, T(Fb + i))
    for g in range(G):
        ps = slots(g)
        px = (b.inp(T(ps.PX)) * r2).mat()
        py = (b.inp(T(ps.PY)) * r2).mat()
        q = [(b.inp(T(ps.QX0 + i)) * r2).mat() for i in range(4)]
        b.out(px, T(ps.PX)), b.out(py, T(ps.PY))
        for i in range(4):
            b.out(q[i], T(ps.QX0 + i))
        px3n = (px * -3).mat()
        m3 = ps.NPX3 is None                 # multi-pair layout: the PX slot itself holds -3 px
        if m3:
            b.outputs = [(v, fx) for v, fx in b.outputs if fx != T(ps.PX)]
            b.out(px3n, T(ps.PX))
            px = px3n
        else:
            b.out(px3n, T(ps.NPX3))
        Tp = ((q[0], q[1]), (q[2], q[3]), (one, b.zero()))
        Qa = ((q[0], q[1]), (q[2], q[3]))
        # Is Q on the twist?  d = qy^2 - qx^3 - 4 xi goes where the accumulator's zero
        # coefficients 1 + 2g, 2 + 2g would: for a point of the curve that IS a zero (possibly
        # in its relaxed form q), and the kernel looks at it right after this segment -- the
        # projective formulas below use the curve equation, the reference's affine ones do not,
        # so a pair with d != 0 is handed to the reference-faithful program (slow_programs.py)
        x3 = cfg.mul2(cfg.sqr2(Qa[0]), Qa[0])
        yy = cfg.sqr2(Qa[1])
        for c in range(2):
            b.out(yy[c] - x3[c] - one * 4, T(Fb + 1 + 2 * g + c))
        # Z = (1, 0): products with the zero imaginary part vanish at trace time
        T2, ld = t_double(cfg, Tp, px3n, py)
        la = None
        if first_add:
            T2 = tuple(tw.f2_mat(c) for c in T2)
            T2, la = t_add(cfg, T2, Qa, px, py, m3)
        _out_t(b, ps, T2, ld, la, zero)
    return b


def seg_body(cfg, cur_add, nxt, G=1, prefix="body", lay=None):
    """One pipelined Miller iteration for G pairs sharing the accumulator:
         f <- f^2 * prod_g LD_g (* LA_g if cur_add)    [lines of the current step]
         (T_g, LD_g, LA_g) <- next step of each T chain [nxt: 0 tangent, 1 tangent+chord,
                                                         2 nothing (last iteration)]"""
    b = Builder("%s_%d%d" % (prefix, cur_add, nxt))
    slots = lay.pair if lay else PairSlots
    Fb = slots(0).F
    f = in12(b, Fb)
    f = tw.f12_sqr(cfg, f)
    for g in range(G):
        ps = slots(g)
        ld = [in2(b, ps.LD + 2 * i) for i in range(3)]
        f = tw.f12_mul_by_014(cfg, f, *ld)
        if cur_add:
            la = [in2(b, ps.LA + 2 * i) for i in range(3)]
            f = tw.f12_mul_by_014(cfg, f, *la)
    out12(b, f, Fb)
    if nxt != 2:
        zero = b.inp(C(C_ZERO))
        for g in range(G):
            ps = slots(g)
            Tp = (in2(b, ps.TX), in2(b, ps.TY), in2(b, ps.TZ))
            Qa = (in2(b, ps.QX0), in2(b, ps.QY0))
            px, py = b.inp(T(ps.PX)), b.inp(T(ps.PY))
            m3 = ps.NPX3 is None             # multi-pair layout: the PX slot holds -3 px
            px3n = px if m3 else b.inp(T(ps.NPX3))
            T2, ldn = t_double(cfg, Tp, px3n, py)
            lan = None
            if nxt == 1:
                T2 = tuple(tw.f2_mat(c) for c in T2)
                T2, lan = t_add(cfg, T2, Qa, px, py, m3)
            _out_t(b, ps, T2, ldn, lan, zero)
    return b


def seg_body_chord(cfg, nxt, G, lay, prefix="mp_body"):
    """A Miller iteration whose step has a chord, for G pairs sharing the accumulator, with the
    chord step fused in:   (T_g, LA_g) <- T_g + Q_g ;  f <- f^2 * prod_g LD_g LA_g ;
    (T_g, LD_g) <- tangent step of T_g (nxt = 0).  The squaring and the tangent-line products
    overlap the chord steps, and the chord lines never need a slot."""
    assert nxt == 0
    b = Builder("%s_c%d" % (prefix, nxt))
    zero = b.inp(C(C_ZERO))
    f = tw.f12_sqr(cfg, in12(b, lay.F))
    st = []
    for g in range(G):
        ps = lay.pair(g)
        Tp = (in2(b, ps.TX), in2(b, ps.TY), in2(b, ps.TZ))
        Qa = (in2(b, ps.QX0), in2(b, ps.QY0))
        px, py = b.inp(T(ps.PX)), b.inp(T(ps.PY))
        ld = [in2(b, ps.LD + 2 * i) for i in range(3)]
        T2, la = t_add(cfg, Tp, Qa, px, py, True)
        st.append((ps, T2, la, ld, px, py))
    for ps, T2, la, ld, px, py in st:                # the tangent lines are there at once,
        f = tw.f12_mul_by_014(cfg, f, *ld)
    for ps, T2, la, ld, px, py in st:                # the chord lines four product levels later
        f = tw.f12_mul_by_014(cfg, f, *la)
    out12(b, f, lay.F)
    for ps, T2, la, ld, px, py in st:
        T3, ldn = t_double(cfg, tuple(tw.f2_mat(c) for c in T2), px, py)
        _out_t(b, ps, T3, ldn, None, zero)
    return b


def miller_script(init="init", prefix="body"):
    """[(segment name)] for the whole loop.  Step p (p = 62 .. 0) multiplies by
    the chord line iff bit p of |x| is set (fields_t.py:1104)."""
    bits = [(NX >> p) & 1 for p in range(62, -1, -1)]
    script = [init]
    for k, bit in enumerate(bits):
        nxt = 2 if k + 1 == len(bits) else bits[k + 1]
        script.append("%s_%d%d" % (prefix, bit, nxt))
    return script, bits[0]


# ------------------------------------------- Fq12 register machine pieces --
# Segments are specialised per register tuple; names encode the operands.
def seg_mul(cfg, d, a):
    b = Builder("mul_%d_%d" % (d, a))     # R[d] <- R[d] * R[a]
    x, y = in12(b, reg(d)), in12(b, reg(a))
    out12(b, tw.f12_mul(cfg, x, y), reg(d))
    return b


CYC_CHUNKS = (16, 8, 4, 2, 1)             # lengths of the chained-squaring segments


def seg_cyc_sqr(cfg, n, d):
    """R[d] <- R[d]^(2^n) (cyclotomic subgroup only).  The n squarings are chained
    lazily: each keeps a materialised copy of its result (for the next one's +-2x
    terms and as the output) while the operand pre-additions of the next squaring
    are taken from the unmaterialised expressions, so that one LIN level sits
    between consecutive MUL rounds: 2n + 1 rounds instead of 4n."""
    b = Builder("cyc_sqr%d_%d" % (n, d))
    lazy = tw.Cfg(m2=cfg.m2, s2=cfg.s2, mat2=False)
    xm = in12(b, reg(d))
    xl = xm
    for _ in range(n):
        y = tw.f12_cyclo_sqr(lazy, xl, ref=xm)
        xm = tuple(tuple(tw.f2_mat(c) for c in h) for h in y)
        xl = y
    out12(b, xm, reg(d))
    return b


def seg_addsub(d, a, sign):
    b = Builder("%s_%d_%d" % ("add" if sign > 0 else "sub", d, a))    # R[d] <- R[d] +- R[a]  (fq12_add / fq12_sub, fields_t.py:339-352)
    x, y = [b.inp(reg(d) + i) for i in range(12)], [b.inp(reg(a) + i) for i in range(12)]
    for i in range(12):
        b.out(x[i] + y[i] if sign > 0 else x[i] - y[i], reg(d) + i)
    return b


def seg_neg(d, a):
    b = Builder("neg_%d_%d" % (d, a))      # R[d] <- -R[a]
    for i in range(12):
        b.out(-b.inp(reg(a) + i), reg(d) + i)
    return b


def seg_copy(d, a):
    b = Builder("copy_%d_%d" % (d, a))    # R[d] <- R[a]
    out12(b, in12(b, reg(a)), reg(d))
    return b


def seg_conj(d, a):
    b = Builder("conj_%d_%d" % (d, a))    # R[d] <- conj(R[a])
    out12(b, tw.f12_conj(in12(b, reg(a))), reg(d))
    return b


def seg_frob(cfg, i, d, a):
    b = Builder("frob%d_%d_%d" % (i, d, a))   # R[d] <- R[a]^(q^i)
    x = in12(b, reg(a))

    def gam(j):
        k = C_GAM + ((i - 1) * 5 + (j - 1)) * 2
        return (b.inp(C(k)), b.inp(C(k + 1)))
    out12(b, tw.f12_frob(cfg, x, i, gam), reg(d), zero=b.inp(C(C_ZERO)))
    return b


def seg_inv12(cfg, d, a):
    b = Builder("inv12_%d_%d" % (d, a))   # R[d] <- R[a]^-1   (0 -> 0)
    out12(b, tw.f12_inv(cfg, in12(b, reg(a))), reg(d))
    return b


def seg_from_mont(d, a):
    b = Builder("from_mont_%d_%d" % (d, a))   # R[d] <- canonical (non-Montgomery) R[a]
    raw1 = b.inp(C(C_RAW1))
    for i in range(12):
        b.out(b.inp(reg(a) + i) * raw1, reg(d) + i)
    return b


def seg_to_mont(d, a):
    b = Builder("to_mont_%d_%d" % (d, a))     # R[d] <- Montgomery form of raw R[a]
    r2 = b.inp(C(C_R2))
    for i in range(12):
        b.out(b.inp(reg(a) + i) * r2, reg(d) + i)
    return b


def seg_set_one(d):
    b = Builder("set_one_%d" % d)         # R[d] <- 1
    one, zero = b.inp(C(C_ONE)), b.inp(C(C_ZERO))
    b.out(one, reg(d))
    for i in range(1, 12):
        b.out(zero, reg(d) + i)
    return b


SEG_FACTORY = {
    "mul": lambda cfg, *r: seg_mul(cfg, *r),
    "cyc_sqr16": lambda cfg, *r: seg_cyc_sqr(cfg, 16, *r),
    "cyc_sqr8": lambda cfg, *r: seg_cyc_sqr(cfg, 8, *r),
    "cyc_sqr4": lambda cfg, *r: seg_cyc_sqr(cfg, 4, *r),
    "cyc_sqr2": lambda cfg, *r: seg_cyc_sqr(cfg, 2, *r),
    "cyc_sqr1": lambda cfg, *r: seg_cyc_sqr(cfg, 1, *r),
    "copy": lambda cfg, *r: seg_copy(*r),
    "add": lambda cfg, *r: seg_addsub(*r, 1),
    "sub": lambda cfg, *r: seg_addsub(*r, -1),
    "neg": lambda cfg, *r: seg_neg(*r),
    "conj": lambda cfg, *r: seg_conj(*r),
    "frob1": lambda cfg, *r: seg_frob(cfg, 1, *r),
    "frob2": lambda cfg, *r: seg_frob(cfg, 2, *r),
    "frob3": lambda cfg, *r: seg_frob(cfg, 3, *r),
    "inv12": lambda cfg, *r: seg_inv12(cfg, *r),
    "from_mont": lambda cfg, *r: seg_from_mont(*r),
    "to_mont": lambda cfg, *r: seg_to_mont(*r),
    "set_one": lambda cfg, *r: seg_set_one(*r),
}


def seg_by_name(cfg, name):
    parts = name.split("_")
    n = len(parts)
    while n > 0 and parts[n - 1].isdigit():
        n -= 1
    return SEG_FACTORY["_".join(parts[:n])](cfg, *[int(p) for p in parts[n:]])


# ---------------------------------------------------- final exponentiation --
def final_exp_script():
    """Segment names, executed in order.  Input and output in register 0.

    easy:  t = conj(f) * f^-1 ;  t = frob2(t) * t
    hard:  y = t^E,  E = ((x-1)^2/3)(x+q)(x^2+q^2-1) + 1,  x = -|x|:
       s = t^e1 ; e1 = (|x|+1)/3                [(x-1)/3 = -e1]
       a = s^|x| * s                            [= t^((x-1)^2/3)]
       b = conj(a^|x|) * frob1(a)               [= a^(x+q)]
       c = (b^|x|)^|x| * frob2(b) * conj(b)     [= b^(x^2+q^2-1)]
       y = c * t
    Register use: r0 accumulator, r1 power base / second operand, r2..r4 saves.
    """
    S = []
    e1 = (NX + 1) // 3
    assert (NX + 1) % 3 == 0

    def pow_acc(e):
        """r0 <- r0^e (cyclotomic), clobbers r1."""
        S.append("copy_1_0")
        run = 0

        def flush(run):
            for n in CYC_CHUNKS:
                while run >= n:
                    S.append("cyc_sqr%d_0" % n)
                    run -= n
        for bit in range(e.bit_length() - 2, -1, -1):
            run += 1
            if (e >> bit) & 1:
                flush(run)
                run = 0
                S.append("mul_0_1")
        flush(run)
    # easy part
    S += ["inv12_2_0", "conj_1_0", "mul_1_2",      # r1 = f^(q^6-1)
          "frob2_0_1", "mul_0_1",                  # r0 = t (cyclotomic from here on)
          "copy_3_0"]                              # r3 = t
    # hard part
    pow_acc(e1)                                    # r0 = s = t^e1, r1 = t
    pow_acc(NX)                                    # r0 = s^|x|,   r1 = s
    S += ["mul_0_1", "copy_4_0"]                   # r0 = r4 = a
    pow_acc(NX)                                    # r0 = a^|x|
    S += ["conj_0_0", "frob1_1_4", "mul_0_1",      # r0 = b = a^x * a^q
          "copy_4_0"]                              # r4 = b
    pow_acc(NX)
    pow_acc(NX)                                    # r0 = b^(x^2)
    S += ["frob2_1_4", "mul_0_1", "conj_1_4", "mul_0_1",   # r0 = c
          "mul_0_3"]                               # r0 = c * t
    return S


# ------------------------------------------------------------- build all ----
def lazy_cfg(cfg):
    """Fq2 products left unmaterialised: their post-additions are folded into the next
    LIN level (fewer, longer LIN rounds -- the emitter splits long ones over lanes).
    Costs a few more live temporaries, so the rare chord-step bodies keep cfg."""
    return tw.Cfg(m2=cfg.m2, s2=cfg.s2, mat2=False)


def build_all(cfg=None, verbose=False):
    """Returns (segments by name, miller script, final-exp script)."""
    cfg = cfg or tw.Cfg()
    lazy = lazy_cfg(cfg)
    mscript, first_add = miller_script()
    builders = [seg_init(lazy, first_add)]
    for name in sorted(set(mscript[1:])):
        builders.append(seg_body(lazy, int(name[5]), int(name[6])))
    fscript = final_exp_script()
    extra = ["from_mont_1_0", "to_mont_0_1", "to_mont_1_1", "set_one_0", "mul_0_1", "copy_0_1", "copy_1_0",
             "mul_0_0", "add_0_1", "sub_0_1", "neg_0_0", "inv12_2_0", "copy_0_2"]      # blsgpu_fq12_op_batch / blsgpu_fq12_pow
    for name in sorted(set(fscript + extra)):
        builders.append(seg_by_name(lazy if name.startswith("mul_") else cfg, name))
    segs = {}
    for b in builders:
        segs[b.name] = schedule(b, temp_base=TEMP0, verbose=verbose)
    return segs, mscript, fscript


def build_multi(cfg=None, G=MP_G, verbose=False, prefix="mp"):
    """Miller-loop programs for G pairs per team sharing one accumulator:
    returns (segments by name, script, layout).  prefix names the segments (mp_*: three pairs, the
    throughput program; mp2_*: two pairs, for calls of a few thousand pairs, where 4096 teams of two
    fill the chip and a team finishes in 3/4 of the time)."""
    cfg = cfg or tw.Cfg()
    lazy = lazy_cfg(cfg)
    lay = MPLayout(G)
    bits = [(NX >> p) & 1 for p in range(62, -1, -1)]
    # step k multiplies the chord line in iff bit k is set (fields_t.py:1104); the chord step
    # itself runs inside that body, so the T chain is a tangent step ahead only
    pb = prefix + "_body"
    script = [prefix + "_init"] + [(pb + "_c0" if bit else (pb + "_02" if k + 1 == len(bits) else pb + "_00"))
                                   for k, bit in enumerate(bits)]
    assert bits[-1] == 0 and not any(a and b_ for a, b_ in zip(bits, bits[1:]))
    plan = [(seg_init(lazy, False, G, prefix + "_init", lay), lay.TEMP_HI, False),
            (seg_body_chord(lazy, 0, G, lay, pb), lay.TEMP_HI, False),
            (seg_body(lazy, 0, 0, G, pb, lay), lay.TEMP_LO, False),
            (seg_body(lazy, 0, 2, G, pb, lay), lay.TEMP_LO, False)]
    segs = {}
    for b, tb, on_demand in plan:
        segs[b.name] = schedule(b, temp_base=tb, verbose=verbose, lazy_lin=on_demand)
        segs[b.name].temp_base = tb
    # the Q window travels in registers between the segments that use it (MPLayout)
    segs[prefix + "_init"].rounds.append({"kind": "save", "K": lay.Q, "lanes": []})
    segs[pb + "_c0"].rounds.insert(0, {"kind": "restore", "K": lay.Q, "lanes": []})
    return segs, script, lay


def mp_team_slots(segs):
    return max(s.temp_base + s.ntemp for s in segs.values())
