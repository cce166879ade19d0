"""Lane-level model and table builder of the WIDE cofactor clearing of hash-to-G2 (round 5; csrc/blsgpu_h2cw.hip): the last
stage of hash_to_point_prehashed_Fq2 (ec.py:528-550: S0 + S1, then the Budroni-Pintore clearing [x^2 - x - 1] P + [x - 1] psi(P) +
psi^2(2 P), ec.py:536-550) for ONE message on ONE wavefront with a field product per lane -- the latency form for BLS.verify of
a single signature, where the wavefront VM's k_h2c_clear took 1.1 ms (a third of the whole verification).

The machine is the wide Miller loop's (vmgen/mlw_model.py: every Fq value in the LDS value file in its multiples 1, -1, 2, -2; a
step = per lane a sum of at most two products of sums of two slots, quad sum, scale, a multiple of q taken off inside the carry
pass).  The clearing is the op script of the register kernels (vmgen/gen_fexp.h2c_clear_script: one accumulator point, five
slot points; ADD / ADDNEG / ST / LD / DBL / PSI), compiled to steps:

  DBL   the complete homogeneous doubling for a = 0 in TWO levels -- the Miller loop's tangent step without its line:
        A = XY, B = Y^2, E = 12 xi Z^2, F = 3 E, YZ;  X3 = 2 A (B - F), Y3 = B^2 + (2 B - E) F, Z3 = 8 B YZ
        (= Renes-Costello-Batina algorithm 9, csrc/fp28.h pdbl: (0 : 1 : 0) and points of order two need no branch)
  ADD   the complete addition (RCB algorithm 7) in TWO levels: x3 = 3 X1 X2, t1 = Y1 Y2, bz = 12 xi Z1 Z2, t3 = X1 Y2 + X2 Y1,
        t4 = Y1 Z2 + Y2 Z1, y3 = 12 xi (X1 Z2 + X2 Z1);  X3 = t3 (t1 - bz) - t4 y3, Y3 = (t1 - bz)(t1 + bz) + y3 x3,
        Z3 = (t1 + bz) t4 + x3 t3 -- every output at most four products of sums of two slots (xi z = (zr - zi, zr + zi) is in the
        choice of the slots), so one product per lane
  PSI   (conj X psi_x, conj Y psi_y, conj Z): one level
  ST / LD  copies between the accumulator's and a slot point's 24 value slots (no arithmetic)

The Jacobian detour of the register kernels (runs of >= 4 doublings) is not taken: a homogeneous doubling is two steps here.
tests/test_h2cw_model.py runs the TABLES digit by digit (column bounds and stored-value range asserted) against the host's
integer hash-to-G2, which tests/test_hostmath_fixtures.py pins to the reference's vectors.
"""
from . import gen_fexp
from .gen_fp28 import Q, R, L, to_limbs, from_limbs
from .mlw_model import Names, Step, Machine, PAGE, VARIANT, T, variant_of

N = Names()
# page 0: the accumulator point, the constants, the write sink and zero
N.put(0, 0, "AX0", "AX1", "AY0", "AY1", "AZ0", "AZ1", "ONE", "PSIX0", "PSIX1", "PSIY0", "PSIY1")
N.put(0, 14, "TRASH", "ZERO")
# page 1: first level of the doubling (ten values) and slot point 4
N.put(1, 0, "A0", "A1", "B0", "B1", "E0", "E1", "F0", "F1", "YZ0", "YZ1")
# page 2: first level of the addition
N.put(2, 0, "X30", "X31", "T10", "T11", "BZ0", "BZ1", "T30", "T31", "T40", "T41", "Y30", "Y31")
POINT = ("X0", "X1", "Y0", "Y1", "Z0", "Z1")
SLOT_AT = {0: (3, 0), 1: (3, 6), 2: (4, 0), 3: (4, 6), 4: (1, 10)}          # (page, first quad) of slot point k
for k, (page, quad) in SLOT_AT.items():
    N.put(page, quad, *["S%d%s" % (k, c) for c in POINT])
PAGES = 5
ACC = ("AX0", "AX1", "AY0", "AY1", "AZ0", "AZ1")


def point_base(index):
    """dword address of the first of the 24 value slots of point `index` (0: the accumulator, 1 + k: slot point k)"""
    return N.slot("AX0") if index == 0 else N.slot("S%dX0" % (index - 1))


def _mul(x0, x1, y0, y1):
    """parts of (x0 + x1 u)(y0 + y1 u); an argument is an operand (list of terms)"""
    neg = lambda op: [(-c, s) for c, s in op]
    return [(x0, y0), (neg(x1), y1)], [(x0, y1), (x1, y0)]


def build_dbl():
    one = lambda n: T((1, n))
    a_re, a_im = _mul(one("AX0"), one("AX1"), one("AY0"), one("AY1"))
    b_re = [(T((1, "AY0"), (1, "AY1")), T((1, "AY0"), (-1, "AY1")))]
    b_im = [(T((2, "AY0")), T((1, "AY1")))]
    zsq = (T((1, "AZ0"), (1, "AZ1")), T((1, "AZ0"), (-1, "AZ1")))
    e_re = [zsq, (T((-2, "AZ0")), T((1, "AZ1")))]
    e_im = [zsq, (T((2, "AZ0")), T((1, "AZ1")))]
    y_re, y_im = _mul(one("AY0"), one("AY1"), one("AZ0"), one("AZ1"))
    l1 = Step("DBL1", [("A0", 1, a_re), ("A1", 1, a_im), ("B0", 1, b_re), ("B1", 1, b_im), ("E0", 12, e_re), ("E1", 12, e_im),
                       ("F0", 36, e_re), ("F1", 36, e_im), ("YZ0", 1, y_re), ("YZ1", 1, y_im)], N)
    bf0, bf1 = T((1, "B0"), (-1, "F0")), T((1, "B1"), (-1, "F1"))
    x_re = [(T((2, "A0")), bf0), (T((-2, "A1")), bf1)]
    x_im = [(T((2, "A0")), bf1), (T((2, "A1")), bf0)]
    be0, be1, nbe1 = T((2, "B0"), (-1, "E0")), T((2, "B1"), (-1, "E1")), T((-2, "B1"), (1, "E1"))
    yy_re = [(T((1, "B0"), (1, "B1")), T((1, "B0"), (-1, "B1"))), (be0, T((1, "F0"))), (nbe1, T((1, "F1")))]
    yy_im = [(T((2, "B0")), T((1, "B1"))), (be0, T((1, "F1"))), (be1, T((1, "F0")))]
    z_re, z_im = _mul(one("B0"), one("B1"), one("YZ0"), one("YZ1"))
    l2 = Step("DBL2", [("AX0", 1, x_re), ("AX1", 1, x_im), ("AY0", 1, yy_re), ("AY1", 1, yy_im), ("AZ0", 8, z_re), ("AZ1", 8, z_im)], N)
    return [l1, l2]


def build_add1(k, sign):
    """first level of acc + sign x (slot point k)"""
    s = "S%d" % k
    X1 = (T((1, "AX0")), T((1, "AX1")))
    Y1 = (T((1, "AY0")), T((1, "AY1")))
    Z1 = (T((1, "AZ0")), T((1, "AZ1")))
    X2 = (T((1, s + "X0")), T((1, s + "X1")))
    Y2 = (T((sign, s + "Y0")), T((sign, s + "Y1")))
    Z2 = (T((1, s + "Z0")), T((1, s + "Z1")))
    # xi z = (zr - zi, zr + zi) as operands of the OTHER factor: w (xi z) with z = (z0, z1)
    xim = lambda z: T((1, z + "0"), (-1, z + "1"))      # zr - zi
    xip = lambda z: T((1, z + "0"), (1, z + "1"))       # zr + zi
    nxip = lambda z: T((-1, z + "0"), (-1, z + "1"))
    x3_re, x3_im = _mul(*X1, *X2)
    t1_re, t1_im = _mul(*Y1, *Y2)
    # bz = 12 xi Z1 Z2: re = Z1r (Z2r - Z2i) - Z1i (Z2i + Z2r), im = Z1r (Z2r + Z2i) + Z1i (Z2r - Z2i)
    z2 = s + "Z"
    bz_re = [(Z1[0], xim(z2)), (Z1[1], nxip(z2))]
    bz_im = [(Z1[0], xip(z2)), (Z1[1], xim(z2))]
    a, b = _mul(*X1, *Y2)
    c, d = _mul(*X2, *Y1)
    t3_re, t3_im = a + c, b + d
    a, b = _mul(*Y1, *Z2)
    c, d = _mul(*Y2, *Z1)
    t4_re, t4_im = a + c, b + d
    # y3 = 12 xi (X1 Z2 + X2 Z1)
    y3_re = [(X1[0], xim(z2)), (X1[1], nxip(z2)), (X2[0], xim("AZ")), (X2[1], nxip("AZ"))]
    y3_im = [(X1[0], xip(z2)), (X1[1], xim(z2)), (X2[0], xip("AZ")), (X2[1], xim("AZ"))]
    return Step("ADD1_%d%s" % (k, "n" if sign < 0 else "p"),
                [("X30", 3, x3_re), ("X31", 3, x3_im), ("T10", 1, t1_re), ("T11", 1, t1_im), ("BZ0", 12, bz_re), ("BZ1", 12, bz_im),
                 ("T30", 1, t3_re), ("T31", 1, t3_im), ("T40", 1, t4_re), ("T41", 1, t4_im), ("Y30", 12, y3_re), ("Y31", 12, y3_im)], N)


def build_add2():
    m0, m1 = T((1, "T10"), (-1, "BZ0")), T((1, "T11"), (-1, "BZ1"))        # t1 - bz
    nm1 = T((-1, "T11"), (1, "BZ1"))
    p0, p1 = T((1, "T10"), (1, "BZ0")), T((1, "T11"), (1, "BZ1"))          # t1 + bz
    np1 = T((-1, "T11"), (-1, "BZ1"))
    o = lambda n, c=1: T((c, n))
    x_re = [(o("T30"), m0), (o("T31", -1), m1), (o("T40", -1), o("Y30")), (o("T41"), o("Y31"))]
    x_im = [(o("T30"), m1), (o("T31"), m0), (o("T40", -1), o("Y31")), (o("T41", -1), o("Y30"))]
    y_re = [(m0, p0), (nm1, p1), (o("Y30"), o("X30")), (o("Y31", -1), o("X31"))]
    y_im = [(m0, p1), (m1, p0), (o("Y30"), o("X31")), (o("Y31"), o("X30"))]
    z_re = [(p0, o("T40")), (np1, o("T41")), (o("X30"), o("T30")), (o("X31", -1), o("T31"))]
    z_im = [(p0, o("T41")), (p1, o("T40")), (o("X30"), o("T31")), (o("X31"), o("T30"))]
    return Step("ADD2", [("AX0", 1, x_re), ("AX1", 1, x_im), ("AY0", 1, y_re), ("AY1", 1, y_im), ("AZ0", 1, z_re), ("AZ1", 1, z_im)], N)


def build_psi():
    """(conj X psix, conj Y psiy, conj Z):  conj(x) p = (xr pr + xi pi) + (xr pi - xi pr) u"""
    def cm(x, p):
        return ([(T((1, x + "0")), T((1, p + "0"))), (T((1, x + "1")), T((1, p + "1")))],
                [(T((1, x + "0")), T((1, p + "1"))), (T((-1, x + "1")), T((1, p + "0")))])
    xr, xi_ = cm("AX", "PSIX")
    yr, yi = cm("AY", "PSIY")
    one = T((1, "ONE"))
    return Step("PSI", [("AX0", 1, xr), ("AX1", 1, xi_), ("AY0", 1, yr), ("AY1", 1, yi),
                        ("AZ0", 1, [(T((1, "AZ0")), one)]), ("AZ1", 1, [(T((-1, "AZ1")), one)])], N)


SCRIPT = gen_fexp.h2c_clear_script()
_adds = sorted({(sl, 1 if op == 1 else -1) for op, sl in SCRIPT if op in (1, 2)})
KINDS = build_dbl() + [build_add2(), build_psi()] + [build_add1(k, s) for k, s in _adds]
KIND = {s.name: i for i, s in enumerate(KINDS)}
COPY = 0x8000                                      # program word of a copy: COPY | source point << 4 | destination point
END = 0xFFFF


def program():
    """the clearing as one word per step (kind | routine << 8, as mlw_model.programs) or copy; ends with END"""
    out = []
    st = lambda name: KIND[name] | (variant_of(KINDS[KIND[name]]) << 8)
    for op, sl in SCRIPT:
        if op == 0:
            break
        if op in (1, 2):
            out += [st("ADD1_%d%s" % (sl, "p" if op == 1 else "n")), st("ADD2")]
        elif op == 3:
            out.append(COPY | (0 << 4) | (1 + sl))
        elif op == 4:
            out.append(COPY | ((1 + sl) << 4) | 0)
        elif op in (5, 8):
            out += [st("DBL1"), st("DBL2")]
        elif op == 6:
            out.append(st("PSI"))
        elif op in (7, 9):
            pass                                   # (the register kernels' Jacobian detour: a doubling is two steps here either way)
        else:
            raise AssertionError(op)
    return out + [END]


def clear(S0, S1, psix, psiy):
    """S0, S1: homogeneous points ((x0, x1), (y0, y1), (z0, z1)) (residues); psix, psiy: the endomorphism's constants.
    Runs the tables; returns the homogeneous result and the largest |stored value| / q."""
    m = Machine(N, KINDS)
    for name, v in (("PSIX0", psix[0]), ("PSIX1", psix[1]), ("PSIY0", psiy[0]), ("PSIY1", psiy[1])):
        m.store_value(name, v)
    for base, P in (("A", S0), ("S0", S1)):
        for c, v in zip(POINT, (P[0][0], P[0][1], P[1][0], P[1][1], P[2][0], P[2][1])):
            m.store_value(base + c, v % Q)
    for w in program():
        if w == END:
            break
        if w & COPY:
            src, dst = point_base((w >> 4) & 0xF), point_base(w & 0xF)
            for i in range(24):
                m.vf[dst + i] = list(m.rd(src + i))
        else:
            m.step(w & 0x3f)
    return tuple((m.value("A" + c + "0"), m.value("A" + c + "1")) for c in "XYZ"), m.max_abs


def horner(points, cbits):
    """the window Horner of a G2 sum on the same tables (csrc/blsgpu_h2cw.hip k_msm_horner_wide2: the accumulator and slot point 0,
    kinds DBL1 / DBL2 / ADD1_0p / ADD2): points = homogeneous ((x0, x1), (y0, y1), (z0, z1)), index 0 the lowest; returns
    sum_i 2^(cbits i) points[i] (homogeneous) and the largest |stored value| / q"""
    m = Machine(N, KINDS)

    def put(base, P):
        for c, v in zip(POINT, (P[0][0], P[0][1], P[1][0], P[1][1], P[2][0], P[2][1])):
            m.store_value(base + c, v % Q)
    put("A", points[-1])
    for P in reversed(points[:-1]):
        for _ in range(cbits):
            m.step(KIND["DBL1"])
            m.step(KIND["DBL2"])
        put("S0", P)
        m.step(KIND["ADD1_0p"])
        m.step(KIND["ADD2"])
    return tuple((m.value("A" + c + "0"), m.value("A" + c + "1")) for c in "XYZ"), m.max_abs

