"""Lane-level model and table builder of the WIDE tail of the sorted-bucket G1 sum (round 5; csrc/blsgpu_g1w.hip
k_msm_horner_wide): result = sum_i 2^(c i) P_i over a short list of projective G1 points by Horner from the top -- the window
sums W_w = sum_b 2^b S_(w,b) (c = 1, one wavefront per window) and the sum over the windows sum_w 2^(13 w) W_w (c = 13, one
wavefront) of BLS.aggregate_pub_keys(secure) at scale (bls.py:203-223; the reference's double-and-add summed over the points,
fields_t.py:705-740).  247 doublings and 19 additions of ONE point are a dependent chain: nothing to batch.  The wavefront VM ran it
on one team with every linear combination a round of its own (k_msm_pip_horner<1>: 1.31 ms, a fifth of the whole sum).

The machine is the wide Miller loop's (vmgen/mlw_model.py: every Fq value in the LDS value file in its multiples 1, -1, 2, -2; a
step = per lane ONE product of sums of two slots, quad sum, scale, a multiple of q taken off inside the carry pass).  The
formulas are the complete ones of Renes-Costello-Batina for a = 0, b = 4 (csrc/fp28.h pdbl / padd: infinity (0 : 1 : 0),
a doubling inside an addition and P + (-P) need no branch), each in TWO levels:

  DBL   A = XY, B = Y^2, E = 12 Z^2, F = 3 E, YZ;   X3 = 2 A (B - F), Y3 = B^2 + (2 B - E) F, Z3 = 8 B YZ
  ADD   x3 = 3 X1 X2, t1 = Y1 Y2, bz = 12 Z1 Z2, t3 = X1 Y2 + X2 Y1, t4 = Y1 Z2 + Y2 Z1, y3 = 12 (X1 Z2 + X2 Z1);
        X3 = t3 (t1 - bz) - t4 y3, Y3 = (t1 - bz)(t1 + bz) + y3 x3, Z3 = (t1 + bz) t4 + x3 t3

(vmgen/h2cw_model.py's steps with the field Fq instead of Fq2).  tests/test_g1w_model.py runs the TABLES digit by digit (column
bounds and stored-value range asserted) against the host's integer curve arithmetic.
"""
from .gen_fp28 import Q
from .mlw_model import Names, Step, Machine, T

N = Names()
# page 0: the accumulator point, the addend, the constant one, the write sink and zero
N.put(0, 0, "AX", "AY", "AZ", "SX", "SY", "SZ", "ONE")
N.put(0, 14, "TRASH", "ZERO")
# page 1: first levels of the doubling and of the addition
N.put(1, 0, "A", "B", "E", "F", "YZ", "X3", "T1", "BZ", "T3", "T4", "Y3")
PAGES = 2
o = lambda n, c=1: T((c, n))


def build_dbl():
    l1 = Step("DBL1", [("A", 1, [(o("AX"), o("AY"))]), ("B", 1, [(o("AY"), o("AY"))]), ("E", 12, [(o("AZ"), o("AZ"))]),
                       ("F", 36, [(o("AZ"), o("AZ"))]), ("YZ", 1, [(o("AY"), o("AZ"))])], N)
    l2 = Step("DBL2", [("AX", 1, [(o("A", 2), T((1, "B"), (-1, "F")))]),
                       ("AY", 1, [(o("B"), o("B")), (T((2, "B"), (-1, "E")), o("F"))]),
                       ("AZ", 8, [(o("B"), o("YZ"))])], N)
    return [l1, l2]


def build_add():
    l1 = Step("ADD1", [("X3", 3, [(o("AX"), o("SX"))]), ("T1", 1, [(o("AY"), o("SY"))]), ("BZ", 12, [(o("AZ"), o("SZ"))]),
                       ("T3", 1, [(o("AX"), o("SY")), (o("SX"), o("AY"))]), ("T4", 1, [(o("AY"), o("SZ")), (o("SY"), o("AZ"))]),
                       ("Y3", 12, [(o("AX"), o("SZ")), (o("SX"), o("AZ"))])], N)
    m, p = T((1, "T1"), (-1, "BZ")), T((1, "T1"), (1, "BZ"))
    l2 = Step("ADD2", [("AX", 1, [(o("T3"), m), (o("T4", -1), o("Y3"))]),
                       ("AY", 1, [(m, p), (o("Y3"), o("X3"))]),
                       ("AZ", 1, [(p, o("T4")), (o("X3"), o("T3"))])], N)
    return [l1, l2]


KINDS = build_dbl() + build_add()
KIND = {s.name: i for i, s in enumerate(KINDS)}


def horner(points, cbits):
    """points: homogeneous (X, Y, Z) residues, index 0 the lowest; returns sum_i 2^(cbits i) points[i] (homogeneous) and the
    largest |stored value| / q, by the tables"""
    m = Machine(N, KINDS)
    for c, v in zip("XYZ", points[-1]):
        m.store_value("A" + c, v % Q)
    for P in reversed(points[:-1]):
        for _ in range(cbits):
            m.step(KIND["DBL1"])
            m.step(KIND["DBL2"])
        for c, v in zip("XYZ", P):
            m.store_value("S" + c, v % Q)
        m.step(KIND["ADD1"])
        m.step(KIND["ADD2"])
    return tuple(m.value("A" + c) for c in "XYZ"), m.max_abs
