"""The reference-faithful Miller loop as a VM program ("slow path").

The fast Miller programs (programs.py) run the twist point in projective coordinates and
scale every line by factors the final exponentiation removes.  That is the reference's
result only while no step degenerates.  The reference itself (fields_t.py:1091-1111) is
total: it is a fixed sequence of affine formulas in which 0^-1 := 0 (fq_invert,
fields_t.py:47-55), with three data-dependent branches in the chord step
(fq2_add_line_eval :1062-1065, fq2_add_points :673-686).  For a low-order, off-curve or
zero-coordinate Q those branches and the 0^-1 rule decide the result (an
order-13 Q gives a non-zero value in the reference, zero from the projective program).

This module restates that total function branch-free, step by step, on AFFINE coordinates:

  tangent step  (fq2_double_line_eval :1035-1049, fq2_double_point :641-646)
      lam = 3 rx^2 * inv(2 ry)                       inv(0) = 0
      line = py - px lam / w - (ry - lam rx) / w^3   exactly the reference's Fq12 value:
             1/w = xi^-1 v^2 w, 1/w^3 = xi^-1 v w  (w^6 = xi), xi^-1 = (1 - u)/2
      R <- (lam^2 - 2 rx, lam (rx - xr) - ry)
  chord step    (fq2_add_line_eval :1052-1078, fq2_add_points :673-686)
      D = inv(qx - rx);  n1 = [rx != qx], n3 = [ry != qy], m1 = [rx != -qx], m2 = [ry != -qy]
      as 0/1 field values  nz(z) = N(z) * inv(N(z)),  N(z) = z0^2 + z1^2  (zero only for z = 0,
      since q = 3 mod 4)
      vert = (1-m1)(1-m2)        the branch  P.x - R12.x  of :1062-1065.  NOTE its test: the code
                                 negates BOTH coordinates of untwist(Q) (:1060-1062), so it fires
                                 for R = (-qx, -qy) -- on the curve only when qx = 0 -- and not for
                                 R = -Q = (qx, -qy), which falls through to the formulas below
      mu = (qy - ry) D, nu = -(qy rx - ry qx) D         both 0 when rx = qx  (0^-1 = 0)
      line = vert (px - rx / w^2)  +  (1 - vert) (py - mu px / w - nu / w^3)
      same = (1-n1)(1-n3)        R == Q: the doubling inside fq2_add_points       (:678-679)
      R <- n1 * chord(R, Q) + same * double(R)          [x1 == x2 otherwise -> (0,0), :680-681]
      R <- qinf * R_old + (1 - qinf) * R                [a flagged Q leaves R alone, :676-677]
  R's own flag never matters: every chord step follows a doubling, which returns
  inf = False (:646), and P's flag is never read.

All selections are products with exact 0/1 values, so every lane runs the same rounds for
every input.  The value left in the accumulator IS fq_miller_loop's (not a multiple of it),
which is what blsgpu_miller_loop_batch returns; pairs the fast kernels flag as degenerate
are recomputed with this program (csrc/blsgpu_kernels.hip k_miller_slow).
"""
from . import tower as tw
from .core import Builder, schedule
from . import programs as P
from .programs import C_HALF, C_ONE, C_R2, C_ZERO, F, NX, PX, PY, QX0, QY0, TX, TY, TZ, in12, in2, out12

RX, RY = TX, TY                            # the affine twist point R
QINF = TZ                                  # Q's infinity flag as a field value 0 / 1 (set by the kernel)
SLOW_TEMP0 = P.TEMP0


def _xi_inv(half, a):
    """a * xi^-1 = a (1 - u) / 2"""
    return (((a[0] + a[1]).mat() * half).mat(), ((a[1] - a[0]).mat() * half).mat())


def _norm_inv(z):
    """(N(z) materialised, its inverse): the one field inversion behind inv(z) and nz(z)"""
    n = (z[0] * z[0] + z[1] * z[1]).mat()
    return n, n.inv()


def _f2_inv_from(z, ni):
    return ((z[0] * ni).mat(), (-(z[1] * ni)).mat())


def _double_point(cfg, rx, ry, inv2ry):
    lam = cfg.mul2(tw.f2_scale(cfg.sqr2(rx), 3), inv2ry)
    xr = tw.f2_mat(tw.f2_sub(cfg.sqr2(lam), tw.f2_scale(rx, 2)))
    yr = tw.f2_mat(tw.f2_sub(cfg.mul2(lam, tw.f2_sub(rx, xr)), ry))
    return lam, (xr, yr)


def _line12(b, c00, c02, c11, c12):
    """the Fq12 value  c00 + c02 v^2 + c11 v w + c12 v^2 w  (c00 in Fq)"""
    z2 = (b.zero(), b.zero())
    return (((c00, b.zero()), z2, c02), (z2, c11, c12))


def seg_slow_init():
    """raw inputs -> Montgomery form; R = Q; f = 1"""
    b = Builder("slow_init")
    r2, one, zero = b.inp(C_R2), b.inp(C_ONE), b.inp(C_ZERO)
    b.out(one, F)
    for i in range(1, 12):
        b.out(zero, F + i)
    for s in (PX, PY):
        b.out((b.inp(s) * r2).mat(), s)
    for i in range(4):
        v = (b.inp(QX0 + i) * r2).mat()
        b.out(v, QX0 + i)
        b.out(v, RX + i)
    return b


def _tangent(b, cfg, rx, ry, px, py, half):
    """fq2_double_line_eval (fields_t.py:1035-1049) and fq2_double_point (:641-646) at R"""
    two_ry = tw.f2_scale(ry, 2)
    _, ni = _norm_inv(two_ry)
    lam, Rn = _double_point(cfg, rx, ry, _f2_inv_from(two_ry, ni))
    l0 = tw.f2_sub(cfg.mul2(lam, rx), ry)              # -(ry - lam rx)
    l1 = tw.f2_neg(tw.f2_mul_fq(lam, px))
    return _line12(b, py, (b.zero(), b.zero()), _xi_inv(half, l0), _xi_inv(half, l1)), Rn


def seg_slow_dbl(cfg):
    b = Builder("slow_dbl")
    half = b.inp(C_HALF)
    px, py = b.inp(PX), b.inp(PY)
    rx, ry = in2(b, RX), in2(b, RY)
    f = in12(b, F)
    line, Rn = _tangent(b, cfg, rx, ry, px, py, half)
    f = tw.f12_mul(cfg, tuple(tuple(tw.f2_mat(c) for c in h) for h in tw.f12_sqr(cfg, f)), line)
    out12(b, f, F, zero=b.inp(C_ZERO))
    zero = b.inp(C_ZERO)
    P._out2z(b, Rn[0], RX, zero)
    P._out2z(b, Rn[1], RY, zero)
    return b


def _chord(b, cfg, rx, ry, qx, qy, px, py, half, one):
    """fq2_add_line_eval (fields_t.py:1052-1078) -> (line, what the point update shares with it)"""
    d = tw.f2_sub(qx, rx)
    u = tw.f2_sub(ry, qy)
    two_ry = tw.f2_scale(ry, 2)

    def is_zero(z):                                    # [z == 0] as a field value
        n, ni = _norm_inv(z)
        return (one - (n * ni).mat()).mat()
    nd, nid = _norm_inv(d)
    _, niy = _norm_inv(two_ry)
    D = _f2_inv_from(d, nid)
    n1 = (nd * nid).mat()
    same = ((one - n1).mat() * is_zero(u)).mat()
    vert = (is_zero(tw.f2_add(rx, qx)) * is_zero(tw.f2_add(ry, qy))).mat()
    nvert = (one - vert).mat()
    mu = cfg.mul2(tw.f2_neg(u), D)
    nu = tw.f2_neg(cfg.mul2(tw.f2_sub(cfg.mul2(qy, rx), cfg.mul2(ry, qx)), D))
    c00 = py + vert * (px - py)
    c02 = _xi_inv(half, tw.f2_neg(tw.f2_mul_fq(rx, vert)))
    c11 = _xi_inv(half, tw.f2_neg(tw.f2_mul_fq(nu, nvert)))
    c12 = _xi_inv(half, tw.f2_neg(tw.f2_mul_fq(tw.f2_mat(tw.f2_mul_fq(mu, px)), nvert)))
    return _line12(b, c00.mat(), c02, c11, c12), (mu, n1, same, two_ry, niy)


def seg_slow_add(cfg):
    b = Builder("slow_add")
    half, one = b.inp(C_HALF), b.inp(C_ONE)
    px, py = b.inp(PX), b.inp(PY)
    rx, ry = in2(b, RX), in2(b, RY)
    qx, qy = in2(b, QX0), in2(b, QY0)
    qinf = b.inp(QINF)
    f = in12(b, F)
    line, (mu, n1, same, two_ry, niy) = _chord(b, cfg, rx, ry, qx, qy, px, py, half, one)
    f = tw.f12_mul(cfg, f, line)
    out12(b, f, F, zero=b.inp(C_ZERO))
    # R + Q
    xc = tw.f2_mat(tw.f2_sub(tw.f2_sub(cfg.sqr2(mu), rx), qx))
    yc = tw.f2_mat(tw.f2_sub(cfg.mul2(mu, tw.f2_sub(rx, xc)), ry))
    _, (xd, yd) = _double_point(cfg, rx, ry, _f2_inv_from(two_ry, niy))
    zero = b.inp(C_ZERO)
    for old, c, dd, slot in ((rx, xc, xd, RX), (ry, yc, yd, RY)):
        new = tw.f2_mat(tw.f2_add(tw.f2_mul_fq(c, n1), tw.f2_mul_fq(dd, same)))
        fin = tw.f2_add(new, tw.f2_mul_fq(tw.f2_sub(old, new), qinf))
        P._out2z(b, fin, slot, zero)
    return b


def seg_line(cfg, add):
    """One line evaluation on raw inputs: the reference's fq2_double_line_eval(R, P) /
    fq2_add_line_eval(R, Q, P) as exported by its native module (fields_t.py:1218-1263):
    accumulator <- the Fq12 line value (Montgomery form)."""
    b = Builder("line_add" if add else "line_dbl")
    r2, half, one = b.inp(C_R2), b.inp(C_HALF), b.inp(C_ONE)

    def mont(s):
        return (b.inp(s) * r2).mat()
    px, py = mont(PX), mont(PY)
    rx, ry = (mont(RX), mont(RX + 1)), (mont(RY), mont(RY + 1))
    if add:
        qx, qy = (mont(QX0), mont(QX0 + 1)), (mont(QY0), mont(QY0 + 1))
        line, _ = _chord(b, cfg, rx, ry, qx, qy, px, py, half, one)
    else:
        line, _ = _tangent(b, cfg, rx, ry, px, py, half)
    out12(b, line, F, zero=b.inp(C_ZERO))
    return b


def slow_script():
    bits = [(NX >> p) & 1 for p in range(62, -1, -1)]
    s = ["slow_init"]
    for bit in bits:
        s.append("slow_dbl")
        if bit:
            s.append("slow_add")
    return s


def build(verbose=False):
    """-> (segments by name, script)"""
    cfg = tw.Cfg()
    segs = {}
    for b in (seg_slow_init(), seg_slow_dbl(cfg), seg_slow_add(cfg), seg_line(cfg, False), seg_line(cfg, True)):
        segs[b.name] = schedule(b, temp_base=SLOW_TEMP0, verbose=verbose)
    return segs, slow_script()


def team_slots(segs):
    return SLOW_TEMP0 + max(s.ntemp for s in segs.values())
