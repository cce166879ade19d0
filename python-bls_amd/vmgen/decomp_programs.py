"""VM programs for batched point decompression:
PublicKey.from_bytes (keys.py:28-40) and Signature.from_bytes (signature.py:21-38)
from the masked x coordinate onwards: y_for_x (ec.py:255-269) with the field square
roots of fields.py:199-205 / 463-482 and the reference's choice between y and -y.

Branch-free restatement (the reference raises ValueError on bad input; here a
validity flag comes out beside the point):
  G1: u = x^3 + 4, z = u^((q-3)/4), r = z u (root candidate), chi = r z (Legendre
      symbol).  valid <=> chi = 1, as the exact indicator (chi^2 + chi)/2 (u = 0 gives
      chi = 0: the reference rejects y = 0).  Result: the larger of r, -r iff the
      sign bit is set (keys.py:33-38 sorts the two roots).
  G2: u = x^3 + 4(1+i) = a0 + a1 i; alpha = a0^2 + a1^2; complex method: r = sqrt(alpha),
      delta = (a0 +- r)/2, x0 = sqrt(delta), x1 = a1 / (2 x0).  Both delta candidates are
      exponentiated side by side and the reference's pick (the first one unless its
      Legendre symbol is -1) is an arithmetic select.  The reference's separate branch
      for a1 = 0 returns an element of Fq (fields.py:466-467), which y_for_x cannot use:
      it ends in ValueError (a0 no square of Fq, or u = 0) or in the AffinePoint constructor's
      Exception('x,y should be field elements') -- Signature.from_bytes rejects EVERY encoding
      whose u has zero imaginary part (tests/golden/g2_real_u.json, reference-generated):
          valid = ind(chi(alpha)) * nz(a1).
      Choice of the root (signature.py:31-35): the other root -y is taken iff
      (big and (-y).c1 > q/2) or (not big and (-y).c1 < q/2), i.e. with g = [y.c1 > q/2]
      and z = [y.c1 = 0] = 1 - nz(a1):  flip = big xor (g + z).
Outputs are canonical (non-Montgomery) integers: x, y and the flag (1 or 0).
"""
from . import tower as tw
from .core import Builder, schedule
from .h2c_programs import EXP_E, HC_END, HC_INV2
from .programs import C_ONE, C_R2, C_RAW1, C_ZERO


class D1Layout:
    """G1: NE points per team."""

    def __init__(self, NE):
        self.NE = NE
        o = HC_END
        self.X = o; o += NE                # raw x (masked), later Montgomery
        self.BIG = o; o += NE              # sign bit of the encoding as Montgomery 0 / 1
        self.U = o; o += NE
        self.ACC = o; o += NE
        self.BASE = o; o += NE
        self.OUT = o; o += 3 * NE          # x, y, valid (canonical)
        self.TEMP0 = o


class D2Layout:
    """G2: NE points per team."""

    def __init__(self, NE):
        self.NE = NE
        o = HC_END
        self.X = o; o += 2 * NE
        self.BIG = o; o += NE
        self.U = o; o += 2 * NE
        self.AL = o; o += NE               # alpha
        self.NZ = o; o += NE               # nz(a1)
        self.VA = o; o += NE               # ind(chi(alpha))
        self.ACC = o; o += 2 * NE
        self.BASE = o; o += 2 * NE
        self.OUT = o; o += 5 * NE          # x.c0, x.c1, y.c0, y.c1, valid (canonical)
        self.TEMP0 = o


def _ind(chi, inv2):
    """1 if chi = 1, 0 if chi in {0, -1}."""
    return ((chi * chi + chi) * inv2).mat()


def _exp_script(sq, mu):
    sc = []
    for ch in bin(EXP_E)[3:]:
        sc.append(sq)
        if ch == "1":
            sc.append(mu)
    return sc


def _pow_segs(prefix, L, cnt, done):
    b = Builder(prefix + "_sqr")
    for k in range(cnt):
        a = b.inp(L.ACC + k)
        b.out(a * a, L.ACC + k)
    done(b)
    b = Builder(prefix + "_mul")
    for k in range(cnt):
        b.out(b.inp(L.ACC + k) * b.inp(L.BASE + k), L.ACC + k)
    done(b)


def build_d1(NE, verbose=False):
    L = D1Layout(NE)
    segs = {}

    def done(b):
        segs[b.name] = schedule(b, temp_base=L.TEMP0, verbose=verbose)
    b = Builder("d1_a")
    r2, one = b.inp(C_R2), b.inp(C_ONE)
    for e in range(NE):
        x = (b.inp(L.X + e) * r2).mat()
        u = ((x * x).mat() * x + one * 4).mat()
        b.out(x, L.X + e), b.out(u, L.U + e), b.out(u, L.ACC + e), b.out(u, L.BASE + e)
    done(b)
    _pow_segs("d1", L, NE, done)
    b = Builder("d1_c")
    inv2, raw1, zero = b.inp(HC_INV2), b.inp(C_RAW1), b.inp(C_ZERO)
    for e in range(NE):
        z, u, big = b.inp(L.ACC + e), b.inp(L.U + e), b.inp(L.BIG + e)
        r = (z * u).mat()
        valid = _ind(r * z, inv2)
        g = r.sgn()
        flip = (big + g - (big * g) * 2).mat()
        y = (r - (flip * r) * 2).mat()
        for k, v in enumerate((b.inp(L.X + e), y, valid)):
            o = v * raw1
            b.out(o, L.OUT + 3 * e + k)
    done(b)
    return segs, L, ["d1_a"] + _exp_script("d1_sqr", "d1_mul") + ["d1_c"]


def build_d2(NE, cfg=None, verbose=False):
    cfg = cfg or tw.Cfg()
    L = D2Layout(NE)
    segs = {}

    def done(b):
        segs[b.name] = schedule(b, temp_base=L.TEMP0, verbose=verbose)
    b = Builder("d2_a")
    r2, one = b.inp(C_R2), b.inp(C_ONE)
    for e in range(NE):
        x = ((b.inp(L.X + 2 * e) * r2).mat(), (b.inp(L.X + 2 * e + 1) * r2).mat())
        u = cfg.mul2(cfg.sqr2(x), x)
        u = tw.f2_mat((u[0] + one * 4, u[1] + one * 4))
        al = (u[0] * u[0] + u[1] * u[1]).mat()
        nz = (u[1] * u[1].inv()).mat()
        b.out(x[0], L.X + 2 * e), b.out(x[1], L.X + 2 * e + 1)
        b.out(u[0], L.U + 2 * e), b.out(u[1], L.U + 2 * e + 1)
        b.out(al, L.AL + e), b.out(nz, L.NZ + e)
        b.out(al, L.ACC + e), b.out(al, L.BASE + e)
    done(b)
    _pow_segs("d2p", L, NE, done)            # alpha^E on the first NE accumulators
    b = Builder("d2_b")
    inv2 = b.inp(HC_INV2)
    for e in range(NE):
        z, al, a0 = b.inp(L.ACC + e), b.inp(L.AL + e), b.inp(L.U + 2 * e)
        r = (z * al).mat()
        b.out(_ind(r * z, inv2), L.VA + e)
        for j, d in enumerate((((a0 + r) * inv2).mat(), ((a0 - r) * inv2).mat())):
            b.out(d, L.ACC + 2 * e + j), b.out(d, L.BASE + 2 * e + j)
    done(b)
    _pow_segs("d2q", L, 2 * NE, done)         # delta+-^E
    b = Builder("d2_c")
    one, inv2, raw1 = b.inp(C_ONE), b.inp(HC_INV2), b.inp(C_RAW1)
    for e in range(NE):
        sq, chi = [], []
        for j in range(2):
            z, d = b.inp(L.ACC + 2 * e + j), b.inp(L.BASE + 2 * e + j)
            s = (z * d).mat()
            sq.append(s)
            chi.append((s * z).mat())
        nz, va, big, a1 = b.inp(L.NZ + e), b.inp(L.VA + e), b.inp(L.BIG + e), b.inp(L.U + 2 * e + 1)
        # the reference keeps delta+ unless its symbol is -1: c = 1 - ind(-chi+)
        c = (one - ((chi[0] * chi[0] - chi[0]) * inv2)).mat()
        x0 = (sq[1] + c * (sq[0] - sq[1])).mat()
        x1 = (a1 * (x0 * 2).inv()).mat()
        valid = (va * nz).mat()
        t = (x1.sgn() + one - nz).mat()                           # g + z
        flip = (big + t - (big * t) * 2).mat()
        y0 = (x0 - (flip * x0) * 2).mat()
        y1 = (x1 - (flip * x1) * 2).mat()
        for k, v in enumerate((b.inp(L.X + 2 * e), b.inp(L.X + 2 * e + 1), y0, y1, valid)):
            b.out(v * raw1, L.OUT + 5 * e + k)
    done(b)
    script = ["d2_a"] + _exp_script("d2p_sqr", "d2p_mul") + ["d2_b"] + _exp_script("d2q_sqr", "d2q_mul") + ["d2_c"]
    return segs, L, script
