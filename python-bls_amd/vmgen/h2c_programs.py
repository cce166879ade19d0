"""VM programs for hash-to-G2 after the SHA-256 step: the reference's
hash_to_point_prehashed_Fq2 (ec.py:528-550) = two Shallue-van de Woestijne
encodings (sw_encode, ec.py:449-507), their sum, and the Budroni-Pintore
cofactor clearing with psi.  SURVEY.md section 8(f) rank 1.

The reference's control flow (try/except around square roots, index selection,
sign flip) is restated branch-free:
  * "x has a y" <=> the norm N(x^3 + b') is a square in Fq; with
    z = n^((q-3)/4) one gets sqrt candidate r = z n and Legendre symbol chi = r z
    from ONE exponentiation, and c = (1 + chi)/2 in {0, 1} drives arithmetic
    selects  v = B + c (A - B);
  * Fq2 square root by the complex method of fields.py:463-482: alpha = sqrt(N),
    delta = (a0 +/- alpha)/2 -- both candidates are exponentiated side by side
    and the one the reference would pick is selected by its Legendre symbol;
  * sign: SGN rounds evaluate "imaginary part > (q-1)/2" (ec.py:94-100).
A candidate x whose u = x^3 + b' has ZERO imaginary part is skipped by the reference although u is a
square of Fq2: Fq2.modsqrt returns an Fq there (fields.py:466-467), y_for_x fails on it and sw_encode's bare
`except` counts the candidate as invalid (ec.py:489-498).  Restated: the norm handed to the symbol test is
n' = N(u), set to 0 when a1 = 0 -- a zero test the stage-0 kernel does itself between h1_a and the
exponentiation (real_u_step below is its statement for the interpreters) -- and the selector is the exact
indicator of chi(n') = 1, so a1 = 0 (which includes u = 0) gives 0.  (If the LAST candidate x3 is of that kind, or x1 is and u2, u3 are non-squares, the reference's
sw_encode raises -- ec.py:503; there is nothing to reproduce then, and hashed inputs never get there.)
Reference-generated vectors: tests/golden/g2_real_u.json.
The reference's early exit for t = 0 (ec.py:450-452) is kept as a flag
nz(t) = t (1/t) in {0, 1} (the VM's inversion maps 0 to 0): t = 0 gives the point
at infinity, Z = 0 in the projective triple handed to kernel H2, whose additions
are complete.  Its other exit, w = t^2 + b' + 1 = 0 (ec.py:466-473), cannot
happen over Fq2: -(5 + 4i) is not a square there (tests/test_vm_h2c.py).  Not covered: an encoding whose
LAST candidate has x^3 + b' with zero imaginary part (the reference raises there).

Kernel H1 (h1_*):  NE encodings per team -> affine points S_e.
Kernel H2 (h2_*):  NM messages per team: P = S_0 + S_1, cofactor clearing, affine bytes.
"""
from . import tower as tw
from .core import Builder, schedule
from .msm_programs import FA, padd, pdbl
from .programs import C_GAM, C_ONE, C_R2, C_RAW1, C_ZERO, NCONST, NX, _fq2_pow, const_table
from .sim import Q, to_m

SQRT_N3 = 1586958781458431025242759403266842894121773480562120986020912974854563298150952611241517463240701
SQRT_N3M1O2 = 793479390729215512621379701633421447060886740281060493010456487427281649075476305620758731620350
assert (SQRT_N3 * SQRT_N3 + 3) % Q == 0 and (2 * SQRT_N3M1O2 + 1 - SQRT_N3) % Q == 0

# extra constants.  The hashing / decompression programs never touch the Frobenius constants of
# the final exponentiation: in their scratchpads the extra constants follow C_K1 directly (the
# kernel copies entries [0, C_GAM) and [NCONST, NCONST + 9) of its constant table), 30 slots less
# per team.
HC_S3, HC_H, HC_SINV, HC_INV2 = C_GAM, C_GAM + 1, C_GAM + 2, C_GAM + 3
HC_PSIX, HC_PSIY = C_GAM + 4, C_GAM + 6             # Fq2 each
HC_R3 = C_GAM + 8                                   # content R^3: raw x -> x R^2 (the 2^384 place of a wide input)
HC_END = C_GAM + 9
HC_TBL0 = NCONST                                    # first extra constant in the kernel's constant table
EXP_E = (Q - 3) // 4                                  # n^E: sqrt candidate n^E n, symbol n^E (n^E n)


def h2c_const_table():
    g2, g3 = _fq2_pow((1, 1), 2 * (Q - 1) // 6), _fq2_pow((1, 1), 3 * (Q - 1) // 6)

    def inv2(a):
        f = pow((a[0] * a[0] + a[1] * a[1]) % Q, Q - 2, Q)
        return (a[0] * f % Q, -a[1] * f % Q)
    px, py = inv2(g2), inv2(g3)                       # w^(2-2q), w^(3-3q)  (ec.py:440-444)
    vals = [SQRT_N3, SQRT_N3M1O2, pow(SQRT_N3, Q - 2, Q), pow(2, Q - 2, Q), px[0], px[1], py[0], py[1]]
    return [to_m(v) for v in vals] + [pow(1 << 384, 3, Q)]


def h2c_scratch_consts():
    """contents of the constant slots [0, HC_END) of these programs' scratchpads"""
    return const_table()[:C_GAM] + h2c_const_table()


class H1Layout:
    """NE encodings per team."""

    def __init__(self, NE):
        self.NE = NE
        o = HC_END
        self.T = o; o += 2 * NE            # raw t (c0, c1), later Montgomery
        self.TH = o; o += 2 * NE           # wide inputs: bits 384.. of the 512-bit hash value
        self.PAR = o; o += NE              # parity flag of t
        self.X = o; o += 6 * NE            # x1, x2, x3 (Fq2 each); later X[0:2] = chosen x
        self.U = o; o += 6 * NE            # u_i = x_i^3 + b
        self.N = o; o += 3 * NE            # norms
        self.ACC = o; o += 3 * NE          # exponentiation accumulators
        self.BASE = o; o += 3 * NE         # exponentiation bases
        self.A1 = o; o += NE               # imaginary part of the chosen u
        self.FT = o; o += NE               # nz(t)
        self.S = o; o += 5 * NE            # result: (x, y, z.c0) Montgomery, z.c0 = nz(t)
        self.TEMP0 = o


class H2Layout:
    """NM messages per team; points are projective triples of Fq2 (6 slots).  Ordered by
    lifetime (the scratchpad size decides how many teams a compute unit holds): the points of
    the double-and-add loop first; the two encodings S are read by the first segment only and
    the affine result OUT is written by the last one only, so both share the area behind the
    points, which is temporaries for every segment in between."""

    def __init__(self, NM):
        self.NM = NM
        o = HC_END
        self.P = o; o += 6 * NM
        self.A = o; o += 6 * NM            # running accumulator
        self.T0 = o; o += 6 * NM           # [x]P
        self.TEMP_LOOP = o                 # first temporary of the segments that touch neither S nor OUT
        self.S = o                         # the two encodings (x, y, z.c0; Montgomery)
        self.OUT = o                       # canonical affine result
        self.TEMP_START = o + 10 * NM
        self.TEMP_OUT = o + 4 * NM

    def temp_base(self, name):
        return {"h2_start": self.TEMP_START, "h2_affine": self.TEMP_OUT}.get(name, self.TEMP_LOOP)


def h2_team_slots(segs, lay):
    return max(lay.temp_base(n) + sg.ntemp for n, sg in segs.items())


def _c2(b, base):
    return (b.inp(base), b.inp(base + 1))


def _sel(c, A, Bv):
    """c in {0,1}: A if c else B  (component-wise, one product each)."""
    return tuple((y + c * (x - y)) for x, y in zip(A, Bv))


def build_h1(NE, cfg=None, verbose=False, wide=False):
    """wide: t comes as the 512-bit hash512 value itself (ec.py:531-534 reduce it with
    `% q`): low 384 bits in T, the rest in TH; t R = lo R + hi R^2 = MUL(lo, R^2) + MUL(hi, R^3).
    Only the first segment differs (h1w_a); the others are shared with the narrow form."""
    cfg = cfg or tw.Cfg()
    L = H1Layout(NE)
    segs = {}

    def done(b):
        segs[b.name] = schedule(b, temp_base=L.TEMP0, verbose=verbose)
    # ---- a: candidates and their norms
    b0name = "h1w_a" if wide else "h1_a"
    b = Builder(b0name)
    r2, one, zero = b.inp(C_R2), b.inp(C_ONE), b.inp(C_ZERO)
    s3, hh, sinv = b.inp(HC_S3), b.inp(HC_H), b.inp(HC_SINV)
    r3 = b.inp(HC_R3) if wide else None
    for e in range(NE):
        if wide:
            t = tuple((b.inp(L.T + 2 * e + c) * r2 + b.inp(L.TH + 2 * e + c) * r3).mat() for c in range(2))
        else:
            t = ((b.inp(L.T + 2 * e) * r2).mat(), (b.inp(L.T + 2 * e + 1) * r2).mat())
        b.out(t[0], L.T + 2 * e), b.out(t[1], L.T + 2 * e + 1)
        b.out(t[1].sgn(), L.PAR + e)                              # parity: t.c1 > (-t).c1
        tt = cfg.sqr2(t)
        w = tw.f2_mat((tt[0] + one * 5, tt[1] + one * 4))         # t^2 + b' + 1
        wt = cfg.mul2(w, t)
        iwt = tw.f2_inv(cfg, wt)
        wi, ti = cfg.mul2(t, iwt), cfg.mul2(w, iwt)               # 1/w, 1/t
        b.out(cfg.mul2(t, ti)[0].mat(), L.FT + e)                 # 1, or 0 when t = 0
        wt2 = cfg.mul2(wi, t)
        wp = tw.f2_mat((wt2[0] * s3, wt2[1] * s3))                # w' = sqrt(-3) t / w
        wpt = cfg.mul2(wp, t)
        x1 = tw.f2_mat((hh - wpt[0], -wpt[1]))
        x2 = tw.f2_mat((-one - x1[0], -x1[1]))
        wti = cfg.mul2(w, ti)
        wpi = tw.f2_mat((wti[0] * sinv, wti[1] * sinv))           # 1/w'
        q3 = cfg.sqr2(wpi)
        x3 = tw.f2_mat((q3[0] + one, q3[1]))
        for i, x in enumerate((x1, x2, x3)):
            u = cfg.mul2(cfg.sqr2(x), x)
            u = tw.f2_mat((u[0] + one * 4, u[1] + one * 4))
            n = (u[0] * u[0] + u[1] * u[1]).mat()                  # N(u); real_u_step() zeroes it when a1 = 0
            k = 3 * e + i
            b.out(x[0], L.X + 2 * k), b.out(x[1], L.X + 2 * k + 1)
            b.out(u[0], L.U + 2 * k), b.out(u[1], L.U + 2 * k + 1)
            b.out(n, L.N + k), b.out(n, L.ACC + k), b.out(n, L.BASE + k)
    done(b)
    # ---- exponentiation steps on the first `cnt` accumulators
    for tag, cnt in (("3", 3 * NE), ("2", 2 * NE)):
        b = Builder("h1_sqr" + tag)
        for k in range(cnt):
            a = b.inp(L.ACC + k)
            b.out(a * a, L.ACC + k)
        done(b)
        b = Builder("h1_mul" + tag)
        for k in range(cnt):
            b.out(b.inp(L.ACC + k) * b.inp(L.BASE + k), L.ACC + k)
        done(b)
    # ---- b: pick the candidate, set up the two delta exponentiations
    b = Builder("h1_b")
    one, inv2 = b.inp(C_ONE), b.inp(HC_INV2)
    for e in range(NE):
        c, rr = [], []
        for i in range(3):
            k = 3 * e + i
            z, n = b.inp(L.ACC + k), b.inp(L.N + k)
            r = (z * n).mat()
            chi = (r * z).mat()
            c.append(((chi * chi + chi) * inv2).mat())             # 1 iff chi = 1 (0 for chi = -1 and for n' = 0)
            rr.append(r)
        xs = [_c2(b, L.X + 2 * (3 * e + i)) for i in range(3)]
        us = [_c2(b, L.U + 2 * (3 * e + i)) for i in range(3)]
        # x1 if c1 else (x2 if c2 else x3)        (index rule of ec.py:489-500)
        pack = [xs[i] + us[i] + (rr[i],) for i in range(3)]
        inner = tuple(v.mat() for v in _sel(c[1], pack[1], pack[2]))
        ch = tuple(v.mat() for v in _sel(c[0], pack[0], inner))
        x, u, r = ch[0:2], ch[2:4], ch[4]
        b.out(x[0], L.X + 6 * e), b.out(x[1], L.X + 6 * e + 1)
        b.out(u[1], L.A1 + e)
        dp = ((u[0] + r) * inv2).mat()
        dm = ((u[0] - r) * inv2).mat()
        for j, d in enumerate((dp, dm)):
            b.out(d, L.ACC + 2 * e + j), b.out(d, L.BASE + 2 * e + j)
    done(b)
    # ---- c: square root, sign, result
    b = Builder("h1_c")
    one, inv2 = b.inp(C_ONE), b.inp(HC_INV2)
    for e in range(NE):
        sq, chi = [], []
        for j in range(2):
            z, d = b.inp(L.ACC + 2 * e + j), b.inp(L.BASE + 2 * e + j)
            s = (z * d).mat()
            sq.append(s)
            chi.append(s * z)
        c = ((one + chi[0]) * inv2).mat()
        x0 = (sq[1] + c * (sq[0] - sq[1])).mat()
        a1 = b.inp(L.A1 + e)
        x1c = (a1 * (x0 * 2).inv()).mat()
        g = x1c.sgn()
        p = b.inp(L.PAR + e)
        flip = (g + p - (g * p) * 2).mat()          # g xor p
        y0 = (x0 - (flip * x0) * 2).mat()
        y1 = (x1c - (flip * x1c) * 2).mat()
        ft = b.inp(L.FT + e)
        pt = (b.inp(L.X + 6 * e), b.inp(L.X + 6 * e + 1), y0, y1)
        # t = 0: the projective point at infinity (0, 1, 0)        (ec.py:450-452)
        res = ((ft * pt[0]).mat(), (ft * pt[1]).mat(), (one + ft * (pt[2] - one)).mat(), (ft * pt[3]).mat(), ft)
        for k, v in enumerate(res):
            b.out(v, L.S + 5 * e + k)
    done(b)
    bits = bin(EXP_E)[3:]                              # below the leading one
    def exp_script(tag):
        sc = []
        for ch in bits:
            sc.append("h1_sqr" + tag)
            if ch == "1":
                sc.append("h1_mul" + tag)
        return sc
    script = [b0name] + exp_script("3") + ["h1_b"] + exp_script("2") + ["h1_c"]
    return segs, L, script


def real_u_step(team, L, NE):
    """The step the stage-0 kernel performs itself after h1_a / h1w_a (k_h2c_stage<0>, blsgpu_h2c.hip):
    a candidate whose u = x^3 + b' has zero imaginary part gets n' = 0, so that h1_b never picks it.
    Fq2.modsqrt returns an Fq for such u and y_for_x raises (fields.py:466-467, ec.py:93-104); sw_encode's
    bare except moves on to the next candidate (ec.py:489-500).  Vectors: tests/golden/g2_real_u.json."""
    from .sim import Q
    for k in range(3 * NE):
        if team[L.U + 2 * k + 1] % Q == 0:
            team[L.N + k] = 0


def _pt(b, base, m):
    o = base + 6 * m
    return tuple((b.inp(o + 2 * k), b.inp(o + 2 * k + 1)) for k in range(3))


def _out_pt(b, P, base, m, zero):
    o = base + 6 * m
    for k in range(3):
        for i in range(2):
            e = P[k][i]
            b.out(e if not e.is_zero() else zero, o + 2 * k + i)


def build_h2(NM, cfg=None, verbose=False):
    cfg = cfg or tw.Cfg(mat2=False)        # unmaterialised Fq2 products: 437 instead of 718 LIN rounds
    F = FA(2, cfg)
    L = H2Layout(NM)
    segs = {}

    def done(b):
        segs[b.name] = schedule(b, temp_base=L.temp_base(b.name), verbose=verbose)

    def psi(b, P):
        cx, cy = _c2(b, HC_PSIX), _c2(b, HC_PSIY)
        return (cfg.mul2(tw.f2_conj(P[0]), cx), cfg.mul2(tw.f2_conj(P[1]), cy), tw.f2_conj(P[2]))

    def neg(P):
        return (P[0], tw.f2_neg(P[1]), P[2])

    def matp(P):
        return tuple(tw.f2_mat(c) for c in P)
    b = Builder("h2_start")                            # P = S0 + S1 ; A = P
    one, zero = b.inp(C_ONE), b.inp(C_ZERO)
    for m in range(NM):
        s = [[b.inp(L.S + 10 * m + 5 * j + k) for k in range(5)] for j in range(2)]
        S0, S1 = [((v[0], v[1]), (v[2], v[3]), (v[4], b.zero())) for v in s]
        P = matp(padd(F, S0, S1))
        _out_pt(b, P, L.P, m, zero)
        _out_pt(b, P, L.A, m, zero)
    done(b)
    b = Builder("h2_dbl")
    zero = b.inp(C_ZERO)
    for m in range(NM):
        _out_pt(b, pdbl(F, _pt(b, L.A, m)), L.A, m, zero)
    done(b)
    for name, src in (("h2_add_p", "P"), ("h2_add_t0", "T0")):
        b = Builder(name)
        zero = b.inp(C_ZERO)
        for m in range(NM):
            _out_pt(b, padd(F, _pt(b, L.A, m), _pt(b, getattr(L, src), m)), L.A, m, zero)
        done(b)
    b = Builder("h2_save_t0")                          # T0 = A  (A keeps its value)
    zero = b.inp(C_ZERO)
    for m in range(NM):
        _out_pt(b, _pt(b, L.A, m), L.T0, m, zero)
    done(b)
    # The closing combination  R = ([x^2]P + [x]P - P) - psi([x]P + P) + psi^2(2P)  (ec.py:540-550)
    # runs as a chain of ONE point operation per message and segment, reusing the loop's
    # segments where it can: as one wide segment it needed three times the temporaries of the
    # double-and-add loop and so dictated the scratchpad size (and with it how many messages a
    # team and how many teams a compute unit hold) for 6 % of the rounds.
    # here A = [x^2]P, T0 = [x]P:   A += T0;  A -= P;  T0 += P;  T0 = psi(T0);  A -= T0;
    #                                T0 = 2P;  T0 = psi(psi(T0));  A += T0;  affine(A)
    for name, dst, fn in (
            ("h2_sub_p", "A", lambda b, m: padd(F, _pt(b, L.A, m), neg(_pt(b, L.P, m)))),
            ("h2_t0_add_p", "T0", lambda b, m: padd(F, _pt(b, L.T0, m), _pt(b, L.P, m))),
            ("h2_psi_t0", "T0", lambda b, m: psi(b, _pt(b, L.T0, m))),
            ("h2_sub_t0", "A", lambda b, m: padd(F, _pt(b, L.A, m), neg(_pt(b, L.T0, m)))),
            ("h2_dbl_p", "T0", lambda b, m: pdbl(F, _pt(b, L.P, m))),
            ("h2_psi2_t0", "T0", lambda b, m: psi(b, matp(psi(b, _pt(b, L.T0, m)))))):
        b = Builder(name)
        zero = b.inp(C_ZERO)
        for m in range(NM):
            _out_pt(b, fn(b, m), getattr(L, dst), m, zero)
        done(b)
    b = Builder("h2_affine")
    zero, raw1 = b.inp(C_ZERO), b.inp(C_RAW1)
    for m in range(NM):
        R = _pt(b, L.A, m)
        zi = tw.f2_inv(cfg, R[2])
        xa, ya = cfg.mul2(R[0], zi), cfg.mul2(R[1], zi)
        for k, v in enumerate((xa, ya)):
            for i in range(2):
                e = v[i].mat() * raw1
                b.out(e if not e.is_zero() else zero, L.OUT + 4 * m + 2 * k + i)
    done(b)

    def mul_x(add_seg):
        sc = []
        for ch in bin(NX)[3:]:
            sc.append("h2_dbl")
            if ch == "1":
                sc.append(add_seg)
        return sc
    script = ["h2_start"] + mul_x("h2_add_p") + ["h2_save_t0"] + mul_x("h2_add_t0") + \
             ["h2_add_t0", "h2_sub_p", "h2_t0_add_p", "h2_psi_t0", "h2_sub_t0", "h2_dbl_p", "h2_psi2_t0", "h2_add_t0", "h2_affine"]
    return segs, L, script
