"""Lane-level model and table builder of the WIDE Miller loop (round 5; csrc/blsgpu_mlw.hip): the loop of ONE pair
(fq_miller_loop, fields_t.py:1091-1111; lines :1035-1078, twist point steps :641-686) on a workgroup of TWO wavefronts with
a field product per lane -- the latency form for calls of a few pairs (BLS.verify of one signature is two pairs,
bls.py:197-201), where the wavefront VM's k_miller took 0.58 ms per pair whatever the count.

What a small call waits for is the number of instructions one wavefront issues (a lone wavefront issues an instruction
every ~5 cycles whatever it is).  So the two dependency chains of the loop run on two wavefronts (two SIMDs of one CU)
side by side and every step gives each lane ONE sum of two products:

  wave 1 (chain)  T <- 2T (+ Q) and the line coefficients, one loop iteration AHEAD of the accumulator
  wave 0 (acc)    f <- f^2, f <- f l (sparse product, positions 0, 2, 3 of the basis Fq2[w]/(w^6 - xi))

  * every Fq value lives in LDS (the "value file"): limb j of slot s at dword  s + 64 j  (+ 896 per page), so a lane reads any
    value with seven ds_read2st64_b32 and 64 lanes reading 64 different slots never meet in a bank.  A value is stored in the
    multiples 1, -1, 2, -2 (four consecutive slots: a "quad"), written by the four lanes that computed it, so the small
    coefficients of the formulas are in the CHOICE of the source slot, as in the one-result final exponentiation
    (fexpw_model.py) -- but with the values in LDS any lane reaches any value, and two wavefronts share them.
  * a step: every lane forms  (V[a1] + V[a2]) (V[b1] + V[b2]) + (V[a3] + V[a4]) (V[b3] + V[b4])  with ONE Montgomery reduction
    (fp28_dot2), the four lanes of a quad add their results (two DPP adds per limb), every lane multiplies the sum by its own
    scale (the step's constant times its multiple), takes a multiple of q off (read from the top limb: the stored value lies
    in (-q/64, q + q/64) whatever the scale) and normalises the limbs while doing so, and stores its slot.
  * the wavefronts meet at one barrier per loop iteration: the chain wave writes the lines of iteration i + 1 into one of
    two line buffers while the accumulator wave consumes those of iteration i from the other.

A pair for which the fast formulas are not the reference's value (Q flagged, Q off the twist, final Z = 0: DESIGN.md 2f) is
reported by the kernel and recomputed by k_ml_lines_exact / k_ml_small like a block of the wavefront-VM kernels.

This file builds the per-lane operand tables from the formulas (checked symbolically mod q against linestream_model's
tangent / chord), and executes the TABLES digit by digit with the multiplier's 64-bit column bounds asserted
(gen_fp28.model_dot); tests/test_mlw_model.py pins the result to the reference's vectors.
"""
from . import linestream_model as LS
from .gen_fp28 import Q, R, L, W, MASK, to_limbs, from_limbs, model_dot

LANES = 64
PAGE = 14 * 64                                     # dwords per page of 64 slots
VARIANT = (1, -1, 2, -2)
QTOP = Q >> (W * (L - 1))                          # 106513: the top limb of q
M47 = (1 << 47) // QTOP                            # k = (top * M47) >> 47 ~ top / QTOP
QD = to_limbs(Q)


# ---- the value file: names -> (page, quad) ------------------------------------------------------------------------------
class Names:
    """a value file layout: names -> (page, quad); "ZERO" and "TRASH" must be among them"""

    def __init__(self):
        self.at = {}

    @property
    def zero(self):
        return self.slot("ZERO")

    @property
    def trash(self):
        return self.slot("TRASH")

    def src(self, term):
        """dword address of one (coefficient, source) term of an operand"""
        c, s = term
        if isinstance(s, tuple):                   # ("line", buffer, kind, coefficient, part): the value itself only
            assert c == 1
            return line_slot(*s[1:])
        return self.slot(s, c)

    def put(self, page, quad, *names):
        for i, n in enumerate(names):
            assert n not in self.at and quad + i < 16
            self.at[n] = (page, quad + i)

    def slot(self, name, c=1):
        """dword address of limb 0 of c x name"""
        page, quad = self.at[name]
        return page * PAGE + 4 * quad + VARIANT.index(c)


N = Names()
# page 0: the accumulator in the one-result final exponentiation's layout (fexpw_model: quad 2 k + part)
N.put(0, 0, *["f%d%d" % (k, p) for k in range(6) for p in range(2)])
# page 1: the twist point, the pair's constants, zero and the write sink
N.put(1, 0, "X0", "X1", "Y0", "Y1", "Z0", "Z1", "ONE", "PX3N", "PY", "PY3", "XQ0", "XQ1", "YQ0", "YQ1", "TRASH", "ZERO")
# page 2: first level of the tangent step
N.put(2, 0, "A0", "A1", "B0", "B1", "E0", "E1", "F0", "F1", "XX0", "XX1", "YZ0", "YZ1", "CK0", "CK1", "D0", "D1")
# page 3: the chord step
N.put(3, 0, "TH0", "TH1", "LA0", "LA1", "CC0", "CC1", "DD0", "DD1", "CE0", "CE1", "H0", "H1", "GG0", "GG1")
# page 4: two line buffers of (tangent l0 l2 l3, chord l0 l2 l3), parts interleaved; only the value itself is stored, one
# slot per value: buffer b, line kind t (0 tangent, 1 chord), coefficient c, part p at slot 12 b + 6 t + 2 c + p
# page 5: the accumulator's second copy (three-wavefront form: the two accumulator waves read one copy and write the other)
N.put(5, 0, *["g%d%d" % (k, p) for k in range(6) for p in range(2)])
PAGES = 6
LINE_PAGE = 4
ZERO = N.slot("ZERO")
TRASH = N.slot("TRASH")


def line_slot(buf, kind, c, part):
    return LINE_PAGE * PAGE + 12 * buf + 6 * kind + 2 * c + part


# ---- formulas: an output is (destination, scale, [product, ...]); a product is (A, B), an operand a list of at most two
# (coefficient, source) with coefficient in +-1, +-2 and source a value name or ("line", buf, kind, c, part) -----------------
def T(*terms):
    return list(terms)


class Step:
    """one step kind: up to 16 outputs, each on a lane quad; the per-lane table it compiles to"""

    def __init__(self, name, outputs, names=None, group=4):
        names = names or N
        self.name, self.outputs, self.names, self.group = name, outputs, names, group
        assert group in (4, 8) and len(outputs) <= LANES // group
        # the shape the kernel specialises on: K products per lane (1 when no output has more products than its group has lanes), and
        # whether every second operand is ONE slot (the sparse products: a line coefficient).  group = 8 (the three-wavefront form):
        # an output's up to eight products on eight lanes, one each; its four multiples are written by the group's first four lanes
        self.K = 1 if all(len(prods) <= group for _, _, prods in outputs) else 2
        assert group == 4 or self.K == 1
        self.b_single = all(len(b) <= 1 for _, _, prods in outputs for _, b in prods)
        self.rec = []                              # per lane: (a1, a2, b1, b2, a3, a4, b3, b4, dst, scale)
        for g in range(LANES // group):
            if g < len(outputs):
                dst, scale, prods = outputs[g]
                assert len(prods) <= 8, (name, dst, len(prods))
            else:
                dst, scale, prods = None, 0, []
            for r in range(group):
                ops = []
                for t in range(2):
                    i = self.K * r + t if t < self.K else None
                    p = prods[i] if i is not None and i < len(prods) else ([], [])
                    for operand in p:
                        assert len(operand) <= 2
                        s = [names.src(x) for x in operand] + [names.zero, names.zero]
                        ops += s[:2]
                if dst is None or r >= 4:
                    d = names.trash
                elif isinstance(dst, tuple):       # a line coefficient: the value itself only
                    d = line_slot(*dst[1:]) if r == 0 else names.trash
                else:
                    d = names.slot(dst, VARIANT[r])
                self.rec.append(tuple(ops) + (d, scale * VARIANT[r & 3]))

    def symbolic(self, val):
        """evaluate mod q on a dict name -> residue (lines: key = the tuple); returns {dst: value}"""
        def opv(operand):
            return sum(c * val[s] for c, s in operand) % Q
        return {dst: scale * sum(opv(a) * opv(b) for a, b in prods) % Q for dst, scale, prods in self.outputs}


def _fq2_mul(x, y, cx=1):
    """products of the two parts of (cx x) y for names x, y (parts x0, x1)"""
    re = [(T((cx, x + "0")), T((1, y + "0"))), (T((-cx, x + "1")), T((1, y + "1")))]
    im = [(T((cx, x + "0")), T((1, y + "1"))), (T((cx, x + "1")), T((1, y + "0")))]
    return re, im


def _fq2_sqr(x):
    re = [(T((1, x + "0"), (1, x + "1")), T((1, x + "0"), (-1, x + "1")))]
    im = [(T((2, x + "0")), T((1, x + "1")))]
    return re, im


def build_sqr(part_of=None, group=4, name="SQR", src="f", dst="f"):
    """f <- f^2 in Fq2[w]/(w^6 - xi): c_k = sum over unordered {s, t}, s + t = k mod 6 (xi when s + t >= 6); part_of = (lo, hi):
    only the outputs lo .. hi - 1 of the twelve (the three-wavefront form splits them over two wavefronts)"""
    outs = []
    for k in range(6):
        for part in range(2):
            prods = []
            for s in range(6):
                t = (k - s) % 6
                if s > t:
                    continue
                wrap = s + t >= 6
                a, b = "%s%d" % (src, s), "%s%d" % (src, t)
                if s == t:
                    sq = (T((1, a + "0"), (1, a + "1")), T((1, a + "0"), (-1, a + "1")))           # re of f_s^2
                    cr = lambda c: (T((c, a + "0")), T((1, a + "1")))                              # c/2 x im of f_s^2
                    if not wrap:
                        prods += [sq] if part == 0 else [cr(2)]
                    else:                          # xi z = (zr - zi, zr + zi)
                        prods += [sq, cr(-2)] if part == 0 else [sq, cr(2)]
                elif not wrap:
                    re, im = _fq2_mul(a, b, 2)
                    prods += re if part == 0 else im
                else:                              # 2 (xi f_s) f_t, xi f_s = (s0 - s1, s0 + s1)
                    fr = T((2, a + "0"), (-2, a + "1"))
                    fi = T((2, a + "0"), (2, a + "1"))
                    nfi = T((-2, a + "0"), (-2, a + "1"))
                    if part == 0:
                        prods += [(fr, T((1, b + "0"))), (nfi, T((1, b + "1")))]
                    else:
                        prods += [(fr, T((1, b + "1"))), (fi, T((1, b + "0")))]
            outs.append(("%s%d%d" % (dst, k, part), 1, prods))
    if part_of:
        outs = outs[part_of[0]:part_of[1]]
    return Step(name, outs, group=group)


def build_mul_line(buf, kind, part_of=None, group=4, suffix="", src="f", dst="f"):
    """f <- f l for the line in buffer `buf` (kind 0 tangent, 1 chord): c_k = F_k l0 + F_{k-2} l2 + F_{k-3} l3,
    F_i = f_i (i >= 0) or xi f_{i+6}"""
    outs = []
    for k in range(6):
        for part in range(2):
            prods = []
            for c, j in enumerate(LS.LINE_POS):
                s, wrap = (k - j) % 6, j > k
                a = "%s%d" % (src, s)
                y0, y1 = ("line", buf, kind, c, 0), ("line", buf, kind, c, 1)
                if wrap:
                    fr, fi, nfi = T((1, a + "0"), (-1, a + "1")), T((1, a + "0"), (1, a + "1")), T((-1, a + "0"), (-1, a + "1"))
                else:
                    fr, fi, nfi = T((1, a + "0")), T((1, a + "1")), T((-1, a + "1"))
                if part == 0:
                    prods += [(fr, T((1, y0))), (nfi, T((1, y1)))]
                else:
                    prods += [(fr, T((1, y1))), (fi, T((1, y0)))]
            outs.append(("%s%d%d" % (dst, k, part), 1, prods))
    if part_of:
        outs = outs[part_of[0]:part_of[1]]
    return Step("MUL%d%d%s" % (buf, kind, suffix), outs, group=group)


def build_l1():
    """first level of the tangent step (vmgen/programs.t_double): A = X Y, B = Y^2, E = 12 xi Z^2, F = 3 E, XX = X^2, YZ = Y Z"""
    a_re, a_im = _fq2_mul("X", "Y")
    b_re, b_im = _fq2_sqr("Y")
    zsq = (T((1, "Z0"), (1, "Z1")), T((1, "Z0"), (-1, "Z1")))
    e_re = [zsq, (T((-2, "Z0")), T((1, "Z1")))]     # re of xi Z^2 = re - im
    e_im = [zsq, (T((2, "Z0")), T((1, "Z1")))]
    x_re, x_im = _fq2_sqr("X")
    y_re, y_im = _fq2_mul("Y", "Z")
    return Step("L1", [("A0", 1, a_re), ("A1", 1, a_im), ("B0", 1, b_re), ("B1", 1, b_im), ("E0", 12, e_re), ("E1", 12, e_im),
                       ("F0", 36, e_re), ("F1", 36, e_im), ("XX0", 1, x_re), ("XX1", 1, x_im), ("YZ0", 1, y_re), ("YZ1", 1, y_im)])


def build_l2(buf):
    """second level: X3 = 2 A (B - F), Y3 = B^2 + (2 B - E) F  (= (B + F)^2 - 12 E^2 with F = 3 E), Z3 = 8 B YZ;
    line (B - E, XX (-3 px), 2 YZ py) into buffer `buf`"""
    bf0, bf1 = T((1, "B0"), (-1, "F0")), T((1, "B1"), (-1, "F1"))
    x_re = [(T((2, "A0")), bf0), (T((-2, "A1")), bf1)]
    x_im = [(T((2, "A0")), bf1), (T((2, "A1")), bf0)]
    be0, be1, nbe1 = T((2, "B0"), (-1, "E0")), T((2, "B1"), (-1, "E1")), T((-2, "B1"), (1, "E1"))
    y_re = [(T((1, "B0"), (1, "B1")), T((1, "B0"), (-1, "B1"))), (be0, T((1, "F0"))), (nbe1, T((1, "F1")))]
    y_im = [(T((2, "B0")), T((1, "B1"))), (be0, T((1, "F1"))), (be1, T((1, "F0")))]
    z_re, z_im = _fq2_mul("B", "YZ")
    one = T((1, "ONE"))
    ln = lambda c, p: ("line", buf, 0, c, p)
    return Step("L2%d" % buf, [
        ("X0", 1, x_re), ("X1", 1, x_im), ("Y0", 1, y_re), ("Y1", 1, y_im), ("Z0", 8, z_re), ("Z1", 8, z_im),
        (ln(0, 0), 1, [(T((1, "B0"), (-1, "E0")), one)]), (ln(0, 1), 1, [(T((1, "B1"), (-1, "E1")), one)]),
        (ln(1, 0), 1, [(T((1, "XX0")), T((1, "PX3N")))]), (ln(1, 1), 1, [(T((1, "XX1")), T((1, "PX3N")))]),
        (ln(2, 0), 1, [(T((2, "YZ0")), T((1, "PY")))]), (ln(2, 1), 1, [(T((2, "YZ1")), T((1, "PY")))])])


def build_chord(buf):
    """the chord step T <- T + Q (vmgen/programs.t_add, px_is_m3) in four levels; line times 3 =
    (3 (th xq - la yq), th (-3 px), la 3 py) into buffer `buf`"""
    one = T((1, "ONE"))
    c1 = Step("C1", [
        ("TH0", 1, [(T((1, "Y0")), one), (T((-1, "YQ0")), T((1, "Z0"))), (T((1, "YQ1")), T((1, "Z1")))]),
        ("TH1", 1, [(T((1, "Y1")), one), (T((-1, "YQ0")), T((1, "Z1"))), (T((-1, "YQ1")), T((1, "Z0")))]),
        ("LA0", 1, [(T((1, "X0")), one), (T((-1, "XQ0")), T((1, "Z0"))), (T((1, "XQ1")), T((1, "Z1")))]),
        ("LA1", 1, [(T((1, "X1")), one), (T((-1, "XQ0")), T((1, "Z1"))), (T((-1, "XQ1")), T((1, "Z0")))])])
    c_re, c_im = _fq2_sqr("TH")
    d_re, d_im = _fq2_sqr("LA")
    ln = lambda c, p: ("line", buf, 1, c, p)
    l0_re = [(T((1, "TH0")), T((1, "XQ0"))), (T((-1, "TH1")), T((1, "XQ1"))), (T((-1, "LA0")), T((1, "YQ0"))), (T((1, "LA1")), T((1, "YQ1")))]
    l0_im = [(T((1, "TH0")), T((1, "XQ1"))), (T((1, "TH1")), T((1, "XQ0"))), (T((-1, "LA0")), T((1, "YQ1"))), (T((-1, "LA1")), T((1, "YQ0")))]
    c2 = Step("C2%d" % buf, [
        ("CC0", 1, c_re), ("CC1", 1, c_im), ("DD0", 1, d_re), ("DD1", 1, d_im),
        (ln(0, 0), 3, l0_re), (ln(0, 1), 3, l0_im),
        (ln(1, 0), 1, [(T((1, "TH0")), T((1, "PX3N")))]), (ln(1, 1), 1, [(T((1, "TH1")), T((1, "PX3N")))]),
        (ln(2, 0), 1, [(T((1, "LA0")), T((1, "PY3")))]), (ln(2, 1), 1, [(T((1, "LA1")), T((1, "PY3")))])])
    e_re, e_im = _fq2_mul("LA", "DD")
    g_re, g_im = _fq2_mul("X", "DD")
    lx0, lx1, nlx1 = T((1, "LA0"), (-2, "X0")), T((1, "LA1"), (-2, "X1")), T((-1, "LA1"), (2, "X1"))
    h_re = [(lx0, T((1, "DD0"))), (nlx1, T((1, "DD1"))), (T((1, "Z0")), T((1, "CC0"))), (T((-1, "Z1")), T((1, "CC1")))]
    h_im = [(lx0, T((1, "DD1"))), (lx1, T((1, "DD0"))), (T((1, "Z0")), T((1, "CC1"))), (T((1, "Z1")), T((1, "CC0")))]
    c3 = Step("C3", [("CE0", 1, e_re), ("CE1", 1, e_im), ("H0", 1, h_re), ("H1", 1, h_im), ("GG0", 1, g_re), ("GG1", 1, g_im)])
    x_re, x_im = _fq2_mul("LA", "H")
    z_re, z_im = _fq2_mul("Z", "CE")
    gh0, gh1 = T((1, "GG0"), (-1, "H0")), T((1, "GG1"), (-1, "H1"))
    y_re = [(T((1, "TH0")), gh0), (T((-1, "TH1")), gh1), (T((-1, "CE0")), T((1, "Y0"))), (T((1, "CE1")), T((1, "Y1")))]
    y_im = [(T((1, "TH0")), gh1), (T((1, "TH1")), gh0), (T((-1, "CE0")), T((1, "Y1"))), (T((-1, "CE1")), T((1, "Y0")))]
    c4 = Step("C4", [("X0", 1, x_re), ("X1", 1, x_im), ("Y0", 1, y_re), ("Y1", 1, y_im), ("Z0", 1, z_re), ("Z1", 1, z_im)])
    return [c1, c2, c3, c4]


def build_check():
    """Q on the twist: d = yq^2 - xq^3 - 4 (1 + u) = 0  (two levels: CK = xq^2, then D)"""
    k_re, k_im = _fq2_sqr("XQ")
    four = (T((-2, "ONE")), T((2, "ONE")))
    d_re = [(T((1, "YQ0"), (1, "YQ1")), T((1, "YQ0"), (-1, "YQ1"))), (T((-1, "CK0")), T((1, "XQ0"))), (T((1, "CK1")), T((1, "XQ1"))), four]
    d_im = [(T((2, "YQ0")), T((1, "YQ1"))), (T((-1, "CK0")), T((1, "XQ1"))), (T((-1, "CK1")), T((1, "XQ0"))), four]
    return [Step("CK1", [("CK0", 1, k_re), ("CK1", 1, k_im)]), Step("CK2", [("D0", 1, d_re), ("D1", 1, d_im)])]


# step kinds, in the order of the generated table
KINDS = [build_sqr(), build_mul_line(0, 0), build_mul_line(0, 1), build_mul_line(1, 0), build_mul_line(1, 1),
         build_l1(), build_l2(0), build_l2(1)]
_ch0, _ch1 = build_chord(0), build_chord(1)
KINDS += [_ch0[0], _ch0[1], _ch1[1], _ch0[2], _ch0[3]] + build_check()
# the three-wavefront form (calls of so few pairs that three SIMDs per pair are free): the accumulator's steps split over TWO
# wavefronts, an output's products on eight lanes -- one product per lane (A: outputs 0 .. 7 of the square, 0 .. 5 of a sparse
# product; B: the rest); the chain wave's kinds are the same
# -- and since the two run side by side, every step reads one copy of f (pages 0 / 5: "f" / "g") and writes the OTHER: no wave
# overwrites what its partner is still reading.  Kind names end in the direction: ...fg reads f, writes g.
for _src, _dst in (("f", "g"), ("g", "f")):
    _d = _src + _dst
    KINDS += [build_sqr((0, 8), 8, "SQRA" + _d, _src, _dst), build_sqr((8, 12), 8, "SQRB" + _d, _src, _dst)]
    for _b in range(2):
        for _k in range(2):
            KINDS += [build_mul_line(_b, _k, (0, 6), 8, "A" + _d, _src, _dst), build_mul_line(_b, _k, (6, 12), 8, "B" + _d, _src, _dst)]
KIND = {s.name: i for i, s in enumerate(KINDS)}
NOP = 0x3f
LAST = 0x80                                        # flag on the last step of a phase
# the routine a step kind runs (bits 8 - 9 of a program word): 0 two products per lane, 1 two products with one-slot second
# operands, 2 one product per lane
def variant_of(step):
    """0 two products per lane, 1 two products with one-slot second operands, 2 one product per lane, 3 one product per lane and
    eight lanes per output"""
    return 3 if step.group == 8 else (2 if step.K == 1 else (1 if step.b_single else 0))


def programs():
    """(acc program, chain program): one word per step -- bits 0 - 5 the kind, bit 7 = last step of its phase (the wavefronts meet
    at a barrier after every phase), bits 8 - 9 the routine (variant_of); NOP | LAST = nothing to do in this phase.  Phase p: the
    chain wave runs iteration p of the loop (p <= 62; lines into buffer p & 1), the accumulator wave iteration p - 1 (p >= 1) --
    and in phase 0, where it has nothing to multiply yet, the test "Q on the twist"."""
    chord_at = [s for s, kind in LS.line_schedule() if kind == "c"]
    iters = LS.NX.bit_length() - 1
    acc, chain = [], []
    for p in range(iters + 1):
        a, c = [], []
        if p >= 1:
            s = p - 1
            a = [KIND["SQR"], KIND["MUL%d0" % (s & 1)]] + ([KIND["MUL%d1" % (s & 1)]] if s in chord_at else [])
        if p == 0:
            a = [KIND["CK1"], KIND["CK2"]]
        if p < iters:
            c += [KIND["L1"], KIND["L2%d" % (p & 1)]]
            if p in chord_at:
                c += [KIND["C1"], KIND["C2%d" % (p & 1)], KIND["C3"], KIND["C4"]]
        for prog, steps in ((acc, a), (chain, c)):
            steps = [k | (variant_of(KINDS[k]) << 8) for k in steps] or [NOP]
            prog += steps[:-1] + [steps[-1] | LAST]
    return acc, chain


def programs3():
    """(acc A, acc B, chain) programs of the three-wavefront form.  Super-phase p: the chain wave runs iteration p of the loop, the
    two accumulator waves iteration p - 1 -- one step each per SUB-phase (a barrier after every accumulator step: each of the two needs
    all of f), the chain's steps dealt out over the sub-phases."""
    chord_at = [s for s, kind in LS.line_schedule() if kind == "c"]
    iters = LS.NX.bit_length() - 1
    progs = ([], [], [])
    cur = "f"                                      # the copy that holds the accumulator (starts as 1 in page 0)
    for p in range(iters + 1):
        a, b, c = [], [], []
        if p == 0:
            a, b = [KIND["CK1"], KIND["CK2"]], [NOP, NOP]
        if p >= 1:
            s = p - 1
            names = ["SQR%s", "MUL%d0%%s" % (s & 1)] + (["MUL%d1%%s" % (s & 1)] if s in chord_at else [])
            for n in names:
                nxt = "g" if cur == "f" else "f"
                a.append(KIND[n % "A" + cur + nxt])
                b.append(KIND[n % "B" + cur + nxt])
                cur = nxt
        if p < iters:
            c += [KIND["L1"], KIND["L2%d" % (p & 1)]]
            if p in chord_at:
                c += [KIND["C1"], KIND["C2%d" % (p & 1)], KIND["C3"], KIND["C4"]]
        n = len(a)
        per = -(-len(c) // n)
        for i in range(n):                         # sub-phase i
            for prog, steps in ((progs[0], [a[i]]), (progs[1], [b[i]]), (progs[2], c[i * per:(i + 1) * per])):
                steps = [k | (variant_of(KINDS[k]) << 8) if k != NOP else NOP for k in steps] or [NOP]
                prog += steps[:-1] + [steps[-1] | LAST]
    return progs + (cur,)


# ---- digit-level interpreter ---------------------------------------------------------------------------------------------
def scale_reduce_norm(t, s):
    """what every lane does with the quad's sum t (14 limbs) and its scale s: limbs of s t - k q, normalised, with
    k = floor(s t[13] M47 / 2^47) ~ s t / q"""
    top = t[L - 1] * s
    assert -(1 << 31) <= top < (1 << 31)
    k = (top * M47) >> 47
    out, c = [], 0
    for j in range(L - 1):
        c += t[j] * s - k * QD[j]
        assert -(1 << 63) <= c < (1 << 63)
        out.append(c & MASK)
        c >>= W
    out.append(c + t[L - 1] * s - k * QD[L - 1])
    assert -(1 << 31) <= out[-1] < (1 << 31)
    return out


class Machine:
    """the value file of one workgroup; step() executes one step kind's table for all 64 lanes of a wavefront"""

    def __init__(self, names=None, kinds=None):
        self.names, self.kinds = names or N, kinds or KINDS
        self.vf = {}                               # slot address -> 14 digits
        self.max_abs = 0.0                         # largest |stored value| / q seen
        self.store_value("ZERO", 0)
        self.store_value("TRASH", 0)
        self.store_value("ONE", 1)

    def rd(self, a):
        return self.vf.get(a, [0] * L)

    def store_value(self, name, x):
        """x (a residue) in Montgomery form, in its four multiples, the way the kernel stores an input"""
        d = to_limbs(x * R % Q)
        for r, c in enumerate(VARIANT):
            self.vf[self.names.slot(name, c)] = scale_reduce_norm(d, c)

    def value(self, name):
        return from_limbs(self.rd(self.names.slot(name))) * pow(R, -1, Q) % Q

    def line(self, buf, kind):
        rinv = pow(R, -1, Q)
        return tuple((from_limbs(self.rd(line_slot(buf, kind, c, 0))) * rinv % Q, from_limbs(self.rd(line_slot(buf, kind, c, 1))) * rinv % Q)
                     for c in range(3))

    def step(self, kind):
        rec = self.kinds[kind].rec
        TRASH = self.names.trash
        P = []
        for lane in range(LANES):
            a1, a2, b1, b2, a3, a4, b3, b4, dst, scale = rec[lane]
            add = lambda x, y: [u + v for u, v in zip(self.rd(x), self.rd(y))]
            P.append(model_dot([(add(a1, a2), add(b1, b2)), (add(a3, a4), add(b3, b4))]))
        writes = []
        grp = self.kinds[kind].group
        for lane in range(LANES):
            q0 = lane & ~(grp - 1)
            t = [sum(P[q0 + r][j] for r in range(grp)) for j in range(L)]
            v = scale_reduce_norm(t, rec[lane][9])
            if rec[lane][8] != TRASH:
                self.max_abs = max(self.max_abs, abs(from_limbs(v)) / Q)
            writes.append((rec[lane][8], v))
        for a, v in writes:                        # every lane reads before any lane writes
            self.vf[a] = v
        return writes


def is_zero_stored(d):
    """the kernel's test of a stored value (in (-q/64, q + q/64)): its digits are those of 0 or of q"""
    return all(x == 0 for x in d) or list(d) == QD


def miller(P, Qa, waves=2, reverse_waves=False):
    """the programs of the two- or three-wavefront form on one pair: f (six (re, im) residues, w-power order) and the validity of
    the fast formulas"""
    m = Machine()
    px, py = P
    (xq0, xq1), (yq0, yq1) = Qa
    for name, x in (("PX3N", -3 * px), ("PY", py), ("PY3", 3 * py), ("XQ0", xq0), ("XQ1", xq1), ("YQ0", yq0), ("YQ1", yq1),
                    ("X0", xq0), ("X1", xq1), ("Y0", yq0), ("Y1", yq1), ("Z0", 1), ("Z1", 0), ("f00", 1)):
        m.store_value(name, x % Q)
    final = "f"
    if waves == 2:
        progs = programs()
    else:
        *progs, final = programs3()
    order = list(range(len(progs)))
    if reverse_waves:
        order.reverse()                            # within a phase the waves touch disjoint values: any order gives the same digits
    at = [0] * len(progs)
    while at[0] < len(progs[0]):
        for w in order:                            # one phase: every wave's steps up to its LAST flag, then the barrier
            while True:
                k = progs[w][at[w]]
                at[w] += 1
                if k & 0x3f != NOP:
                    m.step(k & 0x3f)
                if k & LAST:
                    break
    assert all(a == len(p) for a, p in zip(at, progs))
    on_twist = is_zero_stored(m.rd(N.slot("D0"))) and is_zero_stored(m.rd(N.slot("D1")))
    z_zero = is_zero_stored(m.rd(N.slot("Z0"))) and is_zero_stored(m.rd(N.slot("Z1")))
    f = [(m.value("%s%d0" % (final, k)), m.value("%s%d1" % (final, k))) for k in range(6)]
    return f, on_twist and not z_zero, m.max_abs
