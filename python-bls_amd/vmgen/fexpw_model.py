"""Lane-level model of the ONE-RESULT-PER-WAVEFRONT final exponentiation (round 4; csrc/blsgpu_fexpw.hip).

fq12_final_exp (fields_t.py:44, 1124-1128) of a SINGLE result is a chain of 314 cyclotomic squarings and 59 dense
products: nothing to batch.  The wavefront VM runs it on one wavefront with a field product per lane and its linear
combinations as separate rounds (958 rounds, ~550 k dependent instructions, 1.25 ms -- the floor under every call of the
engine); six lanes per result (blsgpu_fexp.hip) is a throughput form, ten results per wavefront.  Here ONE result owns
the wavefront and every Fq product of a step sits on its own lane:

  * quad o = lane / 4 (o < 12) is the HOME of the Fq value number o = 2 k + part of f = sum_k f_k w^k (part 0: real);
    its four lanes hold the value times 1, -1, 2, -2 (14 signed 28-bit limbs each, csrc/fp28.h);  quad 12 holds the
    constant 1/3, quad 15 zero.
  * a step gives every lane K products  (X[s1] + X[s2]) * (Y[s3] + Y[s4])  -- operands fetched from other lanes'
    registers with ds_bpermute and added limb-wise (no carries); the small coefficients c of c1 x + c2 x' are in the
    CHOICE of the source lane (the variant c x of the value, or the zero quad) -- summed by ONE Montgomery reduction
    (fp28_dot1 / fp28_dot3), the four lanes of a quad add their results (two DPP adds per limb), every lane scales the
    sum by the step's constant times ITS variant's factor and normalises the limbs: the quad's new value.
      CSQ   Granger-Scott squaring: lanes 0 .. 2 of quad (k, part) take the three products of
            fexp_model.cyc_sqr_lane_forms, lane 3 the term -+2 f_k as (-+2 f_k)(1/3); times 3 after the sum     [K = 1]
      MUL   dense product by the value G (a second register bank, loaded from a slot): c_k = sum_t F_{k-t} g_t, twelve
            products per part = three per lane                                                                [K = 3]
      FROB  lane-local products by gamma_{i,k} (G loaded with the constants), CONJ as a product by 1          [K = 3 / 1]
  * slots (ST / LD, the operand of MUL) live in LDS, one 64-byte row per lane.

The same script as fexp_model.script() drives it.  A cyclotomic squaring is then ~620 dependent instructions instead of
~1900 (six lanes per result) or a MUL round plus linear rounds of the VM; values stay below 16 q without any
modular correction (range model below).

This file is the table generator's source (gen_fexpw.py) and an interpreter that executes the TABLES digit by digit with
the 64-bit column bounds of the generated multiplier asserted (gen_fp28.model_dot); tests/test_fexpw_model.py pins it to
the reference's final-exponentiation vectors.
"""
from . import fexp_model as F
from .gen_fp28 import Q, R, L, W, MASK, to_limbs, from_limbs, model_dot

LANES = 64
HOME_QUADS = 12
THIRD_QUAD, ZERO_QUAD = 12, 15                     # quad 12: 1/3 (Montgomery form) in its variants, quad 15: zero
THIRD = pow(3, -1, Q) * R % Q
VARIANT = (1, -1, 2, -2)                           # lane r of a quad holds VARIANT[r] times the quad's value
ZERO_LANE = 4 * ZERO_QUAD


def home(o, r=0):
    return 4 * o + r


def src(c, o):
    """the lane that holds c times the value of quad o (c = 0: a lane of the zero quad)"""
    return ZERO_LANE if c == 0 else 4 * o + VARIANT.index(c)


def quad_of(k, part):
    return 2 * k + part


# ---- step tables: per lane a list of K products ((s1, s2), (s3, s4)) of source LANES: (X[s1] + X[s2]) (Y[s3] + Y[s4]); the
# first operand always from the accumulator bank V, the second from V (kind "vv") or from G (kind "vg": Y[s3] alone) -----
ZERO = ((ZERO_LANE, ZERO_LANE), (ZERO_LANE, ZERO_LANE))


def _operand(terms, srcs):
    """terms: {name: coefficient} over at most two names -> (lane1, lane2)"""
    items = [(c, n) for n, c in terms.items() if c]
    assert 1 <= len(items) <= 2 and sum(abs(c) for c, _ in items) <= 2
    (c1, n1) = items[0]
    (c2, n2) = items[1] if len(items) > 1 else (0, items[0][1])
    return (src(c1, srcs[n1]), src(c2, srcs[n2]))


def csq_forms(k, part):
    """the products of fexp_model.cyc_sqr_lane_forms for output (k, part) and the sign of the 2 f_k term"""
    if k % 2 == 0:
        if part == 0:
            return [({"xr": 1, "xi": 1}, {"xr": 1, "xi": -1}), ({"yr": 1, "yi": -1}, {"yr": 1, "yi": -1}), ({"yi": -2}, {"yi": 1})], -1
        return [({"xr": 2}, {"xi": 1}), ({"yr": 1, "yi": 1}, {"yr": 1, "yi": 1}), ({"yi": -2}, {"yi": 1})], -1
    if k == 1:
        if part == 0:
            return [({"xr": 2}, {"yr": 1, "yi": -1}), ({"xi": -2}, {"yi": 1, "yr": 1})], 1
        return [({"xr": 2}, {"yr": 1, "yi": 1}), ({"xi": 2}, {"yr": 1, "yi": -1})], 1
    if part == 0:
        return [({"xr": 2}, {"yr": 1}), ({"xi": -2}, {"yi": 1})], 1
    return [({"xr": 2}, {"yi": 1}), ({"xi": 2}, {"yr": 1})], 1


def table_csq():
    T = [[ZERO] for _ in range(LANES)]
    for k in range(6):
        p = F.CYC_PAIR[k]
        for part in range(2):
            srcs = {"xr": quad_of(p, 0), "xi": quad_of(p, 1), "yr": quad_of(p + 3, 0), "yi": quad_of(p + 3, 1)}
            forms, sgn = csq_forms(k, part)
            o = quad_of(k, part)
            for r, (a, b) in enumerate(forms):
                T[home(o, r)] = [(_operand(a, srcs), _operand(b, srcs))]
            # lane 3: (sgn 2 f_k) (1/3)
            T[home(o, 3)] = [((src(2 * sgn, o), ZERO_LANE), (src(1, THIRD_QUAD), ZERO_LANE))]
    return T


def table_unit(sign_of):
    """V <- sign_of(k) V as a product by one: lane 0 of every home quad (sign f)(1/3), times 3 after the sum"""
    T = [[ZERO] for _ in range(LANES)]
    for k in range(6):
        for part in range(2):
            o = quad_of(k, part)
            T[home(o, 0)] = [((src(sign_of(k), o), ZERO_LANE), (src(1, THIRD_QUAD), ZERO_LANE))]
    return T


def table_conj():
    return table_unit(lambda k: -1 if k & 1 else 1)


def table_mul():
    """c_k = sum_t F_{k-t} g_t, F_i = f_i (i >= 0) or xi f_{i+6}: real part sum_t Fr gr + (-Fi) gi, imaginary part
    sum_t Fr gi + Fi gr; xi f = (fr - fi, fr + fi).  The twelve products of a part: three per lane of the quad."""
    T = [[ZERO] * 3 for _ in range(LANES)]
    for k in range(6):
        for part in range(2):
            prods = []
            for t in range(6):
                s_, wrap = (k - t) % 6, t > k
                fr, fi = quad_of(s_, 0), quad_of(s_, 1)
                Fr = (src(1, fr), src(-1, fi)) if wrap else (src(1, fr), ZERO_LANE)
                Fi = (src(1, fr), src(1, fi)) if wrap else (src(1, fi), ZERO_LANE)
                nFi = (src(-1, fr), src(-1, fi)) if wrap else (src(-1, fi), ZERO_LANE)
                gr, gi = quad_of(t, 0), quad_of(t, 1)
                if part == 0:
                    prods += [(Fr, gr), (nFi, gi)]
                else:
                    prods += [(Fr, gi), (Fi, gr)]
            o = quad_of(k, part)
            for r in range(4):
                T[home(o, r)] = [(a, (src(1, g), ZERO_LANE)) for a, g in prods[3 * r:3 * r + 3]]
    return T


def table_frob(conj):
    """f_k -> conj^i(f_k) gamma_k with G = (gamma_k.re, gamma_k.im) in the home quads of k: lane 0 takes both products"""
    T = [[ZERO] * 3 for _ in range(LANES)]
    s_ = -1 if conj else 1
    for k in range(6):
        fr, fi = quad_of(k, 0), quad_of(k, 1)
        # re = fr gr - (s fi) gi ; im = fr gi + (s fi) gr
        T[home(fr, 0)] = [((src(1, fr), ZERO_LANE), (src(1, fr), ZERO_LANE)), ((src(-s_, fi), ZERO_LANE), (src(1, fi), ZERO_LANE)), ZERO]
        T[home(fi, 0)] = [((src(1, fr), ZERO_LANE), (src(1, fi), ZERO_LANE)), ((src(s_, fi), ZERO_LANE), (src(1, fr), ZERO_LANE)), ZERO]
    return T


# step kinds: name -> (table, K, second operand's bank, scale after the quad sum)
KINDS = {
    "CSQ": (table_csq(), 1, "V", 3),
    "CONJ": (table_conj(), 1, "V", 3),
    "MUL": (table_mul(), 3, "G", 1),
    "FROBC": (table_frob(True), 3, "G", 1),
    "FROB": (table_frob(False), 3, "G", 1),
}


def frob_constants(j):
    """the G bank for FROB j: gamma_{i,k} (i = 1, 2, 4), Montgomery digits, in lane 0 of the home quads (zero elsewhere)"""
    i = F.FROB_POW[j]
    G = [to_limbs(0) for _ in range(LANES)]
    for k in range(6):
        g = F.GAMMA[i][k]
        for part in range(2):
            G[home(quad_of(k, part))] = to_limbs(g[part] * R % Q)
    return G


# ---- digit-level interpreter ---------------------------------------------------------------------------------------
def norm(d):
    out, c = [], 0
    for j in range(L - 1):
        t = d[j] + c
        out.append(t & MASK)
        c = t >> W
    out.append(d[L - 1] + c)
    assert -(1 << 31) <= out[-1] < (1 << 31)
    return out


def _sum2(x1, x2):
    return [a + b for a, b in zip(x1, x2)]


def variants(value_mont):
    """the four lanes of a quad for a value given as an integer (Montgomery form, any representative)"""
    return [norm([c * d for d in to_limbs(value_mont)]) for c in VARIANT]


class Wave:
    """the register state of the wavefront: V and G (64 x 14 digits) and the LDS slots"""

    def __init__(self):
        self.V = [to_limbs(0) for _ in range(LANES)]
        for r, v in enumerate(variants(THIRD)):
            self.V[home(THIRD_QUAD, r)] = v
        self.G = [to_limbs(0) for _ in range(LANES)]
        self.slots = {}
        self.max_abs = 0                               # largest |value| seen in a home quad, in units of q

    def load_acc(self, f):
        """f: six (re, im) residues -> Montgomery digits in the home quads"""
        for k in range(6):
            for part in range(2):
                for r, v in enumerate(variants(f[k][part] * R % Q)):
                    self.V[home(quad_of(k, part), r)] = v

    def load_g(self, f):
        for k in range(6):
            for part in range(2):
                for r, v in enumerate(variants(f[k][part] * R % Q)):
                    self.G[home(quad_of(k, part), r)] = v

    def value(self):
        """the accumulator as residues (checks that the four lanes of every quad agree)"""
        rinv = pow(R, -1, Q)
        out = []
        for k in range(6):
            c = []
            for part in range(2):
                o = quad_of(k, part)
                assert all(from_limbs(self.V[home(o, r)]) == VARIANT[r] * from_limbs(self.V[home(o)]) for r in range(4))
                c.append(from_limbs(self.V[home(o)]) * rinv % Q)
            out.append(tuple(c))
        return out

    def step(self, kind):
        table, K, bank, scale = KINDS[kind]
        Y = self.V if bank == "V" else self.G
        P = []
        for lane in range(LANES):
            terms = []
            for (s1, s2), (s3, s4) in table[lane]:
                terms.append((_sum2(self.V[s1], self.V[s2]), _sum2(Y[s3], Y[s4])))
            assert len(terms) == K
            P.append(model_dot(terms))                 # asserts the 64-bit column bounds
        newV = []
        for lane in range(LANES):
            q0 = lane & ~3
            s = [scale * VARIANT[lane & 3] * sum(P[q0 + r][j] for r in range(4)) for j in range(L)]
            newV.append(norm(s))
        for lane in range(4 * HOME_QUADS):
            self.V[lane] = newV[lane]
            if lane & 3 == 0:
                self.max_abs = max(self.max_abs, abs(from_limbs(newV[lane])) / Q)

    def tinv(self):
        """the accumulator holds t in Fq2 in coefficient 0: V <- 1/t = conj(t) / (tr^2 + ti^2), 0 -> 0 (fields_t.py:47-55)"""
        tr, ti = self.V[home(0)], self.V[home(1)]
        n = model_dot([(tr, tr), (ti, ti)])
        rinv = pow(R, -1, Q)
        nv = from_limbs(n) * rinv % Q
        ni = to_limbs(pow(nv, Q - 2, Q) * R % Q)
        a = model_dot([(tr, ni)])
        b = model_dot([([-v for v in ti], ni)])
        zero = to_limbs(0)
        for lane in range(4 * HOME_QUADS):
            o = lane >> 2
            v = a if o == 0 else (b if o == 1 else zero)
            self.V[lane] = norm([VARIANT[lane & 3] * d for d in v])

    def run(self, script=None):
        for op, a in (script or F.script()):
            if op == F.MUL:
                self.G = [list(v) for v in self.slots[a]]
                self.step("MUL")
            elif op == F.CSQ:
                for _ in range(a):
                    self.step("CSQ")
            elif op == F.ST:
                self.slots[a] = [list(v) for v in self.V]
            elif op == F.LD:
                self.V = [list(v) for v in self.slots[a]]
            elif op == F.CONJ:
                self.step("CONJ")
            elif op == F.FROB:
                self.G = frob_constants(a)
                self.step("FROBC" if F.FROB_POW[a] % 2 else "FROB")
            elif op == F.TINV:
                self.tinv()
            else:
                break


def final_exp(f):
    """f (six (re, im) residues, w-power order) -> f^((q^12 - 1)/n) through the lane tables"""
    w = Wave()
    w.load_acc(f)
    w.run()
    return w.value(), w.max_abs
