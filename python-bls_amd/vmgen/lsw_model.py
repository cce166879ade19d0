"""Lane-level model and table builder of the WIDE point chains of the line-stream stage (round 5; csrc/blsgpu_lsw.hip
k_ml_lines_wide): the twist-point chain T <- 2T (+ Q) of fq_miller_loop (fields_t.py:1091-1111; lines :1035-1078, point steps
:641-686) with SIXTEEN LANES PER PAIR, four pairs per wavefront, for calls of a few thousand pairs -- where the register forms (one
pair per lane quad, k_ml_lines4) leave half the SIMDs empty and every call waits for one quad's chain of 68 steps (0.9 ms).

Same machine idea as the wide Miller loop (vmgen/mlw_model.py): every Fq value of a pair lives in LDS (limb j of slot s of pair i
at dword  896 (s / 16) + 4 (s % 16) + i + 64 j), a step gives every lane ONE output -- a sum of up to K products of sums of two
slots with one Montgomery reduction (fp28_dotK), scaled, a multiple of q taken off inside the carry pass -- and what the lanes read
and write are tables generated here from the SAME formulas (mlw_model.build_l1 / build_l2 / build_chord / build_check: checked
against linestream_model's tangent / chord).  Differences from the two-wavefront machine:

  * no lane sums: an output's products (at most four) all sit on its own lane -- twelve outputs per level, twelve of a pair's
    sixteen lanes busy;
  * a value is stored as itself and, where the formulas read it so, as its negative (a limb-wise negation: no carry pass) and as
    its double (a second scale-and-reduce of the same sum); a coefficient 2 on a lone operand is the same slot read twice;
  * the line coefficients go straight to the pair's line record in HBM (the records of k_ml_lines2 / k_ml_lines4: same field
    elements, here with normalised digits), the chain's state never leaves LDS.

Per loop iteration a wavefront issues two steps (~975 + ~1235 instructions) for FOUR pairs; a lane quad of k_ml_lines4 ~4.2 k for one.
tests/test_lsw_model.py runs the tables digit by digit (64-bit column bounds and stored-value range asserted) against
linestream_model.pair_lines.
"""
from . import linestream_model as LS
from . import mlw_model as W
from .gen_fp28 import Q, R, L, to_limbs, from_limbs, model_dot

LANES_PER_PAIR = 16
ROW = 14 * 64                                      # dwords per row of 16 slots x 4 pairs
# step kinds in table order and the formulas they come from (mlw_model's: the line buffers are irrelevant here)
_ch = W.build_chord(0)
_ck = W.build_check()
SOURCES = [("L1", W.build_l1()), ("L2", W.build_l2(0)), ("C1", _ch[0]), ("C2", _ch[1]), ("C3", _ch[2]), ("C4", _ch[3]),
           ("CK1", _ck[0]), ("CK2", _ck[1])]
INPUTS = ("PX3N", "PY", "PY3", "XQ0", "XQ1", "YQ0", "YQ1", "ONE", "X0", "X1", "Y0", "Y1", "Z0", "Z1")


def _refs(operand):
    """an operand (list of (coefficient, name)) -> list of at most two (name, variant) slot references"""
    out = []
    for c, s in operand:
        if abs(c) == 1:
            out.append((s, c))
        elif abs(c) == 2 and len(operand) == 1:
            out += [(s, c // 2)] * 2               # 2 x alone: the same slot twice
        elif abs(c) == 2:
            out.append((s, c))                     # beside another term: the stored double
        else:
            raise AssertionError(c)
    assert len(out) <= 2, operand
    return out


class Kind:
    def __init__(self, name, step):
        self.name, self.outputs = name, step.outputs
        assert len(self.outputs) <= LANES_PER_PAIR
        self.K = max(len(prods) for _, _, prods in self.outputs)


KINDS = [Kind(n, s) for n, s in SOURCES]
KIND = {k.name: i for i, k in enumerate(KINDS)}

# ---- which multiples of which value are ever read -> slots ------------------------------------------------------------------
NEEDED = {}
for k in KINDS:
    for dst, scale, prods in k.outputs:
        for a, b in prods:
            for name, v in _refs(a) + _refs(b):
                NEEDED.setdefault(name, set()).add(v)
for _n in ("Z0", "Z1", "D0", "D1"):                # read by the kernel itself (the tests "Q on the twist", "final Z = 0")
    NEEDED.setdefault(_n, set()).add(1)
SLOT = {("ZERO", 1): 0, ("TRASH", 1): 1}
for _n in sorted(NEEDED):
    for _v in sorted(NEEDED[_n], key=lambda v: (abs(v), -v)):
        SLOT.setdefault((_n, _v), len(SLOT))
NSLOTS = len(SLOT)
ROWS = -(-NSLOTS // 16)


def rel_addr(slot):
    """dword address of limb 0 of a slot, relative to the pair (the pair's lane group adds its index 0 .. 3)"""
    return ROW * (slot // 16) + 4 * (slot % 16)


def _dsts(name):
    """(+1, -1, +2, -2) slot addresses of a value (TRASH where that multiple is never read)"""
    t = SLOT[("TRASH", 1)]
    return tuple(SLOT.get((name, v), t) for v in (1, -1, 2, -2))


class Rec:
    """one lane's record of a step kind: K products of (a1, a2, b1, b2) slots, the destinations of the four multiples, the scale,
    and the line coefficient it writes (0 .. 5: l0.re l0.im l2.re l2.im l3.re l3.im; None)"""

    def __init__(self, prods, dsts, scale, lineout):
        self.prods, self.dsts, self.scale, self.lineout = prods, dsts, scale, lineout


def compile_kind(kind):
    z = SLOT[("ZERO", 1)]
    recs = []
    for lane in range(LANES_PER_PAIR):
        if lane >= len(kind.outputs):
            recs.append(Rec([(z, z, z, z)] * kind.K, (SLOT[("TRASH", 1)],) * 4, 0, None))
            continue
        dst, scale, prods = kind.outputs[lane]
        ps = []
        for a, b in prods:
            ra = [SLOT[r] for r in _refs(a)] + [z, z]
            rb = [SLOT[r] for r in _refs(b)] + [z, z]
            ps.append((ra[0], ra[1], rb[0], rb[1]))
        ps += [(z, z, z, z)] * (kind.K - len(ps))
        if isinstance(dst, tuple):                 # ("line", buffer, kind, coefficient, part)
            recs.append(Rec(ps, (SLOT[("TRASH", 1)],) * 4, scale, 2 * dst[3] + dst[4]))
        else:
            recs.append(Rec(ps, _dsts(dst), scale, None))
    return recs


RECS = [compile_kind(k) for k in KINDS]
HAS2 = [any(r.dsts[2] != SLOT[("TRASH", 1)] or r.dsts[3] != SLOT[("TRASH", 1)] for r in recs) for recs in RECS]


def input_records():
    """how a pair's inputs enter its value file: lane r handles one source (0 px, 1 py, 2 .. 5 xq.re xq.im yq.re yq.im, 6 the
    constant one) and stores it (times a scale) at up to two values' slots: [(source, [(value name, scale), ...])]"""
    return [(0, [("PX3N", -3)]), (1, [("PY", 1), ("PY3", 3)]), (2, [("XQ0", 1), ("X0", 1)]), (3, [("XQ1", 1), ("X1", 1)]),
            (4, [("YQ0", 1), ("Y0", 1)]), (5, [("YQ1", 1), ("Y1", 1)]), (6, [("ONE", 1), ("Z0", 1)])]


# ---- digit-level interpreter (one pair) ---------------------------------------------------------------------------------------
class Pair:
    def __init__(self):
        self.vf = {}
        self.max_abs = 0.0

    def rd(self, s):
        return self.vf.get(s, [0] * L)

    def store(self, name, digits_sum, scale):
        """the four multiples of scale x (the value whose un-normalised digits are digits_sum), as a step stores them"""
        self.store_at(_dsts(name), digits_sum, scale)

    def store_at(self, dsts, t, scale):
        trash = SLOT[("TRASH", 1)]
        v1 = W.scale_reduce_norm(t, scale)
        self.max_abs = max(self.max_abs, abs(from_limbs(v1)) / Q)
        if dsts[0] != trash:
            self.vf[dsts[0]] = v1
        if dsts[1] != trash:
            self.vf[dsts[1]] = [-d for d in v1]
        if dsts[2] != trash or dsts[3] != trash:
            v2 = W.scale_reduce_norm(t, 2 * scale)
            if dsts[2] != trash:
                self.vf[dsts[2]] = v2
            if dsts[3] != trash:
                self.vf[dsts[3]] = [-d for d in v2]
        return v1

    def step(self, kind):
        """returns {line coefficient index: digits} of the lanes that write one"""
        outs, writes = {}, []
        for rec in RECS[kind]:
            add = lambda x, y: [u + v for u, v in zip(self.rd(x), self.rd(y))]
            p = model_dot([(add(a1, a2), add(b1, b2)) for a1, a2, b1, b2 in rec.prods])
            writes.append((rec, p))
        for rec, p in writes:                      # every lane reads before any lane writes
            v1 = self.store_at(rec.dsts, p, rec.scale)
            if rec.lineout is not None:
                outs[rec.lineout] = v1
        return outs

    def value(self, name):
        return from_limbs(self.rd(SLOT[(name, 1)])) * pow(R, -1, Q) % Q


def pair_lines(P, Qa):
    """the 68 line records of one pair as the kernel writes them (three Fq2 residues each) and the validity of the fast formulas"""
    px, py = P
    (xq0, xq1), (yq0, yq1) = Qa
    src = [px, py, xq0, xq1, yq0, yq1, 1]
    pr = Pair()
    for s, outs in input_records():
        d = to_limbs(src[s] % Q * R % Q)
        for name, scale in outs:
            pr.store(name, d, scale)
    rinv = pow(R, -1, Q)
    res = lambda d: from_limbs(d) * rinv % Q
    pr.step(KIND["CK1"])
    pr.step(KIND["CK2"])
    on_twist = W.is_zero_stored(pr.rd(SLOT[("D0", 1)])) and W.is_zero_stored(pr.rd(SLOT[("D1", 1)]))
    lines = []
    for s, kind in LS.line_schedule():
        if kind == "t":
            pr.step(KIND["L1"])
            o = pr.step(KIND["L2"])
        else:
            pr.step(KIND["C1"])
            o = pr.step(KIND["C2"])
            pr.step(KIND["C3"])
            pr.step(KIND["C4"])
        lines.append(tuple((res(o[2 * c]), res(o[2 * c + 1])) for c in range(3)))
    z_zero = W.is_zero_stored(pr.rd(SLOT[("Z0", 1)])) and W.is_zero_stored(pr.rd(SLOT[("Z1", 1)]))
    return lines, on_twist and not z_zero, pr.max_abs
