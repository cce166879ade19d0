"""Schedule compiler for the field-arithmetic VM that the HIP kernels interpret.

The GPU hot path (python-bls_amd/csrc/blsgpu_kernels.hip) does not hard-code the
pairing formulas.  One wavefront ("team") owns one pairing; its Fq values live
in an LDS scratchpad of 48-byte slots, and the 64 lanes execute *rounds*:

  MUL round   every active lane:  dst <- A * B            (Montgomery product)
  LIN round   every active lane:  dst <- sum of +/- coef * slot, coef = 1..31
  INV round   every active lane:  dst <- A^-1             (0 -> 0, like the
                                                           reference's fq_invert)
  SGN round   every active lane:  dst <- 1 if canonical(A) > (q-1)/2 else 0
                                  (the "lexicographically larger" test of ec.py:94-100)

This module traces straight-line formulas written over `E` expressions into a
DAG, list-schedules the DAG into rounds (<= 64 lanes each), allocates LDS slots
by live range and emits the per-lane operand tables.  The same tables are
executed by `sim.py` with Python integers, so the whole schedule is checked on
the CPU against the oracle before it ever reaches a GPU.
"""
from collections import defaultdict

LANES = 64

# A slot reference is simply the slot's index inside the team's scratchpad
# (constants, named registers and temporaries all live there).
NOSLOT = 0xFFFF



class V:
    """A materialised Fq value (one slot)."""
    __slots__ = ("id", "kind", "a", "b", "terms", "fixed", "name")

    def __init__(self, id, kind, a=None, b=None, terms=None, fixed=None, name=None):
        self.id, self.kind, self.a, self.b = id, kind, a, b
        self.terms, self.fixed, self.name = terms, fixed, name

    def __repr__(self):
        return "V%d:%s" % (self.id, self.kind)


class E:
    """Lazy integer-linear combination of materialised values."""
    __slots__ = ("b", "t")

    def __init__(self, b, t):
        self.b, self.t = b, {v: c for v, c in t.items() if c != 0}

    def is_zero(self):
        return not self.t

    def __add__(self, o):
        if isinstance(o, int) and o == 0:
            return self
        t = dict(self.t)
        for v, c in o.t.items():
            t[v] = t.get(v, 0) + c
        return E(self.b, t)

    __radd__ = __add__

    def __sub__(self, o):
        t = dict(self.t)
        for v, c in o.t.items():
            t[v] = t.get(v, 0) - c
        return E(self.b, t)

    def __neg__(self):
        return E(self.b, {v: -c for v, c in self.t.items()})

    def __mul__(self, o):
        if isinstance(o, int):
            return E(self.b, {v: c * o for v, c in self.t.items()})
        return self.b.mul(self, o)

    __rmul__ = __mul__

    def mat(self):
        """Force materialisation; returns an E with a single unit term."""
        if self.is_zero():
            return self
        v = self.b.materialise(self)
        return E(self.b, {v: 1})

    def inv(self):
        return self.b.inv(self)

    def sgn(self):
        return self.b.sgn(self)


class Builder:
    """Traces one segment."""

    def __init__(self, name):
        self.name = name
        self.vals = []
        self.outputs = []      # (V, fixed_ref)
        self._lin_cache = {}
        self._mul_cache = {}

    def _new(self, kind, **kw):
        v = V(len(self.vals), kind, **kw)
        self.vals.append(v)
        return v

    def inp(self, fixed, name=None):
        """A value that is live on entry in a fixed slot reference."""
        return E(self, {self._new("in", fixed=fixed, name=name): 1})

    def zero(self):
        return E(self, {})

    def materialise(self, e):
        assert not e.is_zero(), "cannot materialise the constant 0 (use a const slot)"
        items = tuple(sorted(((v.id, c) for v, c in e.t.items())))
        if len(items) == 1 and items[0][1] == 1:
            return next(iter(e.t))
        if items in self._lin_cache:
            return self._lin_cache[items]
        v = self._new("lin", terms=[(c, v) for v, c in sorted(e.t.items(), key=lambda x: x[0].id)])
        self._lin_cache[items] = v
        return v

    def mul(self, x, y):
        if x.is_zero() or y.is_zero():
            return self.zero()
        a, b = self.materialise(x), self.materialise(y)
        key = (min(a.id, b.id), max(a.id, b.id))
        if key not in self._mul_cache:
            self._mul_cache[key] = self._new("mul", a=a, b=b)
        return E(self, {self._mul_cache[key]: 1})

    def inv(self, x):
        a = self.materialise(x)
        return E(self, {self._new("inv", a=a): 1})

    def sgn(self, x):
        a = self.materialise(x)
        return E(self, {self._new("sgn", a=a): 1})

    def out(self, e, fixed):
        """Declare that expression e must be left in the fixed slot reference."""
        assert not e.is_zero(), "segment outputs must be non-constant-zero (copy from ZERO const instead)"
        v = self.materialise(e)
        self.outputs.append((v, fixed))


# ---------------------------------------------------------------------------
MAX_COEF = 31          # a micro-op adds coef * x, x = slot or its 384-bit complement
MAX_LIN_MAG = 160      # sum of |coefficients| per linear combination
# cost of a linear round beyond its steps, in steps (15 VALU each): the sign flip, a merge level
FLIP_COST, LEVEL_COST = 2.5, 3.0


def lin_round_cost(nps):
    """What emit.plan_lin_round will make of a round with these (negative, positive) term counts:
    (steps + flip + merge levels, levels, lanes per combination) of the cheapest way to split the
    combinations over 1, 2 or 4 adjacent lanes within 64 lanes; None if they do not fit."""
    best = None
    nmax = max(n for n, _ in nps)
    pmax = max(p for _, p in nps)
    for lv in (0, 1, 2):
        gmax = 1 << lv
        for tn in range(0, nmax + 1):
            for tp in range(0, pmax + 1):
                c = tn + tp + (FLIP_COST if tn else 0) + LEVEL_COST * lv
                if best is not None and c >= best[0]:
                    break
                gs, tot = [], 0
                for n, p in nps:
                    g = 1
                    while (-(-n // g) > tn or -(-p // g) > tp) and g < gmax:
                        g *= 2
                    if -(-n // g) > tn or -(-p // g) > tp:
                        tot = None
                        break
                    gs.append(g)
                    tot += g
                if tot is not None and tot <= LANES:
                    best = (c, lv, gs)
                    break
    return best
FOLD_COPY_MAX_OPS = 48  # a linear round takes folded output copies up to this many combinations
K1_SLOT = None         # set by programs.py: slot holding -(2^384 - 1) mod q


def lower_lin(terms):
    """[(coef, V)] -> list of (neg, coef, V|"K1"): the SCHEDULE-LEVEL form of a linear
    combination, which vmgen/sim.py executes exactly.  It models subtraction with
    non-negative quantities only: a negative term -c*x is c * (2^384 - 1 - x) (limb-wise
    complement) and the surplus c * (2^384 - 1) is cancelled by one extra micro-op
    N * K1, K1 = -(2^384 - 1) mod q, N = sum of the negative coefficients.
    The PACKED tables the kernel runs re-sort these terms (emit.plan_lin_round: negative
    terms first as plain sums, one sign flip, positive terms; the K1 entries are dropped);
    vmgen/tablesim.py executes that form."""
    uops = []
    mag = nmag = 0
    for c, v in terms:
        m = abs(c)
        mag += m
        if c < 0:
            nmag += m
        while m > 0:
            k = min(m, MAX_COEF)
            uops.append((1 if c < 0 else 0, k, v))
            m -= k
    assert mag <= MAX_LIN_MAG, "linear combination too large (%d)" % mag
    while nmag > 0:
        k = min(nmag, MAX_COEF)
        uops.append((0, k, "K1"))
        nmag -= k
    return uops


class Round:
    def __init__(self, kind):
        self.kind = kind       # 'mul' | 'lin' | 'inv'
        self.ops = []          # list of V (one per lane)


class Segment:
    """A scheduled, slot-allocated segment ready for emission / simulation."""

    def __init__(self, name):
        self.name = name
        self.rounds = []       # list of dict(kind, K, lanes=[...])
        self.ntemp = 0
        self.stats = {}


def schedule(b, temp_base=0, lanes=LANES, verbose=False, fold_copies=True, lazy_lin=False):
    """List-schedule builder b into rounds and allocate slots.

    Returns a Segment.  Temporaries are allocated upwards from slot temp_base.
    """
    vals = b.vals
    live_out = {v.id for v, _ in b.outputs}
    # prune dead values
    needed = set()
    stack = [v for v, _ in b.outputs]
    while stack:
        v = stack.pop()
        if v.id in needed:
            continue
        needed.add(v.id)
        if v.kind in ("mul",):
            stack += [v.a, v.b]
        elif v.kind in ("inv", "sgn"):
            stack.append(v.a)
        elif v.kind == "lin":
            stack += [t for _, t in v.terms]
    ops = [v for v in vals if v.id in needed and v.kind != "in"]

    def srcs(v):
        if v.kind == "mul":
            return [v.a, v.b]
        if v.kind in ("inv", "sgn"):
            return [v.a]
        return [t for _, t in v.terms]

    users = defaultdict(list)
    for v in ops:
        for s in srcs(v):
            users[s.id].append(v)
    # critical-path priority (cost: mul 10, inv 60, lin 1+terms/4)
    cost = {}
    for v in ops:
        cost[v.id] = 10 if v.kind == "mul" else (60 if v.kind == "inv" else (8 if v.kind == "sgn" else 1 + len(v.terms) // 4))
    prio = {}
    for v in reversed(ops):
        prio[v.id] = cost[v.id] + max([prio[u.id] for u in users[v.id]], default=0)

    # ---- levelled list scheduling ------------------------------------------
    # "heavy" ops (mul, inv) define levels: asap[v] = number of heavy ops on
    # the longest input->v path.  A heavy op may run at any level in
    # [asap, alap] without lengthening the schedule; ops with slack are used
    # to fill lanes of rounds that must run anyway.
    heavy = lambda v: v.kind in ("mul", "inv", "sgn")
    asap = {v.id: 0 for v in vals if v.kind == "in"}
    for v in ops:
        asap[v.id] = max([asap[s.id] for s in srcs(v)], default=0) + (1 if heavy(v) else 0)
    depth = max([asap[v.id] for v in ops], default=0)
    alap = {}
    for v in reversed(ops):
        lim = depth
        for u in users[v.id]:
            lim = min(lim, alap[u.id] - (1 if heavy(u) else 0))
        alap[v.id] = lim
    done = {v.id for v in vals if v.kind == "in"}
    pending = list(ops)
    rounds = []
    round_of = {}

    def emit(kind, chunk):
        r = Round(kind)
        r.ops = chunk
        for v in chunk:
            round_of[v.id] = len(rounds)
            done.add(v.id)
            pending.remove(v)
        rounds.append(r)

    def emit_lins(rl):
        rl.sort(key=lambda v: -prio[v.id])
        if len(rl) <= lanes:
            emit("lin", rl)
        else:
            # more ops than lanes: the longest combinations together in the first round(s) -- where
            # the emitter splits them over lanes -- and the short ones in the last, which then
            # needs neither a split nor its merge levels (dealing them out evenly cost 120 more
            # merge levels per three pairs for the same number of steps)
            nr = (len(rl) + lanes - 1) // lanes
            by_len = sorted(rl, key=lambda v: -len(v.terms))
            if nr == 2:
                # the cut that costs least (steps + flips + merge levels of both rounds)
                nps = [(sum(1 for c, _ in v.terms if c < 0), sum(1 for c, _ in v.terms if c > 0)) for v in by_len]
                best = None
                for cut in range(len(by_len) - lanes, lanes + 1):
                    if cut <= 0 or cut >= len(by_len):
                        continue
                    ca, cb = lin_round_cost(nps[:cut]), lin_round_cost(nps[cut:])
                    if ca is not None and cb is not None and (best is None or ca[0] + cb[0] < best[0]):
                        best = (ca[0] + cb[0], cut)
                emit("lin", by_len[:best[1]])
                emit("lin", by_len[best[1]:])
            else:
                tail = by_len[len(by_len) - lanes:]
                head = by_len[:len(by_len) - lanes]
                per = (len(head) + nr - 2) // (nr - 1)
                for r in range(nr - 1):
                    emit("lin", head[r * per:(r + 1) * per])
                emit("lin", tail)

    def choose_heavy(ready_ids, level, kinds=("inv", "sgn", "mul")):
        """the heavy ops of the next level among those whose sources are in ready_ids"""
        out = []
        for kind in kinds:
            rh = [v for v in pending if v.kind == kind and all(s.id in ready_ids for s in srcs(v))]
            must = [v for v in rh if alap[v.id] <= level]
            if not must:
                continue
            cap = lanes * ((len(must) + lanes - 1) // lanes)
            opt = sorted([v for v in rh if alap[v.id] > level], key=lambda v: (alap[v.id], -prio[v.id]))
            chosen = must + opt[:cap - len(must)]
            chosen.sort(key=lambda v: -prio[v.id])
            out.append((kind, chosen))
        return out

    level = 0
    while pending:
        if lazy_lin and any(heavy(v) for v in pending):
            # LIN phase, on demand: decide the next level's heavy ops first (looking through
            # the linear ops that COULD run now), then run only the linear ops they read;
            # the others wait for their own consumers, which spreads the linear work over
            # the levels instead of piling it up in front of the first one
            reach = set(done)
            grew = True
            lin_pending = [v for v in pending if v.kind == "lin"]
            while grew:
                grew = False
                for v in lin_pending:
                    if v.id not in reach and all(s.id in reach for s in srcs(v)):
                        reach.add(v.id)
                        grew = True
            level += 1
            plan = choose_heavy(reach, level)
            need, stack = set(), [s for _, ch in plan for v in ch for s in srcs(v)]
            while stack:
                x = stack.pop()
                if x.id in done or x.id in need or x.kind != "lin":
                    continue
                need.add(x.id)
                stack += srcs(x)
            while True:
                rl = [v for v in pending if v.id in need and all(s.id in done for s in srcs(v))]
                if not rl:
                    break
                emit_lins(rl)
            for kind, chosen in plan:
                for i in range(0, len(chosen), lanes):
                    emit(kind, chosen[i:i + lanes])
            assert level <= depth + 2 * len(ops), "scheduler failed to progress"
            continue
        # LIN phase: all ready linear ops, sub-level by sub-level
        while True:
            rl = [v for v in pending if v.kind == "lin" and all(s.id in done for s in srcs(v))]
            if not rl:
                break
            emit_lins(rl)
        if not pending:
            break
        level += 1
        for only in ("inv", "sgn", "mul"):               # a kind sees the results of the kinds before it
            for kind, chosen in choose_heavy(done, level, (only,)):
                for i in range(0, len(chosen), lanes):
                    emit(kind, chosen[i:i + lanes])
        assert level <= depth + 2 * len(ops), "scheduler failed to progress"
    # ---- live ranges -----------------------------------------------------
    last_use = {}
    for v in ops:
        for s in srcs(v):
            last_use[s.id] = max(last_use.get(s.id, -1), round_of[v.id])
    nrounds = len(rounds)
    # ---- slot assignment ---------------------------------------------------
    slot = {}
    for v in vals:
        if v.kind == "in":
            slot[v.id] = v.fixed
    out_fixed = {}
    for v, fx in b.outputs:
        out_fixed.setdefault(v.id, []).append(fx)
    # when does a fixed (input) slot become dead?  after the last use of the
    # input value that lives there (or never used -> -1)
    fixed_busy_until = {}
    for v in vals:
        if v.kind == "in" and v.id in needed:
            fixed_busy_until[v.fixed] = last_use.get(v.id, -1)
    free = []
    ntemp = 0
    copies = []                # (src_ref, dst_ref) needed at the end
    release_at = defaultdict(list)
    fixed_claimed = {}
    for ri, r in enumerate(rounds):
        # a slot whose last reader is THIS round can already take a result of this
        # round: inside a round every lane reads before any lane writes
        for off in release_at.pop(ri, []):
            free.append(off)
        for v in r.ops:
            want = out_fixed.get(v.id, [])
            chosen = None
            for fx in want:
                # may write straight into its output slot if whatever lives
                # there is dead by now (read-before-write inside a round is safe)
                if fixed_busy_until.get(fx, -1) <= ri and fx not in fixed_claimed:
                    chosen = fx
                    fixed_claimed[fx] = v.id
                    break
            if chosen is None:
                if free:
                    off = free.pop()
                else:
                    off = ntemp
                    ntemp += 1
                chosen = temp_base + off
                lu = last_use.get(v.id, ri)
                if v.id in live_out:
                    lu = nrounds      # must survive until the final copy
                else:
                    release_at[max(lu, ri + 1)].append(off)
            slot[v.id] = chosen
            for fx in want:
                if fx != chosen:
                    copies.append((v, fx))
    # outputs that are plain inputs (pass-through to another slot)
    for v, fx in b.outputs:
        if v.kind == "in" and v.fixed != fx:
            copies.append((v, fx))
    seg = Segment(b.name)
    for r in rounds:
        lanes_out = []
        if r.kind == "mul":
            for v in r.ops:
                lanes_out.append((slot[v.a.id], slot[v.b.id], slot[v.id]))
            seg.rounds.append({"kind": "mul", "K": 0, "lanes": lanes_out})
        elif r.kind in ("inv", "sgn"):
            for v in r.ops:
                lanes_out.append((slot[v.a.id], slot[v.id]))
            seg.rounds.append({"kind": r.kind, "K": 0, "lanes": lanes_out})
        else:
            K = 0
            for v in r.ops:
                u = [(neg, cf, (K1_SLOT if s == "K1" else slot[s.id])) for neg, cf, s in lower_lin(v.terms)]
                K = max(K, len(u))
                lanes_out.append((u, slot[v.id]))
            seg.rounds.append({"kind": "lin", "K": K, "lanes": lanes_out})
    # A copy temp -> output slot fits into an EXISTING linear round r when the value is
    # there (its producer ran before r) and the slot's old content has no reader after r
    # (inside a round every lane reads before any lane writes): no round of its own.
    # Pass-through copies (source = an input slot) keep their final round.
    if copies and fold_copies:
        passthrough_src = {v.fixed for v, _ in copies if v.kind == "in"}
        rest = []
        for v, fx in copies:
            best = None
            if v.kind != "in" and fx not in passthrough_src:
                r0 = max(round_of[v.id] + 1, fixed_busy_until.get(fx, -1))
                for ri in range(r0, nrounds):
                    r = seg.rounds[ri]
                    if r["kind"] == "lin" and len(r["lanes"]) < FOLD_COPY_MAX_OPS and (best is None or len(r["lanes"]) <= len(seg.rounds[best]["lanes"])):
                        best = ri
            if best is None:
                rest.append((v, fx))
            else:
                seg.rounds[best]["lanes"].append(([(0, 1, slot[v.id])], fx))
        copies = rest
    # final copy rounds (dst <- src as a 1-uop LIN); a chain of copies whose
    # destination is another copy's source must be ordered: do them in one
    # round, which is safe because every lane reads before any lane writes.
    if copies:
        # a copy must not overwrite a slot that a LATER copy round still reads
        srcs_all = {slot[v.id] for v, _ in copies}
        for i in range(0, len(copies), lanes):
            chunk = copies[i:i + lanes]
            later = {slot[v.id] for v, _ in copies[i + lanes:]}
            assert not ({fx for _, fx in chunk} & later), "copy rounds would clobber a pending source"
            seg.rounds.append({"kind": "lin", "K": 1,
                               "lanes": [([(0, 1, slot[v.id])], fx) for v, fx in chunk]})
    seg.ntemp = ntemp
    nm = sum(len(r["lanes"]) for r in seg.rounds if r["kind"] == "mul")
    rm = sum(1 for r in seg.rounds if r["kind"] == "mul")
    nl = sum(len(r["lanes"]) for r in seg.rounds if r["kind"] == "lin")
    rl = sum(1 for r in seg.rounds if r["kind"] == "lin")
    uops = sum(r["K"] for r in seg.rounds if r["kind"] == "lin")
    seg.stats = {"mul_ops": nm, "mul_rounds": rm, "lin_ops": nl, "lin_rounds": rl,
                 "lin_uop_depth": uops, "inv_rounds": sum(1 for r in seg.rounds if r["kind"] == "inv"),
                 "ntemp": ntemp, "copies": len(copies)}
    if verbose:
        print("%-14s %s" % (b.name, seg.stats))
    return seg
