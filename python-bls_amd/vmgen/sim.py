"""Python interpreter for the SCHEDULED programs (CPU check of formulas, scheduling and
slot allocation; the packed encoding the kernel decodes is checked by tablesim.py).

Works in the GPU's own value domain: every slot holds the Montgomery content
x*R mod q (R = 2^384), MUL is a*b*R^-1 mod q, a LIN op is an exact sum of
coef * slot (negative terms through the 384-bit complement) reduced mod q once, INV is the Montgomery inverse with 0 -> 0.
"""

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 1 << 384
RINV = pow(R, -1, Q)


def to_m(x):
    return x * R % Q


def from_m(x):
    return x * RINV % Q


class Machine:
    """One team: a scratchpad of Fq slots (Montgomery contents)."""

    def __init__(self, consts, team_slots, const_base=0):
        self.team = [0] * team_slots
        self.nconst = len(consts)
        self.const_base = const_base
        for i, c in enumerate(consts):
            self.team[const_base + i] = c
        self.rounds_run = {"mul": 0, "lin": 0, "inv": 0, "sgn": 0}
        self.lane_ops = {"mul": 0, "lin": 0, "inv": 0, "sgn": 0}
        self.uop_depth = 0

    def rd(self, r):
        return self.team[r]

    def wr(self, r, val):
        assert not (self.const_base <= r < self.const_base + self.nconst), "write to a constant slot"
        self.team[r] = val

    def run(self, seg):
        for rnd in seg.rounds:
            kind = rnd["kind"]
            self.rounds_run[kind] = self.rounds_run.get(kind, 0) + 1
            self.lane_ops[kind] = self.lane_ops.get(kind, 0) + len(rnd["lanes"])
            writes = []
            if kind in ("save", "restore"):
                # the 12-slot window at slot rnd["K"] <-> the team's stash registers (programs.MPLayout)
                self.rounds_run[kind] = self.rounds_run.get(kind, 0)
                if kind == "save":
                    self.stash = self.team[rnd["K"]:rnd["K"] + 12]
                else:
                    self.team[rnd["K"]:rnd["K"] + 12] = self.stash
                continue
            if kind == "mul":
                for a, b, d in rnd["lanes"]:
                    writes.append((d, self.rd(a) * self.rd(b) * RINV % Q))
            elif kind == "inv":
                for a, d in rnd["lanes"]:
                    x = self.rd(a)
                    # content of inverse: (x R^-1)^-1 R = x^-1 R^2 ; 0 -> 0
                    writes.append((d, (pow(x, -1, Q) * R * R) % Q if x else 0))
            elif kind == "sgn":
                for a, d in rnd["lanes"]:
                    writes.append((d, to_m(1) if from_m(self.rd(a)) > (Q - 1) // 2 else 0))
            else:
                self.uop_depth += rnd["K"]
                for uops, d in rnd["lanes"]:
                    # exactly the GPU's non-negative accumulation (core.lower_lin)
                    acc = 0
                    for neg, cf, s in uops:
                        x = self.rd(s)
                        acc += cf * ((R - 1 - x) if neg else x)
                    assert 0 <= acc < (1 << 392)
                    writes.append((d, acc % Q))
            # all lanes read before any lane writes
            for d, val in writes:
                self.wr(d, val)
