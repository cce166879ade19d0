"""Bit-exact Python interpreter for the VM tables (CPU check of the schedules).

Works in the GPU's own value domain: every slot holds the Montgomery content
x*R mod q (R = 2^384), MUL is a*b*R^-1 mod q, LIN micro-ops are modular
add/sub/double, INV is the Montgomery inverse with 0 -> 0.
"""
from .core import (OFF_BITS, OFF_MASK, SEL_CONST, SEL_TEAM, UOP_ADD, UOP_DBL,
                   UOP_NOP, UOP_SUB)

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 1 << 384
RINV = pow(R, -1, Q)


def to_m(x):
    return x * R % Q


def from_m(x):
    return x * RINV % Q


class Machine:
    def __init__(self, consts, team_slots):
        self.const = list(consts)
        self.team = [0] * team_slots
        self.rounds_run = {"mul": 0, "lin": 0, "inv": 0}
        self.lane_ops = {"mul": 0, "lin": 0, "inv": 0}
        self.uop_depth = 0

    def _resolve(self, r, bases):
        sel, off = r >> OFF_BITS, r & OFF_MASK
        if sel == SEL_CONST:
            return self.const, off
        if sel == SEL_TEAM:
            return self.team, off
        return self.team, bases[sel - 2] + off

    def rd(self, r, bases):
        arr, i = self._resolve(r, bases)
        return arr[i]

    def wr(self, r, bases, val):
        arr, i = self._resolve(r, bases)
        assert arr is self.team, "write to the constant region"
        arr[i] = val

    def run(self, seg, bases=(0, 0, 0, 0)):
        for rnd in seg.rounds:
            kind = rnd["kind"]
            self.rounds_run[kind] += 1
            self.lane_ops[kind] += len(rnd["lanes"])
            writes = []
            if kind == "mul":
                for a, b, d in rnd["lanes"]:
                    writes.append((d, self.rd(a, bases) * self.rd(b, bases) * RINV % Q))
            elif kind == "inv":
                for a, d in rnd["lanes"]:
                    x = self.rd(a, bases)
                    # content of inverse: (x R^-1)^-1 R = x^-1 R^2 ; 0 -> 0
                    writes.append((d, (pow(x, -1, Q) * R * R) % Q if x else 0))
            else:
                self.uop_depth += rnd["K"]
                for uops, d in rnd["lanes"]:
                    acc = 0
                    for op, s in uops:
                        if op == UOP_ADD:
                            acc = (acc + self.rd(s, bases)) % Q
                        elif op == UOP_SUB:
                            acc = (acc - self.rd(s, bases)) % Q
                        elif op == UOP_DBL:
                            acc = (acc * 2) % Q
                        else:
                            assert op == UOP_NOP
                    writes.append((d, acc))
            # all lanes read before any lane writes
            for d, val in writes:
                self.wr(d, bases, val)
