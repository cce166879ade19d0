"""Interpreter for the PACKED tables (the u16 records and round headers that go into
csrc/vm_tables.h), mirroring the kernel's decoding in csrc/blsgpu_kernels.hip
run_rounds: record layout, negative-then-positive phases with the sign flip,
combinations split over adjacent lanes and their DPP-style merge.  Values are residues mod q as in vmgen.sim; what
this adds over vmgen.sim is a CPU check of emit.py's encoding."""
from .emit import INACTIVE, kpad
from .sim import Q, R, RINV, to_m, from_m

T0 = sum((1 << 40) << (32 * j) for j in range(12))
BIAS = T0 + (-T0) % Q           # = 0 mod q; every 64-bit limb is 2^40 + (a 32-bit digit)
assert BIAS % Q == 0
PERM_L1 = (1, 1, 3, 3)          # quad_perm 0xF5
PERM_L2 = (2, 2, 2, 2)          # quad_perm 0xAA


class TableMachine:
    def __init__(self, consts, nslots, data, k1_slot=None):
        self.team = [0] * nslots
        for i, c in enumerate(consts):
            self.team[i] = c
        self.data = data

    def run(self, rounds, light=False):
        """light: the program is one of the kernel's light ones (products and combinations only);
        there kinds 2 / 3 are SAVE / RESTORE of the 12-slot window at slot K (programs.MPLayout)."""
        d16, team = self.data, self.team
        for off, meta in rounds:
            kind, K, levels, mn = meta & 3, (meta >> 8) & 0xFF, (meta >> 16) & 3, (meta >> 18) & 0xFF
            writes = []
            if light and kind >= 2:
                if kind == 2:
                    self.stash = team[K:K + 12]
                else:
                    team[K:K + 12] = self.stash
                continue
            if kind != 1:
                for lane in range(64):
                    a, b, d, _ = d16[off + 4 * lane: off + 4 * lane + 4]
                    if d == INACTIVE:
                        continue
                    assert a % 3 == 0 and b % 3 == 0 and d % 3 == 0
                    x = team[a // 3]
                    if kind == 0:
                        val = x * team[b // 3] * RINV % Q
                    elif kind == 2:
                        val = (pow(x, -1, Q) * R * R) % Q if x else 0
                    else:
                        val = to_m(1) if from_m(x) > (Q - 1) // 2 else 0
                    writes.append((d // 3, val))
            else:
                rl = kpad(K)
                acc, dst, w1s = [], [], []
                for lane in range(64):
                    rec = d16[off + rl * lane: off + rl * (lane + 1)]
                    dst.append(rec[0])
                    w1s.append(rec[1])
                    t = 0
                    for p in range(K):
                        if p == mn and mn:
                            t = BIAS - t                      # the kernel's in-place sign flip
                            assert t >= 0
                        u = rec[2 + p]
                        assert u < (1 << 15)
                        t += ((u >> 10) & 31) * team[u & 1023]
                    if mn == K:
                        t = BIAS - t
                    acc.append(t)
                for lv, perm, bit in ((1, PERM_L1, 14), (2, PERM_L2, 15)):
                    if levels >= lv:
                        acc = [acc[l] + (acc[(l & ~3) + perm[l & 3]] if (w1s[l] >> bit) & 1 else 0) for l in range(64)]
                for lane in range(64):
                    if dst[lane] != INACTIVE:
                        assert dst[lane] % 3 == 0 and 0 <= acc[lane] < (1 << 396)
                        writes.append((dst[lane] // 3, acc[lane] % Q))
            for d, val in writes:
                team[d] = val
