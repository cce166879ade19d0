"""Emit the scheduled VM programs as a C header for the HIP library.

Layout consumed by csrc/blsgpu_kernels.hip:
  BLSVM_MILLER_FLAT / BLSVM_FEXP_FLAT: the scripts expanded into the flat
      sequence of rounds the kernel walks: {data_off (u16 units), meta},
      meta = kind (0 MUL, 1 LIN, 2 INV, 3 SGN; in the light multi-pair programs 2 SAVE, 3 RESTORE of
             the 12-slot window at slot K) | K << 8   (K = micro-ops per lane, LIN)
  BLSVM_SEG_FLAT: the same for single segments the kernels call directly
      (BLSVM_SEGF_<NAME>_OFF / _LEN index it)
  BLSVM_DATA[] u16, per round a record per lane:
     MUL / INV / SGN: {a, b, dst, 0}                (4 u16)
     LIN      : {dst, uop_0 .. uop_{K-1}, pad..}    (roundup4(K + 1) u16)
  slot reference  = slot_index * 3 (byte offset / 16 inside the team scratchpad)
  micro-op        = neg << 15 | coef << 10 | slot_index:  acc += coef * (neg ? ~slot : slot)
  padding = 0 (coef 0); inactive lane: dst = 0xFFFF
  BLSVM_CONSTS: Montgomery contents of the constant slots, 12 LE u32 limbs each
"""
import hashlib
import re
import os

from . import decomp_programs as DP
from . import h2c_programs as HP
from . import msm_programs as MP
from . import core
from . import programs as P
from . import slow_programs as SP
from .core import LANES

KIND = {"mul": 0, "lin": 1, "inv": 2, "sgn": 3}
INACTIVE = 0xFFFF


def sref(slot):
    r = slot * 3
    assert 0 <= r < (1 << 14)
    return r


def kpad(K):
    """u16 per lane record of a LIN round: destination, merge flags, K micro-ops"""
    return (K + 2 + 3) & ~3


def plan_lin_round(lanes):
    """Lane records of one LIN round: ([(dst | None, flags, negatives, positives)], MN, MP, levels).

    * Every lane first sums its NEGATIVE terms (as plain products), then all lanes flip
      the sign of their accumulators at the same step (acc <- BIAS - acc, BIAS = 0 mod q
      with every 64-bit limb above any partial sum), then sum the positive terms: no
      per-term complement, no compensation constants.  MN / MP = longest negative /
      positive share of the round; shorter shares are padded with zero micro-ops.
      (The compensation micro-ops of core.lower_lin are dropped here.)
    * A long combination is split over 2 or 4 ADJACENT lanes (aligned), negatives and
      positives dealt out evenly; the kernel adds the partial accumulators across the
      group (DPP, `levels` steps) and the group's first lane reduces and stores.  Exact:
      the accumulators are integers and each flipped share adds one BIAS.
      flags: bit 14 = add the odd neighbour at level 1, bit 15 = add two lanes up at level 2.
    Chosen to minimise MN + MP + FLIP_COST [MN > 0] + LEVEL_COST levels."""
    ops = []
    for uops, d in lanes:
        real = [u for u in uops if u[2] != P.C_K1]
        assert sum(cf for neg, cf, s in real if neg) == sum(cf for neg, cf, s in uops if s == P.C_K1), "compensation does not match"
        ops.append((d, [u for u in real if u[0]], [u for u in real if not u[0]]))

    best = core.lin_round_cost([(len(n), len(p)) for _, n, p in ops])
    assert best is not None
    _, lv, gs = best
    plan = []
    for g in (4, 2, 1):                                   # widest groups first keeps every group aligned
        for (d, n, p), gg in zip(ops, gs):
            if gg != g:
                continue
            for part in range(g):
                flags = (1 << 14 if g >= 2 and part % 2 == 0 else 0) | (1 << 15 if g == 4 and part == 0 else 0)
                plan.append((d if part == 0 else None, flags, n[part::g], p[part::g]))
    mn = max(len(x[2]) for x in plan)
    mp = max(len(x[3]) for x in plan)
    if mn + mp == 0:
        mp = 1
    return plan, mn, mp, lv


def pack(segs, order):
    """-> (per-segment list of (data_off, meta), data)"""
    seg_rounds, data = {}, []
    for name in order:
        seg = segs[name]
        lst = []
        def mv(s):                                    # (every program is built in the kernel's own slot numbers)
            return s
        for rnd in seg.rounds:
            off = len(data)
            assert off % 4 == 0
            kind = rnd["kind"]
            lanes = rnd["lanes"]
            assert len(lanes) <= LANES
            if kind in ("save", "restore"):
                # light programs only: the 12-slot window at slot K <-> the stash registers; no
                # lane records (the look-ahead fetch reads the start of the data, harmlessly)
                assert name.startswith(("mp_", "mp2_")) and rnd["K"] < 256
                lst.append((0, (2 if kind == "save" else 3) | (rnd["K"] << 8)))
                continue
            if kind in ("mul", "inv", "sgn"):
                for ln in range(LANES):
                    if ln < len(lanes):
                        if kind == "mul":
                            a, b, d = lanes[ln]
                        else:
                            (a, d), b = lanes[ln], 0
                        data += [sref(mv(a)), sref(mv(b)), sref(mv(d)), 0]
                    else:
                        data += [0, 0, INACTIVE, 0]
            else:
                plan, mn, mp, lv = plan_lin_round(lanes)
                K = mn + mp
                assert K < 256 and len(plan) <= LANES
                kp = kpad(K)
                for ln in range(LANES):
                    rec = [0] * kp
                    if ln < len(plan):
                        d, flags, negs, poss = plan[ln]
                        rec[0] = sref(mv(d)) if d is not None else INACTIVE
                        rec[1] = flags
                        for base, lst_ in ((2, negs), (2 + mn, poss)):
                            for k, (neg, cf, s) in enumerate(lst_):
                                assert 0 < cf < 32 and 0 <= s < 1024
                                rec[base + k] = (cf << 10) | mv(s)
                    else:
                        rec[0] = INACTIVE
                    data += rec
                lst.append((off, KIND[kind] | (K << 8) | (lv << 16) | (mn << 18)))
                continue
            lst.append((off, KIND[kind]))
        seg_rounds[name] = lst
    return seg_rounds, data


POW_WINDOW = 3


def pow_windows(e, w=POW_WINDOW):
    """Left-to-right sliding-window schedule of b^e: [(squarings, k)] -- do `squarings`
    squarings, then multiply by the odd power b^(2k+1) (k = 255: no product).  The first
    entry has no squarings: it loads b^(2k+1)."""
    bits = bin(e)[2:]
    out, i, pend = [], 0, 0
    while i < len(bits):
        if bits[i] == "0":
            pend += 1
            i += 1
            continue
        L = min(w, len(bits) - i)
        while bits[i + L - 1] == "0":
            L -= 1
        v = int(bits[i:i + L], 2)
        out.append((pend + L if out else 0, (v - 1) // 2))
        assert out[-1][0] < 255
        pend = 0
        i += L
    if pend:
        out.append((pend, 255))
    # replay with integers
    acc = None
    for nsq, k in out:
        acc = 2 * k + 1 if acc is None else (acc << nsq) + (0 if k == 255 else 2 * k + 1)
    assert acc == e
    return out


def limbs32(v):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)]


DIRECT_SEGS = ["mul_0_1", "from_mont_1_0", "to_mont_0_1", "copy_1_0", "line_dbl", "line_add",     # called by name from the kernels
               "to_mont_1_1", "set_one_0", "mul_0_0", "add_0_1", "sub_0_1", "neg_0_0", "inv12_2_0", "copy_0_2",
               "h1_a", "h1w_a", "h1_b", "h1_c", "d1_a", "d1_c", "d2_a", "d2_b", "d2_c"]
POW_SEG = re.compile(r"^(h1_(sqr|mul)[23]|d1_(sqr|mul)|d2[pq]_(sqr|mul))$")
MSM_NP = {1: 6, 2: 2}                                         # points per team in the MSM kernels
HORNER_NP = {2: 5}                                            # sums per team in the batch Horner (k_msm_horner_np)
H1_NE, H2_NM = 12, 5                                          # encodings / messages per team (hash to G2)
D1_NE, D2_NE = 32, 16                                         # points per team (decompression)


def build_tables(verbose=False):
    """Everything the header is made of, as Python objects (also what tablesim runs)."""
    segs, mscript, fscript = P.build_all(verbose=verbose)
    mpsegs, mpscript, mplay = P.build_multi(verbose=verbose)
    segs.update(mpsegs)
    mp2segs, mp2script, mp2lay = P.build_multi(G=2, verbose=verbose, prefix="mp2")
    segs.update(mp2segs)
    msm = {}
    for deg, NP in MSM_NP.items():
        msegs, lay = MP.build(deg, NP, verbose=verbose)
        msm[deg] = (msegs, lay)
        segs.update(msegs)
    hmsm = {}
    for deg, NP in HORNER_NP.items():
        hsegs, hlay = MP.build_horner(deg, NP, verbose=verbose)
        hmsm[deg] = (hsegs, hlay)
        segs.update(hsegs)
    h1segs, h1lay, h1script = HP.build_h1(H1_NE, verbose=verbose)
    h1wsegs, _, h1wscript = HP.build_h1(H1_NE, verbose=verbose, wide=True)
    segs["h1w_a"] = h1wsegs["h1w_a"]
    h1segs = dict(h1segs, h1w_a=h1wsegs["h1w_a"])
    h2segs, h2lay, h2script = HP.build_h2(H2_NM, verbose=verbose)
    segs.update(h1segs)
    segs.update(h2segs)
    slowsegs, slowscript = SP.build(verbose=verbose)
    segs.update(slowsegs)
    d1segs, d1lay, d1script = DP.build_d1(D1_NE, verbose=verbose)
    d2segs, d2lay, d2script = DP.build_d2(D2_NE, verbose=verbose)
    segs.update(d1segs)
    segs.update(d2segs)
    # the fixed-exponent powers run in a register kernel (k_pow), not in the VM: their
    # squaring / multiplication segments stay out of the packed tables
    order = sorted(n for n in segs if not POW_SEG.match(n))
    seg_rounds, data = pack(segs, order)
    return dict(segs=segs, mscript=mscript, fscript=fscript, mpsegs=mpsegs, mpscript=mpscript, mplay=mplay, msm=msm,
                h1=(h1segs, h1lay, h1script), h1w=(h1segs, h1lay, h1wscript), h2=(h2segs, h2lay, h2script),
                d1=(d1segs, d1lay, d1script), d2=(d2segs, d2lay, d2script), order=order,
                slow=(slowsegs, slowscript), mp2=(mp2segs, mp2script, mp2lay), hmsm=hmsm,
                seg_rounds=seg_rounds, data=data)


def generate(path=None, verbose=False):
    tb = build_tables(verbose)
    segs, mscript, fscript, mpsegs, mpscript, msm = (tb[k] for k in ("segs", "mscript", "fscript", "mpsegs", "mpscript", "msm"))
    (h1segs, h1lay, h1script), (h2segs, h2lay, h2script) = tb["h1"], tb["h2"]
    order, seg_rounds, data = tb["order"], tb["seg_rounds"], tb["data"]
    team_slots = P.TEMP0 + max(s.ntemp for n, s in segs.items()
                               if not n.startswith(("g", "mp_", "mp2_", "h", "d1", "d2", "slow_", "line_")))
    mplay = tb["mplay"]
    # (the kernel's fallback for special pairs runs the single-pair program in the same scratchpad)
    mp_team_slots = max(team_slots, P.mp_team_slots(mpsegs))
    for deg, (msegs, lay) in msm.items():
        team_slots = max(team_slots, lay.TEMP0 + max(s.ntemp for n, s in msegs.items()))
    consts = P.const_table()
    mflat = [r for n in mscript for r in seg_rounds[n]]
    fflat = [r for n in fscript for r in seg_rounds[n]]
    out = []
    w = out.append
    w("/* generated by python-bls_amd/vmgen/emit.py -- do not edit */\n#pragma once\n#include <stdint.h>\n")
    w("#define BLSVM_NDATA %d\n" % len(data))
    maxk = max(sum(plan_lin_round(r["lanes"])[1:3]) for s in segs.values() for r in s.rounds if r["kind"] == "lin")
    w("#define BLSVM_MAX_LIN_K %d\n" % maxk)
    w("#define BLSVM_TEAM_SLOTS %d\n#define BLSVM_NCONST %d\n" % (team_slots, P.NCONST))
    for nm in ("PX", "PY", "QX0", "TX", "LD", "LA", "NPX3", "REG0", "NREG", "TEMP0", "C_ZERO", "C_ONE", "C_R2", "C_RAW1", "C_K1"):
        w("#define BLSVM_SLOT_%s %d\n" % (nm, getattr(P, nm)))

    def flat(name, lst):
        w("#define %s_LEN %d\nstatic const uint32_t %s[%d][2] = {\n" % (name, len(lst), name, max(1, len(lst))))
        for i in range(0, len(lst), 4):
            w("  " + " ".join("{%d,0x%x}," % x for x in lst[i:i + 4]) + "\n")
        w("};\n")
    w("#define BLSVM_MP_TEAM_SLOTS %d\n#define BLSVM_MP_NCONST %d\n#define BLSVM_MP_G %d\n" % (mp_team_slots, P.C_GAM, P.MP_G))
    w("/* multi-pair scratchpad (programs.MPLayout): accumulator, pair g's PX PY at CORE + 14 g, its Q at Q + 4 g */\n")
    w("#define BLSVM_MP_F %d\n#define BLSVM_MP_CORE %d\n#define BLSVM_MP_Q %d\n" % (mplay.F, mplay.CORE, mplay.Q))
    flat("BLSVM_MP_FLAT", [r for n in mpscript for r in seg_rounds[n]])
    mp2segs, mp2script, mp2lay = tb["mp2"]
    w("/* the same with TWO pairs per team (calls of a few thousand pairs) */\n")
    w("#define BLSVM_MP2_TEAM_SLOTS %d\n#define BLSVM_MP2_F %d\n#define BLSVM_MP2_CORE %d\n#define BLSVM_MP2_Q %d\n#define BLSVM_MP2_INIT_LEN %d\n"
      % (max(team_slots, P.mp_team_slots(mp2segs)), mp2lay.F, mp2lay.CORE, mp2lay.Q, len(seg_rounds[mp2script[0]])))
    flat("BLSVM_MP2_FLAT", [r for n in mp2script for r in seg_rounds[n]])
    w("/* rounds of the first segment: the kernels look at the on-curve residual (accumulator coefficients 1 .. 2 G) after it */\n")
    w("#define BLSVM_MILLER_INIT_LEN %d\n#define BLSVM_MP_INIT_LEN %d\n" % (len(seg_rounds[mscript[0]]), len(seg_rounds[mpscript[0]])))
    slowsegs, slowscript = tb["slow"]
    w("/* the reference-faithful Miller program (vmgen/slow_programs.py): affine R in the TX/TY slots, Q's flag as 0/1 in QINF */\n")
    w("#define BLSVM_SLOW_SLOTS %d\n#define BLSVM_SLOT_QINF %d\n" % (SP.team_slots(slowsegs), SP.QINF))
    flat("BLSVM_SLOW_FLAT", [r for n in slowscript for r in seg_rounds[n]])
    assert all(n.startswith("slow_") for n in slowscript)
    w("#define BLSVM_H1_NE %d\n#define BLSVM_H1_SLOTS %d\n" % (H1_NE, h1lay.TEMP0 + max(s.ntemp for s in h1segs.values())))
    w("#define BLSVM_H2_NM %d\n#define BLSVM_H2_SLOTS %d\n" % (H2_NM, HP.h2_team_slots(h2segs, h2lay)))
    w("#define BLSVM_H1_T %d\n#define BLSVM_H1_TH %d\n#define BLSVM_H1_S %d\n#define BLSVM_H2_S %d\n#define BLSVM_H2_OUT %d\n" % (h1lay.T, h1lay.TH, h1lay.S, h2lay.S, h2lay.OUT))
    w("#define BLSVM_H1_U %d\n#define BLSVM_H1_N %d\n" % (h1lay.U, h1lay.N))
    w("#define BLSVM_H1_ACC %d\n#define BLSVM_H1_BASE %d\n#define BLSVM_H1_STATE0 %d\n#define BLSVM_H1_STATE1 %d\n" % (h1lay.ACC, h1lay.BASE, h1lay.T, h1lay.TEMP0))
    w("#define BLSVM_NCONST_H2C %d\n#define BLSVM_HC_PSIX %d\n#define BLSVM_HC_PSIY %d\n" % (HP.HC_END, HP.HC_PSIX, HP.HC_PSIY))
    w("/* scratchpad slot of an extra constant -> its entry in the constant table */\n")
    w("#define BLSVM_HC_SLOT0 %d\n#define BLSVM_HC_TBL0 %d\n" % (P.C_GAM, HP.HC_TBL0))
    for tag, ne, (dsegs, dlay, dscript) in (("D1", D1_NE, tb["d1"]), ("D2", D2_NE, tb["d2"])):
        w("#define BLSVM_%s_NE %d\n#define BLSVM_%s_SLOTS %d\n" % (tag, ne, tag, dlay.TEMP0 + max(s.ntemp for s in dsegs.values())))
        w("#define BLSVM_%s_X %d\n#define BLSVM_%s_BIG %d\n#define BLSVM_%s_OUT %d\n" % (tag, dlay.X, tag, dlay.BIG, tag, dlay.OUT))
        w("#define BLSVM_%s_ACC %d\n#define BLSVM_%s_BASE %d\n#define BLSVM_%s_STATE0 %d\n#define BLSVM_%s_STATE1 %d\n" % (tag, dlay.ACC, tag, dlay.BASE, tag, dlay.X, tag, dlay.TEMP0))
    w("static const uint32_t BLSVM_POW_E[12] = {%s};   /* (q - 3) / 4 */\n#define BLSVM_POW_E_BITS %d\n" % (
        ", ".join("0x%08xu" % ((HP.EXP_E >> (32 * i)) & 0xFFFFFFFF) for i in range(12)), HP.EXP_E.bit_length()))
    pw = pow_windows(HP.EXP_E)
    w("/* sliding-window schedule of the same power: {squarings, k}: a <- a^(2^squarings) * b^(2k+1); k = 255: no product */\n")
    w("#define BLSVM_POW_WINDOW %d\n#define BLSVM_POW_STEPS %d\n" % (POW_WINDOW, len(pw)))
    w("static const uint8_t BLSVM_POW_WIN[%d][2] = {%s};\n" % (len(pw), ", ".join("{%d, %d}" % x for x in pw)))
    flat("BLSVM_H2_FLAT", [r for n in h2script for r in seg_rounds[n]])
    assert h2script[-1] == "h2_affine" and all(r["kind"] in ("mul", "lin") for n in h2script[:-1] for r in segs[n].rounds)
    w("#define BLSVM_H2_FINAL_LEN %d   /* rounds of the last segment (the only one with an inversion) */\n" % len(seg_rounds["h2_affine"]))
    flat("BLSVM_MILLER_FLAT", mflat)
    flat("BLSVM_FEXP_FLAT", fflat)
    # the one inversion of the final exponentiation sits in its first segment: the kernels walk that segment with
    # the full interpreter and everything after it with the light one (products and combinations only)
    assert fscript[0] == "inv12_2_0" and all(r["kind"] in ("mul", "lin") for n in fscript[1:] for r in segs[n].rounds)
    w("#define BLSVM_FEXP_HEAD_LEN %d\n" % len(seg_rounds[fscript[0]]))
    for deg, (msegs, lay) in msm.items():
        for nm in ("NP", "IN", "R", "A", "S", "PR0", "PR1", "OUT"):
            w("#define BLSVM_MSM%d_%s %d\n" % (deg, nm, getattr(lay, nm)))
    for deg, (hsegs, hlay) in tb["hmsm"].items():
        for nm in ("NP", "R", "S", "OUT"):
            w("#define BLSVM_HMSM%d_%s %d\n" % (deg, nm, getattr(hlay, nm)))
        w("#define BLSVM_HMSM%d_SLOTS %d\n" % (deg, hlay.TEMP0 + max(s.ntemp for s in hsegs.values())))
    direct = []
    for n in DIRECT_SEGS + sorted(n for n in segs if n.startswith(("g1_", "g2_", "g1h_", "g2h_"))):
        w("#define BLSVM_SEGF_%s_OFF %d\n#define BLSVM_SEGF_%s_LEN %d\n" % (n.upper(), len(direct), n.upper(), len(seg_rounds[n])))
        direct += seg_rounds[n]
    flat("BLSVM_SEG_FLAT", direct)
    w("static const uint16_t BLSVM_DATA[BLSVM_NDATA] = {\n")
    for i in range(0, len(data), 16):
        w("  " + ",".join(str(x) for x in data[i:i + 16]) + ",\n")
    w("};\nstatic const uint32_t BLSVM_CONSTS[%d][12] = {   /* [0, NCONST) shared, then the extra constants of the hashing programs */\n" % (P.NCONST + len(HP.h2c_const_table())))
    for c in consts + HP.h2c_const_table():
        w("  {" + ",".join("0x%08xu" % x for x in limbs32(c)) + "},\n")
    w("};\n")
    text = "".join(out)
    w_hash = hashlib.sha256(text.encode()).hexdigest()[:16]
    text += "#define BLSVM_TABLE_HASH \"%s\"\n" % w_hash
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc", "vm_tables.h")
    with open(path, "w") as f:
        f.write(text)
    return {"segments": len(order), "miller_rounds": len(mflat), "fexp_rounds": len(fflat), "data_u16": len(data),
            "team_slots": team_slots, "path": path, "hash": w_hash}


if __name__ == "__main__":
    import sys
    print(generate(verbose="-v" in sys.argv))
