"""The scheduled VM programs, executed by the Python interpreter (vmgen.sim) in
the GPU's own Montgomery domain, must reproduce the reference bit for bit.
CPU only: this is what lets a schedule change be validated without a GPU."""
import os
import re

import pytest

from conftest import ROOT
from vmgen import emit, programs as P, sim


@pytest.fixture(scope="module")
def built():
    segs, ms, fs = P.build_all()
    nslots = P.TEMP0 + max(s.ntemp for s in segs.values())
    return segs, ms, fs, nslots, P.const_table()


def run_pairing(built, g1, g2):
    segs, ms, fs, nslots, consts = built
    m = sim.Machine(consts, nslots)
    vals = [int.from_bytes(g1[i * 48:(i + 1) * 48], "big") for i in range(2)] + \
           [int.from_bytes(g2[i * 48:(i + 1) * 48], "big") for i in range(4)]
    for i, v in enumerate(vals):
        m.team[P.PX + i] = v
    for name in ms:
        m.run(segs[name])
    miller = [m.team[P.F + i] for i in range(12)]
    for name in fs:
        m.run(segs[name])
    m.run(segs["from_mont_1_0"])
    return miller, b"".join(m.team[P.reg(1) + i].to_bytes(48, "big") for i in range(12))


def test_pairing_of_generators(built, golden):
    g = golden("pairing.json")["gen"]
    _, out = run_pairing(built, bytes.fromhex(g["g1"]), bytes.fromhex(g["g2"]))
    assert out.hex() == g["final_exp"]


def test_small_multiples_and_product(built, golden, oracle):
    v = golden("pairing.json")["small4"]
    segs, ms, fs, nslots, consts = built
    acc = None
    for a, b in zip(v["g1"], v["g2"]):
        miller, out = run_pairing(built, bytes.fromhex(a), bytes.fromhex(b))
        # single pairing equals the oracle's
        assert out == oracle.pairing_multi(bytes.fromhex(a), bytes.fromhex(b), 1)
        if acc is None:
            acc = miller
        else:                       # fold with the mul_0_1 segment
            m = sim.Machine(consts, nslots)
            for i in range(12):
                m.team[P.reg(0) + i] = acc[i]
                m.team[P.reg(1) + i] = miller[i]
            m.run(segs["mul_0_1"])
            acc = [m.team[P.reg(0) + i] for i in range(12)]
    m = sim.Machine(consts, nslots)
    for i in range(12):
        m.team[P.reg(0) + i] = acc[i]
    for name in fs:
        m.run(segs[name])
    m.run(segs["from_mont_1_0"])
    out = b"".join(m.team[P.reg(1) + i].to_bytes(48, "big") for i in range(12))
    assert out.hex() == v["out"]


def test_final_exp_any_element(built, golden):
    segs, ms, fs, nslots, consts = built
    recs = golden("pairing.json")["final_exp"] + [{"in": "00" * 576, "out": "00" * 576}]
    for rec in recs:
        x = bytes.fromhex(rec["in"])
        m = sim.Machine(consts, nslots)
        for i in range(12):
            m.team[P.reg(0) + i] = sim.to_m(int.from_bytes(x[i * 48:(i + 1) * 48], "big"))
        for name in fs:
            m.run(segs[name])
        m.run(segs["from_mont_1_0"])
        out = b"".join(m.team[P.reg(1) + i].to_bytes(48, "big") for i in range(12))
        assert out.hex() == rec["out"]


def test_hard_part_exponent_identity():
    """E = (q^4-q^2+1)/n == ((x-1)^2/3)(x+q)(x^2+q^2-1) + 1 with x = -|x| (fields_t.py:44)."""
    q, x = sim.Q, -P.NX
    n = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    assert (q ** 4 - q ** 2 + 1) % n == 0 and (x - 1) % 3 == 0
    assert (q ** 4 - q ** 2 + 1) // n == ((x - 1) ** 2 // 3) * (x + q) * (x * x + q * q - 1) + 1


def test_schedule_invariants(built):
    segs = built[0]
    nslots = built[3]
    for name, seg in segs.items():
        for rnd in seg.rounds:
            assert 0 < len(rnd["lanes"]) <= 64
            dsts = [ln[-1] for ln in rnd["lanes"]]
            assert len(set(dsts)) == len(dsts), "two lanes write one slot in " + name
            for d in dsts:
                assert P.NCONST <= d < nslots, "write outside the scratchpad or into constants"


def test_committed_tables_are_current(tmp_path):
    """csrc/vm_tables.h is generated; the committed copy must match the generator."""
    path = os.path.join(ROOT, "python-bls_amd", "csrc", "vm_tables.h")
    if not os.path.exists(path):
        pytest.skip("tables not generated yet")
    info = emit.generate(path=str(tmp_path / "t.h"))
    with open(path) as f:
        have = re.search(r'BLSVM_TABLE_HASH "([0-9a-f]+)"', f.read()).group(1)
    assert have == info["hash"]


def test_packed_tables_decode_like_the_kernel(golden):
    """The PACKED tables (what vm_tables.h holds), decoded the way run_rounds decodes
    them -- record layout, folded compensation counts, combinations split over
    adjacent lanes -- reproduce the reference pairing: single-pair program, then the
    3-pairs-per-team program in its shifted address space."""
    from vmgen import tablesim
    tb = emit.build_tables()
    sr, data = tb["seg_rounds"], tb["data"]
    consts = P.const_table()
    g = golden("pairing.json")["gen"]
    g1, g2 = bytes.fromhex(g["g1"]), bytes.fromhex(g["g2"])
    vals = [int.from_bytes(g1[i * 48:(i + 1) * 48], "big") for i in range(2)] + \
           [int.from_bytes(g2[i * 48:(i + 1) * 48], "big") for i in range(4)]
    fexp = [r for n in tb["fscript"] for r in sr[n]] + sr["from_mont_1_0"]

    def finish(f12):
        m = tablesim.TableMachine(consts, 400, data, P.C_K1)
        for i in range(12):
            m.team[P.F + i] = f12[i]
        m.run(fexp)
        return b"".join(m.team[P.reg(1) + i].to_bytes(48, "big") for i in range(12))
    m = tablesim.TableMachine(consts, 400, data, P.C_K1)
    for i, v in enumerate(vals):
        m.team[P.PX + i] = v
    m.run([r for n in tb["mscript"] for r in sr[n]])
    assert finish([m.team[P.F + i] for i in range(12)]).hex() == g["final_exp"]
    # three pairs per team: small4[0..2]; expected = product of the three single pairings
    v = golden("pairing.json")["small4"]
    m = tablesim.TableMachine(consts[:P.C_GAM], P.mp_team_slots(tb["mpsegs"]), data, P.C_K1)
    singles = []
    for k in range(3):
        a, b = bytes.fromhex(v["g1"][k]), bytes.fromhex(v["g2"][k])
        pv = [int.from_bytes(a[i * 48:(i + 1) * 48], "big") for i in range(2)] + \
             [int.from_bytes(b[i * 48:(i + 1) * 48], "big") for i in range(4)]
        ps = tb["mplay"].pair(k)                     # the kernel puts P and Q of pair k here
        m.team[ps.PX], m.team[ps.PY] = pv[0], pv[1]
        for i in range(4):
            m.team[ps.QX0 + i] = pv[2 + i]
        s = tablesim.TableMachine(consts, 400, data, P.C_K1)
        for i, x in enumerate(pv):
            s.team[P.PX + i] = x
        s.run([r for n in tb["mscript"] for r in sr[n]])
        singles.append([s.team[P.F + i] for i in range(12)])
    m.run([r for n in tb["mpscript"] for r in sr[n]], light=True)
    got = finish([m.team[tb["mplay"].F + i] for i in range(12)])
    acc = singles[0]
    for nxt in singles[1:]:
        t = tablesim.TableMachine(consts, 400, data, P.C_K1)
        for i in range(12):
            t.team[P.reg(0) + i], t.team[P.reg(1) + i] = acc[i], nxt[i]
        t.run(sr["mul_0_1"])
        acc = [t.team[P.reg(0) + i] for i in range(12)]
    assert got == finish(acc)


def test_fixed_power_window_schedule():
    """The sliding-window schedule k_pow walks (emit.pow_windows) spells the exponent (q - 3) / 4 and,
    replayed on integers, gives the same power as pow()."""
    from vmgen import h2c_programs as HP
    q = sim.Q
    assert HP.EXP_E == (q - 3) // 4
    for w in (1, 2, 3, 4, 5):
        sched = emit.pow_windows(HP.EXP_E, w)            # asserts the exponent identity itself
        assert sched[0][0] == 0 and all(k == 255 or k < (1 << (w - 1)) for _, k in sched)
    sched = emit.pow_windows(HP.EXP_E)
    b = 0x1234567890abcdef1234567890abcdef % q
    odd = [pow(b, 2 * k + 1, q) for k in range(1 << (emit.POW_WINDOW - 1))]
    acc = odd[sched[0][1]]
    for nsq, k in sched[1:]:
        for _ in range(nsq):
            acc = acc * acc % q
        if k != 255:
            acc = acc * odd[k] % q
    assert acc == pow(b, HP.EXP_E, q)


def test_squaring_columns_of_the_generated_kernel():
    """gen_fqmul.generate_sqr takes each cross product once against the doubled operand
    (d'[k] = a[k] << 1 next to the diagonal, d[k] = (a[k] << 1) | (a[k-1] >> 31) above): the
    column sums must add up to a^2 for every a < 2^383 (2a fits twelve limbs)."""
    import random
    rng = random.Random(5)
    cases = [0, 1, (1 << 383) - 1, 4 * sim.Q - 1 if 4 * sim.Q < (1 << 383) else (1 << 383) - 1, 0x80000000 << 64]
    cases += [rng.randrange(1 << 383) for _ in range(300)]
    M = 0xFFFFFFFF
    for a in cases:
        A = [(a >> (32 * i)) & M for i in range(12)]
        e = [0] + [(A[k] << 1) & M for k in range(1, 12)]
        d = [0] + [e[k] | (A[k - 1] >> 31) for k in range(1, 12)]
        tot = 0
        for i in range(23):
            for j in range(max(0, i - 11), (i + 1) // 2):
                k = i - j
                tot += A[j] * (e[k] if k == j + 1 else d[k]) << (32 * i)
            if i % 2 == 0:
                tot += A[i // 2] ** 2 << (32 * i)
        assert tot == a * a


def test_scratchpad_budgets():
    """The scratchpad sizes that buy the measured occupancy: the multi-pair Miller
    programs must fit 16 teams into a compute unit's 160 KB of LDS (512-byte allocation granules), the
    cofactor clearing 12; the single-pair program is what the multi-pair kernel's fallback runs in the
    same scratchpad."""
    tb = emit.build_tables()
    from vmgen import h2c_programs as HP

    def teams(slots):
        return (160 * 1024) // (-(-slots * 48 // 512) * 512)
    mp = P.mp_team_slots(tb["mpsegs"])
    single = P.TEMP0 + max(s.ntemp for n, s in tb["segs"].items()
                           if not n.startswith(("g", "mp_", "mp2_", "h", "d1", "d2", "slow_", "line_")))
    assert teams(max(mp, single)) >= 16, (mp, single)
    assert P.mp_team_slots(tb["mp2"][0]) <= max(mp, single)            # the two-pair program shares the scratchpad size
    h2segs, h2lay, _ = tb["h2"]
    assert teams(HP.h2_team_slots(h2segs, h2lay)) >= 12
    # the Q window of the multi-pair layout is what three stash registers per lane hold
    assert 4 * tb["mplay"].G * 12 <= 3 * 64


def test_field_op_segments(built, golden):
    """add / sub / neg / squaring / inversion segments behind blsgpu_fq12_op_batch, on the reference's Fq12 KATs"""
    segs, ms, fs, nslots, consts = built
    rec = golden("fields.json")["12"]
    ops = [bytes.fromhex(x) for x in rec["operands"]]

    def load(m, r, x):
        for i in range(12):
            m.team[P.reg(r) + i] = sim.to_m(int.from_bytes(x[48 * i:48 * (i + 1)], "big"))

    def read(m, r):
        return b"".join(sim.from_m(m.team[P.reg(r) + i]).to_bytes(48, "big") for i in range(12))
    for op, seg in (("add", "add_0_1"), ("sub", "sub_0_1"), ("mul", "mul_0_1")):
        for e in rec[op]:
            m = sim.Machine(consts, nslots)
            load(m, 0, ops[e["i"]]), load(m, 1, ops[e["j"]])
            m.run(segs[seg])
            assert read(m, 0).hex() == e["r"], (op, e)
    for i in range(4):
        m = sim.Machine(consts, nslots)
        load(m, 0, ops[i])
        m.run(segs["neg_0_0"])
        assert read(m, 0).hex() == rec["neg"][i]
        m = sim.Machine(consts, nslots)
        load(m, 0, ops[i])
        m.run(segs["inv12_2_0"])
        m.run(segs["copy_0_2"])
        assert read(m, 0).hex() == rec["inv"][i]
        m = sim.Machine(consts, nslots)
        load(m, 0, ops[i]), load(m, 1, ops[i])
        m.run(segs["mul_0_0"])
        want = sim.Machine(consts, nslots)
        load(want, 0, ops[i]), load(want, 1, ops[i])
        want.run(segs["mul_0_1"])
        assert read(m, 0) == read(want, 0)


def test_two_pair_program_packed(golden):
    """the two-pairs-per-team Miller program (k_miller_mp<2>), packed tables decoded like the kernel:
    accumulator == product of the two single-pair Miller values after the final exponentiation"""
    from vmgen import tablesim
    tb = emit.build_tables()
    sr, data = tb["seg_rounds"], tb["data"]
    consts = P.const_table()
    mp2segs, mp2script, lay = tb["mp2"]
    fexp = [r for n in tb["fscript"] for r in sr[n]] + sr["from_mont_1_0"]

    def finish(f12):
        m = tablesim.TableMachine(consts, 400, data, P.C_K1)
        for i in range(12):
            m.team[P.F + i] = f12[i]
        m.run(fexp)
        return b"".join(m.team[P.reg(1) + i].to_bytes(48, "big") for i in range(12))
    v = golden("pairing.json")["small4"]
    m = tablesim.TableMachine(consts[:P.C_GAM], P.mp_team_slots(mp2segs), data, P.C_K1)
    singles = []
    for k in range(2):
        a, b = bytes.fromhex(v["g1"][k + 1]), bytes.fromhex(v["g2"][k + 1])
        pv = [int.from_bytes(a[i * 48:(i + 1) * 48], "big") for i in range(2)] + \
             [int.from_bytes(b[i * 48:(i + 1) * 48], "big") for i in range(4)]
        ps = lay.pair(k)
        m.team[ps.PX], m.team[ps.PY] = pv[0], pv[1]
        for i in range(4):
            m.team[ps.QX0 + i] = pv[2 + i]
        s = tablesim.TableMachine(consts, 400, data, P.C_K1)
        for i, x in enumerate(pv):
            s.team[P.PX + i] = x
        s.run([r for n in tb["mscript"] for r in sr[n]])
        singles.append([s.team[P.F + i] for i in range(12)])
    m.run([r for n in mp2script for r in sr[n]], light=True)
    t = tablesim.TableMachine(consts, 400, data, P.C_K1)
    for i in range(12):
        t.team[P.reg(0) + i], t.team[P.reg(1) + i] = singles[0][i], singles[1][i]
    t.run(sr["mul_0_1"])
    assert finish([m.team[lay.F + i] for i in range(12)]) == finish([t.team[P.reg(0) + i] for i in range(12)])
