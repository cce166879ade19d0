"""CPU: the lane tables and step programs of the wide Miller loop (vmgen/mlw_model.py -- what csrc/mlw_tables_gfx950.h holds and
csrc/blsgpu_mlw.hip executes), formulas checked mod q against the line-stream model's tangent / chord steps
(fields_t.py:1035-1078, 641-686 up to the scalings the final exponentiation removes), and the tables run digit by digit -- the
multiplier's 64-bit column bounds and the stored-value range asserted -- against the reference's vectors
(tests/golden/pairing.json; fields_t.py:1091-1121)."""
import json
import os
import random

from conftest import GOLDEN, cat
from vmgen import linestream_model as LS, mlw_model as M

Q = LS.Q


def _pairs(g1, g2, n):
    I = lambda b: int.from_bytes(b, "big")
    out = []
    for i in range(n):
        a, b = g1[96 * i:96 * (i + 1)], g2[192 * i:192 * (i + 1)]
        out.append(((I(a[:48]), I(a[48:])), ((I(b[:48]), I(b[48:96])), (I(b[96:144]), I(b[144:])))))
    return out


def _bytes(flat):
    return b"".join(x.to_bytes(48, "big") for x in flat)


def _env(rnd, T, Qa, px, py):
    val = {"ONE": 1, "ZERO": 0, "PX3N": -3 * px % Q, "PY": py, "PY3": 3 * py % Q}
    for n, v in (("X", T[0]), ("Y", T[1]), ("Z", T[2]), ("XQ", Qa[0]), ("YQ", Qa[1])):
        val[n + "0"], val[n + "1"] = v
    return val


def test_step_formulas_against_the_line_stream_model():
    rnd = random.Random(5)
    r2 = lambda: (rnd.randrange(Q), rnd.randrange(Q))
    for _ in range(3):
        T, Qa, px, py = (r2(), r2(), r2()), (r2(), r2()), rnd.randrange(Q), rnd.randrange(Q)
        val = _env(rnd, T, Qa, px, py)
        val.update(M.KINDS[M.KIND["L1"]].symbolic(val))
        out = M.KINDS[M.KIND["L20"]].symbolic(val)
        T2, line = LS.tangent(T, -3 * px % Q, py)
        assert ((out["X0"], out["X1"]), (out["Y0"], out["Y1"]), (out["Z0"], out["Z1"])) == T2
        assert tuple((out[("line", 0, 0, c, 0)], out[("line", 0, 0, c, 1)]) for c in range(3)) == line
        # the chord step on the doubled point
        val = _env(rnd, T2, Qa, px, py)
        for name in ("C1", "C21", "C3"):
            val.update(M.KINDS[M.KIND[name]].symbolic(val))
        out = M.KINDS[M.KIND["C4"]].symbolic(val)
        T3, cl = LS.chord(T2, Qa, -3 * px % Q, py)
        assert ((out["X0"], out["X1"]), (out["Y0"], out["Y1"]), (out["Z0"], out["Z1"])) == T3
        assert tuple((val[("line", 1, 1, c, 0)], val[("line", 1, 1, c, 1)]) for c in range(3)) == cl
        # the accumulator's steps
        f = [r2() for _ in range(6)]
        fv = {"f%d%d" % (k, p): f[k][p] for k in range(6) for p in range(2)}
        sq = M.KINDS[M.KIND["SQR"]].symbolic(fv)
        assert [(sq["f%d0" % k], sq["f%d1" % k]) for k in range(6)] == LS.mul_dense(f, f)
        fv.update({("line", 1, 1, c, p): cl[c][p] for c in range(3) for p in range(2)})
        ml = M.KINDS[M.KIND["MUL11"]].symbolic(fv)
        assert [(ml["f%d0" % k], ml["f%d1" % k]) for k in range(6)] == LS.mul_sparse(f, cl)
        # Q on the twist <=> D = 0
        val = _env(rnd, T, Qa, px, py)
        val.update(M.KINDS[M.KIND["CK1"]].symbolic(val))
        d = M.KINDS[M.KIND["CK2"]].symbolic(val)
        want = LS.sub2(LS.sub2(LS.mul2(Qa[1], Qa[1]), LS.mul2(LS.mul2(Qa[0], Qa[0]), Qa[0])), (4, 4))
        assert (d["D0"], d["D1"]) == want


def test_programs_have_the_loop_shape():
    acc, chain = M.programs()
    assert sum(1 for k in acc if k & M.LAST) == sum(1 for k in chain if k & M.LAST) == 64
    assert [k & 0x3f for k in acc[:2]] == [M.KIND["CK1"], M.KIND["CK2"]]      # the idle accumulator wave tests "Q on the twist"
    steps = lambda prog: [k & 0x3f for k in prog if k & 0x3f != M.NOP]
    assert sum(1 for k in steps(acc) if k == M.KIND["SQR"]) == 63
    assert len(steps(acc)) == 2 + 63 + 68                                  # a squaring per iteration, a sparse product per line
    assert sum(1 for k in steps(chain) if k == M.KIND["L1"]) == 63 and sum(1 for k in steps(chain) if k == M.KIND["C4"]) == 5


def test_tables_give_the_reference_pairing(oracle, seeded_pairs):
    with open(os.path.join(GOLDEN, "pairing.json")) as f:
        g = json.load(f)
    P, Qa = _pairs(bytes.fromhex(g["gen"]["g1"]), bytes.fromhex(g["gen"]["g2"]), 1)[0]
    f, ok, mx = M.miller(P, Qa)
    assert ok and mx < 1.02                                            # every stored value in (-q/64, q + q/64)
    assert oracle.final_exp(_bytes(LS.to_flat12(f))).hex() == g["gen"]["final_exp"]
    g1, g2 = seeded_pairs
    i = 7
    P, Qa = _pairs(g1[96 * i:], g2[192 * i:], 1)[0]
    f, ok, mx = M.miller(P, Qa)
    assert ok and mx < 1.02
    assert oracle.final_exp(_bytes(LS.to_flat12(f))) == oracle.pairing_multi(g1[96 * i:96 * (i + 1)], g2[192 * i:192 * (i + 1)], 1)


def test_three_wavefront_form(oracle, seeded_pairs):
    """the accumulator's steps split over two wavefronts (eight lanes per output, one product per lane; every step reads one copy
    of f and writes the other): the same field element whichever of the waves of a phase runs first"""
    pa, pb, pc, final = M.programs3()
    assert sum(1 for k in pa if k & M.LAST) == sum(1 for k in pb if k & M.LAST) == sum(1 for k in pc if k & M.LAST) == 2 + 2 * 63 + 5
    steps = lambda prog: [k & 0x3f for k in prog if k & 0x3f != M.NOP]
    assert len(steps(pa)) == 2 + 63 + 68 and len(steps(pb)) == 63 + 68 and len(steps(pc)) == 2 * 63 + 4 * 5
    assert all(M.KINDS[k].group == 8 for k in steps(pb)) and all(M.KINDS[k].K == 1 for k in steps(pa) + steps(pb) + steps(pc))
    # a step never writes the copy of f it reads (its partner wave may still be reading it)
    for k in set(steps(pb)) | set(steps(pa)[2:]):
        name = M.KINDS[k].name
        src, dst = name[-2], name[-1]
        assert {src, dst} == {"f", "g"}
        assert all(d[0] == dst for d, _, _ in M.KINDS[k].outputs)
        assert all(t[1][0] == src for _, _, prods in M.KINDS[k].outputs for a, _ in prods for t in a)
    g1, g2 = seeded_pairs
    i = 3
    P, Qa = _pairs(g1[96 * i:], g2[192 * i:], 1)[0]
    want = oracle.pairing_multi(g1[96 * i:96 * (i + 1)], g2[192 * i:192 * (i + 1)], 1)
    for rev in (False, True):
        f, ok, mx = M.miller(P, Qa, waves=3, reverse_waves=rev)
        assert ok and mx < 1.02
        assert oracle.final_exp(_bytes(LS.to_flat12(f))) == want


def test_degenerate_pairs_are_reported():
    """Q off the twist, or of an order that ends the chain at Z = 0: the kernel's two tests say so"""
    with open(os.path.join(GOLDEN, "pairing_degenerate.json")) as f:
        cases = json.load(f)["cases"]
    for name in ("ord13", "off_curve"):
        c = cases[name]
        P, Qa = _pairs(cat(c["g1"]), cat(c["g2"]), 1)[0]
        assert M.miller(P, Qa)[1] is False, name
        assert LS.pair_lines(P, Qa)[1] is False


def test_generated_tables_are_current():
    from vmgen import gen_mlw
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        p = gen_mlw.generate(os.path.join(td, "t.h"))
        assert open(p).read() == open(os.path.join(root, "python-bls_amd", "csrc", "mlw_tables_gfx950.h")).read()
