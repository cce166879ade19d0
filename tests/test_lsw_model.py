"""CPU: the lane tables of the wide point chains of the line-stream stage (vmgen/lsw_model.py -- what csrc/lsw_tables_gfx950.h holds
and csrc/blsgpu_lsw.hip executes: sixteen lanes per pair, one output per lane) run digit by digit, the multiplier's 64-bit column
bounds and the stored-value range asserted, against the line-stream model's own lines (vmgen/linestream_model.pair_lines, the
formulas of k_ml_lines2; fields_t.py:1035-1078, 641-686 up to the scalings the final exponentiation removes)."""
import json
import os

from conftest import GOLDEN, cat
from vmgen import linestream_model as LS, lsw_model as M


def _pairs(g1, g2, n):
    I = lambda b: int.from_bytes(b, "big")
    out = []
    for i in range(n):
        a, b = g1[96 * i:96 * (i + 1)], g2[192 * i:192 * (i + 1)]
        out.append(((I(a[:48]), I(a[48:])), ((I(b[:48]), I(b[48:96])), (I(b[96:144]), I(b[144:])))))
    return out


def test_layout_and_shapes():
    assert M.ROWS * 16 >= M.NSLOTS and len(set(M.SLOT.values())) == M.NSLOTS
    assert [k.K for k in M.KINDS] == [2, 3, 3, 4, 4, 4, 1, 4] and [k.name for k in M.KINDS][:2] == ["L1", "L2"]
    # twelve outputs per level of the tangent step, one per lane; the line's six parts come out of its second level
    assert len(M.KINDS[0].outputs) == 12 and len(M.KINDS[1].outputs) == 12
    assert sorted(r.lineout for r in M.RECS[M.KIND["L2"]] if r.lineout is not None) == list(range(6))
    assert sorted(r.lineout for r in M.RECS[M.KIND["C2"]] if r.lineout is not None) == list(range(6))
    # a doubled multiple is stored only where a formula reads it beside another term
    assert {n for n, v in M.NEEDED.items() if 2 in v or -2 in v} == {"B0", "B1", "X0", "X1"}


def test_tables_give_the_lines_of_the_line_stream_model(seeded_pairs):
    with open(os.path.join(GOLDEN, "pairing.json")) as f:
        g = json.load(f)
    pairs = _pairs(bytes.fromhex(g["gen"]["g1"]), bytes.fromhex(g["gen"]["g2"]), 1) + _pairs(seeded_pairs[0][96 * 11:], seeded_pairs[1][192 * 11:], 2)
    for P, Qa in pairs:
        lines, ok, mx = M.pair_lines(P, Qa)
        want, ok2 = LS.pair_lines(P, Qa)
        assert ok and ok2 and mx < 1.02                # every stored value in (-q/64, q + q/64)
        assert lines == [tuple(l) for l in want]       # all 68 records, the same field elements as k_ml_lines2 writes


def test_degenerate_pairs_are_reported():
    with open(os.path.join(GOLDEN, "pairing_degenerate.json")) as f:
        cases = json.load(f)["cases"]
    for name in ("ord13", "off_curve", "qy_zero"):
        c = cases[name]
        P, Qa = _pairs(cat(c["g1"]), cat(c["g2"]), 1)[0]
        assert M.pair_lines(P, Qa)[1] is False and LS.pair_lines(P, Qa)[1] is False, name


def test_generated_tables_are_current():
    from vmgen import gen_lsw
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        p = gen_lsw.generate(os.path.join(td, "t.h"))
        assert open(p).read() == open(os.path.join(root, "python-bls_amd", "csrc", "lsw_tables_gfx950.h")).read()
