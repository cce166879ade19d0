"""CPU: the lane tables of the one-result-per-wavefront final exponentiation (vmgen/fexpw_model.py -- what
csrc/fexpw_tables_gfx950.h holds and csrc/blsgpu_fexpw.hip executes), run digit by digit with the 64-bit column bounds of the
generated multiplier asserted, against the reference's vectors (tests/golden/pairing.json, fields_t.py:1124-1128) and
the six-lane model."""
import json
import os
import random

from conftest import GOLDEN
from vmgen import fexp_model as F, fexpw_model as M


def _ints(b):
    return [int.from_bytes(b[48 * i:48 * (i + 1)], "big") for i in range(12)]


def _bytes(flat):
    return b"".join(x.to_bytes(48, "big") for x in flat)


def test_tables_give_the_reference_vectors():
    with open(os.path.join(GOLDEN, "pairing.json")) as f:
        g = json.load(f)
    worst = 0
    for rec in [{"in": g["gen"]["miller"], "out": g["gen"]["final_exp"]}] + g["final_exp"]:
        got, mx = M.final_exp(F.from_flat12(_ints(bytes.fromhex(rec["in"]))))
        assert _bytes(F.to_flat12(got)).hex() == rec["out"]
        worst = max(worst, mx)
    assert worst < 16                                  # |value| of every quad after every step, in units of q
    got, _ = M.final_exp(F.from_flat12([0] * 12))
    assert all(c == (0, 0) for c in got)               # 0 -> 0 (fields_t.py:47-55)


def test_steps_against_the_six_lane_model():
    rnd = random.Random(23)
    f = F.from_flat12([rnd.randrange(F.Q) for _ in range(12)])
    g = F.from_flat12([rnd.randrange(F.Q) for _ in range(12)])
    w = M.Wave()
    w.load_acc(f)
    w.load_g(g)
    w.step("MUL")
    assert w.value() == F.mul_dense(f, g)
    w.load_acc(f)
    w.step("CONJ")
    assert w.value() == F.conj6(f)
    for j, i in enumerate(F.FROB_POW):
        w.load_acc(f)
        w.G = M.frob_constants(j)
        w.step("FROBC" if i % 2 else "FROB")
        assert w.value() == F.frob(f, i)
    t = F.mul_dense(F.conj6(f), F.inverse(f))
    t = F.mul_dense(F.frob(t, 2), t)                   # in the cyclotomic subgroup
    w.load_acc(t)
    for _ in range(40):                                # a run of squarings: the limbs stay normalised, the values bounded
        w.step("CSQ")
        t = F.cyc_sqr(t)
        assert w.value() == t
    assert w.max_abs < 16


def test_generated_tables_are_current():
    from vmgen import gen_fexpw
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        p = gen_fexpw.generate(os.path.join(td, "t.h"))
        assert open(p).read() == open(os.path.join(root, "python-bls_amd", "csrc", "fexpw_tables_gfx950.h")).read()
