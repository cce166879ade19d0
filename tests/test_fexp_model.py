"""CPU: the integer model of the six-lanes-per-result final exponentiation (vmgen/fexp_model.py -- formulas, lane
operand tables and the op script of csrc/blsgpu_fexp.hip) against the reference's vectors and the oracle."""
import json
import os
import random

from conftest import GOLDEN
from vmgen import fexp_model as F


def _ints(b):
    return [int.from_bytes(b[48 * i:48 * (i + 1)], "big") for i in range(12)]


def _bytes(flat):
    return b"".join(x.to_bytes(48, "big") for x in flat)


def test_script_equals_reference_vectors():
    with open(os.path.join(GOLDEN, "pairing.json")) as f:
        g = json.load(f)
    for rec in [{"in": g["gen"]["miller"], "out": g["gen"]["final_exp"]}] + g["final_exp"]:
        f12 = F.from_flat12(_ints(bytes.fromhex(rec["in"])))
        assert _bytes(F.to_flat12(F.run_script(f12))).hex() == rec["out"]
        assert F.run_script(f12) == F.final_exp(f12)
    assert all(c == (0, 0) for c in F.run_script(F.from_flat12([0] * 12)))       # 0 -> 0 (fields_t.py:47-55)


def test_script_equals_oracle_on_random_elements(oracle):
    rnd = random.Random(7)
    for _ in range(3):
        v = [rnd.randrange(F.Q) for _ in range(12)]
        assert _bytes(F.to_flat12(F.run_script(F.from_flat12(v)))) == oracle.final_exp(_bytes(v))


def test_pieces():
    rnd = random.Random(3)
    f = F.from_flat12([rnd.randrange(F.Q) for _ in range(12)])
    assert F.mul_dense(f, F.inverse(f)) == F.one6()
    t = F.mul_dense(F.conj6(f), F.inverse(f))
    t = F.mul_dense(F.frob(t, 2), t)                                 # in the cyclotomic subgroup
    assert F.cyc_sqr(t) == F.mul_dense(t, t) == F.cyc_sqr_lane_forms(t)
    assert F.frob(F.frob(f, 2), 2) == F.frob(f, 4)
    x = f
    for _ in range(6):
        x = F.frob(x, 2)
    assert x == f and F.frob(F.frob(F.frob(f, 2), 2), 2) == F.conj6(f)      # q^12 = identity, q^6 = conjugation


def test_generated_tables_are_current():
    from vmgen import gen_fexp
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        p = gen_fexp.generate(os.path.join(td, "t.h"))
        assert open(p).read() == open(os.path.join(root, "python-bls_amd", "csrc", "fexp_tables_gfx950.h")).read()
