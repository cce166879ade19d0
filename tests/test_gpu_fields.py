"""SURVEY 8c F-FIELD on the GPU: the reference's own field KATs (bls_py/tdata.py operands and
results, captured in tests/golden/fields.json) through blsgpu_fq12_op_batch / blsgpu_fq12_pow_batch
-- the Montgomery product, the linear rounds, the inversion and the tower formulas of the HIP
engine checked directly, not only through pairings.  Fq, Fq2 and Fq6 elements ride as Fq12
elements whose other coefficients are zero.  Needs an MI355X."""
import pytest

pytestmark = pytest.mark.gpu
DEG = {"1": 1, "2": 2, "6": 6, "12": 12}


def emb(x: bytes, d: int) -> bytes:
    assert len(x) == 48 * d
    return x + bytes(48 * (12 - d))


def test_field_kats_on_gpu(engine, golden):
    f = golden("fields.json")
    for key, d in DEG.items():
        rec = f[key]
        ops = [emb(bytes.fromhex(x), d) for x in rec["operands"]]
        for op in ("add", "mul", "sub"):
            a = b"".join(ops[e["i"]] for e in rec[op])
            b = b"".join(ops[e["j"]] for e in rec[op])
            got = engine.fq12_op(op, a, b)
            for k, e in enumerate(rec[op]):
                assert got[576 * k:576 * (k + 1)] == emb(bytes.fromhex(e["r"]), d), (d, op, e["i"], e["j"])
        neg = engine.fq12_op("neg", b"".join(ops))
        inv = engine.fq12_op("inv", b"".join(ops))
        for i in range(4):
            assert neg[576 * i:576 * (i + 1)] == emb(bytes.fromhex(rec["neg"][i]), d), (d, "neg", i)
            assert inv[576 * i:576 * (i + 1)] == emb(bytes.fromhex(rec["inv"][i]), d), (d, "inv", i)
        assert engine.fq12_op("inv", inv) == b"".join(ops)           # round trip, tests.py:49-52
    assert engine.fq12_op("inv", bytes(576)) == bytes(576)                 # 0^-1 := 0, fields_t.py:47-55
    assert engine.fq12_op("mul", b"", b"") == b""


def test_fq12_pow_on_gpu(engine, golden, oracle):
    f = golden("fields.json")
    x = bytes.fromhex(f["12"]["operands"][1])
    for e in f["fq12_pow"]:
        assert engine.fq12_pow(x, int(e["e"], 16)).hex() == e["r"], e["e"]
    y = bytes.fromhex(f["12"]["operands"][2])
    e = 0x1234567890abcdef1234567890abcdef
    assert engine.fq12_pow(x + y, e) == oracle.fq12_pow(x, e) + oracle.fq12_pow(y, e)
