"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE -- see bls381_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libbls381_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        c_p = ctypes.c_char_p
        L.oracle_miller_loop.argtypes = [c_p, c_p, ctypes.c_int, c_p]
        L.oracle_final_exp.argtypes = [c_p, c_p]
        L.oracle_line_eval.argtypes = [c_p, c_p, c_p, c_p]
        L.oracle_pairing_multi.argtypes = [c_p, c_p, c_p, ctypes.c_size_t, c_p]
        L.oracle_pairing_multi_mt.argtypes = [c_p, c_p, c_p, ctypes.c_size_t, ctypes.c_int, c_p]
        L.oracle_pairing_multi_fast.argtypes = [c_p, c_p, ctypes.c_size_t, ctypes.c_int, c_p]
        L.oracle_field_op.argtypes = [ctypes.c_int, ctypes.c_int, c_p, c_p, c_p]
        L.oracle_qi_pow.argtypes = [ctypes.c_int, c_p, ctypes.c_int, c_p]
        L.oracle_fq12_pow.argtypes = [c_p, c_p, ctypes.c_size_t, c_p]
        L.oracle_g1_msm.argtypes = [c_p, c_p, ctypes.c_size_t, ctypes.c_size_t, c_p, c_p]
        L.oracle_g2_msm.argtypes = [c_p, c_p, ctypes.c_size_t, ctypes.c_size_t, c_p, c_p]
        L.oracle_version.restype = ctypes.c_char_p
        _LIB = L
    return _LIB


def _out(n):
    return ctypes.create_string_buffer(n)


def miller_loop(g1: bytes, g2: bytes, qinf: bool = False) -> bytes:
    o = _out(576)
    assert lib().oracle_miller_loop(g1, g2, int(qinf), o) == 0
    return o.raw


def line_eval(r: bytes, q, p: bytes) -> bytes:
    """fq2_double_line_eval(R, P) when q is None, else fq2_add_line_eval(R, Q, P)."""
    o = _out(576)
    assert lib().oracle_line_eval(r, q, p, o) == 0
    return o.raw


def final_exp(x: bytes) -> bytes:
    o = _out(576)
    assert lib().oracle_final_exp(x, o) == 0
    return o.raw


def pairing_multi(g1: bytes, g2: bytes, n: int, threads: int = 1, inf: bytes = None) -> bytes:
    """inf: n x (pinf, qinf) bytes or None (all False)."""
    assert len(g1) == 96 * n and len(g2) == 192 * n
    assert inf is None or len(inf) == 2 * n
    o = _out(576)
    assert lib().oracle_pairing_multi_mt(g1, g2, inf, n, threads, o) == 0
    return o.raw


def pairing_multi_fast(g1: bytes, g2: bytes, n: int, threads: int = 1) -> bytes:
    """the fast-algorithm CPU flavour (projective twist point, sparse lines, one shared squaring per thread): ordinary
    pairs only -- a timing baseline, same result as pairing_multi for them"""
    assert len(g1) == 96 * n and len(g2) == 192 * n
    o = _out(576)
    assert lib().oracle_pairing_multi_fast(g1, g2, n, threads, o) == 0
    return o.raw


_OPS = {"add": 0, "sub": 1, "mul": 2, "neg": 3, "inv": 4}


def field_op(degree: int, op: str, a: bytes, b: bytes = None) -> bytes:
    o = _out(48 * degree)
    assert lib().oracle_field_op(degree, _OPS[op], a, b if b is not None else a, o) == 0
    return o.raw


def qi_pow(degree: int, a: bytes, i: int) -> bytes:
    o = _out(48 * degree)
    assert lib().oracle_qi_pow(degree, a, i, o) == 0
    return o.raw


def fq12_pow(x: bytes, e: int) -> bytes:
    eb = e.to_bytes(max(1, (e.bit_length() + 7) // 8), "big")
    o = _out(576)
    assert lib().oracle_fq12_pow(x, eb, len(eb), o) == 0
    return o.raw


def g1_msm(pts: bytes, scalars, n: int, slen: int = 32):
    """scalars: list of ints or None (plain sum). Returns (96 bytes, inf)."""
    sb = None if scalars is None else b"".join(int(s).to_bytes(slen, "big") for s in scalars)
    o = _out(96)
    inf = _out(1)
    assert lib().oracle_g1_msm(pts, sb, slen, n, o, inf) == 0
    return o.raw, bool(inf.raw[0])


def g2_msm(pts: bytes, scalars, n: int, slen: int = 32):
    sb = None if scalars is None else b"".join(int(s).to_bytes(slen, "big") for s in scalars)
    o = _out(192)
    inf = _out(1)
    assert lib().oracle_g2_msm(pts, sb, slen, n, o, inf) == 0
    return o.raw, bool(inf.raw[0])
