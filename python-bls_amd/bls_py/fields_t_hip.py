"""Drop-in for the pairing functions of the reference's native module
`bls_py.fields_t_c` (extmod/bls_py/fields_t_c.pyx), backed by libblsgpu.so.

Same names, argument shapes and return types as the functions that
bls_py/fields_t.py:1256-1263 re-binds from the native module:

    fq_ate_pairing_multi(Ps, Qs)                    -> 12-tuple of ints   (fields_t.py:1114-1121)
    fq_miller_loop(px, py, pinf, qx, qy, qinf)      -> 12-tuple of ints   (fields_t.py:1091-1111)
    fq12_final_exp(t)                               -> 12-tuple of ints   (fields_t.py:1124-1128)
    fq2_double_line_eval(rx, ry, px, py)            -> 12-tuple of ints   (fields_t.py:1035-1049)
    fq2_add_line_eval(rx, ry, qx, qy, px, py)       -> 12-tuple of ints   (fields_t.py:1052-1078)
    fq12_mul(a, b), fq12_add(a, b), fq12_invert(a), fq12_pow(a, e)        (fields_t.py:321-554)

Ps = tuple of (x, y, inf), Qs = tuple of ((x0, x1), (y0, y1), inf), ints in [0, q).  The flags
are passed on as they are: like the reference, the engine never reads P's, and a flagged Q skips
the chord updates whatever its coordinates are (fields_t.py:676-677).

There is no CPU fallback: if the GPU library is missing these raise.
"""
from . import _native

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def _fq(v):
    return int(v % Q).to_bytes(48, "big")


def _fq2(t):
    return _fq(t[0]) + _fq(t[1])


def _unpack12(b):
    return tuple(int.from_bytes(b[48 * i:48 * (i + 1)], "big") for i in range(12))


def _pack(Ps, Qs):
    n = len(Qs)
    g1 = b"".join(_fq(Ps[i][0]) + _fq(Ps[i][1]) for i in range(n))
    g2 = b"".join(_fq2(Qs[i][0]) + _fq2(Qs[i][1]) for i in range(n))
    inf = bytes(b for i in range(n) for b in (int(bool(Ps[i][2])), int(bool(Qs[i][2]))))
    return g1, g2, (inf if any(inf) else None), n


def fq_ate_pairing_multi(Ps, Qs, device=0):
    g1, g2, inf, n = _pack(Ps, Qs)
    return _unpack12(_native.engine(device).pairing_multi(g1, g2, n, inf))


def fq_miller_loop(px, py, pinf, qx_t, qy_t, qinf, device=0):
    g1, g2, inf, _ = _pack(((px, py, pinf),), ((qx_t, qy_t, qinf),))
    return _unpack12(_native.engine(device).miller_loop_batch(g1, g2, 1, inf))


def fq12_final_exp(t_x, device=0):
    return _unpack12(_native.engine(device).final_exp(b"".join(_fq(v) for v in t_x)))


def fq2_double_line_eval(rx_t, ry_t, px, py, device=0):
    return _unpack12(_native.engine(device).line_eval_batch(_fq2(rx_t) + _fq2(ry_t), None, _fq(px) + _fq(py), 1))


def fq2_add_line_eval(rx_t, ry_t, qx_t, qy_t, px, py, device=0):
    return _unpack12(_native.engine(device).line_eval_batch(_fq2(rx_t) + _fq2(ry_t), _fq2(qx_t) + _fq2(qy_t),
                                                            _fq(px) + _fq(py), 1))


def _fq12(t):
    return b"".join(_fq(v) for v in t)


def fq12_mul(a, b, device=0):
    return _unpack12(_native.engine(device).fq12_op("mul", _fq12(a), _fq12(b)))


def fq12_add(a, b, device=0):
    return _unpack12(_native.engine(device).fq12_op("add", _fq12(a), _fq12(b)))


def fq12_invert(a, device=0):
    return _unpack12(_native.engine(device).fq12_op("inv", _fq12(a)))


def fq12_pow(a, e, device=0):
    return _unpack12(_native.engine(device).fq12_pow(_fq12(a), e))
