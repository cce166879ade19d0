"""Drop-in for the hot functions of the reference's native module
`bls_py.fields_t_c` (extmod/bls_py/fields_t_c.pyx), backed by libblsgpu.so.

Same names, argument shapes and return types as the functions that
bls_py/fields_t.py:1256-1263 re-binds from the native module:

    fq_ate_pairing_multi(Ps, Qs) -> 12-tuple of ints   (fields_t.py:1114-1121)
    fq12_final_exp(t)            -> 12-tuple of ints   (fields_t.py:1124-1128)

Ps = tuple of (x, y, inf), Qs = tuple of ((x0, x1), (y0, y1), inf), ints in
[0, q).  Like the reference, P's flag is not consulted; infinity is the (0,0)
coordinate pair that AffinePoint / to_affine produce (fields_t.py:609-622).
A flagged Q with NON-zero coordinates (never produced by the reference's own
objects) is rejected with ValueError instead of being mis-evaluated.

There is no CPU fallback: if the GPU library is missing these raise.
"""
from . import _native

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def _fq(v):
    return int(v % Q).to_bytes(48, "big")


def _unpack12(b):
    return tuple(int.from_bytes(b[48 * i:48 * (i + 1)], "big") for i in range(12))


def fq_ate_pairing_multi(Ps, Qs, device=0):
    n = len(Qs)
    g1 = bytearray()
    g2 = bytearray()
    for i in range(n):
        px, py, _pinf = Ps[i]
        (x0, x1), (y0, y1), qinf = Qs[i]
        if qinf and (x0 or x1 or y0 or y1):
            raise ValueError("Q[%d] is flagged infinite but has non-zero coordinates" % i)
        g1 += _fq(px) + _fq(py)
        g2 += _fq(x0) + _fq(x1) + _fq(y0) + _fq(y1)
    return _unpack12(_native.engine(device).pairing_multi(bytes(g1), bytes(g2), n))


def fq12_final_exp(t_x, device=0):
    return _unpack12(_native.engine(device).final_exp(b"".join(_fq(v) for v in t_x)))
