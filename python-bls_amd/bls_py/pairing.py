"""Pairing entry points with the reference's signatures (pairing.py:16-92); the computation is
the HIP engine's (bls_py/backend.py).

As in the reference the objects are unwrapped to plain coordinates plus the infinity flag, and
the flag travels with them: the Miller loop ignores P's flag and lets Q's skip the chord updates
(fields_t.py:676-677, 1091-1111), whatever the coordinates are."""
from . import backend
from . import hostmath as H
from .ec import AffinePoint, default_ec
from .fields import Fq, Fq2, Fq12


def _g1(P):
    if type(P) is not AffinePoint or P.FE is not Fq:
        raise Exception("invalid elements")
    return H.fq_bytes(int(P.x)) + H.fq_bytes(int(P.y))


def _g2(Q):
    if type(Q) is not AffinePoint or Q.FE is not Fq2:
        raise Exception("invalid elements")
    return b"".join(H.fq_bytes(int(c)) for c in (Q.x[0], Q.x[1], Q.y[0], Q.y[1]))


def _flags(Ps, Qs):
    f = bytes(b for p, qq in zip(Ps, Qs) for b in (int(bool(p.infinity)), int(bool(qq.infinity))))
    return f if any(f) else None


def double_line_eval(R, P, ec=default_ec):
    """The tangent at R evaluated at P (pairing.py:16-29 -> fq2_double_line_eval)."""
    return Fq12.from_bytes(ec.q, backend.get().line_eval_batch(_g2(R), None, _g1(P), 1))


def add_line_eval(R, Q, P, ec=default_ec):
    """The line through R and Q evaluated at P (pairing.py:32-48 -> fq2_add_line_eval)."""
    return Fq12.from_bytes(ec.q, backend.get().line_eval_batch(_g2(R), _g2(Q), _g1(P), 1))


def miller_loop(P, Q, ec=default_ec):
    """fq_miller_loop(P, Q) (pairing.py:51-65): the reference's own value, before the final exponentiation."""
    return Fq12.from_bytes(ec.q, backend.get().miller_loop_batch(_g1(P), _g2(Q), 1, _flags([P], [Q])))


def final_exponentiation(element, ec=default_ec):
    return Fq12.from_bytes(ec.q, backend.get().final_exp(element.serialize()))


def ate_pairing(P, Q, ec=default_ec):
    return ate_pairing_multi([P], [Q], ec)


def ate_pairing_multi(Ps, Qs, ec=default_ec):
    """prod_i e(P_i, Q_i): one Miller loop per pair, one final exponentiation (pairing.py:84-92)."""
    Ps, Qs = list(Ps), list(Qs)
    if len(Ps) != len(Qs):
        raise Exception("invalid elements")
    g1 = b"".join(_g1(p) for p in Ps)
    g2 = b"".join(_g2(qq) for qq in Qs)
    return Fq12.from_bytes(ec.q, backend.get().pairing_multi(g1, g2, len(Ps), _flags(Ps, Qs)))
