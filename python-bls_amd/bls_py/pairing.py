"""Pairing entry points with the reference's signatures (pairing.py:68-92);
the computation is the HIP engine's (bls_py/backend.py)."""
from . import backend
from . import hostmath as H
from .ec import AffinePoint, default_ec
from .fields import Fq, Fq2, Fq12


def _check(Ps, Qs):
    if len(Ps) != len(Qs):
        raise Exception("invalid elements")
    for p, qq in zip(Ps, Qs):
        if type(p) is not AffinePoint or type(qq) is not AffinePoint or p.FE is not Fq or qq.FE is not Fq2:
            raise Exception("invalid elements")


def ate_pairing_multi(Ps, Qs, ec=default_ec):
    """prod_i e(P_i, Q_i): one Miller loop per pair, one final exponentiation."""
    Ps, Qs = list(Ps), list(Qs)
    _check(Ps, Qs)
    g1 = b"".join(H.g1_affine_bytes(p._aff()) for p in Ps)
    g2 = b"".join(H.g2_affine_bytes(qq._aff()) for qq in Qs)
    return Fq12.from_bytes(ec.q, backend.get().pairing_multi(g1, g2, len(Ps)))


def ate_pairing(P, Q, ec=default_ec):
    return ate_pairing_multi([P], [Q], ec)


def final_exponentiation(element, ec=default_ec):
    return Fq12.from_bytes(ec.q, backend.get().final_exp(element.serialize()))
