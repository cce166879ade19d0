"""Curve point objects with the reference's surface (ec.py:18-188, 394-399,
511-555): AffinePoint / JacobianPoint over Fq (G1) or Fq2 (G2), generators,
hash-to-curve.  Group arithmetic here is host-side integer code
(bls_py/hostmath.py); the pairing of these points runs on the GPU."""
from collections import namedtuple

from . import bls12381 as C
from . import hostmath as H
from .fields import Fq, Fq2
from .util import hash256, hash512

EC = namedtuple("EC", "q a b gx gy g2x g2y n h x k sqrt_n3 sqrt_n3m1o2")
default_ec = EC(C.q, Fq(C.q, 0), Fq(C.q, C.b), Fq(C.q, C.gx), Fq(C.q, C.gy), Fq2(C.q, C.g2x), Fq2(C.q, C.g2y),
                C.n, C.h, C.x, C.k, C.sqrt_n3, C.sqrt_n3m1o2)
default_ec_twist = EC(C.q, Fq2(C.q, 0, 0), Fq2(C.q, C.b_twist), Fq(C.q, C.gx), Fq(C.q, C.gy), Fq2(C.q, C.g2x),
                      Fq2(C.q, C.g2y), C.n, C.h_twist, C.x, C.k, C.sqrt_n3, C.sqrt_n3m1o2)


def _field_of(x):
    if type(x) is Fq:
        return H.F1
    if type(x) is Fq2:
        return H.F2
    raise Exception("x,y should be field elements")


def _raw(F, e):
    return e.Z if F is H.F1 else e.ZT


def _wrap(F, v):
    return Fq(C.q, v) if F is H.F1 else Fq2(C.q, v)


def _scalar(c):
    if isinstance(c, Fq):
        return c.Z
    if isinstance(c, int):
        return c
    raise ValueError("Error, must be int or Fq")


class AffinePoint:
    def __init__(self, x, y, infinity, ec=default_ec):
        if type(x) is not type(y):
            raise Exception("x,y should be field elements")
        self._F = _field_of(x)
        self.FE = type(x)
        self.x, self.y, self.infinity, self.ec = x, y, bool(infinity), ec

    # ---- conversions to/from the plain-integer form used by hostmath
    def _aff(self):
        return None if self.infinity else (_raw(self._F, self.x), _raw(self._F, self.y))

    @staticmethod
    def _from(F, A, ec=None):
        ec = ec or (default_ec if F is H.F1 else default_ec_twist)
        if A is None:
            return AffinePoint(_wrap(F, F.zero), _wrap(F, F.zero), True, ec)
        return AffinePoint(_wrap(F, A[0]), _wrap(F, A[1]), False, ec)

    def is_on_curve(self):
        return H.on_curve(self._F, self._aff())

    def to_jacobian(self):
        F = self._F
        return JacobianPoint(self.x, self.y, _wrap(F, F.one), self.infinity, self.ec)

    def negate(self):
        return AffinePoint(self.x, -self.y, self.infinity, self.ec)

    def __add__(self, other):
        if isinstance(other, int) and other == 0:
            return self
        if type(other) is not AffinePoint:
            raise Exception("Incorrect object")
        F = self._F
        J = H.jac_add(F, H.aff_to_jac(F, self._aff()), H.aff_to_jac(F, other._aff()))
        return AffinePoint._from(F, H.jac_to_affine(F, J), self.ec)

    __radd__ = __add__

    def __sub__(self, other):
        return self + other.negate()

    def __mul__(self, c):
        F = self._F
        J = H.jac_mul(F, H.aff_to_jac(F, self._aff()), _scalar(c))
        return AffinePoint._from(F, H.jac_to_affine(F, J), self.ec)

    __rmul__ = __mul__

    def __eq__(self, other):
        return (type(other) is AffinePoint and self.x == other.x and self.y == other.y
                and self.infinity == other.infinity)

    def __ne__(self, other):
        return not self.__eq__(other)

    def lex_gt_neg(self):
        return H._lex_gt_neg(self._F, _raw(self._F, self.y))

    def serialize(self):
        A = self._aff()
        if A is None:
            A = (self._F.zero, self._F.zero)
            out = bytearray(H.fq_bytes(0) if self._F is H.F1 else H.fq_bytes(0) * 2)
            return bytes(out)
        return H.g1_compress(A) if self._F is H.F1 else H.g2_compress(A)

    def __repr__(self):
        return "AffinePoint(x=%r, y=%r, i=%s)" % (self.x, self.y, self.infinity)

    def __deepcopy__(self, memo):
        return AffinePoint(self.x, self.y, self.infinity, self.ec)


class JacobianPoint:
    def __init__(self, x, y, z, infinity, ec=default_ec):
        self._F = _field_of(x)
        self.FE = type(x)
        self.x, self.y, self.z, self.infinity, self.ec = x, y, z, bool(infinity), ec

    def _jac(self):
        F = self._F
        return None if self.infinity else (_raw(F, self.x), _raw(F, self.y), _raw(F, self.z))

    @staticmethod
    def _from(F, J, ec=None):
        ec = ec or (default_ec if F is H.F1 else default_ec_twist)
        if J is None:
            return JacobianPoint(_wrap(F, F.one), _wrap(F, F.one), _wrap(F, F.zero), True, ec)
        return JacobianPoint(_wrap(F, J[0]), _wrap(F, J[1]), _wrap(F, J[2]), False, ec)

    def is_on_curve(self):
        return self.to_affine().is_on_curve()

    def to_affine(self):
        return AffinePoint._from(self._F, H.jac_to_affine(self._F, self._jac()), self.ec)

    def __add__(self, other):
        if isinstance(other, int) and other == 0:
            return self
        if type(other) is not JacobianPoint:
            raise ValueError("Incorrect object")
        return JacobianPoint._from(self._F, H.jac_add(self._F, self._jac(), other._jac()),
                                   other.ec if self.infinity else self.ec)

    __radd__ = __add__

    def __mul__(self, c):
        return JacobianPoint._from(self._F, H.jac_mul(self._F, self._jac(), _scalar(c)), self.ec)

    __rmul__ = __mul__

    def __eq__(self, other):
        return type(other) is JacobianPoint and self.to_affine() == other.to_affine()

    def __ne__(self, other):
        return not self.__eq__(other)

    def serialize(self):
        return self.to_affine().serialize()

    def __deepcopy__(self, memo):
        return JacobianPoint(self.x, self.y, self.z, self.infinity, self.ec)


def generator_Fq(ec=default_ec):
    return AffinePoint(ec.gx, ec.gy, False, ec)


def generator_Fq2(ec=default_ec_twist):
    return AffinePoint(ec.g2x, ec.g2y, False, ec)


def y_for_x(x, ec=default_ec, FE=Fq):
    if type(x) is not FE:
        x = FE(ec.q, x)
    F = _field_of(x)
    return [_wrap(F, y) for y in H.y_for_x(F, _raw(F, x))]


def sw_encode(t, ec=default_ec, FE=Fq):
    F = _field_of(t)
    A = H.sw_encode(F, _raw(F, t))
    if A is None:
        return JacobianPoint._from(F, None, ec)
    return AffinePoint._from(F, A, ec)


def _as_bytes(m):
    return m if isinstance(m, bytes) else m.encode("utf-8")


def hash_to_point_prehashed_Fq(m, ec=default_ec):
    return AffinePoint._from(H.F1, H.hash_to_g1_prehashed(_as_bytes(m), hash512), ec)


def hash_to_point_Fq(m, ec=default_ec):
    return hash_to_point_prehashed_Fq(hash256(m), ec)


def hash_to_point_prehashed_Fq2(m, ec=default_ec_twist):
    return AffinePoint._from(H.F2, H.hash_to_g2_prehashed(_as_bytes(m), hash512), ec)


def hash_to_points_prehashed_Fq2(ms, ec=default_ec_twist):
    """Batch form of hash_to_point_prehashed_Fq2: the SHA-256 chain on the host, then
    encodings, sum and cofactor clearing for all messages in one GPU call."""
    from . import backend
    ms = [_as_bytes(m) for m in ms]
    if not ms:
        return []
    prov = backend.get()
    if all(len(m) == 32 for m in ms) and hasattr(prov, "hash_to_g2"):
        out = prov.hash_to_g2(b"".join(ms))                 # 32-byte message hashes: SHA-256 chain on the GPU too
    else:
        out = prov.map_to_g2(b"".join(H.g2_hash_field_elements(m, hash512) for m in ms))
    return [AffinePoint._from(H.F2, H.g2_from_abi(out[192 * i:192 * (i + 1)]), ec) for i in range(len(ms))]


def hash_to_point_Fq2(m, ec=default_ec_twist):
    return hash_to_point_prehashed_Fq2(hash256(m), ec)
