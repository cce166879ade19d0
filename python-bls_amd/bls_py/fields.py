"""Thin host-side field element shells (API surface of the reference's fields.py
that the scheme code and its callers touch).  Arithmetic on the verify hot path
does not go through these classes; it runs in libblsgpu.so."""
from . import hostmath as H
from .bls12381 import q as bls12381_q


class Fq:
    """Integer modulo Q (also used with Q = group order, like the reference)."""
    extension = 1
    __slots__ = ("Q", "Z")

    def __init__(self, Q, X):
        if isinstance(X, Fq):
            X = X.Z
        elif not isinstance(X, int):
            raise TypeError("Fq must be constructed from Fq or int")
        self.Q, self.Z = Q, X % Q

    @staticmethod
    def zero(Q):
        return Fq(Q, 0)

    @staticmethod
    def one(Q):
        return Fq(Q, 1)

    @classmethod
    def from_fq(cls, Q, fq):
        return fq

    def _v(self, o):
        if isinstance(o, Fq):
            return o.Z
        if isinstance(o, int):
            return o
        return None

    def __add__(self, o):
        v = self._v(o)
        return NotImplemented if v is None else Fq(self.Q, self.Z + v)

    __radd__ = __add__

    def __sub__(self, o):
        v = self._v(o)
        return NotImplemented if v is None else Fq(self.Q, self.Z - v)

    def __rsub__(self, o):
        v = self._v(o)
        return NotImplemented if v is None else Fq(self.Q, v - self.Z)

    def __mul__(self, o):
        v = self._v(o)
        return NotImplemented if v is None else Fq(self.Q, self.Z * v)

    __rmul__ = __mul__

    def __neg__(self):
        return Fq(self.Q, -self.Z)

    def __invert__(self):
        return Fq(self.Q, pow(self.Z, self.Q - 2, self.Q))

    def __floordiv__(self, o):
        v = self._v(o)
        return NotImplemented if v is None else Fq(self.Q, self.Z * pow(v, self.Q - 2, self.Q))

    __truediv__ = __floordiv__

    def __pow__(self, e):
        return Fq(self.Q, pow(self.Z, e, self.Q))

    def __eq__(self, o):
        v = self._v(o)
        return v is not None and (v % self.Q) == self.Z

    def __ne__(self, o):
        return not self.__eq__(o)

    def __lt__(self, o):
        return self.Z < self._v(o)

    def __gt__(self, o):
        return self.Z > self._v(o)

    def __le__(self, o):
        return self.Z <= self._v(o)

    def __ge__(self, o):
        return self.Z >= self._v(o)

    def __hash__(self):
        return hash((self.Q, self.Z))

    def __int__(self):
        return self.Z

    def __index__(self):
        return self.Z

    def __mod__(self, m):
        return self.Z % m

    def __repr__(self):
        return "Fq(Q, %s)" % hex(self.Z)

    def __deepcopy__(self, memo):
        return Fq(self.Q, self.Z)

    def qi_power(self, i):
        return self

    def serialize(self):
        return self.Z.to_bytes(48, "big")

    def modsqrt(self):
        if self.Q != bls12381_q:
            raise ValueError("modsqrt is only provided for the base field")
        return Fq(self.Q, H.fq_sqrt(self.Z))


class Fq2:
    """c0 + c1 u, u^2 = -1."""
    extension = 2
    __slots__ = ("Q", "ZT")

    def __init__(self, Q, *args):
        if Q != bls12381_q:
            raise TypeError("only the BLS12-381 base field is supported")
        if len(args) == 1:
            a, b = args[0]
        else:
            a, b = args
        self.Q = Q
        self.ZT = (int(a) % Q, int(b) % Q)

    @staticmethod
    def zero(Q):
        return Fq2(Q, 0, 0)

    @staticmethod
    def one(Q):
        return Fq2(Q, 1, 0)

    @classmethod
    def from_fq(cls, Q, fq):
        return Fq2(Q, int(fq), 0)

    def _t(self, o):
        if isinstance(o, Fq2):
            return o.ZT
        if isinstance(o, (Fq, int)):
            return (int(o) % self.Q, 0)
        return None

    def __add__(self, o):
        t = self._t(o)
        return NotImplemented if t is None else Fq2(self.Q, H.f2_add(self.ZT, t))

    __radd__ = __add__

    def __sub__(self, o):
        t = self._t(o)
        return NotImplemented if t is None else Fq2(self.Q, H.f2_sub(self.ZT, t))

    def __rsub__(self, o):
        t = self._t(o)
        return NotImplemented if t is None else Fq2(self.Q, H.f2_sub(t, self.ZT))

    def __mul__(self, o):
        t = self._t(o)
        return NotImplemented if t is None else Fq2(self.Q, H.f2_mul(self.ZT, t))

    __rmul__ = __mul__

    def __neg__(self):
        return Fq2(self.Q, H.f2_neg(self.ZT))

    def __invert__(self):
        return Fq2(self.Q, H.f2_inv(self.ZT))

    def __floordiv__(self, o):
        t = self._t(o)
        return NotImplemented if t is None else Fq2(self.Q, H.f2_mul(self.ZT, H.f2_inv(t)))

    __truediv__ = __floordiv__

    def __pow__(self, e):
        return Fq2(self.Q, H.f2_pow(self.ZT, e))

    def __eq__(self, o):
        t = self._t(o)
        return t is not None and t == self.ZT

    def __ne__(self, o):
        return not self.__eq__(o)

    def __hash__(self):
        return hash(self.ZT)

    def __getitem__(self, i):
        return Fq(self.Q, self.ZT[i])

    def __iter__(self):
        return iter((Fq(self.Q, self.ZT[0]), Fq(self.Q, self.ZT[1])))

    def __repr__(self):
        return "Fq2(Q, %s, %s)" % (hex(self.ZT[0]), hex(self.ZT[1]))

    def __deepcopy__(self, memo):
        return Fq2(self.Q, self.ZT)

    def qi_power(self, i):
        return self if i % 2 == 0 else Fq2(self.Q, H.f2_conj(self.ZT))

    def serialize(self):
        return self.ZT[0].to_bytes(48, "big") + self.ZT[1].to_bytes(48, "big")

    def modsqrt(self):
        return Fq2(self.Q, H.f2_sqrt(self.ZT))


class Fq12:
    """Value holder for pairing results: the 12 coefficients in the reference's
    flat ZT order (fields.py:624-629).  Products of Fq12 elements are computed
    on the GPU, not here."""
    extension = 12
    __slots__ = ("Q", "ZT")

    def __init__(self, Q, zt):
        zt = tuple(int(v) % Q for v in zt)
        if len(zt) != 12:
            raise TypeError("Fq12 needs 12 coefficients")
        self.Q, self.ZT = Q, zt

    @staticmethod
    def one(Q):
        return Fq12(Q, (1,) + (0,) * 11)

    @staticmethod
    def zero(Q):
        return Fq12(Q, (0,) * 12)

    def __eq__(self, o):
        return isinstance(o, Fq12) and o.ZT == self.ZT

    def __ne__(self, o):
        return not self.__eq__(o)

    def __hash__(self):
        return hash(self.ZT)

    def __repr__(self):
        return "Fq12(Q, %s)" % ", ".join(hex(v) for v in self.ZT)

    def serialize(self):
        return b"".join(v.to_bytes(48, "big") for v in self.ZT)

    @staticmethod
    def from_bytes(Q, b):
        return Fq12(Q, tuple(int.from_bytes(b[48 * i:48 * (i + 1)], "big") for i in range(12)))
