"""Key types (keys.py:17-164 of the reference): PublicKey = G1 point with a
48-byte compressed form, PrivateKey = scalar mod n."""
from copy import deepcopy
from random import SystemRandom

from . import hostmath as H
from .aggregation_info import AggregationInfo
from .bls12381 import n as GROUP_ORDER
from .ec import (JacobianPoint, default_ec, default_ec_twist, generator_Fq, hash_to_point_Fq2,
                 hash_to_point_prehashed_Fq2)
from .fields import Fq
from .signature import Signature
from .util import hash256, hmac256

RNG = SystemRandom()


class PublicKey:
    PUBLIC_KEY_SIZE = 48

    def __init__(self, value):
        self.value = value
        self._ser = None

    @staticmethod
    def from_bytes(buffer):
        A = H.g1_decompress(bytes(buffer))
        return PublicKey(JacobianPoint._from(H.F1, H.aff_to_jac(H.F1, A), default_ec))

    @staticmethod
    def from_bytes_batch(buffers):
        """[PublicKey.from_bytes(b) for b in buffers] with the square roots of all keys
        in one GPU call (blsgpu_g1_decompress); ValueError on the first bad encoding,
        as from_bytes raises."""
        from . import backend
        buffers = [bytes(b) for b in buffers]
        if any(len(b) != PublicKey.PUBLIC_KEY_SIZE for b in buffers):
            raise ValueError("public keys are %d bytes" % PublicKey.PUBLIC_KEY_SIZE)
        if not buffers:
            return []
        out, ok = backend.get().g1_decompress(b"".join(buffers))
        if not all(ok):
            raise ValueError("No y for point x")
        return [PublicKey(JacobianPoint._from(H.F1, H.aff_to_jac(H.F1, H.g1_from_abi(out[96 * i:96 * (i + 1)])), default_ec))
                for i in range(len(buffers))]

    @staticmethod
    def from_g1(g1_el):
        assert type(g1_el) is JacobianPoint
        return PublicKey(g1_el)

    def serialize(self):
        if self._ser is None:
            self._ser = self.value.serialize()
        return self._ser

    def get_fingerprint(self):
        return int.from_bytes(hash256(self.serialize())[:4], "big")

    def size(self):
        return self.PUBLIC_KEY_SIZE

    def __eq__(self, other):
        return self.serialize() == other.serialize()

    def __hash__(self):
        return int.from_bytes(self.serialize(), "big")

    def __lt__(self, other):
        return self.serialize() < other.serialize()

    def __repr__(self):
        return "PublicKey(%s)" % self.serialize().hex()

    def __deepcopy__(self, memo):
        return PublicKey.from_g1(deepcopy(self.value, memo))


class PrivateKey:
    PRIVATE_KEY_SIZE = 32

    def __init__(self, value):
        self.value = int(value)

    @staticmethod
    def from_bytes(buffer):
        return PrivateKey(int.from_bytes(buffer, "big"))

    @staticmethod
    def from_seed(seed):
        return PrivateKey(int.from_bytes(hmac256(seed, b"BLS private key seed"), "big") % GROUP_ORDER)

    @staticmethod
    def new_threshold(T, N):
        """Joint-Feldman dealing (keys.py:92-117): a random degree T-1 polynomial,
        commitments g1*c_i and the N fragments P(1..N)."""
        assert 1 <= T <= N
        g1 = generator_Fq()
        poly = [Fq(GROUP_ORDER, RNG.randint(1, GROUP_ORDER - 1)) for _ in range(T)]
        commitments = [g1 * c for c in poly]
        fragments = [sum(c * pow(x, i, GROUP_ORDER) for i, c in enumerate(poly)) for x in range(1, N + 1)]
        return PrivateKey(poly[0]), commitments, fragments

    def get_public_key(self):
        # sk G1 is one engine call (a scalar multiplication: ~2 ms with its round trip); the point is kept, a NEW PublicKey
        # object comes back every time as in keys.py:104-105 (callers that change `value` would lose the cache: __setattr__ below)
        pt = self.__dict__.get("_pk_point")
        if pt is None:
            pt = (self.value * generator_Fq()).to_jacobian()
            self.__dict__["_pk_point"] = pt
        return PublicKey.from_g1(pt)

    def __setattr__(self, name, v):
        if name == "value":
            self.__dict__.pop("_pk_point", None)
        object.__setattr__(self, name, v)

    def sign(self, m):
        # (one key, one message through the batched steps: three engine calls instead of a dozen -- 15.6 -> ~4 ms;
        # the same objects as keys.py:123-126 builds)
        return PrivateKey.sign_batch([self], [m])[0]

    @staticmethod
    def sign_batch(private_keys, messages):
        """[sk.sign(m) for sk, m in zip(private_keys, messages)] with the three heavy steps
        batched on the GPU: public keys sk*G1 (one group sum per key), H(m) (hash to G2)
        and sk*H(m) (one G2 group sum per message).  Same objects as keys.py:123-126 builds."""
        from . import backend
        from .ec import hash_to_points_prehashed_Fq2
        from .util import hash256
        sks = list(private_keys)
        hashes = [hash256(m) for m in messages]
        return PrivateKey._sign_hashes(sks, hashes)

    @staticmethod
    def _sign_hashes(sks, hashes):
        from . import backend
        from .ec import hash_to_points_prehashed_Fq2
        if len(sks) != len(hashes):
            raise ValueError("one message per key")
        if not sks:
            return []
        n = len(sks)
        prov = backend.get()
        g1 = H.g1_affine_bytes(H.G1_GEN)
        pk_bytes, pk_inf = prov.g1_msm(g1 * n, [sk.value for sk in sks], 1, n)
        Hm = hash_to_points_prehashed_Fq2(hashes)
        pts = b"".join(H.g2_affine_bytes(q._aff()) for q in Hm)
        sig_bytes, sig_inf = prov.g2_msm(pts, [sk.value for sk in sks], 1, n)
        out = []
        for i in range(n):
            pk = PublicKey.from_g1(JacobianPoint._from(
                H.F1, None if pk_inf[i] else H.aff_to_jac(H.F1, H.g1_from_abi(pk_bytes[96 * i:96 * (i + 1)])), default_ec))
            sig = JacobianPoint._from(
                H.F2, None if sig_inf[i] else H.aff_to_jac(H.F2, H.g2_from_abi(sig_bytes[192 * i:192 * (i + 1)])), default_ec_twist)
            out.append(Signature.from_g2(sig, AggregationInfo.from_msg_hash(pk, hashes[i])))
        return out

    def sign_prehashed(self, h):
        return PrivateKey._sign_hashes([self], [h])[0]

    def sign_threshold(self, m, player, players):
        from .threshold import Threshold
        assert player in players
        r = hash_to_point_Fq2(m).to_jacobian()
        lam = Threshold.lagrange_coeffs_at_zero(players)[players.index(player)]
        return Signature.from_g2(self.value * (r * lam))

    def serialize(self):
        return self.value.to_bytes(self.PRIVATE_KEY_SIZE, "big")

    def size(self):
        return self.PRIVATE_KEY_SIZE

    def __lt__(self, other):
        return self.value < other.value

    def __eq__(self, other):
        return self.value == other.value

    def __hash__(self):
        return self.value

    def __repr__(self):
        return "PrivateKey(%s)" % hex(self.value)
