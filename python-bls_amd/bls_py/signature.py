"""Signatures: G2 points, 96-byte compressed form (signature.py:21-38, 120-121)."""
from copy import deepcopy

from . import hostmath as H
from .bls12381 import n as GROUP_ORDER
from .ec import JacobianPoint, default_ec_twist


class Signature:
    SIGNATURE_SIZE = 96

    def __init__(self, value, aggregation_info=None):
        self.value = value
        self.aggregation_info = aggregation_info

    @staticmethod
    def from_bytes(buffer, aggregation_info=None):
        A = H.g2_decompress(bytes(buffer))
        return Signature(JacobianPoint._from(H.F2, H.aff_to_jac(H.F2, A), default_ec_twist), aggregation_info)

    @staticmethod
    def from_bytes_batch(buffers, aggregation_infos=None):
        """[Signature.from_bytes(b, info) ...] with all square roots in one GPU call
        (blsgpu_g2_decompress); ValueError on the first bad encoding."""
        from . import backend
        buffers = [bytes(b) for b in buffers]
        if any(len(b) != Signature.SIGNATURE_SIZE for b in buffers):
            raise ValueError("signatures are %d bytes" % Signature.SIGNATURE_SIZE)
        if not buffers:
            return []
        infos = aggregation_infos or [None] * len(buffers)
        out, ok = backend.get().g2_decompress(b"".join(buffers))
        if not all(ok):
            raise ValueError("No y for point x")
        return [Signature(JacobianPoint._from(H.F2, H.aff_to_jac(H.F2, H.g2_from_abi(out[192 * i:192 * (i + 1)])), default_ec_twist),
                          infos[i]) for i in range(len(buffers))]

    @staticmethod
    def from_g2(g2_el, aggregation_info=None):
        return Signature(g2_el, aggregation_info)

    def set_aggregation_info(self, aggregation_info):
        self.aggregation_info = aggregation_info

    def get_aggregation_info(self):
        return self.aggregation_info

    def divide_by(self, divisor_signatures):
        """Remove already-verified parts from an aggregate (signature.py:44-103):
        each divisor must be a subset with unique (message, pk) pairs and a
        consistent exponent quotient."""
        drop = []
        prod = None
        for div in divisor_signatures:
            pks, mhs = div.aggregation_info.public_keys, div.aggregation_info.message_hashes
            if len(pks) != len(mhs):
                raise Exception("Invalid aggregation info")
            quotient = None
            for mh, pk in zip(mhs, pks):
                divisor = div.aggregation_info.tree[(mh, pk)]
                try:
                    dividend = self.aggregation_info.tree[(mh, pk)]
                except KeyError:
                    raise Exception("Signature is not a subset")
                qn = dividend * pow(divisor, GROUP_ORDER - 2, GROUP_ORDER) % GROUP_ORDER
                if quotient is None:
                    quotient = qn
                elif qn != quotient:
                    raise Exception("Cannot divide by aggregate signature,msg/pk pairs are not unique")
                drop.append((mh, pk))
            term = div.value * ((-quotient) % GROUP_ORDER) if quotient is not None else None
            if term is not None:
                prod = term if prod is None else prod + term
        value = self.value if prod is None else self.value + prod
        out = Signature(deepcopy(value), deepcopy(self.aggregation_info))
        for key in drop:
            out.aggregation_info.tree.pop(key, None)
        keys = sorted(out.aggregation_info.tree)
        out.aggregation_info.message_hashes = [k[0] for k in keys]
        out.aggregation_info.public_keys = [k[1] for k in keys]
        return out

    def serialize(self):
        return self.value.serialize()

    def size(self):
        return self.SIGNATURE_SIZE

    def __eq__(self, other):
        return self.serialize() == other.serialize()

    def __hash__(self):
        return int.from_bytes(self.serialize(), "big")

    def __lt__(self, other):
        return self.serialize() < other.serialize()

    def __repr__(self):
        return "Signature(%s)" % self.serialize().hex()
