"""Scheme API (bls.py of the reference): aggregation and verification.

`verify` is the GPU hot path: it assembles Ps = [-G1] + per-message key sums and
Qs = [signature] + H(m) exactly as bls.py:153-201 does and hands them to the
multi-pairing engine.  Group sums here are host integer code for now."""
from . import hostmath as H
from .aggregation_info import AggregationInfo
from .bls12381 import n as GROUP_ORDER
from .ec import (AffinePoint, JacobianPoint, default_ec, generator_Fq,
                 hash_to_point_prehashed_Fq2)
from .fields import Fq12
from .keys import PrivateKey, PublicKey
from .pairing import ate_pairing_multi
from .signature import Signature
from .util import hash_pks


def _g2_zero():
    return JacobianPoint._from(H.F2, None)


def _g1_zero():
    return JacobianPoint._from(H.F1, None)


class BLS:
    @staticmethod
    def aggregate_sigs_simple(signatures):
        """Plain sum; NOT safe for signatures over one message (rogue keys)."""
        acc = _g2_zero()
        for sig in signatures:
            acc = acc + sig.value
        return Signature.from_g2(acc)

    @staticmethod
    def aggregate_sigs_secure(signatures, public_keys, message_hashes):
        if not (len(signatures) == len(public_keys) == len(message_hashes)):
            raise Exception("Invalid number of keys")
        ordered = sorted(zip(message_hashes, public_keys, signatures))
        ts = hash_pks(len(public_keys), public_keys)
        acc = _g2_zero()
        for t, (_, _, sig) in zip(ts, ordered):
            acc = acc + sig.value * t
        return Signature.from_g2(acc)

    @staticmethod
    def aggregate_sigs(signatures):
        infos = []
        for sig in signatures:
            if sig.aggregation_info is None or sig.aggregation_info.empty():
                raise Exception("Each signature must have a valid aggregation info")
            infos.append(sig.aggregation_info)
        colliding = AggregationInfo._colliding_messages(infos)
        if not colliding:
            out = BLS.aggregate_sigs_simple(signatures)
            out.set_aggregation_info(AggregationInfo.merge_infos(infos))
            return out
        hit = [s for s in signatures if any(m in colliding for m in s.aggregation_info.message_hashes)]
        rest = [s for s in signatures if not any(m in colliding for m in s.aggregation_info.message_hashes)]
        hit.sort(key=lambda s: s.aggregation_info)
        keys = sorted((mh, pk) for s in hit
                      for mh, pk in zip(s.aggregation_info.message_hashes, s.aggregation_info.public_keys))
        ts = hash_pks(len(hit), [pk for _, pk in keys])
        acc = _g2_zero()
        for t, sig in zip(ts, hit):
            acc = acc + sig.value * t
        for sig in rest:
            acc = acc + sig.value
        out = Signature.from_g2(acc)
        out.set_aggregation_info(AggregationInfo.merge_infos(infos))
        return out

    @staticmethod
    def verify(signature):
        info = signature.aggregation_info
        by_message = {}
        for mh, pk in zip(info.message_hashes, info.public_keys):
            by_message.setdefault(mh, []).append(pk)
        Ps, Qs = [], []
        for mh, keys in by_message.items():
            total = _g1_zero()
            for pk in set(keys):
                try:
                    exponent = info.tree[(mh, pk)]
                except KeyError:
                    return False
                total = total + pk.value * exponent
            Ps.append(total.to_affine())
            Qs.append(hash_to_point_prehashed_Fq2(mh))
        neg_g1 = generator_Fq() * (GROUP_ORDER - 1)
        res = ate_pairing_multi([neg_g1] + Ps, [signature.value.to_affine()] + Qs, default_ec)
        return res == Fq12.one(default_ec.q)

    @staticmethod
    def aggregate_pub_keys(public_keys, secure):
        if len(public_keys) < 1:
            raise Exception("Invalid number of keys")
        public_keys.sort()                 # in place, like the reference (bls.py:210)
        ts = hash_pks(len(public_keys), public_keys)
        acc = _g1_zero()
        for t, pk in zip(ts, public_keys):
            acc = acc + (pk.value * t if secure else pk.value)
        return PublicKey.from_g1(acc)

    @staticmethod
    def aggregate_priv_keys(private_keys, public_keys, secure):
        if not secure:
            total = sum(sk.value for sk in private_keys) % GROUP_ORDER
        else:
            if not public_keys:
                raise Exception("Must include public keys in secure aggregation")
            if len(private_keys) != len(public_keys):
                raise Exception("Invalid number of keys")
            pairs = sorted(zip(public_keys, private_keys))
            ts = hash_pks(len(private_keys), public_keys)
            total = sum(sk.value * t for t, (_, sk) in zip(ts, pairs)) % GROUP_ORDER
        return PrivateKey.from_bytes(total.to_bytes(32, "big"))
