"""Scheme API (bls.py of the reference): aggregation and verification.

`verify` is the GPU hot path: it assembles Ps = [-G1] + per-message key sums and
Qs = [signature] + H(m) exactly as bls.py:153-201 does and hands them to the
multi-pairing engine.  The group sums (signature aggregation, per-message key
folding, aggregate public keys) are GPU multi-scalar sums as well."""
from . import hostmath as H
from .aggregation_info import AggregationInfo
from .bls12381 import n as GROUP_ORDER
from .ec import (AffinePoint, JacobianPoint, default_ec, generator_Fq,
                 hash_to_points_prehashed_Fq2)
from .fields import Fq12
from .keys import PrivateKey, PublicKey
from .pairing import ate_pairing_multi
from .signature import Signature
from .util import hash_pks


from . import backend


def _g2_sum(points, scalars=None):
    """sum_i scalars[i] * points[i] (JacobianPoint over Fq2) on the GPU."""
    if not points:
        return JacobianPoint._from(H.F2, None)
    pts = b"".join(H.g2_affine_bytes(p.to_affine()._aff()) for p in points)
    out, inf = backend.get().g2_msm(pts, scalars, len(points), 1)
    return JacobianPoint._from(H.F2, None if inf[0] else H.aff_to_jac(H.F2, H.g2_from_abi(out)))


def _g1_sums(groups_of_points, groups_of_scalars):
    """One G1 multi-scalar sum per group (ragged groups are padded with infinity)."""
    if not groups_of_points:
        return []
    k = max(len(g) for g in groups_of_points)
    pts, sc = bytearray(), []
    for g, s in zip(groups_of_points, groups_of_scalars):
        for p in g:
            pts += H.g1_affine_bytes(p.to_affine()._aff())
        pts += bytes(96) * (k - len(g))
        sc += [int(x) for x in s] + [0] * (k - len(g))
    out, inf = backend.get().g1_msm(bytes(pts), sc, k, len(groups_of_points))
    return [JacobianPoint._from(H.F1, None if inf[i] else H.aff_to_jac(H.F1, H.g1_from_abi(out[96 * i:96 * (i + 1)])))
            for i in range(len(groups_of_points))]


def _jac_g1_bytes(J):
    """affine bytes (x || y, (0,0) for infinity) of a G1 JacobianPoint; no inversion when z = 1"""
    if J.infinity:
        return bytes(96)
    if J.z.Z == 1:
        return int(J.x.Z).to_bytes(48, "big") + int(J.y.Z).to_bytes(48, "big")
    return H.g1_affine_bytes(J.to_affine()._aff())


class BLS:
    @staticmethod
    def aggregate_sigs_simple(signatures):
        """Plain sum; NOT safe for signatures over one message (rogue keys)."""
        return Signature.from_g2(_g2_sum([sig.value for sig in signatures]))

    @staticmethod
    def aggregate_sigs_secure(signatures, public_keys, message_hashes):
        if not (len(signatures) == len(public_keys) == len(message_hashes)):
            raise Exception("Invalid number of keys")
        ordered = sorted(zip(message_hashes, public_keys, signatures))
        ts = hash_pks(len(public_keys), public_keys)
        return Signature.from_g2(_g2_sum([sig.value for _, _, sig in ordered], ts))

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
        out = Signature.from_g2(_g2_sum([s.value for s in hit] + [s.value for s in rest], ts + [1] * len(rest)))
        out.set_aggregation_info(AggregationInfo.merge_infos(infos))
        return out

    @staticmethod
    def verify(signature):
        info = signature.aggregation_info
        by_message = {}
        for mh, pk in zip(info.message_hashes, info.public_keys):
            by_message.setdefault(mh, []).append(pk)
        key_groups, exp_groups = [], []
        for mh, keys in by_message.items():
            uniq = list(set(keys))
            try:
                exp_groups.append([info.tree[(mh, pk)] for pk in uniq])
            except KeyError:
                return False
            key_groups.append([pk.value for pk in uniq])
        prov = backend.get()
        # (an infinity SIGNATURE keeps the tuple path: its flag is the one flag the reference's Miller loop reads)
        if hasattr(prov, "verify_pipeline") and not signature.value.infinity and all(len(m) == 32 for m in by_message):
            return BLS._verify_on_device(prov, signature, list(by_message), key_groups, exp_groups)
        Qs = hash_to_points_prehashed_Fq2(list(by_message))
        Ps = [t.to_affine() for t in _g1_sums(key_groups, exp_groups)]
        neg_g1 = generator_Fq() * (GROUP_ORDER - 1)
        res = ate_pairing_multi([neg_g1] + Ps, [signature.value.to_affine()] + Qs, default_ec)
        return res == Fq12.one(default_ec.q)

    _NEG_G1 = None

    @staticmethod
    def _verify_on_device(prov, signature, hashes, key_groups, exp_groups):
        """The same Ps / Qs as above (bls.py:177-199), assembled as bytes and left on the GPU between hash-to-G2, the
        per-message key sums and the multi-pairing (HipProvider.verify_pipeline): no Python point objects per pair."""
        if BLS._NEG_G1 is None:
            BLS._NEG_G1 = H.g1_affine_bytes((generator_Fq() * (GROUP_ORDER - 1))._aff())
        n = len(hashes)
        sig = H.g2_affine_bytes(signature.value.to_affine()._aff())
        if all(len(g) == 1 and e[0] % GROUP_ORDER == 1 for g, e in zip(key_groups, exp_groups)):
            # one key with exponent 1 per message (a plain aggregate): the key itself is the pairing's P
            keys = b"".join(_jac_g1_bytes(g[0]) for g in key_groups)
            out = prov.verify_pipeline(BLS._NEG_G1, sig, b"".join(hashes), n, keys_affine=keys)
        else:
            k = max(len(g) for g in key_groups)
            pts, sc = bytearray(), bytearray()
            for g, e in zip(key_groups, exp_groups):
                for p, x in zip(g, e):
                    pts += _jac_g1_bytes(p)
                    sc += (int(x) % GROUP_ORDER).to_bytes(32, "big")
                pts += bytes(96) * (k - len(g))
                sc += bytes(32) * (k - len(g))
            out = prov.verify_pipeline(BLS._NEG_G1, sig, b"".join(hashes), n, key_pts=bytes(pts), key_scalars=bytes(sc), k=k)
        return out == Fq12.one(default_ec.q).serialize()

    @staticmethod
    def verify_batch(signatures):
        """[BLS.verify(s) for s in signatures] with every GPU step batched across the
        signatures: one hash-to-G2 call for all distinct (signature, message) pairs, one call
        for all per-message key sums, and blsgpu_pairing_multi_batch per distinct pair count
        (independent multi-pairings side by side).  Same results as verify, one by one."""
        from . import backend
        prov = backend.get()
        ONE = Fq12.one(default_ec.q).serialize()
        results = [None] * len(signatures)
        plans = []                                        # (index, message hashes, key groups, exponent groups)
        for i, sig in enumerate(signatures):
            info = sig.aggregation_info
            by_message = {}
            for mh, pk in zip(info.message_hashes, info.public_keys):
                by_message.setdefault(mh, []).append(pk)
            kg, eg = [], []
            try:
                for mh, keys in by_message.items():
                    uniq = list(set(keys))
                    eg.append([info.tree[(mh, pk)] for pk in uniq])
                    kg.append([pk.value for pk in uniq])
            except KeyError:
                results[i] = False                        # bls.py:189-190
                continue
            plans.append((i, list(by_message), kg, eg))
        if not plans:
            return results
        all_hashes = [mh for _, mhs, _, _ in plans for mh in mhs]
        Qs = hash_to_points_prehashed_Fq2(all_hashes)
        Ps = _g1_sums([g for _, _, kg, _ in plans for g in kg], [e for _, _, _, eg in plans for e in eg])
        neg_g1 = H.g1_affine_bytes((generator_Fq() * (GROUP_ORDER - 1))._aff())
        by_size, pos = {}, 0
        for i, mhs, _, _ in plans:
            k = len(mhs)
            g1 = neg_g1 + b"".join(H.g1_affine_bytes(p.to_affine()._aff()) for p in Ps[pos:pos + k])
            g2 = H.g2_affine_bytes(signatures[i].value.to_affine()._aff()) + \
                b"".join(H.g2_affine_bytes(q._aff()) for q in Qs[pos:pos + k])
            pos += k
            by_size.setdefault(k + 1, []).append((i, g1, g2))
        for size, items in by_size.items():
            out = prov.pairing_multi_batch(b"".join(x[1] for x in items), b"".join(x[2] for x in items), size, len(items))
            for j, (i, _, _) in enumerate(items):
                results[i] = out[576 * j:576 * (j + 1)] == ONE
        return results

    @staticmethod
    def aggregate_pub_keys(public_keys, secure):
        if len(public_keys) < 1:
            raise Exception("Invalid number of keys")
        public_keys.sort()                 # in place, like the reference (bls.py:210)
        ts = hash_pks(len(public_keys), public_keys)
        return PublicKey.from_g1(_g1_sums([[pk.value for pk in public_keys]], [ts if secure else [1] * len(public_keys)])[0])

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
