"""How an aggregate signature was put together: a map (message_hash, PublicKey)
-> exponent, plus the two parallel sorted lists the verifier iterates over.
Same observable behaviour as the reference's aggregation_info.py (ordering of
the lists, exponent arithmetic mod n, comparison used for sorting)."""
from copy import deepcopy

from .bls12381 import n as GROUP_ORDER
from .util import hash256, hash_pks


class AggregationInfo:
    def __init__(self, tree, message_hashes, public_keys):
        self.tree = tree
        self.message_hashes = message_hashes
        self.public_keys = public_keys

    # ---- constructors
    @staticmethod
    def from_msg_hash(public_key, message_hash):
        return AggregationInfo({(message_hash, public_key): 1}, [message_hash], [public_key])

    @staticmethod
    def from_msg(pk, message):
        return AggregationInfo.from_msg_hash(pk, hash256(message))

    @staticmethod
    def _from_tree(tree):
        keys = sorted(tree)
        return AggregationInfo(tree, [mh for mh, _ in keys], [pk for _, pk in keys])

    def empty(self):
        return not self.tree

    # ---- ordering: lexicographic on (message hash, pk, exponent) triples
    def _order_key(self):
        """the triples with the key in its serialised form: PublicKey orders by its 48 bytes (keys.py:69), so comparing these tuples
        IS the lexicographic comparison of the triples, the shorter list first on a tie -- kept on the object (sorting n infos
        compared n log n lists built from scratch: 30 ms of a 1024-signature secure aggregation) and rebuilt when the map or the
        lists were replaced (Signature.divide_by)"""
        stamp = (len(self.tree), id(self.message_hashes), id(self.public_keys))
        k = self.__dict__.get("_okey")
        if k is None or k[0] != stamp:
            k = (stamp, tuple((mh, pk.serialize(), self.tree[(mh, pk)]) for mh, pk in zip(self.message_hashes, self.public_keys)))
            self.__dict__["_okey"] = k
        return k[1]

    def __lt__(self, other):
        return self._order_key() < other._order_key()

    def __eq__(self, other):
        return not self.__lt__(other) and not other.__lt__(self)

    def __deepcopy__(self, memo):
        return AggregationInfo(deepcopy(self.tree, memo), deepcopy(self.message_hashes, memo),
                               deepcopy(self.public_keys, memo))

    def __str__(self):
        return "".join("(%s,%s):\n%s\n" % (mh.hex(), pk.serialize().hex(), hex(e))
                       for (mh, pk), e in self.tree.items())

    # ---- merging
    @staticmethod
    def _colliding_messages(infos):
        """Messages that appear in more than one info."""
        seen, colliding = set(), set()
        for info in infos:
            local = {mh for mh, _ in info.tree}
            colliding |= local & seen
            seen |= local
        return colliding

    @staticmethod
    def simple_merge_infos(aggregation_infos):
        tree = {}
        for info in aggregation_infos:
            tree.update(info.tree)
        return AggregationInfo._from_tree(tree)

    @staticmethod
    def secure_merge_infos(colliding_infos):
        """Each info's exponents are scaled by t_i = H(i, all pks) and summed mod n."""
        colliding_infos.sort()
        keys = sorted(k for info in colliding_infos for k in info.tree)
        ts = hash_pks(len(colliding_infos), [pk for _, pk in keys])
        tree = {}
        for t, info in zip(ts, colliding_infos):
            for key, exp in info.tree.items():
                tree[key] = (tree.get(key, 0) + exp * t) % GROUP_ORDER
        return AggregationInfo._from_tree(tree)

    @staticmethod
    def merge_infos(aggregation_infos):
        colliding = AggregationInfo._colliding_messages(aggregation_infos)
        if not colliding:
            return AggregationInfo.simple_merge_infos(aggregation_infos)
        hit = [i for i in aggregation_infos if any(mh in colliding for mh, _ in i.tree)]
        rest = [i for i in aggregation_infos if not any(mh in colliding for mh, _ in i.tree)]
        rest.append(AggregationInfo.secure_merge_infos(hit))
        return AggregationInfo.simple_merge_infos(rest)
