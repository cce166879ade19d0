"""Hash helpers with the reference's constructions (util.py:7-50)."""
import hashlib


def _b(m):
    return m if isinstance(m, (bytes, bytearray)) else m.encode("utf-8")


def hash256(m):
    return hashlib.sha256(_b(m)).digest()


def hash512(m):
    m = bytes(_b(m))
    return hash256(m + b"\x00") + hash256(m + b"\x01")


def hmac256(m, k):
    """HMAC-SHA256 written out as the reference does (64-byte block, key hashed if longer)."""
    m, k = bytes(_b(m)), bytes(_b(k))
    if len(k) > 64:
        k = hash256(k)
    k = k.ljust(64, b"\x00")
    inner = hash256(bytes(c ^ 0x36 for c in k) + m)
    return hash256(bytes(c ^ 0x5c for c in k) + inner)


def hash_pks(num_outputs, public_keys):
    """t_i = sha256(be32(i) || sha256(ser(pk_0) || ser(pk_1) || ...)) mod n  (util.py:36-50)."""
    from .bls12381 import n
    digest = hash256(b"".join(pk.serialize() for pk in public_keys))
    return [int.from_bytes(hash256(i.to_bytes(4, "big") + digest), "big") % n for i in range(num_outputs)]
