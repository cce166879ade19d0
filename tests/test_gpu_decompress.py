"""Batched point decompression on the GPU (blsgpu_g1_decompress / blsgpu_g2_decompress
through the C ABI): reference serialisation vectors, the host integer implementation on
seeded and adversarial encodings, round trips at size.  Bit-exact affine bytes + accept flags."""
import hashlib
import random

import pytest

from bls_py import hostmath as H

pytestmark = pytest.mark.gpu
Q = H.Q


def host(deg, enc):
    try:
        A = (H.g1_decompress if deg == 1 else H.g2_decompress)(enc)
    except (ValueError, H.RealSquareRoot):
        return None
    return H.g1_affine_bytes(A) if deg == 1 else H.g2_affine_bytes(A)


def compare(engine, deg, encs):
    sz = 48 * deg
    out, ok = (engine.g1_decompress if deg == 1 else engine.g2_decompress)(b"".join(encs))
    for i, e in enumerate(encs):
        want = host(deg, e)
        assert ok[i] == (want is not None), i
        if want is not None:
            assert out[2 * sz * i:2 * sz * (i + 1)] == want, i


def test_reference_vectors(engine, golden):
    """Serialised keys / signatures of the reference's own tests (tests.py:118-127 via scheme.json)
    and the compressed golden points: decompress -> the affine points the reference holds."""
    pts = golden("points.json")
    g1 = [bytes.fromhex(r["p"]) for r in pts["g1"]]
    g2 = [bytes.fromhex(r["p"]) for r in pts["g2"]]
    enc1 = [H.g1_compress(H.g1_from_abi(p)) for p in g1]
    enc2 = [H.g2_compress(H.g2_from_abi(p)) for p in g2]
    out, ok = engine.g1_decompress(b"".join(enc1))
    assert all(ok) and out == b"".join(g1)
    out, ok = engine.g2_decompress(b"".join(enc2))
    assert all(ok) and out == b"".join(g2)
    v4 = golden("verify4.json")
    pks = [bytes.fromhex(h) for h in v4["pk"]]
    sigs = [bytes.fromhex(h) for h in v4["sig"]] + [bytes.fromhex(v4["agg_sig"]), bytes.fromhex(v4["tampered_sig"])]
    compare(engine, 1, pks)
    compare(engine, 2, sigs)


def test_random_and_rejected_encodings(engine):
    rng = random.Random(21)
    enc1 = [bytes([rng.randrange(256) for _ in range(48)]) for _ in range(300)]      # flag bits set at random
    enc2 = [bytes([rng.randrange(256) for _ in range(96)]) for _ in range(150)]
    enc1 += [bytes(48), b"\x80" + bytes(47), b"\xff" * 48, (Q + 5).to_bytes(48, "big")]
    enc2 += [bytes(96), b"\x80" + bytes(95), b"\xff" * 96, bytes(47) + b"\x01" + bytes(48), bytes(95) + b"\x01"]
    compare(engine, 1, enc1)
    compare(engine, 2, enc2)
    assert engine.g1_decompress(b"") == (b"", [])


def test_g2_real_u_branch(engine, golden):
    """x^3 + 4(1+i) real: the reference's `a1 == 0` branch of Fq2.modsqrt returns an Fq and
    Signature.from_bytes raises for every such encoding (tests/golden/g2_real_u.json, reference-generated)."""
    from test_vm_decompress import fq2_cube_roots_with_real_u, encode
    xs = fq2_cube_roots_with_real_u(6)
    encs = [encode(2, x, big) for x in xs for big in (False, True)]
    assert not any(host(2, e) is not None for e in encs)
    compare(engine, 2, encs)
    recs = golden("g2_real_u.json")["decompress"]
    assert all(r["reference"] in ("Exception", "ValueError") for r in recs)
    _, ok = engine.g2_decompress(b"".join(bytes.fromhex(r["encoding"]) for r in recs))
    assert not any(ok)


def test_round_trip_at_size(engine, seeded_pairs):
    """1025 golden pairs: compress on the host, decompress on the GPU -> the original bytes,
    in ragged batches (not multiples of the per-team counts)."""
    g1, g2 = seeded_pairs
    p1 = [g1[96 * i:96 * (i + 1)] for i in range(1025)]
    p2 = [g2[192 * i:192 * (i + 1)] for i in range(1025)]
    out, ok = engine.g1_decompress(b"".join(H.g1_compress(H.g1_from_abi(p)) for p in p1))
    assert all(ok) and out == g1[:96 * 1025]
    out, ok = engine.g2_decompress(b"".join(H.g2_compress(H.g2_from_abi(p)) for p in p2))
    assert all(ok) and out == g2[:192 * 1025]


def test_host_mirror_batch_constructors(engine):
    from bls_py.keys import PrivateKey, PublicKey
    from bls_py.signature import Signature
    sks = [PrivateKey.from_seed(bytes([i, 7, 9])) for i in range(5)]
    pks = [sk.get_public_key() for sk in sks]
    sigs = [sk.sign(b"m%d" % i) for i, sk in enumerate(sks)]
    assert [p.serialize() for p in PublicKey.from_bytes_batch([p.serialize() for p in pks])] == [p.serialize() for p in pks]
    back = Signature.from_bytes_batch([s.serialize() for s in sigs])
    assert [s.serialize() for s in back] == [s.serialize() for s in sigs]
    assert [PublicKey.from_bytes(p.serialize()).serialize() for p in pks] == [p.serialize() for p in pks]
    with pytest.raises(ValueError):
        PublicKey.from_bytes_batch([pks[0].serialize(), b"\x00" * 47 + b"\x01"])
