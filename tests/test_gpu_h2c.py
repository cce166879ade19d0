"""Hash-to-G2 on the GPU (blsgpu_map_to_g2 through the C ABI): reference vectors
(tests/golden/hash_to_curve.json, generated from ec.py:528-550) and the host
integer implementation on seeded batches.  Bit-exact affine bytes."""
import hashlib

import pytest

from bls_py import hostmath as H
from bls_py.util import hash512

pytestmark = pytest.mark.gpu


def test_reference_vectors(engine, golden):
    vec = golden("hash_to_curve.json")["hash_to_g2"]
    t = b"".join(H.g2_hash_field_elements(bytes.fromhex(r["msg_hash"]), hash512) for r in vec)
    out = engine.map_to_g2(t)
    assert [out[192 * i:192 * (i + 1)].hex() for i in range(len(vec))] == [r["point"] for r in vec]


def test_whole_hash_from_message_hashes(engine, golden):
    """blsgpu_hash_to_g2: SHA-256 chain + reduction mod q on the GPU as well -- reference
    vectors, then seeded batches against the host implementation."""
    vec = golden("hash_to_curve.json")["hash_to_g2"]
    out = engine.hash_to_g2(b"".join(bytes.fromhex(r["msg_hash"]) for r in vec))
    assert [out[192 * i:192 * (i + 1)].hex() for i in range(len(vec))] == [r["point"] for r in vec]
    for n in (1, 13, 300):
        msgs = [hashlib.sha256(b"whole-%d-%d" % (n, i)).digest() for i in range(n)]
        out = engine.hash_to_g2(b"".join(msgs))
        t = b"".join(H.g2_hash_field_elements(m, hash512) for m in msgs)
        assert out == engine.map_to_g2(t)
        for i in (0, n // 2, n - 1):
            assert out[192 * i:192 * (i + 1)] == H.g2_affine_bytes(H.hash_to_g2_prehashed(msgs[i], hash512))
    assert engine.hash_to_g2(b"") == b""


def test_sw_encode_vectors_through_the_map(engine, golden):
    """t1 = 0 encodes to infinity, so the map returns clear_cofactor(sw_encode(t0))."""
    vec = golden("hash_to_curve.json")["sw_encode_fq2"]
    t = b"".join(bytes.fromhex(r["t"]) + bytes(96) for r in vec)
    out = engine.map_to_g2(t)
    for i, r in enumerate(vec):
        S = H.g2_from_abi(bytes.fromhex(r["point"]))
        assert out[192 * i:192 * (i + 1)] == H.g2_affine_bytes(H.clear_cofactor_g2(H.aff_to_jac(H.F2, S)))
    # infinity + infinity, and t0 = -t1 (the encodings cancel): (0, 0)
    tt = bytes.fromhex(vec[1]["t"])
    neg = b"".join(((H.Q - int.from_bytes(tt[48 * j:48 * j + 48], "big")) % H.Q).to_bytes(48, "big") for j in range(2))
    assert engine.map_to_g2(bytes(192) + tt + neg) == bytes(384)
    assert engine.map_to_g2(b"") == b""


@pytest.mark.parametrize("n", [1, 2, 7, 61])
def test_seeded_batches_match_host(engine, n):
    """Ragged batches (not multiples of the per-team counts)."""
    msgs = [hashlib.sha256(b"h2c-%d-%d" % (n, i)).digest() for i in range(n)]
    out = engine.map_to_g2(b"".join(H.g2_hash_field_elements(m, hash512) for m in msgs))
    for i, m in enumerate(msgs):
        assert out[192 * i:192 * (i + 1)] == H.g2_affine_bytes(H.hash_to_g2_prehashed(m, hash512)), i


def test_large_batch_properties(engine):
    """4096 messages: every output is on the twist and in the order-n subgroup is checked
    for a sample through the pairing-free relation psi(P) = [x]P... kept simple: on-curve
    for all, exact match with the host for a strided sample, permutation invariance."""
    n = 4096
    msgs = [hashlib.sha256(b"h2c-big-%d" % i).digest() for i in range(n)]
    t = [H.g2_hash_field_elements(m, hash512) for m in msgs]
    out = engine.map_to_g2(b"".join(t))
    pts = [out[192 * i:192 * (i + 1)] for i in range(n)]
    for p in pts:
        assert H.on_curve(H.F2, H.g2_from_abi(p))
    for i in range(0, n, 512):
        assert pts[i] == H.g2_affine_bytes(H.hash_to_g2_prehashed(msgs[i], hash512))
    rev = engine.map_to_g2(b"".join(reversed(t)))
    assert [rev[192 * i:192 * (i + 1)] for i in range(n)] == pts[::-1]


def test_register_form_of_the_clearing_on_small_batches(golden):
    """The register forms of the cofactor clearing (lane quads / lane pairs: the default from 8192 messages on;
    BLSGPU_H2C_REG_THRESHOLD), forced for small batches: reference vectors, infinity summands, ragged counts, and equality with the
    VM form."""
    import os
    from bls_py import _native
    old = os.environ.get("BLSGPU_H2C_REG_THRESHOLD")
    os.environ["BLSGPU_H2C_REG_THRESHOLD"] = "1"
    try:
        e = _native.Engine(0)
    finally:
        if old is None:
            del os.environ["BLSGPU_H2C_REG_THRESHOLD"]
        else:
            os.environ["BLSGPU_H2C_REG_THRESHOLD"] = old
    vec = golden("hash_to_curve.json")["hash_to_g2"]
    out = e.hash_to_g2(b"".join(bytes.fromhex(r["msg_hash"]) for r in vec))
    assert [out[192 * i:192 * (i + 1)].hex() for i in range(len(vec))] == [r["point"] for r in vec]
    sw = golden("hash_to_curve.json")["sw_encode_fq2"]
    tt = bytes.fromhex(sw[1]["t"])
    neg = b"".join(((H.Q - int.from_bytes(tt[48 * j:48 * j + 48], "big")) % H.Q).to_bytes(48, "big") for j in range(2))
    assert e.map_to_g2(bytes(192) + tt + neg) == bytes(384)              # infinity + infinity, S + (-S)
    for n in (1, 63, 64, 65, 200):
        msgs = b"".join(hashlib.sha256(b"lane-%d-%d" % (n, i)).digest() for i in range(n))
        got = e.hash_to_g2(msgs)
        assert got == _native.engine(0).hash_to_g2(msgs)                # the VM form (default engine, small batch)
        assert got[:192] == H.g2_affine_bytes(H.hash_to_g2_prehashed(msgs[:32], hash512))


def test_default_selection_at_65536_messages(engine):
    """from 65 536 messages on the default engine runs the lane / symbol encodings and the lane-pair clearing: same bytes
    as an engine with every one of those switched off (wavefront-VM kernels only), a strided sample against the host"""
    import os
    from bls_py import _native
    n = 65536 + 7
    msgs = b"".join(hashlib.sha256(b"h2c-65k-%d" % i).digest() for i in range(n))
    got = engine.hash_to_g2(msgs)
    # the comparison engine runs NONE of the round-3 kernels: encodings on the wavefront VM with five powers each (lanes
    # and symbols off), cofactor clearing on the VM
    knobs = {"BLSGPU_H2C_REG_THRESHOLD": str(1 << 40), "BLSGPU_H2C_LANE_THRESHOLD": str(1 << 40), "BLSGPU_H2C_JACOBI": "0"}
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        vm = _native.Engine(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    assert got == vm.hash_to_g2(msgs)
    for i in range(0, n, 8191):
        assert got[192 * i:192 * (i + 1)] == H.g2_affine_bytes(H.hash_to_g2_prehashed(msgs[32 * i:32 * (i + 1)], hash512))


def test_candidate_with_real_u_is_skipped(engine, golden):
    """tests/golden/g2_real_u.json (reference-generated): t whose first Shallue-van de Woestijne candidate has
    a u with zero imaginary part -- the reference skips it; blsgpu_map_to_g2 must return the reference's point."""
    recs = golden("g2_real_u.json")["sw_encode"]
    out = engine.map_to_g2(b"".join(bytes.fromhex(r["t"]) for r in recs))
    assert out.hex() == "".join(r["point"] for r in recs)
