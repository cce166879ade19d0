"""The kernel forms the engine ships but does not select by default, and the fallback it takes when the line-stream
workspace cannot be allocated (VERDICT r3 item 1 d / e): each is reachable through an environment knob read at context
creation (csrc/blsgpu_api.hip), so each gets the reference's vectors.

  BLSGPU_LS_WIDE_MAX=0      k_ml_lines4 (one pair per lane QUAD) where the default is k_ml_lines_wide (sixteen lanes per pair, up to 10 240 pairs)
  BLSGPU_LS_QUAD_MAX=0      (with the former) k_ml_lines2 (one pair per lane PAIR) for calls of every size: the kernel of bench.py's
                            524 800-pair step, which calls of up to 20 480 pairs -- every vector here -- otherwise leave to the other two
  BLSGPU_LS_MERGE_WIDE_MAX=0  k_ml_merge (six lanes per value) for every merge level (default: k_ml_merge_wide, one wavefront per
                            output, for levels with few outputs)
  BLSGPU_MILLER_WIDE3_MAX=0 the wide Miller loop on two wavefronts per pair for calls of every size it takes (default: three wavefronts,
                            the accumulator split over two of them, up to 256 pairs)
  BLSGPU_MILLER_WIDE_MAX=0  small calls on the wavefront VM's k_miller (four pairs per workgroup + product tree) instead of k_miller_wide
  BLSGPU_VM_EXACT_LANES=0   degenerate blocks of the wavefront-VM kernels recomputed by k_miller_slow
                            (default: k_ml_lines_exact in block mode + k_ml_small in list mode)
  BLSGPU_TEST_LS_NOMEM=1    test hook: launch_miller_ls reports -ENOMEM before touching the device, the call must go
                            through the wavefront-VM kernels and return the same bytes
  BLSGPU_WG256_MAX_WAVES=0  the register kernels launched as 64-thread workgroups whatever the size (default: launches of
                            up to 4096 wavefronts as 256-thread workgroups, csrc/blsgpu_tu.h wave_index())

Expected values: tests/golden/pairing.json (1025-pair seeded batch, edge cases) and pairing_degenerate.json -- generated
by importing the reference -- and the CPU oracle for spliced batches."""
import os

import pytest

from conftest import cat, engine_with_env
from test_gpu_linestream import EDGE, _spliced, flags

pytestmark = pytest.mark.gpu

FORMS = {
    "point_chains_on_lane_pairs": ({"BLSGPU_LS_WIDE_MAX": "0", "BLSGPU_LS_QUAD_MAX": "0"}, True),   # k_ml_lines2 at these sizes
    "point_chains_on_lane_quads": ({"BLSGPU_LS_WIDE_MAX": "0"}, True),                          # k_ml_lines4 (round 4's default below 20 480 pairs)
    "merge_levels_six_lanes_per_value": ({"BLSGPU_LS_MERGE_WIDE_MAX": "0"}, True),          # k_ml_merge for every level
    "vm_slow_program_for_degenerate_blocks": ({"BLSGPU_VM_EXACT_LANES": "0"}, False),
    "wide_miller_on_two_wavefronts": ({"BLSGPU_MILLER_WIDE3_MAX": "0"}, False),               # k_miller_wide<2> at the sizes k_miller_wide<3> takes by default
    "small_calls_on_the_wavefront_vm": ({"BLSGPU_MILLER_WIDE_MAX": "0"}, False),              # k_miller: round 4's default below 4096 pairs
    "line_stream_workspace_unavailable": ({"BLSGPU_TEST_LS_NOMEM": "1"}, True),
    "workgroups_of_64_threads": ({"BLSGPU_WG256_MAX_WAVES": "0"}, True),                    # round 3's launch shape for every size
}
_cache = {}


def form_engine(name):
    from bls_py import _native
    if name not in _cache:
        env, force_ls = FORMS[name]
        e = engine_with_env(env)
        if force_ls:
            e.set_ls_threshold(1, 1)          # every multi-pairing asks for the line-stream kernels
        _cache[name] = e
    return _cache[name]


@pytest.mark.parametrize("name", sorted(FORMS))
def test_reference_vectors(name, golden, seeded_pairs, oracle):
    e = form_engine(name)
    g1, g2 = seeded_pairs
    want = golden("pairing.json")["seeded"]["1025"]["out"]
    assert e.pairing_multi(g1, g2, 1025).hex() == want
    v = golden("pairing.json")["small4"]
    assert e.pairing_multi(cat(v["g1"]), cat(v["g2"]), 4).hex() == v["out"]
    for edge in EDGE:
        v = golden("pairing.json")["edge"][edge]
        assert e.pairing_multi(cat(v["g1"]), cat(v["g2"]), len(v["g1"]), flags(v)).hex() == v["out"], edge
    # every reference-generated degenerate case (all_kinds among them), alone ...
    for case, v in golden("pairing_degenerate.json")["cases"].items():
        assert e.pairing_multi(cat(v["g1"]), cat(v["g2"]), len(v["g1"]), flags(v)).hex() == v["out"], case
    # ... and sprinkled into a 333-pair batch (the oracle is pinned to the same fixtures)
    a, b, inf = _spliced(golden, seeded_pairs)
    n = len(a) // 96
    assert e.pairing_multi(a, b, n, inf) == oracle.pairing_multi(a, b, n, threads=8, inf=inf)
    # a batch of 20 verifications (the batch entry: groups, the per-group product tree / Horner, final exponentiations)
    out = e.pairing_multi_batch(g1 * 20, g2 * 20, 1025, 20)
    assert all(out[576 * g:576 * (g + 1)].hex() == want for g in range(20))


def test_fallback_engine_really_refuses_the_line_stream_path(seeded_pairs):
    """the hook is live: a FRESH context with it answers calls that ask for the line-stream kernels without ever
    allocating the line-stream stage's partial products, a context
    without it does allocate them, and both return the same bytes -- two paths compared, not one with itself"""
    from bls_py import _native
    g1, g2 = seeded_pairs
    _cache.pop("line_stream_workspace_unavailable", None)
    hooked = form_engine("line_stream_workspace_unavailable")
    plain = _native.Engine(0)
    plain.set_ls_threshold(1, 1)
    for n in (1, 7, 300, 1025):
        assert hooked.pairing_multi(g1[:96 * n], g2[:192 * n], n) == plain.pairing_multi(g1[:96 * n], g2[:192 * n], n)
    # (the wavefront-VM kernels keep a scratch of line records for their degenerate blocks, so "lines" is no evidence; the
    # per-line-index partial products exist only on the line-stream path)
    assert hooked.workspace_bytes()["line_products"] == 0 and plain.workspace_bytes()["line_products"] >= 68 * 168 * 4
    ws = plain.workspace_bytes()
    assert ws["total"] == sum(v for k, v in ws.items() if k != "total")


def test_workgroup_shapes_agree_beyond_the_pairing(golden, seeded_pairs):
    """The other kernels that index by wave_index() / the global thread index -- hash to G2 (division-step encodings,
    clearing on lane pairs and quads), batches of G2 sums (window Horner on lane quads), batched final exponentiations
    (six lanes per result), small-group Miller loops -- with 64-thread workgroups against the reference's digests and
    against the default shapes, at sizes that leave the last workgroup ragged."""
    import hashlib
    from bls_py import _native
    e64 = form_engine("workgroups_of_64_threads")
    dflt = _native.Engine(0)
    rec = golden("h2c_20000.json")
    msgs = b"".join(hashlib.sha256(b"bench-h2c-0-%d" % i).digest() for i in range(rec["n"]))
    out = e64.hash_to_g2(msgs)                                   # lanes + symbols + clearing on lane pairs (20 000 > 16 384)
    assert hashlib.sha256(out).hexdigest() == rec["outputs_sha256"]
    assert e64.hash_to_g2(msgs[:32 * 16381]) == out[:192 * 16381] == dflt.hash_to_g2(msgs[:32 * 16381])     # ... on lane quads, ragged
    g1, g2 = seeded_pairs
    # 333 groups of 3 pairs: k_ml_lines4 + k_ml_small (34 wavefronts of ten groups), then 333 final exponentiations
    n = 999
    a, b = (g1 * 2)[:96 * n], (g2 * 2)[:192 * n]
    for eng in (e64, dflt):
        eng.set_ls_threshold(1, 64)
    want = dflt.pairing_multi_batch(a, b, 3, 333)
    assert e64.pairing_multi_batch(a, b, 3, 333) == want
    for eng in (e64, dflt):
        eng.set_fexp_team_threshold(1)                           # six lanes per result from one result on
    assert e64.pairing_multi_batch(a, b, 3, 333) == want == dflt.pairing_multi_batch(a, b, 3, 333)
    v = golden("pairing.json")["small4"]
    assert e64.pairing_multi(cat(v["g1"]), cat(v["g2"]), 4).hex() == v["out"]
