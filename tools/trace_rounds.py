"""Diagnostic (needs csrc/libblsgpu_stamps.so): run the first n rounds of a flat VM program
on the GPU from a given scratchpad image and compare every slot with vmgen/tablesim.py,
bisecting to the first round whose result differs.  Example: the G1 decompression program."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
os.environ["BLSGPU_LIBRARY"] = os.path.join(ROOT, "python-bls_amd", "csrc", "libblsgpu_stamps.so")
from bls_py import _native, hostmath as H
from vmgen import emit, tablesim, programs as P, h2c_programs as HP, sim
Q = sim.Q
e = _native.Engine(0)
e.lib.blsgpu_debug_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint32)]
tb = emit.build_tables()
sr, data = tb["seg_rounds"], tb["data"]
which = {"miller": (0, "mscript"), "mp": (1, "mpscript"), "fexp": (2, "fscript")}
prog = sys.argv[1] if len(sys.argv) > 1 else "d1"
if prog in ("d1", "d2", "h1", "h2"):
    segs, lay, script = tb[prog]
    wid = {"h1": 3, "h2": 4, "d1": 5, "d2": 6}[prog]
else:
    wid, key = which[prog]
    script = tb[key]
rounds = [r for n in script for r in sr[n]]
consts = HP.h2c_scratch_consts()
nslots = 900
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
img0 = [0] * nslots
for i, c in enumerate(consts):
    img0[i] = c
if prog == "d1":
    for k in range(lay.NE):
        enc = H.g1_compress(H.g1_from_abi(g1[96 * k:96 * (k + 1)]))
        img0[lay.X + k] = int.from_bytes(bytes([enc[0] & 0x1f]) + enc[1:], "big")
        img0[lay.BIG + k] = sim.to_m(1) if enc[0] & 0x80 else 0
else:
    import random
    rng = random.Random(1)
    for s in range(len(consts), nslots):
        img0[s] = rng.randrange(Q)

def gpu(n):
    buf = (ctypes.c_uint32 * (nslots * 12))()
    for s, v in enumerate(img0):
        for j in range(12):
            buf[s * 12 + j] = (v >> (32 * j)) & 0xFFFFFFFF
    rc = e.lib.blsgpu_debug_run(e.h, wid, n, nslots, buf)
    assert rc == 0, e.lib.blsgpu_last_error()
    return [sum(buf[s * 12 + j] << (32 * j) for j in range(12)) % Q for s in range(nslots)]

def cpu(n):
    m = tablesim.TableMachine(consts, nslots, data, P.C_K1)
    m.team = list(img0)
    m.run(rounds[:n])
    return [v % Q for v in m.team]

def same(n):
    return gpu(n) == cpu(n)
print("program", prog, "rounds", len(rounds), "full run matches:", same(len(rounds)))
lo, hi = 0, len(rounds)
if not same(hi):
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if same(mid):
            lo = mid
        else:
            hi = mid
    off, meta = rounds[hi - 1]
    a, b = gpu(hi), cpu(hi)
    bad = [s for s in range(nslots) if a[s] != b[s]]
    print("first differing round: #%d kind %d K %d levels %d; slots %s" % (hi - 1, meta & 3, (meta >> 8) & 255, (meta >> 16) & 3, bad[:12]))
    rl = emit.kpad((meta >> 8) & 255) if meta & 3 == 1 else 4
    for lane in range(64):
        rec = data[off + rl * lane: off + rl * (lane + 1)]
        if rec[0] != 0xFFFF and (rec[0] // 3 in bad if meta & 3 == 1 else rec[2] // 3 in bad):
            print("  lane", lane, "record", rec)
