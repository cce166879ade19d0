"""Diagnostic: per-round-kind cycle shares from the -DBLSGPU_STAMPS build."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
os.environ["BLSGPU_LIBRARY"] = os.path.join(ROOT, "python-bls_amd", "csrc", "libblsgpu_stamps.so")
from bls_py import _native
e = _native.Engine(0)
e.lib.blsgpu_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong * 9)]
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()
buf = (ctypes.c_ulonglong * 9)()
for n in (1025, 1025, 4, 8200, 32800):
    reps = (n + 1024) // 1025
    e.pairing_multi((g1 * reps)[:96 * n], (g2 * reps)[:192 * n], n)
    e.lib.blsgpu_debug_stamps(e.h, ctypes.byref(buf))
    cyc, cnt = list(buf[:3]), list(buf[3:6])
    print("   LIN: accumulate %.0f cyc/round, reduce %.0f cyc/round, uops/round %.1f" % (buf[6] / cnt[1], buf[7] / cnt[1], buf[8] / cnt[1]))
    print("n=%d" % n, " ".join("%s: %d rounds, %.0f cyc/round" % (k, c, (t / c if c else 0))
                                 for k, t, c in zip(("MUL", "LIN", "INV"), cyc, cnt)), "total Mcyc %.2f" % (sum(cyc) / 1e6))
