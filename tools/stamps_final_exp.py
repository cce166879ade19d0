"""Diagnostic (-DBLSGPU_STAMPS build): where a lone wavefront's final exponentiation spends its cycles."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
os.environ["BLSGPU_LIBRARY"] = os.path.join(ROOT, "python-bls_amd", "csrc", "libblsgpu_stamps.so")
from bls_py import _native
e = _native.Engine(0)
e.lib.blsgpu_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong * 9)]
buf = (ctypes.c_ulonglong * 9)()
x = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests/golden/pairing.json")))["gen"]["miller"])
for rep in range(3):
    e.final_exp(x)
    e.lib.blsgpu_debug_stamps(e.h, ctypes.byref(buf))
    cyc, cnt = list(buf[:3]), list(buf[3:6])
    print("final exp: " + " ".join("%s: %d rounds, %.0f ticks/round (%.3f Mticks)" % (k, c, (t / c if c else 0), t / 1e6)
                                   for k, t, c in zip(("MUL", "LIN", "INV"), cyc, cnt)),
          "| LIN accumulate %.0f, reduce %.0f ticks/round, %.1f micro-ops/round" % (buf[6] / cnt[1], buf[7] / cnt[1], buf[8] / cnt[1]))
