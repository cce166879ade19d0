"""Rate of the HOST-buffer entry points (PCIe copies included), for DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
from bls_py import _native
e = _native.Engine(0)
g1 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g1.bin"), "rb").read()
g2 = open(os.path.join(ROOT, "tests/golden/pairs_seed1_g2.bin"), "rb").read()
n = 1025
e.pairing_multi(g1, g2, n)
t = time.perf_counter()
for _ in range(20):
    e.pairing_multi(g1, g2, n)
dt = (time.perf_counter() - t) / 20
print("blsgpu_pairing_multi, 1025 pairs from host buffers: %.3f ms  %.0f pairings/s" % (dt * 1e3, n / dt))
B = 32
a, b = g1 * B, g2 * B
e.set_mp_threshold(0)
e.pairing_multi_batch(a, b, n, B)
t = time.perf_counter()
for _ in range(5):
    e.pairing_multi_batch(a, b, n, B)
dt = (time.perf_counter() - t) / 5
print("blsgpu_pairing_multi_batch, 32 x 1025 pairs from host buffers: %.3f ms  %.0f pairings/s" % (dt * 1e3, n * B / dt))
