"""End-to-end BASELINE configs[1] through the host mirror: 1024 (sk, msg) -> sign (GPU
batched) -> aggregate -> verify, with wall times of the Python-visible steps."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
from bls_py.bls import BLS
from bls_py.keys import PrivateKey
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
order = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
sks = [PrivateKey(int.from_bytes(hashlib.sha256(b"blsgpu/a" + (1).to_bytes(4, "big") + i.to_bytes(4, "big")).digest(), "big")
                  % (order - 1) + 1) for i in range(n)]
msgs = [i.to_bytes(4, "big") for i in range(n)]
PrivateKey.sign_batch(sks[:4], msgs[:4])          # warm up (library load, context)
t0 = time.perf_counter(); sigs = PrivateKey.sign_batch(sks, msgs); t1 = time.perf_counter()
agg = BLS.aggregate_sigs(sigs); t2 = time.perf_counter()
ok = BLS.verify(agg); t3 = time.perf_counter()
ok2 = BLS.verify(agg); t4 = time.perf_counter()
print("n=%d sign_batch %.3f s, aggregate_sigs %.3f s, verify %.3f s (again %.3f s) -> %s" % (n, t1 - t0, t2 - t1, t3 - t2, t4 - t3, ok and ok2))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); BLS.verify(agg); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
