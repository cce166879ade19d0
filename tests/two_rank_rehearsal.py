#!/usr/bin/env python3
"""Started by tests/conftest.py (pytest_sessionstart, GPU runs only): two bench.py runs with TWO ranks on GPU 0 over gloo,
one after the other; every run's JSON line goes to <out>/<name>.json, its stderr to <out>/<name>.err.  This process never
touches the GPU itself (bench.py starts its ranks as children before anything does)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--no-latency", "--no-cpu-baseline", "--no-secondary"]
RUNS = {
    # BASELINE configs[2]'s shape at 8192 pairs: ONE multi-pairing sharded two ways (4096 pairs per rank), one partial per
    # rank through the all-gather, the final exponentiation on both ranks; check = the reference's digest
    "c3_8192": ["--config", "c3", "--pairs-total", "8192"] + COMMON,
    # the default (weak) mode: 3 verifications per step, every rank holds 4096 pairs of each, B x 576 bytes per rank
    # through the all-gather; verification 0 is the multi-pairing of the PRF pairs 0 .. 8191: the reference's digest again
    "weak_3x4096": ["--pairs", "4096", "--verifications", "3"] + COMMON,
}


def main():
    out = sys.argv[1]
    rc = 0
    for name, args in RUNS.items():
        with open(os.path.join(out, name + ".json"), "w") as o, open(os.path.join(out, name + ".err"), "w") as e:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=o, stderr=e, cwd=ROOT, timeout=900)
        rc = rc or r.returncode
        with open(os.path.join(out, name + ".rc"), "w") as f:
            f.write(str(r.returncode))
    sys.exit(rc)


if __name__ == "__main__":
    main()
