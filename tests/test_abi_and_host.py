"""CPU-side checks of the boundary: the C-ABI library loads and exports every
function include/blsgpu.h declares; the limb arithmetic of csrc/fq32.h (compiled
for the host) matches Python integers; the product refuses to run without a GPU
instead of falling back."""
import os
import random
import re
import subprocess

import pytest

from conftest import ROOT

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
CSRC = os.path.join(ROOT, "python-bls_amd", "csrc")


def test_header_symbols_exported():
    from bls_py import _native
    with open(os.path.join(ROOT, "include", "blsgpu.h")) as f:
        hdr = f.read()
    declared = set(re.findall(r"\b(blsgpu_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _native.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), "libblsgpu.so does not export " + name
    assert declared == set(_native.SYMBOLS)
    assert lib.blsgpu_version().startswith(b"blsgpu/")


def test_no_cpu_fallback():
    """Without a GPU the engine must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from bls_py import _native
    with pytest.raises(_native.BlsGpuError):
        _native.Engine(0)


HOST_TEST = r'''
#include "fq32.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static void parse(const char* h, uint32_t* l){ for(int i=0;i<12;i++){ char b[9]; memcpy(b,h+8*(11-i),8); b[8]=0; l[i]=(uint32_t)strtoul(b,0,16);} }
static void pr(const uint32_t* l){ for(int i=11;i>=0;i--) printf("%08x", l[i]); printf("\n"); }
int main(){ char op[16], a[200], b[200];
  while(scanf("%15s %199s %199s", op,a,b)==3){ uint32_t x[12],y[12],r[12]; parse(a,x); parse(b,y);
    if(!strcmp(op,"mul")) { bls::fq_mul(r,x,y); pr(r);}
    else if(!strcmp(op,"add")) { bls::fq_add_mod(x,y); pr(x);}
    else if(!strcmp(op,"sub")) { bls::fq_neg_raw(y); bls::fq_add_mod(x,y); pr(x);}
    else if(!strcmp(op,"inv")) { bls::fq_inv(r,x); uint32_t r2[12]; bls::fq_inv_var(r2,x);          // both forms of the inversion
      if(memcmp(r,r2,48)) { printf("inv_var_differs\n"); } else pr(r);}
    else if(!strcmp(op,"jac")) { printf("%d\n", bls::fq_jacobi_var(x)); } }
  return 0; }
'''


def test_fq32_host_build_matches_python_ints(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(HOST_TEST)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", CSRC, "-o", str(exe), str(src)])
    R = 1 << 384
    Ri = pow(R, -1, Q)
    rnd = random.Random(7)
    vals = [0, 1, 2, Q - 1, Q - 2, R % Q] + [rnd.randrange(Q) for _ in range(150)]
    inp, exp = [], []
    for _ in range(400):
        a, b = rnd.choice(vals), rnd.choice(vals)
        for op in ("mul", "add", "sub", "inv"):
            inp.append("%s %096x %096x" % (op, a, b))
            exp.append({"mul": a * b * Ri % Q, "add": (a + b) % Q, "sub": (a - b) % Q,
                        "inv": (pow(a, -1, Q) * R * R) % Q if a else 0}[op])
    # the inversion (safegcd division steps) on many more values, also relaxed ones (q <= a < 2q)
    # and values with long runs of zero / one bits
    more = [rnd.randrange(Q) for _ in range(3000)] + [Q + rnd.randrange(Q) for _ in range(200)]
    more += [(1 << k) % Q for k in range(0, 384, 7)] + [(Q - (1 << k)) % Q for k in range(0, 380, 11)] + [Q, Q + 1, 2 * Q - 1]
    for a in more:
        inp.append("inv %096x %096x" % (a, 0))
        exp.append((pow(a % Q, -1, Q) * R * R) % Q if a % Q else 0)
    # the quadratic character from division steps (fq_jacobi_var) against Euler's criterion: canonical values, corners
    jvals = [0, 1, 2, 3, 4, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2, 1 << 380] + [rnd.randrange(Q) for _ in range(4000)]
    jvals += [rnd.randrange(1 << k) for k in (8, 64, 200, 380) for _ in range(100)]
    ninp = len(inp)
    for a in jvals:
        inp.append("jac %096x %096x" % (a, 0))
    out = subprocess.run([str(exe)], input="\n".join(inp) + "\n", capture_output=True, text=True).stdout.split()
    assert len(out) == len(exp) + len(jvals)
    assert [int(o, 16) for o in out[:ninp]] == exp
    euler = lambda a: {0: 0, 1: 1, Q - 1: -1}[pow(a, (Q - 1) // 2, Q)]
    assert [int(o) for o in out[ninp:]] == [euler(a) for a in jvals]
