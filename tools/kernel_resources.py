"""Print VGPR / scratch / spill figures of the kernels in libblsgpu.so (reads the code object's notes).
usage: python3 tools/kernel_resources.py [substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.path.join(ROOT, "python-bls_amd", "csrc", "libblsgpu.so")
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "lib.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
    notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
    filt = subprocess.check_output(["c++filt"], input=notes, text=True)
for m in re.finditer(r"\.name:\s+(.+?)\n(.*?)(?=\.name:|\Z)", filt, re.S):
    n, b = m.group(1).strip(), m.group(2)
    if sys.argv[1:] and not any(k in n for k in sys.argv[1:]):
        continue
    g = lambda k: (re.search(k + r":\s+(\d+)", b) or [0, "-"])[1]
    print("%-90s vgpr %4s agpr %3s scratch %5s spills %4s lds %6s" % (n[:90], g(r"\.vgpr_count"), g(r"\.agpr_count"), g(r"\.private_segment_fixed_size"), g(r"\.vgpr_spill_count"), g(r"\.group_segment_fixed_size")))
