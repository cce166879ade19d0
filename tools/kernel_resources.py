"""Print VGPR / scratch / spill figures of the kernels in libblsgpu.so (reads the code object's notes).
usage: python3 tools/kernel_resources.py [substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.path.join(ROOT, "python-bls_amd", "csrc", "libblsgpu.so")
filt = ""
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    blob, magic = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]      # one bundle per translation unit
    for i, a in enumerate(starts):
        part, co = os.path.join(td, "fat%d.bin" % i), os.path.join(td, "lib%d.co" % i)
        open(part, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co, "--unbundle"])
        notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
        filt += subprocess.check_output(["c++filt"], input=notes, text=True)
# one "- .key: ..." list entry per kernel under amdhsa.kernels; its keys are sorted, so .name sits in the middle of the entry
for entry in re.split(r"\n\s*- (?=\.)", filt):
    m = re.search(r"\.name:\s+(.+?)\n", entry)
    if not m or ".vgpr_count" not in entry:
        continue
    n, b = m.group(1).strip(), entry
    if sys.argv[1:] and not any(k in n for k in sys.argv[1:]):
        continue
    g = lambda k: (re.search(k + r":\s+(\d+)", b) or [0, "-"])[1]
    print("%-90s vgpr %4s agpr %3s scratch %5s spills %4s lds %6s" % (n[:90], g(r"\.vgpr_count"), g(r"\.agpr_count"), g(r"\.private_segment_fixed_size"), g(r"\.vgpr_spill_count"), g(r"\.group_segment_fixed_size")))
