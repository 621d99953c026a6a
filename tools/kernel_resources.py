#!/usr/bin/env python3
"""VGPRs, scratch bytes (spills) and the waves per SIMD they allow, for every kernel instantiation of cedr_kernels.hip,
read from the gfx950 assembly hipcc generates (no GPU needed). usage: kernel_resources.py [extra hipcc flags ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "compose_amd", "csrc", "cedr_kernels.hip")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-x", "hip", "-S", "--cuda-device-only"]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[1:] + ["-o", out, SRC], stderr=subprocess.DEVNULL)
    text = open(out).read()
res = {}
for m in re.finditer(r"\.set (_Z\w+)\.(num_vgpr|num_agpr|private_seg_size), (\d+)", text):
    res.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
names = subprocess.run(["c++filt"] + list(res), capture_output=True, text=True).stdout.splitlines()
print("flags: %s" % " ".join(sys.argv[1:]) if sys.argv[1:] else "flags: (product build)")
print("%-86s %5s %8s %11s" % ("kernel", "VGPRs", "scratch", "waves/SIMD"))
for mangled, name in sorted(zip(res, names), key=lambda kv: kv[1]):
    r = res[mangled]
    if "num_vgpr" not in r:
        continue
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"^void ", "", short).split("(")[0]
    v = r["num_vgpr"] + r.get("num_agpr", 0)
    print("%-86s %5d %7dB %11d" % (short[:86], v, r.get("private_seg_size", 0), min(8, 512 // max(8, (v + 7) // 8 * 8))))
