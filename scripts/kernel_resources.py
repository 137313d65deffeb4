"""Static resource usage of every kernel in liblance_hip.so (no GPU needed): compiles each .hip to gfx950 assembly and reads
the .amdhsa_* directives -> profiles/rNN_kernel_resources.csv (VGPRs incl. AGPRs, SGPRs, static LDS, scratch = spills,
waves/SIMD allowed by the VGPR allocation: 512 registers per lane, granule 8, at most 8 waves)."""
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r01_kernel_resources.csv")
tmp = tempfile.mkdtemp()
kern = []
for src in sorted(glob.glob(os.path.join(ROOT, "lance_amd", "csrc", "*.hip"))):
    asm = os.path.join(tmp, os.path.basename(src) + ".s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-x", "hip", "-c", src,
                           "--cuda-device-only", "-S", "-o", asm], stderr=subprocess.DEVNULL)
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", open(asm).read(), re.S):
        kern.append((os.path.basename(src), m.group(1), m.group(2)))
names = subprocess.run(["c++filt"], input="\n".join(k[1] for k in kern), capture_output=True, text=True).stdout.split("\n")
rows = []
for (src, _, body), name in zip(kern, names):
    def g(key):
        mm = re.search(r"\.amdhsa_" + key + r" (\d+)", body)
        return int(mm.group(1)) if mm else -1
    vg = g("next_free_vgpr")
    alloc = ((vg + 7) // 8) * 8 if vg > 0 else 8
    rows.append((src, re.sub(r"\(.*", "", name).replace("void ", "").replace("lh::", ""), vg, g("next_free_sgpr"),
                 g("group_segment_fixed_size"), g("private_segment_fixed_size"), min(8, 512 // alloc)))
rows.sort()
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["source", "kernel", "vgprs(arch+acc)", "sgprs", "static_lds_bytes", "scratch_bytes", "waves_per_simd_by_vgprs"])
    w.writerows(rows)
print(len(rows), "kernels,", sum(1 for r in rows if r[5] > 0), "with scratch ->", out)
