#!/usr/bin/env python3
"""Register / LDS / occupancy report of the kernels in one csrc file (compile-time, no GPU):
    scripts/kernel_regs.py spconv_tiles.hip [name-filter] [--rev GIT_REV]      (KREGS_FLAGS="-DX=1 ..." adds compiler flags)"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
src, filt = args[0], (args[1] if len(args) > 1 else "")
path = os.path.join(ROOT, "efg_amd", "csrc", src)
if "--rev" in sys.argv:
    rev = sys.argv[sys.argv.index("--rev") + 1]
    text = subprocess.check_output(["git", "-C", ROOT, "show", "%s:efg_amd/csrc/%s" % (rev, src)])
    path = "/tmp/kregs_" + src
    open(path, "wb").write(text)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
       "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "efg_amd", "csrc"), "-x", "hip",
       "-c", path, "-o", "/tmp/kregs.o", "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("KREGS_FLAGS", "").split()
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("efg::(anonymous namespace)::", "").split("(")[0]
    if filt and filt not in n:
        continue
    print("%-48s vgpr %3d agpr %3d occ %d lds %6d spill %d scratch %d" % (
        n[:48], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("Occupancy", -1), r.get("LDS Size", -1),
        r.get("VGPRs Spill", -1), r.get("ScratchSize", -1)))
