"""Compile libefg_hip.so (gfx950) in-tree with hipcc.  `python -m efg_amd.build`."""
import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libefg_hip.so")

SOURCES = ["lib.cpp", "voxelize.hip", "voxelize_bins.hip", "voxelize_hash.hip", "scatter.hip", "spconv_index.hip", "spconv_conv.hip", "spconv_tiles.hip", "spconv_wgt.hip", "msda.hip", "box_fused.hip", "iou3d_nms.hip", "matcher.hip", "layernorm.hip", "augment.hip", "det_loss.hip", "batchnorm.hip", "colsum.hip", "crop.hip", "attention.hip", "gemm_bf16x3.hip", "topk.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-I" + os.path.join(_ROOT, "include"), "-I" + CSRC]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc every translation unit to an object, then link libefg_hip.so.  Cross-compiles
    without a GPU."""
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(_ROOT, "include", "efg_hip.h"))
    objs = []
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [path] + headers):
            cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    if force or procs or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
