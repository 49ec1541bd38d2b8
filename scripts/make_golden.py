"""Generate tests/golden/*.npz from the REFERENCE (run in the build container only).

Needs /root/reference.  Nothing from the reference is copied: the reference code is imported /
executed in place and only its input/output VECTORS are stored.

  voxelize_*.npz  efg/operators/src/voxelize/voxelization_cpu.cpp (compiled as oracle/_ref) and the
                  numba twin efg/geometry/point_cloud_ops.py:5-53 (imported with an identity-jit shim,
                  numba is not installed) on synthetic clouds.
  msda_*.npz      efg/operators/ms_deform_attn.py:55-76 ms_deform_attn_core_pytorch (forward) and
                  torch autograd through it (backward).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

import oracle  # noqa: E402
from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene  # noqa: E402


def load_numba_twin():
    numba = types.ModuleType("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    numba.jit = jit
    sys.modules["numba"] = numba
    spec = importlib.util.spec_from_file_location("ref_point_cloud_ops", REF + "/efg/geometry/point_cloud_ops.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_msda_ref():
    efg = types.ModuleType("efg")
    efg.__path__ = [REF + "/efg"]
    c = types.ModuleType("efg._C")
    efg._C = c
    sys.modules["efg"] = efg
    sys.modules["efg._C"] = c
    spec = importlib.util.spec_from_file_location("efg.operators.ms_deform_attn",
                                                  REF + "/efg/operators/ms_deform_attn.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.ms_deform_attn_core_pytorch


def voxel_cases():
    rng = np.random.default_rng(7)
    cases = {}
    # (a) config-0 sized cloud, caps not hit
    p, _, _ = make_scene(1000, n_points=16000)
    cases["cfg0_16k"] = (p, VOXEL_SIZE, PC_RANGE, 5, 120000)
    # (b) max_voxels hit early -> `break` semantics (voxelization_cpu.cpp:78), small max_points
    p, _, _ = make_scene(1001, n_points=6000)
    cases["break_6k"] = (p, (0.8, 0.8, 0.5), PC_RANGE, 3, 700)
    # (c) dense small range: many points per voxel, boundary / outside points, exact-boundary values
    q = rng.uniform(-1.2, 1.2, (5000, 4)).astype(np.float32)
    q[:200, 0] = np.round(q[:200, 0] * 10) / 10          # on voxel faces
    q[200:260, 1] = 1.0                                    # on the open upper bound -> outside
    q[260:300, 2] = -1.0                                   # on the closed lower bound -> inside
    cases["dense_boundary"] = (q, (0.1, 0.1, 0.2), (-1.0, -1.0, -1.0, 1.0, 1.0, 1.0), 4, 3000)
    # (d) 4-sweep style 6-feature cloud
    p, _, _ = make_scene(4000, n_points=12000, n_sweeps=4)
    cases["sweep4_12k"] = (p, VOXEL_SIZE, PC_RANGE, 5, 4000)
    return cases


def gen_voxelize():
    twin = load_numba_twin()
    for name, (pts, vs, cr, mp, mv) in voxel_cases().items():
        v, c, n = oracle.hard_voxelize(pts, vs, cr, mp, mv, use_ref=True)
        d = oracle.dynamic_voxelize(pts, vs, cr, use_ref=True)
        if pts.shape[0] <= 6000:  # numba twin is pure python here (~1 ms / point)
            v2, c2, n2 = twin.points_to_voxel(pts, np.array(vs, np.float32), np.array(cr, np.float32), mp, True, mv)
            assert np.array_equal(v, v2) and np.array_equal(c, c2) and np.array_equal(n, n2), name
            print(name, "numba twin == C++ reference")
        np.savez_compressed(os.path.join(OUT, "voxelize_%s.npz" % name), points=pts,
                            voxel_size=np.array(vs, np.float32), coors_range=np.array(cr, np.float32),
                            max_points=mp, max_voxels=mv, voxels=v, coors=c, num_points_per_voxel=n,
                            dynamic_coors=d)
        print(name, pts.shape, "->", v.shape)


def gen_msda():
    core = load_msda_ref()
    g = torch.Generator().manual_seed(0)
    cases = {
        # box-attention shaped: 1 level, 8 heads x 32 ch, 25 points (ConQueR), locations spill past the border
        "box_l1_h8_d32_p25": dict(b=2, shapes=[(20, 18)], h=8, d=32, lq=37, p=25, lo=-0.15, hi=1.15),
        # deformable-DETR shaped: 4 levels x 4 points
        "msda_l4_h4_d8_p4": dict(b=2, shapes=[(12, 10), (6, 5), (3, 3), (2, 1)], h=4, d=8, lq=50, p=4, lo=-0.1, hi=1.1),
        # odd head dim (multiple of 4, not a power of two)
        "msda_l2_h3_d12_p3": dict(b=1, shapes=[(7, 9), (4, 4)], h=3, d=12, lq=11, p=3, lo=0.0, hi=1.0),
    }
    for name, c in cases.items():
        shapes = torch.tensor(c["shapes"], dtype=torch.int64)
        s = int((shapes[:, 0] * shapes[:, 1]).sum())
        start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
        l = len(c["shapes"])
        value = torch.randn(c["b"], s, c["h"], c["d"], generator=g)
        loc = torch.rand(c["b"], c["lq"], c["h"], l, c["p"], 2, generator=g) * (c["hi"] - c["lo"]) + c["lo"]
        attn = torch.softmax(torch.randn(c["b"], c["lq"], c["h"], l * c["p"], generator=g), -1).view(
            c["b"], c["lq"], c["h"], l, c["p"])
        go = torch.randn(c["b"], c["lq"], c["h"] * c["d"], generator=g)
        v64, l64, a64 = (t.double().requires_grad_(True) for t in (value, loc, attn))
        out64 = core(v64, c["shapes"], l64, a64)
        gv, gl, ga = torch.autograd.grad(out64, (v64, l64, a64), go.double())
        out32 = core(value, c["shapes"], loc, attn)
        np.savez_compressed(os.path.join(OUT, "msda_%s.npz" % name), value=value.numpy(), shapes=shapes.numpy(),
                            level_start=start.numpy(), loc=loc.numpy(), attn=attn.numpy(), grad_out=go.numpy(),
                            out_fp32=out32.numpy(), out_fp64=out64.detach().numpy(), grad_value=gv.numpy(),
                            grad_loc=gl.numpy(), grad_attn=ga.numpy())
        print(name, tuple(out32.shape))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_voxelize()
    gen_msda()
