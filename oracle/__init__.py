"""CPU oracle for the EFG hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  ``efg_amd`` never imports it: the product path fails
loudly when its HIP library is missing.

* ``liboracle.so``      -- our plain-C restatement (``oracle/efg_oracle.c``; every function
  cites the reference file:line it follows).
* ``_ref/libefg_ref.so`` -- the reference's *own* ``voxelization_cpu.cpp`` and ``scatter_points_cpu.cpp`` compiled in place from
  ``/root/reference`` by ``oracle/Makefile`` (git-ignored build output; it travels to the GPU box
  like any other built ``.so``).  Used to pin the restatement and as the ``"reference"`` CPU
  baseline for voxelization.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(with_ref=True):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if with_ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(with_ref=False)
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_hard_voxelize.restype = ctypes.c_int
        _LIB.oracle_scatter_forward.restype = ctypes.c_int64
        _LIB.oracle_spconv_out_indices.restype = ctypes.c_int64
        _LIB.oracle_nms.restype = ctypes.c_int
        _LIB.oracle_lsap.restype = ctypes.c_int
    return _LIB


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libefg_ref.so"))


def ref():
    """The reference's own CPU voxelizer (oracle/_ref).  Needs libtorch (import torch first)."""
    global _REF
    if _REF is None:
        import torch  # noqa: F401  (loads libtorch_cpu / libc10 into the process)

        _REF = ctypes.CDLL(os.path.join(_HERE, "_ref", "libefg_ref.so"))
        _REF.ref_hard_voxelize_cpu.restype = ctypes.c_int
        if hasattr(_REF, "ref_dynamic_point_to_voxel_run"):
            _REF.ref_dynamic_point_to_voxel_run.restype = ctypes.c_int
    return _REF


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def _i3(v):
    return (ctypes.c_int * 3)(*[int(x) for x in v])


# ------------------------------------------------------------------------------------------------
# voxelization
# ------------------------------------------------------------------------------------------------
def dynamic_voxelize(points, voxel_size, coors_range, use_ref=False):
    points = _f(points)
    n, f = points.shape
    vs, cr = _f(voxel_size), _f(coors_range)
    coors = np.zeros((n, 3), np.int32)
    if use_ref:
        ref().ref_dynamic_voxelize_cpu(_p(points, _f32p), ctypes.c_long(n), f, _p(vs, _f32p), _p(cr, _f32p),
                                       _p(coors, _i32p))
    else:
        lib().oracle_dynamic_voxelize(_p(points, _f32p), ctypes.c_int64(n), f, _p(vs, _f32p), _p(cr, _f32p),
                                      _p(coors, _i32p))
    return coors


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels, use_ref=False):
    """Returns (voxels[M,max_points,F], coors[M,3] zyx, num_points_per_voxel[M]) like
    efg/operators/voxelize.py:39-49."""
    points = _f(points)
    n, f = points.shape
    vs, cr = _f(voxel_size), _f(coors_range)
    voxels = np.zeros((max_voxels, max_points, f), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    npv = np.zeros((max_voxels,), np.int32)
    if use_ref:
        m = ref().ref_hard_voxelize_cpu(_p(points, _f32p), ctypes.c_long(n), f, _p(vs, _f32p), _p(cr, _f32p),
                                        max_points, max_voxels, _p(voxels, _f32p), _p(coors, _i32p),
                                        _p(npv, _i32p))
    else:
        m = lib().oracle_hard_voxelize(_p(points, _f32p), ctypes.c_int64(n), f, _p(vs, _f32p), _p(cr, _f32p),
                                       max_points, max_voxels, _p(voxels, _f32p), _p(coors, _i32p),
                                       _p(npv, _i32p))
    return voxels[:m], coors[:m], npv[:m]


def voxel_mean(voxels, npv, nfeat=None):
    voxels = _f(voxels)
    m, mp, f = voxels.shape
    nfeat = f if nfeat is None else nfeat
    out = np.zeros((m, nfeat), np.float32)
    npv = _i(npv)
    lib().oracle_voxel_mean(_p(voxels, _f32p), _p(npv, _i32p), ctypes.c_int64(m), mp, f, nfeat, _p(out, _f32p))
    return out


# ------------------------------------------------------------------------------------------------
# dynamic scatter
# ------------------------------------------------------------------------------------------------
_REDUCE = {"sum": 0, "mean": 1, "max": 2}


def scatter_forward(feats, coors, reduce_type):
    feats, coors = _f(feats), _i(coors)
    n, c = feats.shape
    ndim = coors.shape[1]
    vf = np.zeros((max(n, 1), c), np.float32)
    vc = np.zeros((max(n, 1), ndim), np.int32)
    p2v = np.zeros((n,), np.int32)
    cnt = np.zeros((max(n, 1),), np.int32)
    m = lib().oracle_scatter_forward(_p(feats, _f32p), _p(coors, _i32p), ctypes.c_int64(n), c, ndim,
                                     _REDUCE[reduce_type], _p(vf, _f32p), _p(vc, _i32p), _p(p2v, _i32p),
                                     _p(cnt, _i32p))
    return vf[:m], vc[:m], p2v, cnt[:m]


def ref_dynamic_point_to_voxel(points, coors, voxel_size, coors_range):
    """The REFERENCE's own CPU grouping (scatter_points_cpu.cpp:62-119, compiled into oracle/_ref): every point of
    each voxel, voxels in first-occurrence order.  coors i32 [n,3] (z,y,x), all rows inside the grid.
    Returns (voxels [M, max_points, F] zero padded, voxel_coors [M,3], num_points_per_voxel [M])."""
    points, coors = _f(points), _i(coors)
    assert coors.min() >= 0, "the reference kernel indexes its grid with every row: no -1 rows"
    n, f = points.shape
    vs, cr = _f(voxel_size), _f(coors_range)
    mp = ctypes.c_int(0)
    m = ref().ref_dynamic_point_to_voxel_run(_p(points, _f32p), ctypes.c_long(n), f, _p(coors, _i32p), _p(vs, _f32p),
                                             _p(cr, _f32p), ctypes.byref(mp))
    voxels = np.zeros((m, mp.value, f), np.float32)
    vc, npv = np.zeros((m, 3), np.int32), np.zeros((m,), np.int32)
    ref().ref_dynamic_point_to_voxel_fetch(_p(voxels, _f32p), _p(vc, _i32p), _p(npv, _i32p))
    return voxels, vc, npv


def scatter_backward(grad_voxel, feats, voxel_feats, p2v, count, reduce_type):
    feats, grad_voxel, voxel_feats = _f(feats), _f(grad_voxel), _f(voxel_feats)
    p2v, count = _i(p2v), _i(count)
    n, c = feats.shape
    m = voxel_feats.shape[0]
    g = np.zeros((n, c), np.float32)
    lib().oracle_scatter_backward(_p(g, _f32p), _p(grad_voxel, _f32p), _p(feats, _f32p), _p(voxel_feats, _f32p),
                                  _p(p2v, _i32p), _p(count, _i32p), ctypes.c_int64(n), ctypes.c_int64(m), c,
                                  _REDUCE[reduce_type])
    return g


# ------------------------------------------------------------------------------------------------
# sparse convolution
# ------------------------------------------------------------------------------------------------
def spconv_out_indices(in_idx, batch, in_shape, ksize, stride, pad):
    in_idx = _i(in_idx)
    m_in = in_idx.shape[0]
    kvol = int(np.prod(ksize))
    cap = max(m_in * kvol, 1)
    out = np.zeros((cap, 4), np.int32)
    oshape = (ctypes.c_int * 3)()
    m = lib().oracle_spconv_out_indices(_p(in_idx, _i32p), ctypes.c_int64(m_in), batch, _i3(in_shape), _i3(ksize),
                                        _i3(stride), _i3(pad), _p(out, _i32p), ctypes.c_int64(cap), oshape)
    return out[:m].copy(), [int(x) for x in oshape]


def spconv_rulebook(in_idx, out_idx, batch, in_shape, ksize, stride, pad):
    in_idx, out_idx = _i(in_idx), _i(out_idx)
    kvol = int(np.prod(ksize))
    nbr = np.zeros((kvol, max(out_idx.shape[0], 1)), np.int32)[:, : out_idx.shape[0]]
    nbr = np.ascontiguousarray(nbr)
    lib().oracle_spconv_rulebook(_p(in_idx, _i32p), ctypes.c_int64(in_idx.shape[0]), _p(out_idx, _i32p),
                                 ctypes.c_int64(out_idx.shape[0]), batch, _i3(in_shape), _i3(ksize), _i3(stride),
                                 _i3(pad), _p(nbr, _i32p))
    return nbr


def spconv_forward(in_feat, weight, bias, nbr):
    """weight: [cout, kvol, cin] (spconv 2.x KRSC flattened)."""
    in_feat, weight, nbr = _f(in_feat), _f(weight), _i(nbr)
    cout, kvol, cin = weight.shape
    m_out = nbr.shape[1]
    out = np.zeros((m_out, cout), np.float32)
    b = None if bias is None else _f(bias)
    lib().oracle_spconv_forward(_p(in_feat, _f32p), ctypes.c_int64(in_feat.shape[0]), cin, _p(weight, _f32p),
                                _p(b, _f32p) if b is not None else None, cout, kvol, _p(nbr, _i32p),
                                ctypes.c_int64(m_out), _p(out, _f32p))
    return out


def spconv_dgrad(grad_out, weight, nbr, m_in):
    grad_out, weight, nbr = _f(grad_out), _f(weight), _i(nbr)
    cout, kvol, cin = weight.shape
    g = np.zeros((m_in, cin), np.float32)
    lib().oracle_spconv_dgrad(_p(grad_out, _f32p), ctypes.c_int64(grad_out.shape[0]), cout, _p(weight, _f32p), cin,
                              kvol, _p(nbr, _i32p), ctypes.c_int64(m_in), _p(g, _f32p))
    return g


def spconv_wgrad(in_feat, grad_out, nbr):
    in_feat, grad_out, nbr = _f(in_feat), _f(grad_out), _i(nbr)
    cin, cout, kvol = in_feat.shape[1], grad_out.shape[1], nbr.shape[0]
    gw = np.zeros((cout, kvol, cin), np.float32)
    lib().oracle_spconv_wgrad(_p(in_feat, _f32p), ctypes.c_int64(in_feat.shape[0]), cin, _p(grad_out, _f32p),
                              ctypes.c_int64(grad_out.shape[0]), cout, kvol, _p(nbr, _i32p), _p(gw, _f32p))
    return gw


def sparse_to_dense(feat, idx, batch, shape):
    feat, idx = _f(feat), _i(idx)
    m, c = feat.shape
    dense = np.zeros((batch, c, shape[0], shape[1], shape[2]), np.float32)
    lib().oracle_sparse_to_dense(_p(feat, _f32p), _p(idx, _i32p), ctypes.c_int64(m), c, batch, _i3(shape),
                                 _p(dense, _f32p))
    return dense


# ------------------------------------------------------------------------------------------------
# box / deformable attention
# ------------------------------------------------------------------------------------------------
def msda_forward(value, shapes, level_start, loc, attn):
    value, loc, attn = _f(value), _f(loc), _f(attn)
    shapes = np.ascontiguousarray(shapes, np.int64)
    level_start = np.ascontiguousarray(level_start, np.int64)
    b, s, h, d = value.shape
    _, lq, _, l, p, _ = loc.shape
    out = np.zeros((b, lq, h * d), np.float32)
    lib().oracle_msda_forward(_p(value, _f32p), _p(shapes, _i64p), _p(level_start, _i64p), _p(loc, _f32p),
                              _p(attn, _f32p), b, s, h, d, l, lq, p, _p(out, _f32p))
    return out


def msda_backward(value, shapes, level_start, loc, attn, grad_out):
    value, loc, attn, grad_out = _f(value), _f(loc), _f(attn), _f(grad_out)
    shapes = np.ascontiguousarray(shapes, np.int64)
    level_start = np.ascontiguousarray(level_start, np.int64)
    b, s, h, d = value.shape
    _, lq, _, l, p, _ = loc.shape
    gv, gl, ga = np.zeros_like(value), np.zeros_like(loc), np.zeros_like(attn)
    lib().oracle_msda_backward(_p(value, _f32p), _p(shapes, _i64p), _p(level_start, _i64p), _p(loc, _f32p),
                               _p(attn, _f32p), _p(grad_out, _f32p), b, s, h, d, l, lq, p, _p(gv, _f32p),
                               _p(gl, _f32p), _p(ga, _f32p))
    return gv, gl, ga


# ------------------------------------------------------------------------------------------------
# next row n1: rotated BEV IoU / NMS
# ------------------------------------------------------------------------------------------------
def boxes_bev(boxes_a, boxes_b, mode="iou"):
    """[N,7] x [M,7] (x,y,z,dx,dy,dz,heading) -> [N,M] BEV overlap area ("overlap") or IoU ("iou")."""
    a, b = _f(boxes_a), _f(boxes_b)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().oracle_boxes_bev(_p(a, _f32p), a.shape[0], _p(b, _f32p), b.shape[0], 1 if mode == "iou" else 0,
                           _p(out, _f32p))
    return out


def boxes_iou3d(boxes_a, boxes_b):
    """[N,7] x [M,7] -> [N,M] 3-D IoU: BEV overlap x height overlap over the union volume, composed in fp32 in the
    order of efg/operators/iou3d_nms.py:54-87."""
    a, b = _f(boxes_a), _f(boxes_b)
    top = np.minimum((a[:, 2] + a[:, 5] / 2)[:, None], (b[:, 2] + b[:, 5] / 2)[None, :])
    bottom = np.maximum((a[:, 2] - a[:, 5] / 2)[:, None], (b[:, 2] - b[:, 5] / 2)[None, :])
    overlap = boxes_bev(a, b, "overlap") * np.maximum(top - bottom, np.float32(0))
    vol_a = (a[:, 3] * a[:, 4] * a[:, 5])[:, None]
    vol_b = (b[:, 3] * b[:, 4] * b[:, 5])[None, :]
    return (overlap / np.maximum(vol_a + vol_b - overlap, np.float32(1e-6))).astype(np.float32)


def nms(boxes_sorted, thresh, rotated=True):
    """Greedy NMS over boxes already sorted by descending score -> kept indices (int64)."""
    b = _f(boxes_sorted)
    keep = np.zeros((max(b.shape[0], 1),), np.int64)
    n = lib().oracle_nms(_p(b, _f32p), b.shape[0], ctypes.c_float(thresh), 1 if rotated else 0, _p(keep, _i64p))
    return keep[:n]


# ------------------------------------------------------------------------------------------------
# next row "GPU matcher": linear sum assignment
# ------------------------------------------------------------------------------------------------
def lsap(cost, ng=None):
    """cost [nq, g_stride] float32, first `ng` columns valid -> query_of_gt int64 [ng] (-1 = unmatched)."""
    c = _f(cost)
    nq, gs = c.shape
    ng = gs if ng is None else int(ng)
    out = np.full((max(ng, 1),), -1, np.int64)
    rc = lib().oracle_lsap(_p(c, _f32p), nq, gs, ng, _p(out, _i64p))
    if rc != 0:
        raise ValueError("cost matrix is infeasible")
    return out[:ng]


# ------------------------------------------------------------------------------------------------
# TrajectoryFormer point crop: membership of points in vertical cylinders (checker for csrc/crop.hip)
# ------------------------------------------------------------------------------------------------
def cylinder_select(points, point_range, centre_radius, time_col=-1, max_time=1.0):
    """numpy restatement of the membership test of $TF/modules/utils.py:361-402 (fp32: sqrt(dx*dx + dy*dy) <= r,
    optional time gate).  Returns (counts [R], list of index arrays relative to each cylinder's range start)."""
    pts = _f(points)
    counts, lists = [], []
    for (lo, hi), (x, y, r) in zip(np.asarray(point_range, np.int64), _f(centre_radius)):
        seg = pts[lo:hi]
        dx, dy = seg[:, 0] - np.float32(x), seg[:, 1] - np.float32(y)
        inside = np.sqrt(dx * dx + dy * dy, dtype=np.float32) <= np.float32(r)
        if time_col >= 0:
            inside &= seg[:, time_col] < np.float32(max_time)
        idx = np.nonzero(inside)[0].astype(np.int32)
        counts.append(len(idx))
        lists.append(idx)
    return np.asarray(counts, np.int32), lists
