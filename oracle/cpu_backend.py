"""Run the efg_amd MODEL on the CPU with the oracle standing in for every HIP op.

TEST INFRASTRUCTURE ONLY: used by tests/ (model-level parity HIP vs oracle, the config-0 CPU
plumbing case, world_size-2 gloo runs) and by bench.py's `cpu_baseline` leg.  It works by
monkeypatching the few functions of efg_amd that call libefg_hip.so; efg_amd itself contains no
reference to this module and no CPU branch.
"""
import contextlib

import numpy as np
import torch

import oracle


def _np(t):
    return t.detach().cpu().numpy()


class _CpuSiteIndex:
    def __init__(self, indices, batch_size, spatial_shape):
        self.indices = indices
        self.batch_size = batch_size
        self.spatial_shape = tuple(spatial_shape)
        self.index = None
        self.perm = None


@contextlib.contextmanager
def install():
    import efg_amd._lib as L
    import efg_amd.operators.box_attention_func as baf
    import efg_amd.operators.ms_deform_attn as mda
    import efg_amd.operators.scatter_points as sp
    import efg_amd.operators.voxelize as vz
    import efg_amd.spconv.core as core

    saved = []

    def patch(mod, name, fn):
        saved.append((mod, name, getattr(mod, name)))
        setattr(mod, name, fn)

    patch(L, "require_gpu", lambda *a: None)
    patch(baf, "FUSED_ENABLED", False)  # CPU: the reference sequence (grid + softmax + sampling op)

    # ---- voxelization -------------------------------------------------------------------------
    def hard_launch(points, offsets, voxel_size, coors_range, max_points, max_voxels, voxels, coors, npv, voxel_num,
                    mean):
        pts = _np(points)
        base = 0
        for b in range(len(offsets) - 1):
            v, c, n = oracle.hard_voxelize(pts[offsets[b]:offsets[b + 1]], voxel_size, coors_range, max_points,
                                           max_voxels)
            m = v.shape[0]
            voxels[base:base + m] = torch.from_numpy(v)
            if coors.shape[1] == 4:
                coors[base:base + m, 0] = b
                coors[base:base + m, 1:] = torch.from_numpy(c)
            else:
                coors[base:base + m] = torch.from_numpy(c)
            npv[base:base + m] = torch.from_numpy(n)
            if mean is not None:
                mean[base:base + m] = torch.from_numpy(oracle.voxel_mean(v, n))
            voxel_num[b] = m
            base += m

    def dyn(points, coors, voxel_size, coors_range, NDim=3):
        coors.copy_(torch.from_numpy(oracle.dynamic_voxelize(_np(points), voxel_size, coors_range)))

    patch(vz, "_hard_voxelize_launch", hard_launch)
    patch(vz, "dynamic_voxelize", dyn)

    # ---- dynamic scatter ----------------------------------------------------------------------
    def sc_fwd(feats, coors, reduce_type):
        sp._reduce_id(reduce_type)
        vf, vc, p2v, cnt = oracle.scatter_forward(_np(feats), _np(coors), reduce_type)
        return [torch.from_numpy(np.ascontiguousarray(a)) for a in (vf, vc, p2v, cnt)]

    def sc_bwd(grad_feats, gv, feats, vf, p2v, cnt, reduce_type):
        grad_feats.copy_(torch.from_numpy(oracle.scatter_backward(_np(gv), _np(feats), _np(vf), _np(p2v), _np(cnt),
                                                                   reduce_type)))

    patch(sp, "dynamic_point_to_voxel_forward", sc_fwd)
    patch(sp, "dynamic_point_to_voxel_backward", sc_bwd)

    # ---- box / deformable attention -----------------------------------------------------------
    def attn_fwd(value, shapes, start, loc, attn, im2col_step):
        return torch.from_numpy(oracle.msda_forward(_np(value), _np(shapes), _np(start), _np(loc), _np(attn)))

    def attn_bwd(value, shapes, start, loc, attn, grad_output, im2col_step):
        gv, gl, ga = oracle.msda_backward(_np(value), _np(shapes), _np(start), _np(loc), _np(attn), _np(grad_output))
        return [torch.from_numpy(gv), torch.from_numpy(gl).view_as(loc), torch.from_numpy(ga).view_as(attn)]

    for mod in (baf, mda):
        for nm in ("box_attn_forward", "ms_deform_attn_forward"):
            if hasattr(mod, nm):
                patch(mod, nm, attn_fwd)
        for nm in ("box_attn_backward", "ms_deform_attn_backward"):
            if hasattr(mod, nm):
                patch(mod, nm, attn_bwd)

    # ---- sparse convolution -------------------------------------------------------------------
    def site_index(indices, batch_size, spatial_shape, canonical=False):
        return _CpuSiteIndex(indices, batch_size, spatial_shape)

    def downsample(x, ks, st, pad):
        out_idx, oshape = oracle.spconv_out_indices(_np(x.indices), x.batch_size, x.spatial_shape, ks, st, pad)
        t = torch.from_numpy(out_idx)
        return t, _CpuSiteIndex(t, x.batch_size, oshape), oshape

    def build_nbr(si_in, out_indices, m_out, ksize, stride, padding):
        nbr = oracle.spconv_rulebook(_np(si_in.indices), _np(out_indices), si_in.batch_size, si_in.spatial_shape,
                                     ksize, stride, padding)
        return torch.from_numpy(nbr)

    def build_rnbr(nbr, m_out, kvol, m_in):
        r = torch.full((kvol, m_in), -1, dtype=torch.int32)
        k, o = torch.nonzero(nbr >= 0, as_tuple=True)
        r[k, nbr[k, o].long()] = o.int()
        return r

    def conv_fwd(features, w, bias, rb):
        return torch.from_numpy(oracle.spconv_forward(_np(features), _np(w), None if bias is None else _np(bias),
                                                      _np(rb.nbr)))

    def conv_dgrad(grad_out, w, rb):
        return torch.from_numpy(oracle.spconv_dgrad(_np(grad_out), _np(w), _np(rb.nbr), rb.m_in))

    def conv_wgrad(features, grad_out, rb):
        return torch.from_numpy(oracle.spconv_wgrad(_np(features), _np(grad_out), _np(rb.nbr)))

    def to_dense(features, x):
        return torch.from_numpy(oracle.sparse_to_dense(_np(features), _np(x.indices), x.batch_size, x.spatial_shape))

    def to_bev(features, x):
        d = to_dense(features, x)  # [B, C, D, H, W]
        b, c, dd, h, w = d.shape
        return d.view(b, c * dd, h, w).permute(0, 2, 3, 1).contiguous()

    def from_bev(grad_out, x, c):
        b, h, w, cd = grad_out.shape
        g = grad_out.permute(0, 3, 1, 2).reshape(b, c, cd // c, h, w)
        return from_dense(g, x)

    def from_dense(grad_dense, x):
        i = x.indices.long()
        return grad_dense[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]].contiguous()

    # ---- rotated IoU / NMS (TrajectoryFormer's target assignment and linking) -------------------
    import efg_amd.operators.iou3d_nms as iou

    def pairwise(boxes_a, boxes_b, mode):
        return torch.from_numpy(oracle.boxes_iou3d(_np(boxes_a), _np(boxes_b)) if mode == "iou3d"
                                else oracle.boxes_bev(_np(boxes_a), _np(boxes_b), mode))

    def nms(boxes, scores, thresh, pre_maxsize, rotated):
        order = scores.sort(0, descending=True)[1]
        if pre_maxsize is not None:
            order = order[:pre_maxsize]
        keep = torch.from_numpy(oracle.nms(_np(boxes[order]), thresh, rotated))
        return order[keep].contiguous(), None

    def nms_segmented(boxes_sorted, segment, thresh, rotated):
        seg = _np(segment)
        keep = []
        for sid in np.unique(seg):
            rows = np.nonzero(seg == sid)[0]
            keep.append(rows[oracle.nms(_np(boxes_sorted)[rows], thresh, rotated)])
        keep = np.concatenate(keep) if keep else np.zeros((0,), np.int64)
        out = torch.zeros(max(boxes_sorted.shape[0], 1), dtype=torch.int64)
        out[:len(keep)] = torch.from_numpy(keep.astype(np.int64))
        return out, torch.tensor([len(keep)], dtype=torch.int32)

    patch(iou, "_pairwise", pairwise)
    patch(iou, "_nms", nms)
    patch(iou, "_nms_segmented", nms_segmented)

    patch(core, "_site_index_from_indices", site_index)
    patch(core, "_downsample_geometry", downsample)
    patch(core, "_build_nbr", build_nbr)
    patch(core, "_build_rnbr", build_rnbr)
    patch(core, "_conv_forward", conv_fwd)
    patch(core, "_conv_dgrad", conv_dgrad)
    patch(core, "_conv_wgrad", conv_wgrad)
    patch(core, "_to_dense", to_dense)
    patch(core, "_from_dense", from_dense)
    patch(core, "_to_bev", to_bev)
    patch(core, "_from_bev", from_bev)
    try:
        yield
    finally:
        for mod, name, fn in reversed(saved):
            setattr(mod, name, fn)
