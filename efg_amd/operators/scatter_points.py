"""`efg.operators.scatter_points` on MI355X (mirrors efg/operators/scatter_points.py:8-104).

`dynamic_point_to_voxel_forward/backward` have the Python-visible signatures of the reference
bindings (efg/operators/src/voxelize/voxelization.h:96-128).  Voxel order = ascending
linearised coordinate with per-call dims `coors.max(0)+1` (scatter_points_cuda.cu:220).
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import _lib as L

_REDUCE = {"sum": 0, "mean": 1, "max": 2}


def _reduce_id(reduce_type):
    if reduce_type not in _REDUCE:
        raise RuntimeError("do not support reduce type " + str(reduce_type))  # voxelization.h:92
    return _REDUCE[reduce_type]


def dynamic_point_to_voxel_forward(feats, coors, reduce_type):
    """-> [voxel_feats f32[M,C], voxel_coors i32[M,ndim], point2voxel_map i32[N], voxel_points_count i32[M]]"""
    L.require_gpu(feats, coors)
    red = _reduce_id(reduce_type)
    if feats.dtype != torch.float32 or coors.dtype != torch.int32:
        raise RuntimeError("dynamic_point_to_voxel_forward: feats must be float32 and coors int32")
    feats, coors = feats.contiguous(), coors.contiguous()
    n, c = feats.shape
    ndim = coors.shape[1]
    dev = feats.device
    p2v = torch.empty(n, dtype=torch.int32, device=dev)
    if n == 0:
        return [feats.new_zeros((0, c)), coors.new_zeros((0, ndim)), p2v, coors.new_zeros((0,))]
    dims = (coors.max(0)[0] + 1).clamp_(min=0).tolist()  # host read, as the reference (it syncs for M anyway)
    lib = L.lib()
    dims_h = L.host_i32(dims)
    ws_bytes = lib.efg_scatter_workspace_bytes(n, ndim, dims_h)
    if ws_bytes == 0:
        raise RuntimeError("efg_hip: " + lib.efg_last_error().decode())
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    m_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(lib.efg_scatter_index(L.ptr(coors), n, ndim, dims_h, L.ptr(p2v), L.ptr(m_dev), L.ptr(ws), ws_bytes,
                                  L.stream()))
    m = int(m_dev.item())  # sizes the outputs (scatter_points_cuda.cu:253-254 does the same D2H)
    vf = torch.empty((m, c), dtype=torch.float32, device=dev)
    vc = torch.empty((m, ndim), dtype=torch.int32, device=dev)
    cnt = torch.empty((m,), dtype=torch.int32, device=dev)
    ws_bytes = lib.efg_scatter_reduce_workspace_bytes(m, c)   # sum / mean: exact integer accumulators (deterministic)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    L.check(lib.efg_scatter_reduce_f32(L.ptr(feats), L.ptr(coors), L.ptr(p2v), n, c, ndim, red, m, L.ptr(vf),
                                       L.ptr(vc), L.ptr(cnt), L.ptr(ws), ws_bytes, L.stream()))
    return [vf, vc, p2v, cnt]


def dynamic_point_to_voxel_backward(grad_feats, grad_voxel_feats, feats, voxel_feats, point2voxel_map,
                                    voxel_points_count, reduce_type):
    L.require_gpu(grad_feats, grad_voxel_feats, feats, voxel_feats)
    red = _reduce_id(reduce_type)
    n, c = feats.shape
    m = voxel_feats.shape[0]
    ws = torch.empty(max(m * c, 1), dtype=torch.int32, device=feats.device) if red == 2 else None
    L.check(L.lib().efg_scatter_backward_f32(L.ptr(grad_feats), L.ptr(grad_voxel_feats.contiguous()),
                                             L.ptr(feats.contiguous()), L.ptr(voxel_feats.contiguous()),
                                             L.ptr(point2voxel_map), L.ptr(voxel_points_count), n, m, c, red,
                                             L.ptr(ws), 0 if ws is None else ws.numel() * 4, L.stream()))


class _dynamic_scatter(Function):
    """efg/operators/scatter_points.py:8-50."""

    @staticmethod
    def forward(ctx, feats, coors, reduce_type="max"):
        results = dynamic_point_to_voxel_forward(feats, coors, reduce_type)
        (voxel_feats, voxel_coors, point2voxel_map, voxel_points_count) = results
        ctx.reduce_type = reduce_type
        ctx.save_for_backward(feats, voxel_feats, point2voxel_map, voxel_points_count)
        ctx.mark_non_differentiable(voxel_coors)
        return voxel_feats, voxel_coors

    @staticmethod
    def backward(ctx, grad_voxel_feats, grad_voxel_coors=None):
        (feats, voxel_feats, point2voxel_map, voxel_points_count) = ctx.saved_tensors
        grad_feats = torch.empty_like(feats)
        dynamic_point_to_voxel_backward(grad_feats, grad_voxel_feats.contiguous(), feats, voxel_feats,
                                        point2voxel_map, voxel_points_count, ctx.reduce_type)
        return grad_feats, None, None


dynamic_scatter = _dynamic_scatter.apply


class DynamicScatter(nn.Module):
    """efg/operators/scatter_points.py:53-104."""

    def __init__(self, voxel_size, point_cloud_range, average_points: bool):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.average_points = average_points

    def forward_single(self, points, coors):
        reduce = "mean" if self.average_points else "max"
        return dynamic_scatter(points.contiguous(), coors.contiguous(), reduce)

    def forward(self, points, coors):
        if coors.size(-1) == 3:
            return self.forward_single(points, coors)
        batch_size = int(coors[-1, 0]) + 1
        voxels, voxel_coors = [], []
        for i in range(batch_size):
            inds = torch.where(coors[:, 0] == i)
            voxel, voxel_coor = self.forward_single(points[inds], coors[inds][:, 1:])
            voxel_coors.append(nn.functional.pad(voxel_coor, (1, 0), mode="constant", value=i))
            voxels.append(voxel)
        return torch.cat(voxels, dim=0), torch.cat(voxel_coors, dim=0)

    def __repr__(self):
        return (self.__class__.__name__ + "(voxel_size=" + str(self.voxel_size) + ", point_cloud_range=" +
                str(self.point_cloud_range) + ", average_points=" + str(self.average_points) + ")")
