"""`efg.operators.voxelize` on MI355X.

Mirrors efg/operators/voxelize.py:9-96 (`_Voxelization`, `voxelization`, `Voxelization`): same
names, argument meaning, return values and error behaviour (RuntimeError).  The compute is the
HIP path in csrc/voxelize.hip behind the C ABI (include/efg_hip.h); there is no CPU branch.

`voxelize_batch` is the batched entry the model path uses: all scenes of a rank in one call,
outputs already in the `collate` layout (efg/data/datasets/waymo/waymo.py:143-183) with the
per-voxel mean (efg/modeling/readers/voxel_reader.py:14-19) fused in.
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from .. import _lib as L
from .. import _prof


_LOOKBACK_MSG = ("efg_hip: hard_voxelize gave up waiting for a look-back record (csrc/voxelize_bins.hip: a workgroup's prefix "
                 "never arrived -- dispatch-order invariant broken on this device / partition mode); no voxels were written. "
                 "EFG_VOX_IMPL=hash selects the implementation without look-back scans")


def dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3):
    """efg._C.dynamic_voxelize (efg/operators/src/voxelize/voxelization.h:71-83): fills
    coors[N,3] int32 (z,y,x), (-1,-1,-1) outside the range."""
    L.require_gpu(points, coors)
    if NDim != 3:
        raise RuntimeError("dynamic_voxelize: only NDim == 3 is supported")
    if points.dtype != torch.float32 or coors.dtype != torch.int32:
        raise RuntimeError("dynamic_voxelize: points must be float32 and coors int32")
    points = points.contiguous()
    assert coors.is_contiguous() and coors.shape == (points.shape[0], 3)
    L.check(L.lib().efg_dynamic_voxelize_f32(L.ptr(points), points.shape[0], points.shape[1],
                                             L.host_f32(voxel_size, 3), L.host_f32(coors_range, 6), L.ptr(coors),
                                             L.stream()))


def _hard_voxelize_launch(points, offsets, voxel_size, coors_range, max_points, max_voxels, voxels, coors, npv,
                          voxel_num, mean):
    batch = len(offsets) - 1
    n_total = int(offsets[-1])
    lib = L.lib()
    vs, cr = L.host_f32(voxel_size, 3), L.host_f32(coors_range, 6)
    ws_bytes = lib.efg_hard_voxelize_workspace_bytes(n_total, batch, points.shape[1], max_points, max_voxels, vs, cr)
    if ws_bytes == 0:
        raise RuntimeError("hard_voxelize: max_points and max_voxels must be >= 1, batch in [1, 64], points [N, >= 3] "
                           "and a non-empty grid")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=points.device)
    L.check(lib.efg_hard_voxelize_f32(L.ptr(points), L.host_i64(offsets), batch, points.shape[1],
                                      vs, cr, max_points, max_voxels,
                                      L.ptr(voxels), L.ptr(coors), coors.shape[1], L.ptr(npv), L.ptr(voxel_num),
                                      L.ptr(mean), L.ptr(ws), ws_bytes, L.stream()))


def hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, max_points, max_voxels,
                  NDim=3):
    """efg._C.hard_voxelize (voxelization.h:51-69): fills the three caller-allocated, pre-zeroed
    buffers and returns voxel_num (a host int, hence one device sync -- same contract as the
    reference, voxelization_cuda.cu:316-317)."""
    L.require_gpu(points, voxels, coors, num_points_per_voxel)
    if NDim != 3:
        raise RuntimeError("hard_voxelize: only NDim == 3 is supported")
    if points.dtype != torch.float32:
        raise RuntimeError("hard_voxelize: points must be float32")
    points = points.contiguous()
    voxel_num = torch.zeros(1, dtype=torch.int32, device=points.device)
    _hard_voxelize_launch(points, [0, points.shape[0]], voxel_size, coors_range, max_points, max_voxels, voxels, coors,
                          num_points_per_voxel, voxel_num, None)
    n = int(voxel_num.item())
    if n < 0:
        raise RuntimeError(_LOOKBACK_MSG)
    return n


class _Voxelization(Function):
    """efg/operators/voxelize.py:9-49.  No backward (pure indexing), as in the reference."""

    @staticmethod
    def forward(ctx, points, voxel_size, coors_range, max_points=35, max_voxels=20000):
        if max_points == -1 or max_voxels == -1:
            coors = points.new_zeros(size=(points.size(0), 3), dtype=torch.int)
            dynamic_voxelize(points, coors, voxel_size, coors_range, 3)
            return coors
        voxels = points.new_zeros(size=(max_voxels, max_points, points.size(1)))
        coors = points.new_zeros(size=(max_voxels, 3), dtype=torch.int)
        num_points_per_voxel = points.new_zeros(size=(max_voxels,), dtype=torch.int)
        voxel_num = hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, max_points,
                                  max_voxels, 3)
        return voxels[:voxel_num], coors[:voxel_num], num_points_per_voxel[:voxel_num]


voxelization = _Voxelization.apply


class Voxelization(nn.Module):
    """efg/operators/voxelize.py:55-106 (constructor arguments, (train, eval) max_voxels pair,
    grid_size / pcd_shape attributes, repr)."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else _pair(max_voxels)
        point_cloud_range = torch.tensor(point_cloud_range, dtype=torch.float32)
        voxel_size = torch.tensor(voxel_size, dtype=torch.float32)
        grid_size = torch.round((point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size).long()
        self.grid_size = grid_size
        self.pcd_shape = [*grid_size[:2], 1][::-1]

    def forward(self, input):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        return voxelization(input, self.voxel_size, self.point_cloud_range, self.max_num_points, max_voxels)

    def __repr__(self):
        return (self.__class__.__name__ + "(voxel_size=" + str(self.voxel_size) + ", point_cloud_range=" +
                str(self.point_cloud_range) + ", max_num_points=" + str(self.max_num_points) + ", max_voxels=" +
                str(self.max_voxels) + ")")


def voxelize_batch(points_list, voxel_size, coors_range, max_points, max_voxels, with_mean=True):
    """Voxelize every scene of a rank in ONE call.

    points_list: list of [N_b, F] float32 device tensors (or one concatenated tensor plus
    offsets via `voxelize_concat`).  Returns a dict with the keys `collate` produces
    (waymo.py:143-183): voxels [sumM, max_points, F], coordinates [sumM, 4] int32 (b,z,y,x),
    num_points_per_voxel [sumM], num_voxels (host list), plus voxel_mean [sumM, F].
    One host sync (the voxel counts size every downstream tensor).
    """
    offsets = [0]
    for p in points_list:
        offsets.append(offsets[-1] + p.shape[0])
    points = points_list[0] if len(points_list) == 1 else torch.cat(points_list, 0)
    return voxelize_concat(points, offsets, voxel_size, coors_range, max_points, max_voxels, with_mean)


def voxelize_concat(points, offsets, voxel_size, coors_range, max_points, max_voxels, with_mean=True):
    L.require_gpu(points)
    points = points.contiguous()
    batch = len(offsets) - 1
    f = points.shape[1]
    cap = min(batch * max_voxels, max(int(offsets[-1]), 1))
    dev = points.device
    voxels = torch.empty((cap, max_points, f), dtype=torch.float32, device=dev)
    coors = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    npv = torch.empty((cap,), dtype=torch.int32, device=dev)
    mean = torch.empty((cap, f), dtype=torch.float32, device=dev) if with_mean else None
    voxel_num = torch.zeros(batch, dtype=torch.int32, device=dev)
    box = {}
    n_total = int(offsets[-1])
    # algorithmic bytes: read 4F per point; write voxels + coordinates + count (+ mean) per voxel
    cost = lambda: (4 * f * n_total + box.get("m", 0) * (4 * max_points * f + 16 + 4 + (4 * f if with_mean else 0)), 0)  # noqa: E731
    with _prof.timed("hard_voxelize (all kernels, batched)", cost):
        _hard_voxelize_launch(points, offsets, voxel_size, coors_range, max_points, max_voxels, voxels, coors, npv,
                              voxel_num, mean)
    counts = voxel_num.tolist()  # the one sync
    if min(counts, default=0) < 0:
        raise RuntimeError(_LOOKBACK_MSG)
    box["m"] = sum(counts)
    m = sum(counts)
    out = {
        "voxels": voxels[:m],
        "coordinates": coors[:m],
        "num_points_per_voxel": npv[:m],
        "num_voxels": counts,
    }
    if with_mean:
        out["voxel_mean"] = mean[:m]
    return out


def wait_for_points(stream, main, samples, pts):
    """Order `stream` (a model's geometry stream) after the producers of `pts`.  Samples that carry a `ready_event`
    (engine.synthetic_batch, data/loader.py DeviceLoader) promise "the points are complete once this event fires":
    the stream waits for those events only, NOT for the previous step's backward still queued on `main`, so the
    voxel-count read-back that follows returns early and the host prepares step n+1 under step n.  Anything else
    (host arrays uploaded just now, converted dtypes) has unknown provenance: wait for everything queued on `main`."""
    events = [s.get("ready_event") for s in samples]
    converted = any(torch.is_tensor(s["points"]) and (not s["points"].is_cuda or s["points"].dtype != torch.float32)
                    for s in samples)
    if all(e is not None for e in events) and not converted:
        for e, p in zip(events, pts):
            stream.wait_event(e)
            p.record_stream(stream)   # possibly allocated on a loader's stream: keep alive until `stream` has read it
    else:
        stream.wait_stream(main)
