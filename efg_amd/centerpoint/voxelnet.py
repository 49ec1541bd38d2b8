"""CenterPoint `VoxelNet` ($CP1/voxelnet.py:19-226): reader -> SpMiddleResNetFHD -> RPN -> CenterHead.

`forward(batched_inputs)` keeps the reference contract -- a list of `(sample, info)` pairs in, the loss dict
(`"0_loss"`, `"0_hm_loss"`, `"0_loc_loss"`, `"0_num_positive"`) or per-scene detections out -- and, like the
Voxel-DETR model of this package, also accepts samples that carry raw `points` [N, F] on the GPU: they are
voxelized for the whole batch in one call with the per-voxel mean fused (csrc/voxelize.hip), which is the
720k-point-per-sample stress case of BASELINE configs[3].  The sparse middle encoder runs on the HIP sparse-conv
kernels (geometry on the side stream), the RPN / head convolutions are MIOpen calls on the channels-last BEV map."""
import itertools

import numpy as np
import torch
from torch import nn

from ..modeling.backbones.configurable_rpn import RPN
from ..modeling.backbones.sparse_net import SpMiddleResNetFHD
from ..modeling.readers import VoxelMeanFeatureExtractor
from ..operators import voxelize_batch
from ..operators.voxelize import wait_for_points
from ..spconv import core as spconv_core
from .center_head import CenterHead
from .targets import assign_scene


class VoxelNet(nn.Module):
    def __init__(self, config, **kwargs):
        super().__init__()
        self.config = config
        self.device = torch.device(config.model.device)
        self.reader = VoxelMeanFeatureExtractor(**config.model.reader)
        self.backbone = SpMiddleResNetFHD(**config.model.backbone)
        self.neck = RPN(config.model.neck)
        self.center_head = CenterHead(config)
        lc = config.model.loss
        self.out_size_factor, self.gaussian_overlap = lc.out_size_factor, lc.gaussian_overlap
        self._max_objs, self._min_radius = lc.max_objs, lc.min_radius
        self.tasks = [dict(t) for t in config.model.head.tasks]
        self.class_names_plain = list(itertools.chain(*[t["class_names"] for t in self.tasks]))
        vz = config.dataset.processors
        self._vox_cfg = {k: vz[k].Voxelization for k in vz if "Voxelization" in vz[k]} if isinstance(vz, dict) else {}
        pr = torch.tensor(config.dataset.pc_range, dtype=torch.float32)
        vs = torch.tensor(config.dataset.voxel_size, dtype=torch.float32)
        self.grid_size = torch.round((pr[3:] - pr[:3]) / vs).long().tolist()  # (x, y, z)
        self._geo_stream = None
        self.to(self.device)
        if self.device.type == "cuda":
            self.neck.to(memory_format=torch.channels_last)
            self.center_head.to(memory_format=torch.channels_last)

    # ---- inputs -------------------------------------------------------------------------------------------------
    def _geometry_stream(self):
        if self.device.type != "cuda":
            return None
        if self._geo_stream is None:
            from ..streams import side_stream

            self._geo_stream = side_stream(self.device, "geometry", priority=-1)   # one per process and device
        return self._geo_stream

    def _inputs(self, samples):
        """-> (voxel features [M, F], coordinates [M, 4], batch size, grid (x, y, z), pc_range, voxel_size)"""
        if "voxels" in samples[0]:  # reference format: voxelized on the host by the data pipeline
            voxels = torch.as_tensor(np.concatenate([s["voxels"] for s in samples], 0)).to(self.device)
            npv = torch.as_tensor(np.concatenate([s["num_points_per_voxel"] for s in samples], 0)).to(self.device)
            coors = torch.as_tensor(np.concatenate(
                [np.pad(s["coordinates"], ((0, 0), (1, 0)), mode="constant", constant_values=i)
                 for i, s in enumerate(samples)], 0)).to(self.device)
            geo = self._geometry_stream()
            if geo is not None:
                geo.wait_stream(torch.cuda.current_stream())
            return (self.reader(voxels, npv), coors, len(samples), list(samples[0]["shape"]), samples[0]["range"],
                    samples[0]["size"])
        vc = self._vox_cfg["train" if self.training else "val"]
        pts = [torch.as_tensor(s["points"], dtype=torch.float32).to(self.device, non_blocking=True) for s in samples]
        geo = self._geometry_stream()
        nf = self.reader.num_input_features
        if geo is None:
            out = voxelize_batch(pts, vc.voxel_size, vc.pc_range, vc.max_points_in_voxel, vc.max_voxel_num)
            mean = out["voxel_mean"][:, :nf].contiguous()
        else:
            main = torch.cuda.current_stream()
            wait_for_points(geo, main, samples, pts)
            with torch.cuda.stream(geo):
                out = voxelize_batch(pts, vc.voxel_size, vc.pc_range, vc.max_points_in_voxel, vc.max_voxel_num)
                mean = out["voxel_mean"][:, :nf].contiguous()
            main.wait_stream(geo)
            for t in (out["coordinates"], mean):
                t.record_stream(main)
        return mean, out["coordinates"], len(samples), self.grid_size, list(vc.pc_range), list(vc.voxel_size)

    # ---- targets ------------------------------------------------------------------------------------------------
    def label_assign(self, infos, grid_size, pc_range, voxel_size):
        """Host label assignment ($CP1/voxelnet.py:43-194) + ONE asynchronous upload per key."""
        per_scene = [assign_scene(info["annotations"], self.tasks, self.class_names_plain, np.asarray(grid_size),
                                  pc_range, voxel_size, self.out_size_factor, self.gaussian_overlap, self._max_objs,
                                  self._min_radius) for info in infos]
        targets = {}
        for key in ("hm", "anno_box", "ind", "mask", "cat"):
            targets[key] = [torch.from_numpy(np.stack([s[key][t] for s in per_scene])).to(self.device, non_blocking=True)
                            for t in range(len(self.tasks))]
        targets["gt_boxes_and_cls"] = torch.from_numpy(np.stack([s["gt_boxes_and_cls"] for s in per_scene]))
        return targets

    def forward(self, batched_inputs):
        samples = [bi[0] for bi in batched_inputs]
        infos = [bi[1] for bi in batched_inputs]
        with torch.no_grad():
            feats, coors, batch_size, grid, pc_range, voxel_size = self._inputs(samples)
            targets = self.label_assign(infos, grid, pc_range, voxel_size) if self.training else None
        with spconv_core.geometry_stream(self._geometry_stream()):
            x = self.backbone(feats, coors, batch_size, grid)
        x = self.neck(x)
        preds = self.center_head(x)
        if self.training:
            return self.center_head.loss(targets, preds)
        return self.center_head.predict({}, preds, self.config.model.post_process)
