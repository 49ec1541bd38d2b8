"""Golden vectors from the reference's OWN online tracker (`TrajectoryFormer.forward_inference`, evaluation mode), run
frame by frame on CPU in the build container over a short synthetic drive.

Same import-in-place recipe and stand-ins as scripts/make_golden_trajectoryformer.py (rotated IoU / NMS -> the CPU
oracle, `.cuda()` -> identity, deterministic weights from tests/golden_init.py).  Stored per frame: the track ids, boxes,
scores and labels the reference returns -- tests/golden/trajectoryformer_online.npz; inputs are regenerated on both
sides by efg_amd/tracking/synthetic.py:make_tracking_sequence (checksummed in the fixture)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
TF = "/root/reference/playground/tracking.3d/waymo/trajectoryformer/trajectoryformer.centerpoint"


def main():
    import make_golden_full as shim
    from golden_init import ONLINE_SEQUENCE, deterministic_state

    import oracle
    from efg_amd.config import load_config
    from efg_amd.tracking.synthetic import make_tracking_sequence

    shim.install_shims(TF)

    def boxes_iou3d_gpu(a, b):
        return torch.from_numpy(oracle.boxes_iou3d(a.detach().numpy(), b.detach().numpy()))

    def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kw):
        order = scores.sort(0, descending=True)[1]
        if pre_maxsize is not None:
            order = order[:pre_maxsize]
        keep = torch.from_numpy(oracle.nms(boxes[order].detach().numpy(), thresh, True))
        return order[keep].contiguous(), None

    shim._mod("efg.modeling.operators", boxes_iou3d_gpu=boxes_iou3d_gpu, nms_gpu=nms_gpu)
    import trajectoryformer as ref  # the reference model, imported in place

    cfg = load_config(os.path.join(ROOT, "configs", "trajectoryformer_waymo_centerpoint.yaml"),
                      {"model.device": "cpu", "task": "val", "model.eval_class": "VEHICLE"})
    torch.manual_seed(0)
    model = ref.TrajectoryFormer(cfg)
    model.load_state_dict(deterministic_state(model.state_dict()))
    model.eval()
    model.load_motion_module = True
    out = {}
    seq = make_tracking_sequence(**ONLINE_SEQUENCE)
    out["in.checksum"] = np.array([float(np.abs(s[0]["points"]).sum(dtype=np.float64)) +
                                   float(np.abs(i["annotations"]["pred_boxes3d"]).sum(dtype=np.float64)) for s, i in seq])
    with torch.no_grad():
        for f, item in enumerate(seq):
            res = model([item])[0]
            for k, v in res.items():
                out["frame%d.%s" % (f, k)] = v.detach().cpu().numpy().astype(np.int64 if k == "track_ids" else np.float32)
            print("frame", f, "tracks", res["track_ids"].tolist())
    path = os.path.join(ROOT, "tests", "golden", "trajectoryformer_online.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f kB" % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
