"""tests/golden/tf_augment.npz from the REFERENCE TrajectoryFormer augmentations ($TF/aug.py), imported in place behind
the stub modules of scripts/make_golden_augment.py (build container only).  Only vectors are stored."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
TF = "/root/reference/playground/tracking.3d/waymo/trajectoryformer/trajectoryformer.centerpoint"


def main():
    import make_golden_augment as base

    from efg_amd.tracking.synthetic import make_tracking_sample

    ext = base.load_reference_processors()
    # the stub package `efg.data.augmentations` has no __init__ of its own: expose what aug.py imports from it
    sys.modules["efg.data.augmentations"].AugmentationBase = ext.AugmentationBase
    spec = importlib.util.spec_from_file_location("tf_aug_ref", TF + "/aug.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    pc_range = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    out = {}
    for case, seed in enumerate((3, 12, 40)):
        sample, info = make_tracking_sample(600 + case, n_points=20000, n_objects=12, n_false=4)
        pts = sample[0]["points"].copy()
        pts[:, :3] *= 1.2                                   # some points / boxes leave the range
        info["annotations"]["gt_boxes"][:, :2] *= 1.35
        info["sweeps"] = []
        chain = [ref.CusTomRandomFlip3D(p=0.5), ref.CusTomGlobalRotation(rotation=0.78539816),
                 ref.CusTomGlobalScaling(min_scale=0.95, max_scale=1.05), ref.CusTomFilterByRange(pc_range=pc_range)]
        np.random.seed(seed)
        for proc in chain:
            pts, info = proc(pts, info)
        ann = info["annotations"]
        out["case%d.seed" % case] = np.array(seed)
        out["case%d.points" % case] = pts.astype(np.float32)
        for k in ("gt_boxes", "labels", "pred_boxes3d", "pred_scores"):
            out["case%d.%s" % (case, k)] = np.asarray(ann[k])
        print(case, pts.shape, ann["gt_boxes"].shape, ann["pred_boxes3d"].shape)
    out["rng_after"] = np.random.get_state()[1][:8].astype(np.int64)
    path = os.path.join(ROOT, "tests", "golden", "tf_augment.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f kB" % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
