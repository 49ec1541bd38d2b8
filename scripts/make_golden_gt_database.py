"""tests/golden/gt_database.npz from the REFERENCE `DatabaseSampling` processor (build container only).

efg/data/augmentations/extend_3d.py and efg/data/samplers/gt_database_sampler.py are imported IN PLACE behind the stub
modules of scripts/make_golden_augment.py; a small synthetic object database (efg_amd/data/synthetic_db.py, regenerated
identically by the tests) is written to a temporary directory in the reference's on-disk format (one pickle of infos +
one float32 .bin per object).  `efg/geometry/box_ops.py:box_collision_test` is a numba kernel in the reference; numba
is not installed, and as plain Python its `ret[i, j] is True` tests are always false (numpy.bool_ identity), which
would skip the containment branch the compiled kernel executes.  The module is therefore loaded with those identity
tests rewritten to value tests (`== True` / `== False`) -- the semantics numba gives them -- and nothing else changed.
Only input / output vectors are stored."""
import importlib
import os
import pickle
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
REF = "/root/reference"


def main():
    import make_golden_augment as base

    from efg_amd.data.synthetic_db import make_database, make_sampling_scene

    base.install_stubs()               # stub modules + efg package paths
    src = open(REF + "/efg/geometry/box_ops.py").read()
    src = src.replace(" is True", " == True").replace(" is False", " == False")
    mod = types.ModuleType("efg.geometry.box_ops")
    mod.__file__ = REF + "/efg/geometry/box_ops.py"
    sys.modules["efg.geometry.box_ops"] = mod
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    ref = importlib.import_module("efg.data.augmentations.extend_3d")

    infos, clouds = make_database(seed=7)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for path, cloud in clouds.items():
            os.makedirs(os.path.dirname(os.path.join(tmp, path)), exist_ok=True)
            cloud.astype(np.float32).tofile(os.path.join(tmp, path))
        with open(os.path.join(tmp, "db.pkl"), "wb") as f:
            pickle.dump(infos, f)
        np.random.seed(99)
        proc = ref.DatabaseSampling(db_info_path=os.path.join(tmp, "db.pkl"),
                                    sample_groups=[{"VEHICLE": 15}, {"PEDESTRIAN": 10}, {"CYCLIST": 10}], min_points=5,
                                    difficulty=-1)
        for case in range(4):           # consecutive samples: the cursors advance and wrap (40 entries per class)
            pts, info = make_sampling_scene(300 + case)
            info["metadata"] = {"db_path": tmp, "num_point_features": 5}
            n_before = len(info["annotations"]["gt_boxes"])
            out_pts, out_info = proc(pts.copy(), info)
            ann = out_info["annotations"]
            out["case%d.points" % case] = out_pts.astype(np.float32)
            out["case%d.gt_boxes" % case] = ann["gt_boxes"].astype(np.float32)
            out["case%d.gt_names" % case] = np.array([str(n) for n in ann["gt_names"]])
            out["case%d.num_points_in_gt" % case] = np.asarray(ann["num_points_in_gt"], np.int64)
            print("case", case, "scene boxes", n_before, "->", len(ann["gt_boxes"]), "points", len(pts), "->", len(out_pts))
    out["rng_after"] = np.random.get_state()[1][:8].astype(np.int64)
    path = os.path.join(ROOT, "tests", "golden", "gt_database.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f kB" % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
