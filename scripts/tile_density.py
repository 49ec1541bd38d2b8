"""Density of the mask-sorted tile plan: useful (row, offset) pairs / MFMA row slots executed (16 x active offsets per
tile), per backbone stage of a synthetic scene, as a function of the sort chunk.  CPU only (oracle rulebook)."""
import sys

import numpy as np

sys.path.insert(0, ".")
import oracle  # noqa: E402
from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene  # noqa: E402

oracle.build()
pts, _, _ = make_scene(3000, n_points=180000)
_, coors, _ = oracle.hard_voxelize(pts, VOXEL_SIZE, PC_RANGE, 5, 120000)
idx = np.concatenate([np.zeros((len(coors), 1), np.int32), coors.astype(np.int32)], 1)
# canonical order (b, z, y, x)
key = ((idx[:, 1].astype(np.int64) * 2048 + idx[:, 2]) * 2048) + idx[:, 3]
idx = idx[np.argsort(key, kind="stable")]
shape = [41, 1504, 1504]


def density(nbr, chunk, spatial_key=None):
    kvol, m = nbr.shape
    has = nbr >= 0
    mask = np.zeros(m, np.int64)
    for k in range(kvol):
        mask |= has[k].astype(np.int64) << k
    order = np.arange(m)
    pairs = int(has.sum())
    slots = 0
    for s in range(0, m, chunk):
        sl = order[s:s + chunk]
        sl = sl[np.argsort(mask[sl], kind="stable")]
        for t in range(0, len(sl), 16):
            rows = sl[t:t + 16]
            slots += 16 * int(has[:, rows].any(1).sum())
    return pairs / slots, pairs / (m * kvol)


x_idx, x_shape = idx, shape
stages = [("res1 subm (stride 1)", None), ("down 1->2", (3, 2, 1)), ("res2 subm", None), ("down 2->3", (3, 2, 1)), ("res3 subm", None),
          ("down 3->4", (3, 2, 1)), ("res4 subm", None)]
for name, down in stages:
    if down is None:
        nbr = oracle.spconv_rulebook(x_idx, x_idx, 1, x_shape, [3, 3, 3], [1, 1, 1], [1, 1, 1])
    else:
        o_idx, o_shape = oracle.spconv_out_indices(x_idx, 1, x_shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
        nbr = oracle.spconv_rulebook(x_idx, o_idx, 1, x_shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
        x_idx, x_shape = o_idx, o_shape
    res = ["chunk %6d: %.3f" % (c, density(nbr, c)[0]) for c in (16, 256, 1024, 4096, 16384, 1 << 30)]
    print("%-22s rows %7d  fill %.3f | %s" % (name, nbr.shape[1], density(nbr, 1 << 30)[1], "  ".join(res)))
