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


def imbalance(nbr, ny, chunk=1024, r=2, scenes=2, cus=256, per_cu=4, lpt=False):
    """Per-workgroup MFMA work (active sub-tile offsets of a wave tile) of the mask-sorted plan, `scenes` copies of the
    scene, `ny` output-channel groups, and a model of ONE launch on `cus` CUs with `per_cu` workgroup slots: workgroups
    go round-robin to the CUs (first round) and then to whichever slot frees first; a CU's matrix pipes are shared by
    its resident workgroups (processor sharing).  Returns (workgroups, mean, std, max work, balance = ideal / makespan)."""
    import heapq

    kvol, m = nbr.shape
    has = nbr >= 0
    mask = np.zeros(m, np.int64)
    for k in range(kvol):
        mask |= has[k].astype(np.int64) << k
    work = []
    for s in range(0, m, chunk):
        sl = np.arange(s, min(s + chunk, m))
        sl = sl[np.argsort(mask[sl], kind="stable")]
        tiles = [sl[t:t + 16] for t in range(0, len(sl), 16)]
        for t in range(0, len(tiles), r):
            work.append(sum(int(has[:, rows].any(1).sum()) for rows in tiles[t:t + r]))
    work = np.array(work * scenes * ny, np.float64)
    if lpt:
        work = np.sort(work)[::-1]
    # event simulation with processor sharing per CU
    queue = list(work)
    cu = [[] for _ in range(cus)]          # remaining work of resident workgroups
    qi = 0
    for rnd in range(per_cu):
        for c in range(cus):
            if qi < len(queue):
                cu[c].append(queue[qi]); qi += 1
    now = 0.0
    # time to next completion on a CU with k residents: min(rem) * k (each gets 1/k of the pipe)
    while True:
        best, bc = None, -1
        for c in range(cus):
            if cu[c]:
                t = min(cu[c]) * len(cu[c])
                if best is None or t < best:
                    best, bc = t, c
        if best is None:
            break
        for c in range(cus):
            if cu[c]:
                k = len(cu[c])
                cu[c] = [x - best / k for x in cu[c]]
        now += best
        cu[bc] = [x for x in cu[bc] if x > 1e-9]
        while len(cu[bc]) < per_cu and qi < len(queue):
            cu[bc].append(queue[qi]); qi += 1
    return len(work), work.mean(), work.std(), work.max(), work.sum() / cus / now


print("\nload balance of ONE launch (2 scenes, R=2 wave tiles, 4 workgroup slots per CU, matrix pipes shared per CU):")
x_idx, x_shape = idx, shape
ny_of = {"res2 subm": 1, "res3 subm": 1, "res4 subm": 2}
for name, down in stages:
    if down is None:
        if name in ny_of:
            nbr = oracle.spconv_rulebook(x_idx, x_idx, 1, x_shape, [3, 3, 3], [1, 1, 1], [1, 1, 1])
            for kw in ({"lpt": True}, {"r": 1, "per_cu": 6}, {"r": 1, "per_cu": 6, "lpt": True}):
                print("      ", kw, "-> workgroups %d balance %.2f" % (imbalance(nbr, ny_of[name], **kw)[0], imbalance(nbr, ny_of[name], **kw)[4]))
            n, mean, std, mx, eff = imbalance(nbr, ny_of[name])
            print("%-22s (model level: %s) workgroups %5d  work mean %.1f std %.1f max %.0f -> balance %.2f" % (
                name, {"res2 subm": "stem", "res3 subm": "res2", "res4 subm": "res3"}[name], n, mean, std, mx, eff))
    else:
        o_idx, o_shape = oracle.spconv_out_indices(x_idx, 1, x_shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
        x_idx, x_shape = o_idx, o_shape
