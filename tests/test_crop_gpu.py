"""GPU parity of efg_cylinder_select_f32 (csrc/crop.hip) with the oracle restatement: counts and the ordered index
lists, several scenes in one launch, empty scenes, the time gate, padding cylinders; and the vectorised
`crop_current_frame_points` against the PyTorch formulation the host path runs."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _select(points, rng, xyr, time_col, chunks=1):
    from efg_amd import _lib as L

    dev = points.device
    counts = torch.zeros(len(xyr), dtype=torch.int32, device=dev)
    rng_t = torch.as_tensor(rng, device=dev).contiguous()
    xyr_t = torch.as_tensor(xyr, device=dev).contiguous()
    lib = L.lib()
    per_chunk = torch.full((len(xyr), chunks), -3, dtype=torch.int32, device=dev)
    L.check(lib.efg_cylinder_select_f32(L.ptr(points), points.shape[0], points.shape[1], time_col, 1.0, L.ptr(rng_t),
                                        L.ptr(xyr_t), len(xyr), None, L.ptr(counts), None, chunks, L.ptr(per_chunk),
                                        L.stream()))
    c = counts.long()
    starts = (torch.cumsum(c, 0) - c).contiguous()
    total = int(c.sum())
    index = torch.full((max(total, 1),), -7, dtype=torch.int32, device=dev)
    L.check(lib.efg_cylinder_select_f32(L.ptr(points), points.shape[0], points.shape[1], time_col, 1.0, L.ptr(rng_t),
                                        L.ptr(xyr_t), len(xyr), L.ptr(starts), None, L.ptr(index), chunks, L.ptr(per_chunk),
                                        L.stream()))
    return counts.cpu().numpy(), starts.cpu().numpy(), index.cpu().numpy()[:total]


@pytest.mark.parametrize("chunks", [1, 7, 64])
@pytest.mark.parametrize("sizes,cyl,time_col", [((5000,), 16, -1), ((30000, 0, 12345), 48, 5), ((180000, 170000), 320, 5),
                                                 ((257,), 16, 5)])
def test_counts_and_ordered_indices_match_oracle(dev, sizes, cyl, time_col, chunks):
    rng = np.random.default_rng(sum(sizes) + cyl)
    clouds = [np.concatenate([rng.uniform(-40, 40, (n, 3)), rng.uniform(0, 1, (n, 2)),
                              rng.choice([0.0, 0.1, 1.0, 1.2], (n, 1), p=[0.6, 0.2, 0.1, 0.1])], 1).astype(np.float32)
              for n in sizes]
    pts = np.concatenate(clouds)
    base = np.concatenate([[0], np.cumsum(sizes)])
    ranges, xyr = [], []
    for s in range(len(sizes)):
        c = np.concatenate([rng.uniform(-40, 40, (cyl, 2)), rng.uniform(0.2, 6.0, (cyl, 1))], 1).astype(np.float32)
        c[-3:, 2] = -1.0                                       # padding cylinders
        if sizes[s]:
            c[0, :2] = clouds[s][0, :2]                         # a centre exactly on a point
        xyr.append(c)
        ranges.append(np.repeat([[base[s], base[s + 1]]], cyl, 0))
    xyr, ranges = np.concatenate(xyr), np.concatenate(ranges).astype(np.int64)
    counts, starts, index = _select(torch.from_numpy(pts).to(dev), ranges, xyr, time_col, chunks)
    want_counts, want_lists = oracle.cylinder_select(pts, ranges, xyr, time_col, 1.0)
    np.testing.assert_array_equal(counts, want_counts)
    for c in range(len(xyr)):
        np.testing.assert_array_equal(index[starts[c]:starts[c] + counts[c]], want_lists[c])
    assert counts.max() > 128 or sum(sizes) < 20000


def test_vectorised_crop_equals_host_formulation(dev):
    from efg_amd.tracking.geometry import crop_current_frame_points

    rng = np.random.default_rng(4)
    b, n, h, k = 3, 7, 4, 128
    rois = torch.zeros(b, 2, n, h, 8)
    rois[:, 0, :, :, 0:2] = torch.from_numpy(rng.uniform(-30, 30, (b, n, h, 2)).astype(np.float32))
    rois[:, 0, :, :, 2] = -1.0
    rois[:, 0, :, :, 3:6] = torch.from_numpy(rng.uniform(0.5, 14.0, (b, n, h, 3)).astype(np.float32))
    rois[1, 0, 2] = 0                                            # zero boxes (padding tracks)
    clouds = [torch.from_numpy(np.concatenate([rng.uniform(-35, 35, (m, 3)), rng.uniform(0, 1, (m, 2)),
                                               rng.choice([0.0, 1.0], (m, 1), p=[0.9, 0.1])], 1).astype(np.float32))
              for m in (60000, 0, 25000)]
    np.random.seed(11)
    host = crop_current_frame_points(k, rois, clouds)
    state = np.random.get_state()[1][:4].copy()
    np.random.seed(11)
    gpu = crop_current_frame_points(k, rois.to(dev), [c.to(dev) for c in clouds])
    assert torch.equal(gpu.cpu(), host)
    np.testing.assert_array_equal(np.random.get_state()[1][:4], state)   # the generator is left in the same state
    counts = (host[..., 3:].abs().sum(-1) > 0).sum(-1)
    assert int((counts == 0).sum()) > 0 and int((counts == k).sum()) > 0   # empty and crowded boxes both occur
