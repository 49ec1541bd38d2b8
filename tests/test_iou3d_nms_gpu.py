"""GPU parity: efg_boxes_bev_f32 / efg_nms_f32 (through efg_amd.operators.iou3d_nms) vs the oracle restatement
of efg/operators/src/iou3d_nms/iou3d_nms_kernel.cu.  fp32 tolerance 1e-4 on areas / IoUs (only sin/cos/atan2
ULPs differ); NMS keep lists must be identical on inputs whose IoUs are not within 1e-4 of the threshold."""
import numpy as np
import pytest
import torch

import oracle
from test_oracle_iou3d import random_boxes

pytestmark = pytest.mark.gpu


def _ops():
    from efg_amd.operators import iou3d_nms
    return iou3d_nms


@pytest.mark.parametrize("na,nb,extent", [(1, 1, 1.0), (37, 129, 6.0), (300, 257, 15.0), (1000, 1000, 60.0)])
def test_overlap_and_iou(na, nb, extent):
    ops = _ops()
    rng = np.random.default_rng(na * 7 + nb)
    a, b = random_boxes(rng, na, extent, big=True), random_boxes(rng, nb, extent, big=True)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    ov = ops.boxes_overlap_bev(ta, tb).cpu().numpy()
    iou = ops.boxes_iou_bev(ta, tb).cpu().numpy()
    ov_ref, iou_ref = oracle.boxes_bev(a, b, "overlap"), oracle.boxes_bev(a, b, "iou")
    np.testing.assert_allclose(ov, ov_ref, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(iou, iou_ref, atol=1e-4, rtol=1e-4)
    # the early reject returns exactly what the reference's cnt == 0 path returns
    assert ((ov == 0) == (ov_ref == 0)).all()


def test_iou3d_matches_torch_composition():
    """boxes_iou3d_gpu (iou3d_nms.py:54-87) restated with torch ops over the oracle's BEV overlap."""
    ops = _ops()
    rng = np.random.default_rng(11)
    a, b = random_boxes(rng, 200, 8.0), random_boxes(rng, 180, 8.0)
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    ov = torch.from_numpy(oracle.boxes_bev(a, b, "overlap"))
    a_max, a_min = (ta[:, 2] + ta[:, 5] / 2).view(-1, 1), (ta[:, 2] - ta[:, 5] / 2).view(-1, 1)
    b_max, b_min = (tb[:, 2] + tb[:, 5] / 2).view(1, -1), (tb[:, 2] - tb[:, 5] / 2).view(1, -1)
    oh = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    o3 = ov * oh
    va, vb = (ta[:, 3] * ta[:, 4] * ta[:, 5]).view(-1, 1), (tb[:, 3] * tb[:, 4] * tb[:, 5]).view(1, -1)
    ref = o3 / torch.clamp(va + vb - o3, min=1e-6)
    got = ops.boxes_iou3d_gpu(ta.cuda(), tb.cuda()).cpu()
    assert (ref > 0.01).sum() > 50
    torch.testing.assert_close(got, ref, atol=1e-4, rtol=1e-4)


def test_identical_and_touching_boxes():
    ops = _ops()
    rng = np.random.default_rng(5)
    a = random_boxes(rng, 64, 10.0)
    t = torch.from_numpy(a).cuda()
    iou = ops.boxes_iou_bev(t, t).cpu().numpy()
    # self IoU: collinear edges make the reference clipper ill-conditioned, stay within 1e-3 of 1
    np.testing.assert_allclose(np.diag(iou), 1.0, atol=1e-3)
    # edge-sharing axis-aligned neighbours: overlap is only the 1e-2 margin artefacts, never more than 3 % of a box
    b = np.array([[0, 0, 0, 2, 2, 1, 0], [2, 0, 0, 2, 2, 1, 0], [0, 2, 0, 2, 2, 1, 0]], np.float32)
    ov = ops.boxes_overlap_bev(torch.from_numpy(b).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    np.testing.assert_allclose(ov, oracle.boxes_bev(b, b, "overlap"), atol=1e-4)


def test_empty_inputs():
    ops = _ops()
    z = torch.zeros((0, 7), device="cuda")
    b = torch.from_numpy(random_boxes(np.random.default_rng(2), 5)).cuda()
    assert ops.boxes_iou_bev(z, b).shape == (0, 5)
    assert ops.boxes_iou_bev(b, z).shape == (5, 0)
    keep, _ = ops.nms_gpu(z, torch.zeros((0,), device="cuda"), 0.5)
    assert keep.numel() == 0 and keep.dtype == torch.int64


def _nms_ref(boxes, scores, thresh, pre_maxsize, rotated):
    order = np.argsort(-scores, kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep = oracle.nms(boxes[order], thresh, rotated=rotated)
    return order[keep]


@pytest.mark.parametrize("n,extent,pre", [(1, 1.0, None), (63, 5.0, None), (64, 5.0, None), (65, 5.0, None),
                                          (500, 12.0, None), (3000, 30.0, 1000), (5000, 40.0, 4096)])
@pytest.mark.parametrize("rotated", [True, False])
def test_nms(n, extent, pre, rotated):
    ops = _ops()
    rng = np.random.default_rng(n + 13)
    boxes = random_boxes(rng, n, extent)
    scores = rng.permutation(n).astype(np.float32) / n  # distinct scores: sort order is unambiguous
    thresh = 0.25
    fn = ops.nms_gpu if rotated else ops.nms_normal_gpu
    kw = {"pre_maxsize": pre} if rotated else {}
    if not rotated:
        pre = None
    keep, _ = fn(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thresh, **kw)
    ref = _nms_ref(boxes, scores, thresh, pre, rotated)
    assert keep.dtype == torch.int64 and keep.is_cuda
    np.testing.assert_array_equal(keep.cpu().numpy(), ref)
    if n >= 500:
        assert 0.05 * len(ref) < len(ref) < min(n, pre or n)  # suppression actually happened


def test_nms_all_duplicates_and_all_disjoint():
    ops = _ops()
    one = np.array([[1, 2, 0, 4, 2, 1.5, 0.3]], np.float32)
    dup = np.repeat(one, 200, 0)
    dup[:, 0] += np.linspace(0, 0.05, 200, dtype=np.float32)  # near-duplicates
    s = torch.linspace(1, 0, 200).cuda()
    keep, _ = ops.nms_gpu(torch.from_numpy(dup).cuda(), s, 0.5)
    assert keep.tolist() == [0]
    grid = np.zeros((150, 7), np.float32)
    grid[:, 0] = np.arange(150) * 10
    grid[:, 3:6] = 2
    keep, _ = ops.nms_gpu(torch.from_numpy(grid).cuda(), torch.linspace(1, 0, 150).cuda(), 0.1)
    assert keep.tolist() == list(range(150))


@pytest.mark.parametrize("sets,m,extent", [(1, 40, 4.0), (7, 128, 10.0), (44, 128, 14.0), (5, 300, 14.0)])
@pytest.mark.parametrize("rotated", [True, False])
def test_batched_nms_equals_per_set_nms(sets, m, extent, rotated):
    """efg_nms_segmented_f32 (one launch over independent sets, what TrajectoryFormer's 11 per-frame NMS passes per
    sample become) == a loop of single-set NMS calls against the oracle; zero-padded sets and a score threshold."""
    ops = _ops()
    rng = np.random.default_rng(sets * 1000 + m)
    boxes = np.stack([random_boxes(rng, m, extent) for _ in range(sets)])
    scores = rng.uniform(0.05, 1.0, (sets, m)).astype(np.float32)
    for s in range(0, sets, 3):                       # some sets end in zero padding, as the tracker's frames do
        boxes[s, m // 2:] = 0
        scores[s, m // 2:] = 0
    score_thresh = 0.2
    set_id, index, counts = ops.nms_gpu_batched(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.25,
                                                score_thresh=score_thresh, rotated=rotated)
    set_id, index, counts = set_id.cpu().numpy(), index.cpu().numpy(), counts.cpu().numpy()
    assert len(set_id) == counts.sum()
    for s in range(sets):
        valid = np.nonzero(scores[s] >= score_thresh)[0]
        ref = valid[_nms_ref(boxes[s][valid], scores[s][valid], 0.25, None, rotated)]
        np.testing.assert_array_equal(index[set_id == s], ref)
        assert counts[s] == len(ref)
