"""CPU: the oracle's rotated BEV IoU / NMS restatement (oracle/efg_oracle.c, reference
efg/operators/src/iou3d_nms/iou3d_nms_kernel.cu:34-362).

PARITY UNPINNED against reference binaries (iou3d_cpu.cpp needs <cuda.h>, absent here; the reference has no
tests for this operator).  The restatement is instead checked against an INDEPENDENT float64
Sutherland-Hodgman polygon clipper written here, and NMS against a brute-force loop over that clipper's IoU.
"""
import numpy as np
import pytest

import oracle


def _corners(b):
    x, y, dx, dy, a = b[0], b[1], b[3], b[4], b[6]
    c, s = np.cos(a), np.sin(a)
    loc = np.array([[-dx / 2, -dy / 2], [dx / 2, -dy / 2], [dx / 2, dy / 2], [-dx / 2, dy / 2]], np.float64)
    rot = np.array([[c, -s], [s, c]])
    return loc @ rot.T + np.array([x, y])


def _clip(subject, a, b):
    """Keep the part of polygon `subject` on the left of the directed edge a->b."""
    out = []
    n = len(subject)
    for i in range(n):
        p, q = subject[i], subject[(i + 1) % n]
        sp = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        sq = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
        if sp >= 0:
            out.append(p)
        if (sp >= 0) != (sq >= 0):
            t = sp / (sp - sq)
            out.append(p + t * (q - p))
    return out


def _overlap_f64(ba, bb):
    poly = list(_corners(ba.astype(np.float64)))
    cb = _corners(bb.astype(np.float64))
    for i in range(4):
        if not poly:
            return 0.0
        poly = _clip(poly, cb[i], cb[(i + 1) % 4])
    if len(poly) < 3:
        return 0.0
    p = np.array(poly)
    return 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - p[:, 1] * np.roll(p[:, 0], -1)))


def random_boxes(rng, n, extent=20.0, big=False):
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rng.uniform(-extent, extent, (n, 2))
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3] = rng.uniform(1.5, 12.0 if big else 5.0, n)
    b[:, 4] = rng.uniform(0.6, 3.0, n)
    b[:, 5] = rng.uniform(1.0, 3.0, n)
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    return b


def test_overlap_matches_independent_clipper():
    rng = np.random.default_rng(0)
    a, b = random_boxes(rng, 60, 8.0, big=True), random_boxes(rng, 50, 8.0, big=True)
    got = oracle.boxes_bev(a, b, "overlap")
    ref = np.array([[_overlap_f64(x, y) for y in b] for x in a])
    assert (ref > 0.05).sum() > 100  # the case actually exercises the clipper
    # the reference algorithm counts vertices within a 1e-2 margin as inside (iou3d_nms_kernel.cu:56), which
    # inflates touching overlaps by up to ~margin * perimeter
    np.testing.assert_allclose(got, ref, atol=0.08)
    exact = np.abs(got - ref) < 1e-3
    assert exact.mean() > 0.97


def test_known_answers():
    a = np.array([[0, 0, 0, 4, 2, 1, 0.0], [0, 0, 0, 4, 2, 1, np.pi / 2], [1, 0, 0.5, 4, 2, 2, 0.0],
                  [10, 10, 0, 1, 1, 1, 0.3]], np.float32)
    ov = oracle.boxes_bev(a, a, "overlap")
    np.testing.assert_allclose(ov[0, 1], 4.0, rtol=1e-5)  # 2x2 square core of the cross
    np.testing.assert_allclose(ov[0, 2], 6.0, rtol=1e-5)  # shifted by 1 along x: 3 x 2
    assert ov[0, 3] == 0.0 and ov[3, 0] == 0.0
    iou = oracle.boxes_bev(a, a, "iou")
    np.testing.assert_allclose(iou[0, 1], 4.0 / 12.0, rtol=1e-5)
    np.testing.assert_allclose(iou[0, 2], 6.0 / 10.0, rtol=1e-5)
    np.testing.assert_allclose(np.diag(iou), 1.0, atol=1e-5)


def test_empty():
    z = np.zeros((0, 7), np.float32)
    assert oracle.boxes_bev(z, random_boxes(np.random.default_rng(1), 3)).shape == (0, 3)
    assert oracle.nms(z, 0.5).shape == (0,)


@pytest.mark.parametrize("rotated", [True, False])
def test_nms_matches_bruteforce(rotated):
    rng = np.random.default_rng(3)
    b = random_boxes(rng, 300, 12.0)
    if not rotated:
        iou = np.zeros((300, 300))
        for i in range(300):
            for j in range(300):
                l, r = max(b[i, 0] - b[i, 3] / 2, b[j, 0] - b[j, 3] / 2), min(b[i, 0] + b[i, 3] / 2, b[j, 0] + b[j, 3] / 2)
                t, u = max(b[i, 1] - b[i, 4] / 2, b[j, 1] - b[j, 4] / 2), min(b[i, 1] + b[i, 4] / 2, b[j, 1] + b[j, 4] / 2)
                inter = max(r - l, 0) * max(u - t, 0)
                iou[i, j] = inter / max(b[i, 3] * b[i, 4] + b[j, 3] * b[j, 4] - inter, 1e-8)
    else:
        iou = oracle.boxes_bev(b, b, "iou")
    thresh = 0.1
    removed = np.zeros(300, bool)
    keep = []
    for i in range(300):
        if removed[i]:
            continue
        keep.append(i)
        removed[i + 1:] |= iou[i, i + 1:] > thresh
    got = oracle.nms(b, thresh, rotated=rotated)
    assert 20 < len(keep) < 290
    np.testing.assert_array_equal(got, np.array(keep))
