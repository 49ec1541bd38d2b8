"""GPU: fused detection-loss kernels (csrc/det_loss.hip) against the PyTorch composite they replace (the batched
restatement of $CQ/modules/matcher.py:40-80 and $CQ/losses.py:26-108 in detection3d/, itself golden-pinned on CPU).
fp32: values 1e-5, gradients 1e-4 relative to the tensor max."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _problem(seed, L=3, B=2, Q=300, C=3, G=17):
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(L, B, Q, C, generator=g) * 2 - 2).cuda()
    boxes = torch.rand(L, B, Q, 7, generator=g)
    boxes[..., 3:6] = boxes[..., 3:6] * 0.2 + 0.01
    boxes = boxes.cuda()
    tgt_labels = torch.randint(0, C, (B, G), generator=g).cuda()
    tgt_boxes = torch.rand(B, G, 7, generator=g)
    tgt_boxes[..., 3:6] = tgt_boxes[..., 3:6] * 0.2 + 0.01
    return logits, boxes, tgt_labels, tgt_boxes.cuda()


def test_match_cost_matches_composite():
    from efg_amd.detection3d.utils import box_cxcyczlwh_to_xyxyxy, pairwise_box3d_giou
    from efg_amd.operators.det_loss import match_cost

    logits, boxes, tl, tb = _problem(0)
    L, B, Q, C = logits.shape
    G = tl.shape[1]
    got = match_cost(logits, boxes, tl, tb, 1.0, 4.0, 2.0, 4.0).view(L, B, Q, G)
    p = logits.sigmoid()
    neg = 0.75 * p ** 2 * (-(1 - p + 1e-8).log())
    pos = 0.25 * (1 - p) ** 2 * (-(p + 1e-8).log())
    lab = tl[None, :, None, :].expand(L, B, Q, G)
    cc = torch.gather(pos, 3, lab) - torch.gather(neg, 3, lab)
    cb = (boxes[..., None, :6] - tb[None, :, None, :, :6]).abs().sum(-1)
    cr = (boxes[..., None, 6:] - tb[None, :, None, :, 6:]).abs().sum(-1)
    cg = -pairwise_box3d_giou(box_cxcyczlwh_to_xyxyxy(boxes[..., :6]), box_cxcyczlwh_to_xyxyxy(tb[..., :6])[None])
    ref = 4.0 * cb + 1.0 * cc + 2.0 * cg + 4.0 * cr
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)


def _close(a, b, name):
    scale = float(b.abs().max()) + 1e-12
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-4 * scale, err_msg=name)


@pytest.mark.parametrize("shape", [(3, 2, 300, 3), (1, 2, 70688, 1), (2, 1, 5, 4)])
def test_focal_layers(shape):
    from efg_amd.detection3d.utils import sigmoid_focal_loss
    from efg_amd.operators.det_loss import FocalLossLayers, device_scalar

    g = torch.Generator().manual_seed(sum(shape))
    L, B, Q, C = shape
    logits = (torch.randn(shape, generator=g) * 3).cuda().requires_grad_(True)
    tcls = torch.full((L, B, Q), -1, dtype=torch.int32)
    pick = torch.rand(L, B, Q, generator=g) < 0.05
    tcls[pick] = torch.randint(0, C, (int(pick.sum()),), generator=g, dtype=torch.int32)
    tcls = tcls.cuda()
    denom = 37.0
    w = torch.randn(L, generator=g).cuda()
    out = FocalLossLayers.apply(logits, tcls, device_scalar(denom, logits.device), 0.25, 2.0)
    (out * w).sum().backward()
    g_fused, logits.grad = logits.grad.clone(), None
    onehot = F.one_hot(tcls.long().clamp(min=0), C).float() * (tcls >= 0)[..., None]
    ref = sigmoid_focal_loss(logits, onehot, alpha=0.25, gamma=2.0, reduction="none").sum(dim=(1, 2, 3)) / denom
    (ref * w).sum().backward()
    torch.testing.assert_close(out, ref, rtol=2e-5, atol=1e-6)
    _close(g_fused, logits.grad, "grad logits")


def test_focal_sum_over_split_workgroups_is_reproducible():
    """From ~16k elements per layer the focal sum runs on several workgroups per layer; the one that draws the layer's last
    ticket adds the partial sums in index order and zeroes its ring slots: a hundred calls (two shapes interleaved, so the
    slots are reused by calls of another size) give the first call's bits."""
    from efg_amd.operators.det_loss import FocalLossLayers, device_scalar

    g = torch.Generator().manual_seed(1)
    cases = []
    for shape in [(1, 2, 35344, 1), (3, 2, 9000, 3)]:
        logits = (torch.randn(shape, generator=g) * 3).cuda()
        tcls = torch.randint(-1, shape[-1], shape[:-1], generator=g, dtype=torch.int32).cuda()
        cases.append((logits, tcls, device_scalar(11.0, logits.device)))
    first = [FocalLossLayers.apply(lg, tc, dn, 0.25, 2.0).clone() for lg, tc, dn in cases]
    for _ in range(100):
        for (lg, tc, dn), want in zip(cases, first):
            assert torch.equal(FocalLossLayers.apply(lg, tc, dn, 0.25, 2.0), want)


def test_box_loss_layers():
    from efg_amd.detection3d.utils import box_cxcyczlwh_to_xyxyxy, paired_box3d_giou
    from efg_amd.operators.det_loss import BoxLossLayers, device_scalar

    _, boxes, _, tb = _problem(5)
    L, B, Q, _ = boxes.shape
    G = tb.shape[1]
    g = torch.Generator().manual_seed(9)
    l_idx = torch.arange(L).repeat_interleave(B * G)
    b_idx = torch.arange(B).repeat_interleave(G).repeat(L)
    g_idx = torch.arange(G).repeat(L * B)
    q_idx = torch.stack([torch.randperm(Q, generator=g)[:G] for _ in range(L * B)]).flatten()
    # make some pairs overlap strongly / exactly share a face (kinks of min / max)
    boxes = boxes.clone()
    boxes[l_idx[:40], b_idx[:40], q_idx[:40]] = tb[b_idx[:40], g_idx[:40]].cpu().cuda() + 0.01
    boxes.requires_grad_(True)
    idx = [t.cuda() for t in (l_idx, b_idx, q_idx, g_idx)]
    denom = 11.0
    w = torch.randn(L, 3, generator=g).cuda()
    out = BoxLossLayers.apply(boxes, tb, *idx, device_scalar(denom, boxes.device))
    (out * w).sum().backward()
    g_fused, boxes.grad = boxes.grad.clone(), None
    src, tgt = boxes[idx[0], idx[1], idx[2]], tb[idx[1], idx[3]]
    l1 = F.l1_loss(src, tgt, reduction="none")
    giou = 1 - paired_box3d_giou(box_cxcyczlwh_to_xyxyxy(src[:, :6]), box_cxcyczlwh_to_xyxyxy(tgt[:, :6]))
    per = torch.stack((l1[:, :6].sum(1), giou, l1[:, 6:].sum(1)), dim=1)
    ref = per.new_zeros(L, 3).index_add_(0, idx[0], per) / denom
    (ref * w).sum().backward()
    torch.testing.assert_close(out, ref, rtol=2e-5, atol=1e-6)
    _close(g_fused, boxes.grad, "grad boxes")


def test_box_refine_matches_the_reference_formulation():
    dev = torch.device("cuda:0")
    """sigmoid(delta + inverse_sigmoid(anchor)) ($CQ/heads.py:78, $CQ/modules/utils.py:83-87) in one launch each way,
    incl. anchors outside [0, 1] and at the eps clamps."""
    from efg_amd.operators.det_loss import box_refine

    g = torch.Generator().manual_seed(3)
    delta = (torch.randn(2, 1240, 7, generator=g) * 3).to(dev).requires_grad_(True)
    anchor = torch.rand(2, 1240, 7, generator=g)
    anchor[0, :7, 0] = torch.tensor([0.0, 1.0, -0.3, 1.7, 1e-6, 1 - 1e-6, 0.5])
    anchor = anchor.to(dev)
    out = box_refine(delta, anchor)
    x = anchor.clamp(min=0, max=1)
    d2 = delta.detach().clone().requires_grad_(True)
    ref = (d2 + torch.log(x.clamp(min=1e-5) / (1 - x).clamp(min=1e-5))).sigmoid()
    torch.testing.assert_close(out, ref, rtol=2e-6, atol=2e-7)
    w = torch.randn(out.shape, generator=g).to(dev)
    (out * w).sum().backward()
    (ref * w).sum().backward()
    torch.testing.assert_close(delta.grad, d2.grad, rtol=1e-5, atol=1e-7)
    with torch.no_grad():   # the momentum decoder's (graph-captured) use
        torch.testing.assert_close(box_refine(delta, anchor), ref.detach(), rtol=2e-6, atol=2e-7)


def test_box_refine_returns_the_anchor_gradient_of_undetached_boxes():
    """The model's per-layer heads refine the previous layer's boxes WITH their graph ($CQ/voxel_detr.py:171-180): the
    anchor gradient comes out of the same backward launch, the clamps of inverse_sigmoid differentiated as autograd
    differentiates them (zero outside [0, 1] and below the eps clamps)."""
    from efg_amd.operators.det_loss import box_refine

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    delta = (torch.randn(2, 1240, 7, generator=g) * 2).to(dev).requires_grad_(True)
    a0 = torch.rand(2, 1240, 7, generator=g)
    a0[0, :9, 0] = torch.tensor([0.0, 1.0, -0.3, 1.7, 1e-6, 1 - 1e-6, 0.5, 2e-5, 1 - 2e-5])
    anchor = a0.to(dev).requires_grad_(True)
    out = box_refine(delta, anchor)
    assert type(out.grad_fn).__name__ == "BoxRefineFunctionBackward"
    d2, a2 = delta.detach().clone().requires_grad_(True), anchor.detach().clone().requires_grad_(True)
    x = a2.clamp(min=0, max=1)
    ref = (d2 + torch.log(x.clamp(min=1e-5) / (1 - x).clamp(min=1e-5))).sigmoid()
    w = torch.randn(out.shape, generator=g).to(dev)
    (out * w).sum().backward()
    (ref * w).sum().backward()
    torch.testing.assert_close(out, ref, rtol=2e-6, atol=2e-7)
    torch.testing.assert_close(delta.grad, d2.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(anchor.grad, a2.grad, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("n,k", [(35344, 1000), (35344, 900), (5000, 5000), (1500, 1), (70000, 300)])
def test_topk_unsorted_is_exact_and_breaks_ties_by_index(n, k):
    """csrc/topk.hip against torch.topk: the same VALUES (multiset), indices in ascending order; with a plateau of equal
    scores at the cut (every empty BEV cell of a fresh model) the lowest indices of the plateau are taken, every time."""
    from efg_amd.operators.det_loss import topk_unsorted

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n + k)
    x = torch.rand(3, n, generator=g)
    x[1, torch.randperm(n, generator=g)[: n // 2]] = 0.25        # half the row on one plateau: the cut falls inside it
    x[2] = torch.randn(n, generator=g)                           # negative values too
    x[2, :7] = float("-inf")
    x = x.to(dev)
    v, i = topk_unsorted(x, k)
    tv, ti = torch.topk(x, k, dim=1, sorted=True)
    assert torch.equal(torch.sort(v, dim=1, descending=True)[0], tv)          # the same values, exactly
    assert torch.equal(torch.gather(x, 1, i), v)
    assert bool((i[:, 1:] > i[:, :-1]).all()) or k == 1                       # ascending, no duplicates
    for r in range(3):                                                         # tie rule: lowest indices of the cut value
        cut = tv[r, -1]
        eq = torch.nonzero(x[r] == cut).flatten()
        taken = i[r][torch.gather(x[r], 0, i[r]) == cut]
        assert torch.equal(taken, eq[: taken.numel()])
    v2, i2 = topk_unsorted(x, k)
    assert torch.equal(i2, i) and torch.equal(v2, v)
