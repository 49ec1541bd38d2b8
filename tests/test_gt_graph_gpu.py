"""The momentum ("GT") decoder replayed from a HIP graph gives the same outputs as the eager pass."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gt_decoder_graph_matches_eager(monkeypatch):
    from efg_amd.engine import Trainer

    tr = Trainer(device="cuda:0", overrides={"model.transformer.num_queries": 60}, seed=0)
    t = tr.model.transformer
    g = torch.Generator(device="cuda").manual_seed(0)
    memory = torch.randn(2, 188 * 188, 256, device="cuda", generator=g) * 0.1
    shape = torch.tensor([[188, 188]], device="cuda")
    start = torch.zeros(1, dtype=torch.int64, device="cuda")
    for n in (48, 96, 48):  # two shapes, the first one twice (capture, capture, replay)
        props = torch.rand(2, n, 10, device="cuda", generator=g) * 0.8 + 0.1
        grp = torch.arange(n, device="cuda") // 16
        mask = grp[:, None] != grp[None, :]
        with torch.no_grad():
            monkeypatch.setenv("EFG_GT_GRAPH", "0")
            eager = t._run_gt_decoder(memory, shape, start, props, mask)
            monkeypatch.setenv("EFG_GT_GRAPH", "1")
            graphed = t._run_gt_decoder(memory, shape, start, props, mask)
            graphed = [x.clone() for x in graphed]  # static buffers: copy before the next replay
        assert not getattr(t, "_gt_graph_off", False)
        for a, b in zip(eager, graphed):
            assert a.shape == b.shape
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), float((a - b).abs().max())
    assert len(t._gt_graphs) == 2 and t._gt_graph_stats == [1, 2]
    # with autograd on (not the momentum decoder's situation) the eager path is taken: nothing new is captured
    t._run_gt_decoder(memory, shape, start, props, mask)
    assert t._gt_graph_stats == [1, 2]
