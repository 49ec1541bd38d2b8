"""Micro-benchmarks of the individual HIP ops at ConQueR sizes (run on the GPU box).
usage: python scripts/bench_ops.py [msda] [spconv] [voxelize] [dense]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def bench_msda():
    from efg_amd.operators.box_attention_func import box_attn_backward, box_attn_forward

    g = torch.Generator().manual_seed(0)
    for name, b, lq in [("encoder", 2, 188 * 188), ("decoder", 2, 1240)]:
        s, h, d, p = 188 * 188, 8, 32, 25
        value = torch.randn(b, s, h, d, generator=g).to(dev)
        if name == "encoder":  # box self-attention: 5x5 lattice over a ~4.7-cell box around each token
            ys, xs = torch.meshgrid(torch.arange(188.0), torch.arange(188.0), indexing="ij")
            cen = torch.stack([(xs + 0.5) / 188, (ys + 0.5) / 188], -1).view(1, s, 1, 1, 1, 2)
            k = torch.linspace(-2, 2, 5) / 5
            ky, kx = torch.meshgrid(k, k, indexing="ij")
            lat = torch.stack([kx, ky], -1).view(1, 1, 1, 1, 25, 2) * 0.025
            loc = (cen + lat * (1 + 0.1 * torch.rand(b, s, h, 1, 1, 2, generator=g))).contiguous().to(dev)
        else:
            loc = torch.rand(b, lq, h, 1, p, 2, generator=g).to(dev)
        attn = torch.softmax(torch.randn(b, lq, h, p, generator=g), -1).view(b, lq, h, 1, p).to(dev)
        go = torch.randn(b, lq, h * d, generator=g).to(dev)
        shapes = torch.tensor([[188, 188]], device=dev)
        start = torch.zeros(1, dtype=torch.int64, device=dev)
        tf = timeit(lambda: box_attn_forward(value, shapes, start, loc, attn, 64))
        tb = timeit(lambda: box_attn_backward(value, shapes, start, loc, attn, go, 64))
        fwd_bytes = 4 * (b * s * h * d + b * lq * h * p * 3 + b * lq * h * d)
        print("msda %-8s fwd %8.1f us (%6.1f GB/s alg)   bwd %8.1f us" % (name, tf, fwd_bytes / tf / 1e3, tb))


def bench_box():
    """Fused Box3dAttention sampling at the encoder / decoder shapes (csrc/box_fused.hip)."""
    from efg_amd.operators.box_attention_func import BoxAttnFusedFunction

    g = torch.Generator().manual_seed(0)
    s, h, d, p = 188 * 188, 8, 32, 25
    shapes = torch.tensor([[188, 188]], device=dev)
    start = torch.zeros(1, dtype=torch.int64, device=dev)
    k = torch.linspace(-2, 2, 5) / 5
    ky, kx = torch.meshgrid(k, k, indexing="ij")
    kidx = torch.stack([kx, ky], -1).view(-1, 2).to(dev)
    for name, b, lq, v in [("encoder", 2, s, 4), ("decoder", 2, 1240, 5)]:
        value = torch.randn(b, s, h, d, generator=g).to(dev).requires_grad_(True)
        if name == "encoder":
            ys, xs = torch.meshgrid(torch.arange(188.0), torch.arange(188.0), indexing="ij")
            ref = torch.zeros(b, s, 7)
            ref[..., 0], ref[..., 1] = ((xs + 0.5) / 188).reshape(-1), ((ys + 0.5) / 188).reshape(-1)
            ref[..., 3] = ref[..., 4] = 0.025
        else:
            ref = torch.rand(b, lq, 7, generator=g)
            ref[..., 3:5] = ref[..., 3:5] * 0.05 + 0.01
        ref = ref.to(dev)
        # EFG_BOX_OFF_SCALE > 1: boxes grown far beyond their anchors, as after some training (the encoder's backward then
        # leaves its 16 x 16 LDS window for many corners)
        off = (torch.rand(b, lq, h * v, generator=g) * float(os.environ.get("EFG_BOX_OFF_SCALE", "1"))).to(dev).requires_grad_(True)
        logits = torch.randn(b, lq, h * p, generator=g).to(dev).requires_grad_(True)
        go = torch.randn(b, lq, h * d, generator=g).to(dev)
        out = BoxAttnFusedFunction.apply(value, shapes, start, ref, off, logits, kidx, v)
        tf = timeit(lambda: BoxAttnFusedFunction.apply(value, shapes, start, ref, off, logits, kidx, v))
        tb = timeit(lambda: torch.autograd.grad(out, (value, off, logits), go, retain_graph=True))
        alg = 4 * (b * s * h * d + b * lq * (7 + h * (v + p)) + b * lq * h * d)
        print("box fused %-8s fwd %8.1f us (%6.1f GB/s alg)   bwd %8.1f us (incl. zero-fill of grad_value)" % (
            name, tf, alg / tf / 1e3, tb))


def bench_gemm3():
    """Split-precision (bf16 x 3) GEMM against the fp32 library product at the encoder's Linear shapes."""
    from efg_amd.operators import gemm_bf16x3 as G

    g = torch.Generator().manual_seed(0)
    for m, k, n in [(70688, 256, 256), (70688, 256, 1024), (70688, 1024, 256), (70688, 256, 200), (70688, 200, 256),
                    (70688, 256, 32), (70688, 32, 256)]:
        a = torch.randn(m, k, generator=g).to(dev)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
        b = torch.randn(n, generator=g).to(dev)
        pk = G.pack_linear(w, False)
        t3 = timeit(lambda: G.gemm(a, pk, n, bias=b))
        tp = timeit(lambda: G.pack_linear(w, False))
        t32 = timeit(lambda: torch.addmm(b, a, w.t()))
        ref = a[:4096].double() @ w.double().t() + b.double()
        e3 = float((G.gemm(a[:4096], pk, n, bias=b).double() - ref).abs().max() / ref.abs().max())
        e32 = float((torch.addmm(b, a[:4096], w.t()).double() - ref).abs().max() / ref.abs().max())
        hbm = 4 * (m * k + m * n)
        print("gemm %6d x %4d x %4d  bf16x3 %7.1f us (%5.2f TB/s of A + C, %6.1f TFLOP/s fp32-equivalent; pack %5.1f us)   fp32 %7.1f us"
              "   max err / max |c|: %.1e vs %.1e" % (m, k, n, t3, hbm / t3 / 1e6, 2.0 * m * k * n / t3 / 1e6, tp, t32, e3, e32))


def bench_wgrad3():
    """Split-precision weight gradient g^T x against the fp32 products (library, and operators/linear.py's 16-chunk bmm)."""
    from efg_amd.operators import gemm_bf16x3 as G
    from efg_amd.operators.linear import weight_grad

    g = torch.Generator().manual_seed(0)
    for m, n, k in [(70688, 256, 256), (70688, 1024, 256), (70688, 256, 1024), (70688, 200, 256), (70688, 32, 256)]:
        go = torch.randn(m, n, generator=g).to(dev)
        x = torch.randn(m, k, generator=g).to(dev)
        t3 = timeit(lambda: G.wgrad(go, x))
        t32 = timeit(lambda: weight_grad(x, go))
        ref = go.double().t() @ x.double()
        e3 = float((G.wgrad(go, x).double() - ref).abs().max() / ref.abs().max())
        e32 = float((weight_grad(x, go).double() - ref).abs().max() / ref.abs().max())
        print("wgrad %6d rows -> %4d x %4d  bf16x3 %7.1f us (%5.2f TB/s of g + x)   fp32 (16-chunk bmm) %7.1f us   max err / max |dw|: %.1e vs %.1e"
              % (m, n, k, t3, 4.0 * m * (n + k) / t3 / 1e6, t32, e3, e32))


def bench_spconv():
    import efg_amd.spconv as spconv
    from efg_amd import _prof
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene
    from efg_amd.modeling.backbones import build_sparse_resnet_backbone
    from efg_amd.operators import voxelize_batch

    pts = [torch.from_numpy(make_scene(2000 + i)[0]).to(dev) for i in range(2)]
    vox = voxelize_batch(pts, VOXEL_SIZE, PC_RANGE, 5, 120000)
    cfg = dict(depth=18, out_features=["res2", "res3", "res4"], num_groups=1, norm="BN1d",
               activation=dict(type="ReLU", inplace=True), width_per_group=64, res1_out_channels=64,
               stem_out_channels=32)
    net = build_sparse_resnet_backbone(cfg, 5).to(dev)
    net.dense_features = ["res3", "res4"]
    feats, coors = vox["voxel_mean"], vox["coordinates"]
    print("input voxels", feats.shape)

    def fwd_bwd():
        out = net(feats, coors, 2, [1504, 1504, 40])
        (out["res3"].sum() + out["res4"].sum()).backward()

    print("sparse backbone fwd+bwd %.1f us" % timeit(fwd_bwd, iters=5, warm=2))
    _prof.enable(True)
    fwd_bwd()
    torch.cuda.synchronize()
    _prof.enable(False)
    if "--detail" in sys.argv:
        import efg_amd.spconv.core as core
        for name, recs in sorted(_prof._records.items()):
            for (s_, e_, cost) in recs:
                b, f = cost()
                meta = getattr(cost, "meta", "")
                print("    %-40s %8.1f us  %7.2f GFLOP  %6.1f TF/s  %s" % (name, s_.elapsed_time(e_) * 1e3, f / 1e9, f / s_.elapsed_time(e_) / 1e9, meta))
    for k, v in sorted(_prof.summary().items()):
        print("  %-14s launches %3d total %8.2f ms  %7.1f GB/s alg  %6.2f TFLOP/s" % (
            k, v["launches"], v["total_ms"], v["bytes"] / v["total_ms"] / 1e6, v["flops"] / v["total_ms"] / 1e9))
    x = spconv.SparseConvTensor(feats, coors, [41, 1504, 1504], 2)
    lvl = net.stem(x)
    names = ["stem"]
    levels = [lvl]
    for stage, name in net.stages_and_names:
        lvl = stage(lvl)
        levels.append(lvl)
        names.append(name)
    for n, l in zip(names, levels):
        rb = [v for k, v in l.indice_dict.items() if k[1] == n or n == "stem" and k[1] == "stem"]
        pairs = rb[0].num_pairs() if rb else -1
        print("  level %-5s sites %7d channels %3d subm pairs/site %.2f" % (
            n, l.features.shape[0], l.features.shape[1], pairs / max(l.features.shape[0], 1)))


def bench_voxelize():
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene
    from efg_amd.operators.voxelize import _hard_voxelize_launch

    for n, sw, mv, nb in [(180000, 1, 120000, 2), (720000, 4, 200000, 1), (180000, 1, 120000, 8)]:
        scenes = [torch.from_numpy(make_scene(2000 + i, n_points=n, n_sweeps=sw)[0]).to(dev) for i in range(nb)]
        pts = torch.cat(scenes)
        offs = [0]
        for s in scenes:
            offs.append(offs[-1] + s.shape[0])
        f = pts.shape[1]
        cap = nb * mv
        voxels = torch.empty((cap, 5, f), device=dev)
        coors = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        npv = torch.empty(cap, dtype=torch.int32, device=dev)
        mean = torch.empty((cap, f), device=dev)
        num = torch.zeros(nb, dtype=torch.int32, device=dev)
        t = timeit(lambda: _hard_voxelize_launch(pts, offs, VOXEL_SIZE, PC_RANGE, 5, mv, voxels, coors, npv, num, mean))
        m = int(num.sum())
        alg = 4 * f * pts.shape[0] + m * (4 * 5 * f + 16 + 4 + 4 * f)
        print("hard_voxelize %d scenes x %d pts x %d feats -> %d voxels: %8.1f us  (%6.1f GB/s algorithmic)" % (
            nb, n, f, m, t, alg / t / 1e3))


def bench_iou3d():
    """Rotated BEV IoU / NMS at CenterPoint-style post-processing sizes (pre_maxsize 4096)."""
    from efg_amd import _lib
    from efg_amd.operators import iou3d_nms

    rng = np.random.default_rng(0)
    for n, extent in [(1000, 75.0), (4096, 75.0), (4096, 20.0)]:
        b = np.zeros((n, 7), np.float32)
        b[:, 0:2] = rng.uniform(-extent, extent, (n, 2))
        b[:, 3] = rng.uniform(1.5, 5.0, n)
        b[:, 4] = rng.uniform(0.6, 3.0, n)
        b[:, 5] = 1.5
        b[:, 6] = rng.uniform(-np.pi, np.pi, n)
        t_b = torch.from_numpy(b).to(dev)
        scores = torch.rand(n, device=dev)
        t_iou = timeit(lambda: iou3d_nms.boxes_iou_bev(t_b, t_b))
        frac = float((iou3d_nms.boxes_iou_bev(t_b, t_b) > 0).float().mean())
        order = scores.sort(0, descending=True)[1]
        bs = t_b[order].contiguous()
        keep = torch.empty(n, dtype=torch.int64, device=dev)
        num = torch.empty(1, dtype=torch.int32, device=dev)
        wsb = _lib.lib().efg_nms_workspace_bytes(n)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        t_nms = timeit(lambda: _lib.check(_lib.lib().efg_nms_f32(_lib.ptr(bs), n, 0.7, 1, _lib.ptr(keep), _lib.ptr(num),
                                                                 _lib.ptr(ws), wsb, _lib.stream())))
        t_full = timeit(lambda: iou3d_nms.nms_gpu(t_b, scores, 0.7))
        print("iou3d n=%d extent=%.0f: boxes_iou_bev %8.1f us (%.2f Gpair/s, %.2f%% overlapping, %5.1f GB/s out)  "
              "nms kernels %8.1f us  nms_gpu incl. sort+sync %8.1f us  kept %d" % (
                  n, extent, t_iou, n * n / t_iou / 1e3, 100 * frac, 4 * n * n / t_iou / 1e3, t_nms, t_full, int(num)))


def bench_augment():
    """Device augmentation chain (flip + rotate + scale + range filter, then shuffle gather) per 180k-point scene."""
    from efg_amd.data import gpu_pipeline as gp
    from efg_amd.data.synthetic import PC_RANGE, make_scene

    for n, sw in [(180000, 1), (720000, 4)]:
        pts = torch.from_numpy(make_scene(7, n_points=n, n_sweeps=sw)[0]).to(dev)
        f = pts.shape[1]

        def fused():
            dp = gp.DevicePoints(pts)
            dp.queue(gp.NEG_Y)
            dp.queue(gp.ROT_Z, 0.9, 0.43589)
            dp.queue(gp.SCALE, 1.05)
            return dp, dp.materialize(PC_RANGE)

        t = timeit(lambda: fused())
        dp, m = fused()
        idx = torch.randperm(m, device=dev)
        out = torch.empty_like(dp.tensor)
        from efg_amd import _lib as L
        tg = timeit(lambda: L.check(L.lib().efg_points_gather_f32(L.ptr(dp.tensor), L.ptr(idx), m, f, L.ptr(out),
                                                                   L.stream())))
        print("augment %d pts x %d feats: transform+filter %7.1f us (incl. count read-back; %5.1f GB/s alg)  "
              "shuffle gather %7.1f us  kept %d" % (n, f, t, 2 * 4 * f * n / t / 1e3, tg, m))


def bench_copy():
    """Practical HBM ceiling: a device-to-device copy."""
    for mb in (72, 290, 1024, 4096):
        a = torch.empty(mb * 1024 * 1024 // 4, device=dev)
        b = torch.empty_like(a)
        t = timeit(lambda: b.copy_(a))
        print("copy %5d MB: %8.1f us  %6.2f TB/s (read + write)" % (mb, t, 2 * a.numel() * 4 / t / 1e6))


def bench_colsum():
    """Bias-gradient column sums: csrc/colsum.hip vs ATen's sum(0)."""
    from efg_amd.operators.linear import column_sum
    for rows, cols in [(70688, 256), (70688, 1024), (70688, 200), (70688, 32), (2480, 256), (2800, 256), (2480, 1024)]:
        x = torch.randn(rows, cols, device=dev)
        t = timeit(lambda: column_sum(x))
        ta = timeit(lambda: x.sum(0))
        print("colsum %6d x %4d: %7.1f us (%5.2f TB/s)   aten sum(0) %7.1f us" % (rows, cols, t, rows * cols * 4 / t / 1e6, ta))


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["msda", "spconv", "voxelize"]
    for w in which:
        globals()["bench_" + w]()
