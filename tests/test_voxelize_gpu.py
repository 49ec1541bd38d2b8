"""GPU parity: HIP voxelizers (through the C ABI) vs the oracle and the reference's golden
vectors.  Bar: bit-exact (integer indices, copied fp32 payloads)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu
CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "voxelize_*.npz")))


def _run_hard(dev, pts, vs, cr, mp, mv):
    from efg_amd.operators import voxelization

    out = voxelization(torch.from_numpy(pts).to(dev), list(map(float, vs)), list(map(float, cr)), mp, mv)
    return [o.cpu().numpy() for o in out]


@pytest.mark.parametrize("case", CASES)
def test_hard_voxelize_golden(dev, case):
    g = golden(case)
    v, c, n = _run_hard(dev, g["points"], g["voxel_size"], g["coors_range"], int(g["max_points"]),
                        int(g["max_voxels"]))
    assert v.shape == g["voxels"].shape
    assert np.array_equal(c, g["coors"])
    assert np.array_equal(n, g["num_points_per_voxel"])
    assert np.array_equal(v, g["voxels"])


@pytest.mark.parametrize("case", CASES)
def test_dynamic_voxelize_golden(dev, case):
    from efg_amd.operators import voxelization

    g = golden(case)
    coors = voxelization(torch.from_numpy(g["points"]).to(dev), g["voxel_size"].tolist(), g["coors_range"].tolist(),
                         -1, -1)
    assert np.array_equal(coors.cpu().numpy(), g["dynamic_coors"])


@pytest.mark.parametrize("n,sweeps,mp,mv", [(180000, 1, 5, 120000), (180000, 1, 5, 30000), (720000, 4, 5, 200000),
                                            (720000, 4, 5, 50000), (50000, 1, 1, 120000)])
def test_hard_voxelize_full_size_vs_oracle(dev, oracle_mod, n, sweeps, mp, mv):
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene

    pts, _, _ = make_scene(2000 + n % 97 + mv % 13, n_points=n, n_sweeps=sweeps)
    ev, ec, en = oracle_mod.hard_voxelize(pts, VOXEL_SIZE, PC_RANGE, mp, mv)
    v, c, k = _run_hard(dev, pts, VOXEL_SIZE, PC_RANGE, mp, mv)
    assert np.array_equal(c, ec) and np.array_equal(k, en) and np.array_equal(v, ev)


def test_module_api_and_modes(dev, oracle_mod):
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene
    from efg_amd.operators import Voxelization

    pts, _, _ = make_scene(5, n_points=20000)
    m = Voxelization(list(VOXEL_SIZE), list(PC_RANGE), 5, max_voxels=(3000, 9000))
    assert m.grid_size.tolist() == [1504, 1504, 40]
    t = torch.from_numpy(pts).to(dev)
    m.train()
    v, c, n = m(t)
    assert v.shape[0] == 3000 and c.dtype == torch.int32 and n.dtype == torch.int32
    m.eval()
    v2, c2, n2 = m(t)
    e = oracle_mod.hard_voxelize(pts, VOXEL_SIZE, PC_RANGE, 5, 9000)
    assert np.array_equal(v2.cpu().numpy(), e[0]) and np.array_equal(c2.cpu().numpy(), e[1])
    assert "max_voxels=(3000, 9000)" in repr(m)


def test_edge_cases(dev, oracle_mod):
    from efg_amd.operators import voxelization

    vs, cr = [0.5, 0.5, 0.5], [0.0, 0.0, 0.0, 4.0, 4.0, 4.0]
    # empty input
    v, c, n = voxelization(torch.zeros((0, 4), device=dev), vs, cr, 3, 10)
    assert v.shape == (0, 3, 4) and c.shape == (0, 3) and n.shape == (0,)
    # everything outside, NaN counted as outside
    pts = torch.full((9, 4), -3.0, device=dev)
    pts[3, 0] = float("nan")
    v, c, n = voxelization(pts, vs, cr, 3, 10)
    assert v.shape[0] == 0
    assert (voxelization(pts, vs, cr, -1, -1) == -1).all()
    # one voxel, many points (slot ranks must follow point order)
    rng = np.random.default_rng(0)
    p = np.concatenate([rng.uniform(1.0, 1.49, (300, 3)), rng.uniform(0, 1, (300, 1))], 1).astype(np.float32)
    v, c, n = voxelization(torch.from_numpy(p).to(dev), vs, cr, 7, 4)
    e = oracle_mod.hard_voxelize(p, vs, cr, 7, 4)
    assert np.array_equal(v.cpu().numpy(), e[0]) and np.array_equal(n.cpu().numpy(), e[2])
    # CPU tensors are refused loudly (no fallback)
    with pytest.raises(RuntimeError):
        voxelization(torch.zeros((4, 4)), vs, cr, 3, 10)


def test_batched_matches_per_scene(dev, oracle_mod):
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene
    from efg_amd.operators import voxelize_batch

    scenes = [make_scene(300 + i, n_points=n)[0] for i, n in enumerate([40000, 25000, 0, 33000])]
    scenes[2] = np.zeros((0, 5), np.float32)
    mv = 20000
    out = voxelize_batch([torch.from_numpy(s).to(dev) for s in scenes], VOXEL_SIZE, PC_RANGE, 5, mv)
    base = 0
    for b, s in enumerate(scenes):
        ev, ec, en = oracle_mod.hard_voxelize(s, VOXEL_SIZE, PC_RANGE, 5, mv)
        m = out["num_voxels"][b]
        assert m == ev.shape[0]
        sl = slice(base, base + m)
        assert np.array_equal(out["voxels"][sl].cpu().numpy(), ev)
        co = out["coordinates"][sl].cpu().numpy()
        assert (co[:, 0] == b).all() and np.array_equal(co[:, 1:], ec)
        assert np.array_equal(out["num_points_per_voxel"][sl].cpu().numpy(), en)
        np.testing.assert_allclose(out["voxel_mean"][sl].cpu().numpy(), oracle_mod.voxel_mean(ev, en), rtol=1e-6,
                                   atol=1e-6)
        base += m
    assert out["voxels"].shape[0] == base


@pytest.fixture(params=["bins", "hash"])
def impl(request, monkeypatch):
    """Both hard voxelizers behind the one entry point (csrc/voxelize_bins.hip default, voxelize_hash.hip)."""
    monkeypatch.setenv("EFG_VOX_IMPL", request.param)
    return request.param


def _check_vs_oracle(dev, oracle_mod, pts, vs, cr, mp, mv):
    ev, ec, en = oracle_mod.hard_voxelize(pts, vs, cr, mp, mv)
    v, c, k = _run_hard(dev, pts, vs, cr, mp, mv)
    assert v.shape == ev.shape
    assert np.array_equal(c, ec) and np.array_equal(k, en) and np.array_equal(v, ev)


@pytest.mark.parametrize("order", ["shuffled", "scan", "sorted_x"])
def test_both_implementations_any_point_order(dev, oracle_mod, impl, order):
    """The binned voxelizer aggregates a tile's points per supercell: a cloud in scan order (long runs inside one
    supercell) and a shuffled one (every point a different supercell) take different paths through it."""
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene

    pts, _, _ = make_scene(77, n_points=60000)
    if order == "scan":
        pts = pts[np.lexsort((pts[:, 0], np.round(np.arctan2(pts[:, 1], pts[:, 0]), 2)))]
    elif order == "sorted_x":
        pts = pts[np.argsort(pts[:, 0], kind="stable")]
    for mp, mv in [(5, 120000), (5, 9000), (1, 500), (35, 20000)]:
        _check_vs_oracle(dev, oracle_mod, np.ascontiguousarray(pts), VOXEL_SIZE, PC_RANGE, mp, mv)


@pytest.mark.parametrize("f", [3, 4, 6, 8, 9, 13])
def test_feature_widths_and_unaligned_rows(dev, oracle_mod, impl, f):
    """Rows are staged through LDS with 16-byte loads when F <= 8 (any alignment of the first row), read in place
    otherwise; the buffer itself may start 4 bytes off a 16-byte boundary."""
    from efg_amd.operators import voxelization

    rng = np.random.default_rng(f)
    n = 30011
    pts = rng.uniform(-1, 1, (n, f)).astype(np.float32)
    pts[:, :3] = rng.uniform(-12.5, 12.5, (n, 3))          # ~4 % of the points fall outside the range
    vs, cr = [0.25, 0.25, 0.5], [-12.0, -12.0, -12.0, 12.0, 12.0, 12.0]
    ev, ec, en = oracle_mod.hard_voxelize(pts, vs, cr, 4, 7000)
    for shift in (0, 1, 3):
        buf = torch.empty(n * f + 4, device=dev)
        t = buf[shift:shift + n * f].view(n, f)
        t.copy_(torch.from_numpy(pts))
        assert t.data_ptr() % 16 == (buf.data_ptr() + 4 * shift) % 16
        v, c, k = voxelization(t, vs, cr, 4, 7000)
        assert np.array_equal(c.cpu().numpy(), ec) and np.array_equal(k.cpu().numpy(), en)
        assert np.array_equal(v.cpu().numpy(), ev)


def test_crowded_supercells(dev, oracle_mod, impl):
    """Bins far beyond the one-wave limit (256 points) and beyond any LDS list: 70k points inside two supercells, many
    of them in a handful of voxels; plus a grid taller than one supercell (two z-slabs)."""
    rng = np.random.default_rng(5)
    a = rng.uniform(0.0, 1.6, (50000, 3))            # one 16 x 16 column of 0.1 m cells
    b = rng.uniform(3.2, 3.3, (20000, 3))            # one voxel column, ~1 cell wide
    pts = np.concatenate([a, b, rng.uniform(0, 6.4, (3000, 3))]).astype(np.float32)
    pts = np.concatenate([pts, rng.uniform(0, 1, (len(pts), 2)).astype(np.float32)], 1)
    rng.shuffle(pts)
    vs, cr = [0.1, 0.1, 0.1], [0.0, 0.0, 0.0, 6.4, 6.4, 6.4]   # grid 64 x 64 x 64: nsz = 2
    for mp, mv in [(5, 50000), (3, 700), (40, 50000)]:
        _check_vs_oracle(dev, oracle_mod, pts, vs, cr, mp, mv)


def test_batched_cap_and_break_per_scene(dev, oracle_mod, impl):
    """The `break` (voxelization_cpu.cpp:78-79) is per scene: one scene of the batch hits max_voxels, the others do
    not; scenes of very different sizes share the launches."""
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene
    from efg_amd.operators import voxelize_batch

    sizes = [50000, 700, 90000, 1, 12000]
    scenes = [make_scene(900 + i, n_points=n)[0][:n] for i, n in enumerate(sizes)]
    mv = 15000
    out = voxelize_batch([torch.from_numpy(s).to(dev) for s in scenes], VOXEL_SIZE, PC_RANGE, 5, mv)
    base = 0
    for b, s in enumerate(scenes):
        ev, ec, en = oracle_mod.hard_voxelize(s, VOXEL_SIZE, PC_RANGE, 5, mv)
        m = out["num_voxels"][b]
        assert m == ev.shape[0]
        sl = slice(base, base + m)
        assert np.array_equal(out["voxels"][sl].cpu().numpy(), ev)
        assert np.array_equal(out["coordinates"][sl].cpu().numpy()[:, 1:], ec)
        assert np.array_equal(out["num_points_per_voxel"][sl].cpu().numpy(), en)
        base += m
    assert out["num_voxels"][0] == mv and out["num_voxels"][2] == mv and out["num_voxels"][1] < mv


def test_library_state_survives_changing_layouts(dev, oracle_mod, monkeypatch):
    """The binned path keeps its counters in a buffer the library owns per stream, left zero by the call itself (no clear
    launch).  Calls whose grids / batches / sizes differ -- i.e. whose counters and look-back records sit at different offsets
    -- alternate on one stream and every one of them is still bit-exact (the first version kept the records behind the
    counters: the stale records of a small grid were the next, larger layout's counters)."""
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene
    from efg_amd.operators import voxelize_batch

    monkeypatch.setenv("EFG_VOX_IMPL", "bins")
    rng = np.random.default_rng(11)
    small = np.concatenate([rng.uniform(0, 6.4, (70000, 3)), rng.uniform(0, 1, (70000, 2))], 1).astype(np.float32)
    waymo = [make_scene(950 + i, n_points=n)[0][:n] for i, n in enumerate([90000, 300, 120000, 70000, 2000])]
    for rnd in range(2):
        _check_vs_oracle(dev, oracle_mod, small, [0.1, 0.1, 0.1], [0.0, 0.0, 0.0, 6.4, 6.4, 6.4], 5, 50000)
        out = voxelize_batch([torch.from_numpy(s).to(dev) for s in waymo], VOXEL_SIZE, PC_RANGE, 5, 30000)
        base = 0
        for b, s_ in enumerate(waymo):
            ev, ec, en = oracle_mod.hard_voxelize(s_, VOXEL_SIZE, PC_RANGE, 5, 30000)
            m = out["num_voxels"][b]
            assert m == ev.shape[0]
            assert np.array_equal(out["voxels"][base:base + m].cpu().numpy(), ev)
            assert np.array_equal(out["coordinates"][base:base + m].cpu().numpy()[:, 1:], ec)
            base += m
        _check_vs_oracle(dev, oracle_mod, waymo[0], VOXEL_SIZE, PC_RANGE, 5, 120000)


def test_full_size_both_implementations_agree(dev, impl, oracle_mod):
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene

    pts, _, _ = make_scene(2024, n_points=720000, n_sweeps=4)
    _check_vs_oracle(dev, oracle_mod, pts, VOXEL_SIZE, PC_RANGE, 5, 200000)


@pytest.mark.parametrize("impl", ["bins", "hash"])
def test_hard_voxelize_writes_nothing_outside_its_buffers(impl, monkeypatch):
    """Every output and the workspace sit between two 1 MiB guard bands of a known byte: after a batched call at the
    benchmark's size (and a 6-feature, 4-scene one) the bands are untouched -- an out-of-bounds write inside the caching
    allocator's segments would otherwise corrupt a neighbour silently."""
    from efg_amd import _lib as L
    from efg_amd.engine import synthetic_batch

    dev = torch.device("cuda:0")
    monkeypatch.setenv("EFG_VOX_IMPL", impl)
    guard = 1 << 20

    def guarded(nbytes, dtype):
        raw = torch.full((nbytes + 2 * guard,), 0xAB, dtype=torch.uint8, device=dev)
        return raw, raw[guard:guard + nbytes].view(dtype)

    def run(pts, max_points=5, max_voxels=120000):
        offsets = [0]
        for p in pts:
            offsets.append(offsets[-1] + p.shape[0])
        points = torch.cat(pts, 0).contiguous()
        f, n, batch = points.shape[1], offsets[-1], len(pts)
        cap = min(batch * max_voxels, n)
        lib = L.lib()
        vs, cr = L.host_f32([0.1, 0.1, 0.15], 3), L.host_f32([-75.2, -75.2, -2.0, 75.2, 75.2, 4.0], 6)
        ws_bytes = lib.efg_hard_voxelize_workspace_bytes(n, batch, f, max_points, max_voxels, vs, cr)
        specs = {"voxels": (cap * max_points * f * 4, torch.float32), "coors": (cap * 16, torch.int32), "npv": (cap * 4, torch.int32),
                 "mean": (cap * f * 4, torch.float32), "num": (batch * 4, torch.int32), "ws": (ws_bytes, torch.uint8)}
        bufs = {k: guarded(*v) for k, v in specs.items()}
        bufs["num"][1].zero_()
        L.check(lib.efg_hard_voxelize_f32(L.ptr(points), L.host_i64(offsets), batch, f, vs, cr, max_points, max_voxels,
                                          bufs["voxels"][1].data_ptr(), bufs["coors"][1].data_ptr(), 4, bufs["npv"][1].data_ptr(),
                                          bufs["num"][1].data_ptr(), bufs["mean"][1].data_ptr(), bufs["ws"][1].data_ptr(),
                                          ws_bytes, L.stream()))
        torch.cuda.synchronize()
        assert min(bufs["num"][1].tolist()) > 1000
        for k, (nbytes, _) in specs.items():
            raw = bufs[k][0]
            assert bool((raw[:guard] == 0xAB).all()) and bool((raw[guard + nbytes:] == 0xAB).all()), "write outside `%s`" % k

    pts = [s[0]["points"] for s in synthetic_batch(2000, 2, device=dev)]
    run(pts)
    run([torch.cat([p, p[:, :1]], 1) for p in pts] * 2)
    run([pts[0][:16000]])


@pytest.mark.parametrize("impl", ["bins", "hash"])
def test_hard_voxelize_replays_from_a_hip_graph(impl, monkeypatch):
    """The call's launches captured into a HIP graph and replayed give the eager result every time.  (With hipMemsetAsync
    for the counters the SECOND replay ran on dirty counters on this stack -- a memory fault in the binned path, a hang in the
    hash path; both paths clear with a kernel since round 4.)"""
    from efg_amd.engine import synthetic_batch
    from efg_amd.hipgraph import capture
    from efg_amd.operators import voxelize as V

    dev = torch.device("cuda:0")
    monkeypatch.setenv("EFG_VOX_IMPL", impl)
    pts = [s[0]["points"] for s in synthetic_batch(2000, 2, device=dev)]
    points = torch.cat(pts, 0).contiguous()
    offsets = [0, pts[0].shape[0], pts[0].shape[0] + pts[1].shape[0]]
    cap, f = 240000, points.shape[1]

    def buffers():
        return (torch.full((cap, 5, f), -7.0, device=dev), torch.full((cap, 4), -7, dtype=torch.int32, device=dev),
                torch.full((cap,), -7, dtype=torch.int32, device=dev), torch.zeros(2, dtype=torch.int32, device=dev),
                torch.full((cap, f), -7.0, device=dev))

    def launch(bufs):
        bufs[3].zero_()
        V._hard_voxelize_launch(points, offsets, [0.1, 0.1, 0.15], [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0], 5, 120000, *bufs)

    want = buffers()
    launch(want)
    torch.cuda.synchronize()
    m = int(want[3].sum())
    got = buffers()
    graph, _ = capture(lambda: launch(got), dev)
    for _ in range(4):
        for t in (got[0], got[1], got[2], got[4]):
            t.fill_(-7)
        graph.replay()
        torch.cuda.synchronize()
        assert got[3].tolist() == want[3].tolist()
        for a, b in zip(got, want):
            assert torch.equal(a[:m] if a.shape[0] == cap else a, b[:m] if b.shape[0] == cap else b)
