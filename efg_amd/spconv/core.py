"""spconv-style sparse tensors and convolutions on MI355X.

The reference backbone (efg/modeling/backbones/sparse_net.py:6-11) imports third-party
`spconv.pytorch`; this module provides the subset it calls -- `SparseConvTensor`, `SparseModule`,
`SparseSequential`, `SubMConv3d`, `SparseConv3d`, `.dense()`, `.replace_feature()` -- with spconv
2.x constructor signatures and parameter layout (`weight` is [Cout, kd, kh, kw, Cin]), on top of
the HIP kernels in csrc/spconv_index.hip and csrc/spconv_conv.hip.

Semantics: SubMConv3d keeps the input's sites AND row order; SparseConv3d activates
every output site touched by an input site and returns rows in canonical order (ascending
(b,z,y,x) linear index), so two convolutions with equal geometry over the same input are
row-aligned (required by `out.features + shortcut.features`, sparse_net.py:162).
"""
import math
import os

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from .. import _prof


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return tuple(int(x) for x in v)
    return (int(v),) * 3


class SiteIndex:
    """Rank index (bitmap + popcount prefix) of a set of active sites; see csrc/rank_index.h."""

    def __init__(self, index, perm, batch_size, spatial_shape):
        self.index = index            # uint8 storage of uint2[words]
        self.perm = perm              # int32[M] canonical rank -> row, or None if rows are canonical
        self.batch_size = batch_size
        self.spatial_shape = tuple(spatial_shape)

    @staticmethod
    def _alloc(batch_size, spatial_shape, device):
        lib = L.lib()
        shp = L.host_i32(spatial_shape, 3)
        nbytes = lib.efg_spconv_index_bytes(batch_size, shp)
        if nbytes == 0:
            raise RuntimeError("efg_hip: " + lib.efg_last_error().decode())
        ws_bytes = lib.efg_spconv_index_workspace_bytes(batch_size, shp)
        index = torch.empty(nbytes, dtype=torch.uint8, device=device)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        return lib, shp, index, ws, ws_bytes

    @classmethod
    def from_indices(cls, indices, batch_size, spatial_shape, canonical=False):
        return _site_index_from_indices(indices, batch_size, spatial_shape, canonical)


# ---- the only functions of this module that touch libefg_hip.so --------------------------------
def _site_index_from_indices(indices, batch_size, spatial_shape, canonical=False):
    lib, shp, index, ws, ws_bytes = SiteIndex._alloc(batch_size, spatial_shape, indices.device)
    m = indices.shape[0]
    perm = None if canonical else torch.empty(max(m, 1), dtype=torch.int32, device=indices.device)
    L.check(lib.efg_spconv_index_from_indices(L.ptr(indices), m, batch_size, shp, L.ptr(index), L.ptr(perm),
                                              L.ptr(ws), ws_bytes, L.stream()))
    return SiteIndex(index, perm, batch_size, spatial_shape)


def _build_rnbr(nbr, m_out, kvol, m_in):
    r = torch.empty((kvol, max(m_in, 1)), dtype=torch.int32, device=nbr.device)
    L.check(L.lib().efg_spconv_build_rnbr(L.ptr(nbr), m_out, kvol, m_in, L.ptr(r), L.stream()))
    return (r[:, :m_in] if m_in > 0 else r[:, :0]).contiguous()


def _tiled():
    """Mask-sorted row tiles (csrc/spconv_tiles.hip) for forward / dgrad of every window of at most 31 offsets; the
    generation-one kernels (csrc/spconv_conv.hip) remain for larger windows only (the EFG_CONV_TILED=0 switch of rounds 2-5 was
    retired in round 6: profiles/r02_gen1_vs_tiled_per_layer.txt)."""
    return True


_SHAPE_CACHE = {}


def _tile_kernel_name(cin, cout, kvol, m_in, m_out):
    """Label of a tiled launch = the NT (16-column tiles per wave) of the instantiation the library will run, asked
    from the library itself (efg_spconv_tile_shape shares its rule with the launcher)."""
    key = (cin, cout, kvol, m_in == m_out)
    if key not in _SHAPE_CACHE:
        import ctypes

        nt, r, ks = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if L.lib().efg_spconv_small_ok(cin, cout, kvol, ctypes.byref(r), ctypes.byref(nt)):
            _SHAPE_CACHE[key] = "conv_small_kernel<%d,%d>" % (r.value, nt.value)   # the narrow layers (stem): <C16, NT>
            return _SHAPE_CACHE[key]
        L.check(L.lib().efg_spconv_tile_shape(cin, cout, kvol, m_in, m_out, ctypes.byref(nt), ctypes.byref(r),
                                              ctypes.byref(ks)))
        _SHAPE_CACHE[key] = "conv_tile_kernel<%d>" % nt.value
    return _SHAPE_CACHE[key]


def _build_plan(table, m, kvol):
    """Tile plan (uint8 buffer) of a neighbour table int32 [kvol, m]; None for an empty table."""
    if m == 0:
        return None
    lib = L.lib()
    nbytes = lib.efg_spconv_tile_plan_bytes(m, kvol)
    if nbytes == 0:
        raise RuntimeError("efg_hip: tile plan needs kvol <= 31, got %d" % kvol)
    plan = torch.empty(nbytes, dtype=torch.uint8, device=table.device)
    L.check(lib.efg_spconv_tile_plan(L.ptr(table), m, kvol, L.ptr(plan), nbytes, L.stream()))
    return plan


def _build_wgrad_sched(plan, m, cin, cout, kvol):
    """Launch schedule (uint8 buffer) of the plan-walking weight gradient of a (cin -> cout) layer over `plan`."""
    lib = L.lib()
    nbytes = lib.efg_spconv_wgrad_sched_bytes(m, cin, cout, kvol)
    sched = torch.empty(nbytes, dtype=torch.uint8, device=plan.device)
    L.check(lib.efg_spconv_wgrad_sched(L.ptr(plan), m, cin, cout, kvol, L.ptr(sched), nbytes, L.stream()))
    return sched


_WGT_OK = {}


def _wgrad_tiled(cin, cout, kvol, m_out, m_in=0):
    """Does the plan-walking weight gradient take this layer?  (EFG_WGRAD_TILED=0: the table kernel everywhere.)  Cached:
    this sits on the host path of every sparse convolution, and the training step is close to host-bound.
    Size limits of csrc/spconv_wgt.hip (32-bit byte offsets into the feature tensors, 32-bit tile products in the
    schedule kernel): larger layers take the table kernel instead of failing at backward time."""
    if m_out <= 0:
        return False
    if max(m_in, m_out) * max(cin, cout) * 4 >= (1 << 32) or m_out >= (1 << 25):
        return False
    key = (cin, cout, kvol)
    ok = _WGT_OK.get(key)
    if ok is None:
        ok = _WGT_OK[key] = (kvol <= 31 and os.environ.get("EFG_WGRAD_TILED", "1") != "0"
                             and bool(L.lib().efg_spconv_wgrad_tiled_ok(cin, cout, kvol)))
    return ok


_ARM_OK = {}


def _arm_bf16x3(cin, cout, kvol, m_in, m_out):
    """4 (the flag of the pack / tiled-conv entry points) when the split-precision A/B arm is on (operators/linear.py's switch,
    EFG_GEMM_ARM=bf16x3) and the library's tile kernel covers this convolution, else 0."""
    from ..operators import linear as _lin

    if not _lin._ARM_BF16X3 or os.environ.get("EFG_CONV_ARM", "1") == "0":   # (EFG_CONV_ARM=0: the arm without its convolutions)
        return 0
    key = (cin, cout, kvol, m_in == m_out)
    if key not in _ARM_OK:
        _ARM_OK[key] = 4 if L.lib().efg_spconv_tile_bf16x3_ok(cin, cout, kvol, m_in, m_out) else 0
    return _ARM_OK[key]


# ---- packed weights, cached across calls ------------------------------------------------------------------------------
# The MFMA operand order of a layer's weights is a pure function of the weights: re-packing in every forward and every
# backward was ~40 launches per ConQueR step.  The packed copy lives on the PARAMETER object (so it dies with it) and is
# valid while (parameter._version, weight epoch) is unchanged.  The epoch is bumped by a global post-step hook on every
# torch optimizer (the fused AdamW updates parameters without moving `_version`) and by `weights_updated()`, which code
# that writes weights behind autograd's back (`p.data.copy_(...)`) must call..
_WEIGHT_EPOCH = [0]
_PACK_CACHE_ON = True


def weights_updated(*_args, **_kwargs):
    """Invalidate every cached packed weight (called after each optimizer step; call it yourself after writing
    convolution weights through `.data`)."""
    _WEIGHT_EPOCH[0] += 1


try:   # every optimizer of the process, the reference's own (efg/solver) included
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_hook

    _reg_hook(weights_updated)
except ImportError:   # pragma: no cover  (older torch: no global hooks -> no caching)
    _PACK_CACHE_ON = False


class _PackRegistry:
    """Per device: every (parameter, layout) pair that has a cached packed copy, and the device table of 32-byte
    {weight ptr, packed ptr, cout, kvol, cin, flags} records efg_spconv_pack_weights_multi reads.  When the first
    convolution after an optimizer step finds its copy stale, ALL registered copies are refreshed by ONE launch (two
    launches per convolution and step before: ~40 of the step's launches)."""

    def __init__(self):
        self.entries = {}      # (id(owner), flags) -> [weakref(owner), packed tensor, (cout, kvol, cin), data_ptr]
        self.table = None      # int64 [n, 4] on the device, rebuilt when the membership changes
        self.keys = None

    def _live(self):
        # gone parameters, and parameters whose storage moved (module.to(...), load_state_dict(assign=True)): their
        # recorded pointer must never be read again; the next use of such a layer packs and registers it afresh
        dead = [k for k, e in self.entries.items() if e[0]() is None or e[0]().data_ptr() != e[3]]
        for k in dead:
            del self.entries[k]
            self.table = None
        return self.entries

    def refresh_all(self, device):
        """One launch for every registered copy; tags of all owners move to the current (version, epoch)."""
        ent = self._live()
        if self.table is None:
            self.keys = list(ent)
            rows = []
            for k in self.keys:
                _, packed, (cout, kvol, cin), wptr = ent[k]
                rows.append([wptr, packed.data_ptr(), cout | (kvol << 32), cin | (k[1] << 32)])
            self.table = torch.tensor(rows, dtype=torch.int64).to(device)
        L.check(L.lib().efg_spconv_pack_weights_multi(L.ptr(self.table), len(self.keys), L.stream()))
        for k in self.keys:
            owner = ent[k][0]()
            owner.__dict__["_efg_packed"][k[1]][0] = (owner._version, _WEIGHT_EPOCH[0], ent[k][3])


_PACK_REGISTRY = {}


def _packed_weight(w, flags, owner=None):
    """MFMA-order copy of w [cout, kvol, cin] for `flags` (bit 0: data-gradient layout, bit 1: natural channel order,
    bit 2: split-precision arm), cached on `owner` (the layer's Parameter) when given."""
    lib = L.lib()
    cout, kvol, cin = w.shape
    cached = owner is not None and _PACK_CACHE_ON and w.is_cuda
    if cached:
        tag = (owner._version, _WEIGHT_EPOCH[0], w.data_ptr())
        cache = owner.__dict__.setdefault("_efg_packed", {})
        ent = cache.get(flags)
        if ent is not None and ent[0] == tag:
            return ent[1]
        reg = _PACK_REGISTRY.setdefault(w.device.index, _PackRegistry())
        key = (id(owner), flags)
        known = reg.entries.get(key)
        if ent is not None and known is not None and known[3] == w.data_ptr() and known[0]() is owner:
            # stale (an optimizer step happened): refresh EVERY registered copy of the device now, in one launch
            reg.refresh_all(w.device)
            return ent[1]
    packed = torch.empty(lib.efg_spconv_packed_weight_bytes(cout, kvol, cin, flags & 1), dtype=torch.uint8, device=w.device)
    L.check(lib.efg_spconv_pack_weight_f32(L.ptr(w), cout, kvol, cin, flags, L.ptr(packed), L.stream()))
    if cached:
        import weakref

        cache[flags] = [tag, packed]
        reg.entries[key] = [weakref.ref(owner), packed, (cout, kvol, cin), w.data_ptr()]
        reg.table = None
    return packed


def _conv_forward(features, w, bias, rb, owner=None):
    """features [m_in,cin], w [cout,kvol,cin] -> [m_out,cout]"""
    lib = L.lib()
    cout, kvol, cin = w.shape
    tiled = _tiled() and kvol <= 31 and rb.m_out > 0
    nat = _arm_bf16x3(cin, cout, kvol, rb.m_in, rb.m_out) if tiled else 0   # (bit 2 of the flag word: the split-precision arm)
    packed = _packed_weight(w, 0 | nat, owner)
    out = torch.empty((rb.m_out, cout), dtype=torch.float32, device=features.device)
    if tiled:
        plan = rb.plan_fwd()
        with _prof.timed(_tile_kernel_name(cin, cout, kvol, rb.m_in, rb.m_out), _Cost(rb, cin, cout, "fwd")):
            L.check(lib.efg_spconv_forward_tiled_f32(L.ptr(features), rb.m_in, cin, L.ptr(packed), L.ptr(bias), cout,
                                                     kvol, L.ptr(plan), rb.m_out, 0 | nat, L.ptr(out), L.stream()))
        return out
    with _prof.timed(_fwd_kernel_name(cout, rb.m_out, kvol), _Cost(rb, cin, cout, "fwd")):
        L.check(lib.efg_spconv_forward_f32(L.ptr(features), rb.m_in, cin, L.ptr(packed), L.ptr(bias), cout, kvol,
                                           L.ptr(rb.nbr), rb.m_out, L.ptr(out), L.stream()))
    return out


def _conv_dgrad(grad_out, w, rb, owner=None):
    lib = L.lib()
    cout, kvol, cin = w.shape
    tiled = _tiled() and kvol <= 31 and rb.m_in > 0
    nat = _arm_bf16x3(cout, cin, kvol, rb.m_out, rb.m_in) if tiled else 0
    packed = _packed_weight(w, 1 | nat, owner)
    grad_in = torch.empty((rb.m_in, cin), dtype=torch.float32, device=grad_out.device)
    if tiled:
        plan, flip = rb.plan_dgrad()
        with _prof.timed(_tile_kernel_name(cout, cin, kvol, rb.m_out, rb.m_in), _Cost(rb, cin, cout, "dgrad")):
            L.check(lib.efg_spconv_forward_tiled_f32(L.ptr(grad_out), rb.m_out, cout, L.ptr(packed), None, cin, kvol,
                                                     L.ptr(plan), rb.m_in, flip | nat, L.ptr(grad_in), L.stream()))
        return grad_in
    rnbr = rb.rnbr
    order = rb.dgrad_order()
    with _prof.timed(_fwd_kernel_name(cin, rb.m_in, kvol), _Cost(rb, cin, cout, "dgrad")):
        L.check(lib.efg_spconv_dgrad_f32(L.ptr(grad_out), rb.m_out, cout, L.ptr(packed), cin, kvol, L.ptr(rnbr),
                                         rb.m_in, L.ptr(order), L.ptr(grad_in), L.stream()))
    return grad_in


def _conv_wgrad(features, grad_out, rb):
    lib = L.lib()
    cin, cout, kvol = features.shape[1], grad_out.shape[1], rb.kvol
    if _wgrad_tiled(cin, cout, kvol, rb.m_out, rb.m_in):
        # over the layer's forward tile plan: MFMA operands straight from the feature rows (csrc/spconv_wgt.hip)
        plan, sched, ws_bytes = rb.plan_fwd(), *rb.wgrad_sched(cin, cout)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=grad_out.device)
        grad_w = torch.empty((cout, kvol, cin), dtype=torch.float32, device=grad_out.device)
        with _prof.timed("conv_wgrad_tile_kernel+wgt_reduce_kernel", _Cost(rb, cin, cout, "wgrad")):
            L.check(lib.efg_spconv_wgrad_tiled_f32(L.ptr(features), rb.m_in, cin, L.ptr(grad_out), rb.m_out, cout, kvol,
                                                   L.ptr(plan), L.ptr(sched), L.ptr(grad_w), L.ptr(ws), ws_bytes, L.stream()))
        return grad_w
    ws_bytes = lib.efg_spconv_wgrad_workspace_bytes(rb.m_out, cin, cout, kvol)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=grad_out.device)
    grad_w = torch.empty((cout, kvol, cin), dtype=torch.float32, device=grad_out.device)
    with _prof.timed("conv_wgrad_kernel+wgrad_reduce_kernel", _Cost(rb, cin, cout, "wgrad")):
        L.check(lib.efg_spconv_wgrad_f32(L.ptr(features), rb.m_in, cin, L.ptr(grad_out), rb.m_out, cout, kvol,
                                         L.ptr(rb.nbr), L.ptr(grad_w), L.ptr(ws), ws_bytes, L.stream()))
    return grad_w


def _pair_ok(rb, w_a, w_b, cin, cout):
    """Can the two convolutions (weights [cout, kvol, cin] each) over `rb` run as the pair launches of csrc/spconv_tiles.hip /
    spconv_wgt.hip?  (the tiled fp32 path with the default weight order; EFG_CONV_PAIR=0: two launches each, A/B)"""
    return (os.environ.get("EFG_CONV_PAIR", "1") != "0" and _tiled() and rb.kvol <= 31 and rb.m_out > 0 and rb.m_in > 0
            and w_a.shape == w_b.shape
            and not _arm_bf16x3(cin, cout, rb.kvol, rb.m_in, rb.m_out) and not _arm_bf16x3(cout, cin, rb.kvol, rb.m_out, rb.m_in))


def _conv_forward_pair(features, w_a, w_b, rb, owner_a, owner_b):
    """(features (x) w_a, features (x) w_b) over one rulebook in ONE launch (efg_spconv_tiled_pair_f32, "N pair")."""
    cout, kvol, cin = w_a.shape
    pa, pb = _packed_weight(w_a, 0, owner_a), _packed_weight(w_b, 0, owner_b)
    out_a = torch.empty((rb.m_out, cout), dtype=torch.float32, device=features.device)
    out_b = torch.empty_like(out_a)
    plan = rb.plan_fwd()
    with _prof.timed(_tile_kernel_name(cin, cout, kvol, rb.m_in, rb.m_out), _Cost(rb, cin, 2 * cout, "fwd pair")):
        L.check(L.lib().efg_spconv_tiled_pair_f32(L.ptr(features), None, rb.m_in, cin, L.ptr(pa), L.ptr(pb), cout, kvol,
                                                  L.ptr(plan), rb.m_out, 0, L.ptr(out_a), L.ptr(out_b), L.stream()))
    return out_a, out_b


def _conv_dgrad_pair(go_a, go_b, w_a, w_b, rb, owner_a, owner_b):
    """dgrad(go_a, w_a) + dgrad(go_b, w_b) accumulated in ONE pass (efg_spconv_tiled_pair_f32, "K pair")."""
    cout, kvol, cin = w_a.shape
    pa, pb = _packed_weight(w_a, 1, owner_a), _packed_weight(w_b, 1, owner_b)
    grad_in = torch.empty((rb.m_in, cin), dtype=torch.float32, device=go_a.device)
    plan, flip = rb.plan_dgrad()
    with _prof.timed(_tile_kernel_name(cout, cin, kvol, rb.m_out, rb.m_in), _Cost(rb, cin, 2 * cout, "dgrad pair")):
        L.check(L.lib().efg_spconv_tiled_pair_f32(L.ptr(go_a), L.ptr(go_b), rb.m_out, cout, L.ptr(pa), L.ptr(pb), cin, kvol,
                                                  L.ptr(plan), rb.m_in, flip, L.ptr(grad_in), None, L.stream()))
    return grad_in


def _conv_wgrad_pair(features, go_a, go_b, rb):
    """The two weight gradients in one launch + one fold (efg_spconv_wgrad_tiled_pair_f32); bit-identical to two calls."""
    lib = L.lib()
    cin, cout, kvol = features.shape[1], go_a.shape[1], rb.kvol
    plan, sched, ws_one = rb.plan_fwd(), *rb.wgrad_sched(cin, cout)
    ws_bytes = 2 * ws_one + 512
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=go_a.device)
    gw_a = torch.empty((cout, kvol, cin), dtype=torch.float32, device=go_a.device)
    gw_b = torch.empty_like(gw_a)
    with _prof.timed("conv_wgrad_tile_kernel+wgt_reduce_kernel", _Cost(rb, cin, 2 * cout, "wgrad pair")):
        L.check(lib.efg_spconv_wgrad_tiled_pair_f32(L.ptr(features), rb.m_in, cin, L.ptr(go_a), L.ptr(go_b), rb.m_out, cout, kvol,
                                                    L.ptr(plan), L.ptr(sched), L.ptr(gw_a), L.ptr(gw_b), L.ptr(ws), ws_bytes,
                                                    L.stream()))
    return gw_a, gw_b


def _fwd_kernel_name(n_out_channels, n_rows, kvol=27):
    """Symbol of the forward/dgrad instantiation csrc/spconv_conv.hip:run_conv picks (same rule)."""
    ntiles = (n_out_channels + 15) // 16
    row_waves = (n_rows + 15) // 16
    fill = 1400 if kvol >= 8 else 2048
    nt = 16
    while nt > 1 and (nt // 2 >= ntiles or row_waves * ((ntiles + nt - 1) // nt) < fill):
        nt >>= 1
    if nt > ntiles:
        nt = 16 if ntiles >= 16 else 8 if ntiles >= 8 else 4 if ntiles >= 4 else 2 if ntiles >= 2 else 1
    while nt < ntiles and nt < 16 and ntiles % nt != 0:
        nt <<= 1
    return "conv_fwd_kernel<%d>" % nt


class _Cost:
    """Deferred algorithmic cost of one launch (evaluated after the run) + a label for detailed listings."""

    def __init__(self, rb, cin, cout, kind):
        self.rb, self.cin, self.cout = rb, cin, cout
        self.meta = "%s m_in=%d m_out=%d cin=%d cout=%d kvol=%d %s" % (kind, rb.m_in, rb.m_out, cin, cout, rb.kvol,
                                                                      "subm" if rb.subm else "strided")

    def __call__(self):
        return _conv_cost(self.rb, self.cin, self.cout)


def _conv_cost(rb, cin, cout):
    """Algorithmic (bytes, flops) of one sparse-conv launch, SURVEY.md §8(d):
    4*Cin*M_in + 4*Cout*M_out + 4*K*Cin*Cout + 8*pairs bytes, 2*pairs*Cin*Cout flops."""
    pairs = rb.num_pairs()
    return (4 * cin * rb.m_in + 4 * cout * rb.m_out + 4 * rb.kvol * cin * cout + 8 * pairs,
            2 * pairs * cin * cout)


def _to_dense(features, x):
    si = x.site_index()
    c = features.shape[1]
    d, h, w = x.spatial_shape
    dense = torch.empty((x.batch_size, c, d, h, w), dtype=torch.float32, device=features.device)
    L.check(L.lib().efg_sparse_to_dense_f32(L.ptr(features), c, L.ptr(si.index), L.ptr(si.perm), x.batch_size,
                                            L.host_i32(x.spatial_shape, 3), L.ptr(dense), L.stream()))
    return dense


def _to_bev(features, x):
    """[B, H, W, C*D] channels-last BEV map (channel = c*D + d)."""
    si = x.site_index()
    c = features.shape[1]
    d, h, w = x.spatial_shape
    out = torch.empty((x.batch_size, h, w, c * d), dtype=torch.float32, device=features.device)
    L.check(L.lib().efg_sparse_to_bev_f32(L.ptr(features), c, L.ptr(si.index), L.ptr(si.perm), x.batch_size,
                                          L.host_i32(x.spatial_shape, 3), L.ptr(out), L.stream()))
    return out


def _from_bev(grad_out, x, c):
    m = x.indices.shape[0]
    g = torch.empty((m, c), dtype=torch.float32, device=grad_out.device)
    L.check(L.lib().efg_bev_to_sparse_f32(L.ptr(grad_out), c, L.ptr(x.indices), m, x.batch_size,
                                          L.host_i32(x.spatial_shape, 3), L.ptr(g), L.stream()))
    return g


def _from_dense(grad_dense, x):
    m, c = x.indices.shape[0], grad_dense.shape[1]
    g = torch.empty((m, c), dtype=torch.float32, device=grad_dense.device)
    L.check(L.lib().efg_dense_to_sparse_f32(L.ptr(grad_dense), c, L.ptr(x.indices), m, x.batch_size,
                                            L.host_i32(x.spatial_shape, 3), L.ptr(g), L.stream()))
    return g


# ---- geometry stream ------------------------------------------------------------------------------------
# Sparse-conv geometry (site discovery, neighbour tables) depends only on voxel coordinates, never on
# features or weights, but sizes every downstream tensor: each strided level reads its site count back to
# the host.  On the main stream that read waits for everything queued before it -- in a training loop the
# whole backward of the previous step -- so the host cannot run ahead.  Inside a `geometry_stream(s)` scope
# the geometry launches (and their count readbacks) go to the side stream `s` instead: the readback only
# waits for the few geometry kernels, the main stream gets a device-side dependency, and tensors that cross
# streams are handed to the caching allocator with record_stream.  Outside a scope nothing changes.
_GEO = None


class geometry_stream:
    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        global _GEO
        self.prev, _GEO = _GEO, self.stream
        return self

    def __exit__(self, *exc):
        global _GEO
        _GEO = self.prev
        return False


class _on_geometry_stream:
    """`with _on_geometry_stream() as main:` -- main is None when no scope is active.  wait=False: the main stream does
    NOT wait for what was queued (the caller records an event and waits where the result is used)."""

    def __init__(self, wait=True):
        self.wait = wait

    def __enter__(self):
        # (set_stream directly: the torch.cuda.stream() context asks for the current stream once more on entry -- ~60 such
        # scopes per step, each query building a Stream object through _get_device_index)
        self.main = None
        if _GEO is not None:
            self.main = torch.cuda.current_stream()
            torch.cuda.set_stream(_GEO)
        return self.main

    def __exit__(self, *exc):
        if self.main is not None:
            torch.cuda.set_stream(self.main)
            if self.wait:
                self.main.wait_stream(_GEO)  # device-side dependency, the host does not block
        return False


def _hand_over(main, *tensors):
    """Tensors allocated on the geometry stream that main-stream kernels will read."""
    if main is not None:
        for t in tensors:
            if t is not None:
                t.record_stream(main)


class Rulebook:
    """Output-stationary neighbour table of one convolution geometry."""

    def __init__(self, nbr, m_in, m_out, kvol, subm, in_indices=None):
        self.nbr = nbr          # int32 [kvol, m_out]
        self.m_in, self.m_out, self.kvol, self.subm = m_in, m_out, kvol, subm
        self.in_indices = in_indices  # int32 [m_in, 4] of a strided geometry (for the dgrad row order)
        self._rnbr = None
        self._pairs = None
        self._order = None
        self._plan_fwd = None
        self._plan_stream = None
        self._plan_dgrad = None
        self._wgrad_sched = {}

    def plan_fwd(self):
        """Tile plan of `nbr` (csrc/spconv_tiles.hip), built on first use on the geometry stream when one is active."""
        if self._plan_fwd is None:
            with _on_geometry_stream() as main:
                self._plan_fwd = _build_plan(self.nbr, self.m_out, self.kvol)
                _hand_over(main, self._plan_fwd)
                self._plan_stream = ((_GEO if main is not None else torch.cuda.current_stream())
                                     if self._plan_fwd is not None else None)
        return self._plan_fwd

    def prepare_wgrad(self, cin, cout):
        """Queue the launch schedule of the plan-walking weight gradient of a (cin -> cout) layer over plan_fwd()
        (csrc/spconv_wgt.hip) on the geometry stream -- the forward pass of a training step calls this and does NOT wait
        for it: only the backward pass reads the schedule (wgrad_sched waits for the event recorded here)."""
        key = (cin, cout)
        if key in self._wgrad_sched or not _wgrad_tiled(cin, cout, self.kvol, self.m_out, self.m_in):
            return
        plan = self.plan_fwd()
        with _on_geometry_stream(wait=False) as main:
            cur = _GEO if main is not None else torch.cuda.current_stream()
            if self._plan_stream is not None and self._plan_stream != cur:
                cur.wait_stream(self._plan_stream)   # (a plan built outside the scope this runs in)
            sched = _build_wgrad_sched(plan, self.m_out, cin, cout, self.kvol)
            _hand_over(main, sched)
            event = None
            if main is not None:
                event = torch.cuda.Event()
                event.record(_GEO)
        ws_bytes = L.lib().efg_spconv_wgrad_tiled_workspace_bytes(self.m_out, cin, cout, self.kvol)
        self._wgrad_sched[key] = [sched, ws_bytes, event]

    def wgrad_sched(self, cin, cout):
        """(schedule, workspace bytes) of the layer shape: the schedule queued by prepare_wgrad (built now if the forward
        pass did not), ready for the current stream (the event is waited for once: every later use is on that stream)."""
        key = (cin, cout)
        if key not in self._wgrad_sched:
            self.prepare_wgrad(cin, cout)
        ent = self._wgrad_sched[key]
        if ent[2] is not None:
            torch.cuda.current_stream().wait_event(ent[2])
            ent[2] = None
        return ent[0], ent[1]

    def plan_dgrad(self):
        """(plan, flip): submanifold -> the forward plan walked with reversed offsets (the transposed table of a
        symmetric window is the table with its offsets reversed); strided -> the plan of `rnbr`."""
        if self.subm:
            return self.plan_fwd(), 1
        if self._plan_dgrad is None:
            rnbr = self.rnbr
            with _on_geometry_stream() as main:
                self._plan_dgrad = _build_plan(rnbr, self.m_in, self.kvol)
                _hand_over(main, self._plan_dgrad)
        return self._plan_dgrad, 0

    def dgrad_order(self):
        """int32 [m_in] row order for the dgrad of a strided geometry: input rows grouped by coordinate parity, so the
        16 rows of a tile share their reachable offsets (csrc/spconv_conv.hip: efg_spconv_parity_order); None for
        submanifold geometries (every row reaches every offset)."""
        if self.subm or self.in_indices is None or self.m_in == 0:
            return None
        if self._order is None:
            idx = self.in_indices
            order = torch.empty(self.m_in, dtype=torch.int32, device=idx.device)
            ws = torch.empty(64, dtype=torch.uint8, device=idx.device)
            L.check(L.lib().efg_spconv_parity_order(L.ptr(idx), self.m_in, L.ptr(order), L.ptr(ws), 64, L.stream()))
            self._order = order
        return self._order

    @property
    def rnbr(self):
        """int32 [kvol, m_in]: transpose of nbr, for dgrad."""
        if self._rnbr is None:
            if self.subm:
                self._rnbr = self.nbr.flip(0).contiguous()  # symmetric window: rnbr[k] = nbr[kvol-1-k]
            else:
                self._rnbr = _build_rnbr(self.nbr, self.m_out, self.kvol, self.m_in)
        return self._rnbr

    def num_pairs(self):
        """Number of (input, output) pairs (one host read, cached; used only for roofline accounting)."""
        if self._pairs is None:
            self._pairs = int((self.nbr >= 0).sum().item())
        return self._pairs


class SparseConvTensor:
    """spconv.pytorch.SparseConvTensor(features[M,C], indices int32[M,4] (b,z,y,x), spatial_shape[3],
    batch_size) as constructed at sparse_net.py:289,531."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None, indice_dict=None,
                 benchmark=False, _site_index=None):
        L.require_gpu(features, indices)
        if indices.dtype != torch.int32:
            raise RuntimeError("SparseConvTensor: indices must be int32")
        if indices.dim() != 2 or indices.shape[1] != 4 or indices.shape[0] != features.shape[0]:
            raise RuntimeError("SparseConvTensor: indices must be [M, 4] (b,z,y,x) matching features")
        self.features = features
        self.indices = indices.contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}
        self._site_index = _site_index
        self._conv_cache = {}

    def replace_feature(self, feature):
        new = SparseConvTensor.__new__(SparseConvTensor)
        new.__dict__.update(self.__dict__)  # shares indices, site index, rulebook caches
        new.features = feature
        return new

    @property
    def spatial_size(self):
        return int(math.prod(self.spatial_shape))

    def site_index(self):
        if self._site_index is None:
            self._site_index = SiteIndex.from_indices(self.indices, self.batch_size, self.spatial_shape)
        return self._site_index

    def dense(self, channels_first=True):
        """[B, C, D, H, W] (sparse_net.py:304); zeros where inactive."""
        out = _ToDense.apply(self.features, self)
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()

    def dense_bev(self):
        """The BEV map the backbones build with `dense()` + `view(N, C*D, H, W)` (sparse_net.py:304-306,
        541-545): same logical [B, C*D, H, W] tensor and values, produced in ONE pass and stored
        channels-last (the layout the transformer's [B, H*W, C] tokens need)."""
        return _ToBev.apply(self.features, self).permute(0, 3, 1, 2)


class _ToBev(Function):
    @staticmethod
    def forward(ctx, features, x):
        ctx.x = x
        ctx.c = features.shape[1]
        return _to_bev(features.contiguous(), x)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        return _from_bev(grad_out.contiguous(), ctx.x, ctx.c), None


class _ToDense(Function):
    @staticmethod
    def forward(ctx, features, x):
        ctx.x = x
        return _to_dense(features.contiguous(), x)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_dense):
        return _from_dense(grad_dense.contiguous(), ctx.x), None


class _SparseConvFunction(Function):
    """features[M_in,Cin] x weight[Cout,kvol,Cin] (+bias) -> [M_out,Cout] through a Rulebook."""

    @staticmethod
    def forward(ctx, features, weight, bias, rb, grad_on=True):
        features = features.contiguous()
        cout, cin = weight.shape[0], weight.shape[-1]
        w = weight.reshape(cout, rb.kvol, cin).contiguous()
        # (the oracle-backed CPU path of the tests patches _conv_* with three-argument stand-ins: no `owner` there)
        out = _conv_forward(features, w, bias, rb, weight) if features.is_cuda else _conv_forward(features, w, bias, rb)
        ctx.owner = weight if features.is_cuda else None
        # the weight gradient's launch schedule: only when a backward can follow (needs_input_grad is True under no_grad
        # too, and grad mode is always off inside forward(): the caller passes what it saw)
        if ctx.needs_input_grad[1] and features.is_cuda and grad_on:
            rb.prepare_wgrad(cin, cout)   # the weight gradient's launch schedule, on the geometry stream, ahead of the backward
        ctx.save_for_backward(features, w)
        ctx.rb = rb
        ctx.has_bias = bias is not None
        ctx.wshape = weight.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        features, w = ctx.saved_tensors
        rb = ctx.rb
        grad_out = grad_out.contiguous()
        grad_in = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            grad_in = _conv_dgrad(grad_out, w, rb, ctx.owner) if ctx.owner is not None else _conv_dgrad(grad_out, w, rb)
        if ctx.needs_input_grad[1]:
            grad_w = _conv_wgrad(features, grad_out, rb).view(ctx.wshape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_b = grad_out.sum(0)
        return grad_in, grad_w, grad_b, None, None


class _ConvBnActFunction(Function):
    """relu?(bn(conv(features)) + residual?) as ONE autograd node (efg_amd/_fuse.py): the convolution and the fused
    BatchNorm of operators/batchnorm.py run back to back in every layer of the sparse backbones (sparse_net.py:85-95,
    120-165); as two nodes they cost the host two Function applications per direction and layer (40 of the step's ~250)."""

    @staticmethod
    def forward(ctx, features, weight, rb, grad_on, residual, gamma, beta, running_mean, running_var, num_batches_tracked,
                momentum, eps, relu):
        from .._fuse import Ctx, pack
        from ..operators.batchnorm import BatchNormActFunction

        conv = Ctx()
        conv.needs_input_grad = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], False, False, False)
        out = _SparseConvFunction.forward(conv, features, weight, None, rb, grad_on)
        bn = Ctx()
        y = BatchNormActFunction.forward(bn, out, residual, gamma, beta, running_mean, running_var, num_batches_tracked,
                                         momentum, eps, relu)
        conv_t, bn_t = pack(ctx, conv, "conv"), pack(ctx, bn, "bn")
        ctx.n_conv = len(conv_t)
        ctx.save_for_backward(*conv_t, *bn_t)
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, num_batches_tracked) if t is not None])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        from .._fuse import unpack
        from ..operators.batchnorm import BatchNormActFunction

        saved = ctx.saved_tensors
        bn = unpack(ctx, saved[ctx.n_conv:], "bn")
        dx, dres, dgamma, dbeta = BatchNormActFunction.backward(bn, dy)[:4]
        conv = unpack(ctx, saved[:ctx.n_conv], "conv", (ctx.needs_input_grad[0], ctx.needs_input_grad[1], False, False, False))
        grad_in, grad_w = _SparseConvFunction.backward(conv, dx)[:2]
        return grad_in, grad_w, None, None, dres, dgamma, dbeta, None, None, None, None, None, None


class _ConvPairBnActFunction(Function):
    """(relu?(bn_a(conv_a(x))), relu?(bn_b(conv_b(x)))) for TWO convolutions of one shape over one rulebook -- the main and the
    shortcut SparseConv3d of a residual stage's first block (sparse_net.py:125-165) -- as ONE autograd node: one launch for
    both forward products, one for the joint data gradient (the two gradients of x are never separate tensors, so nothing adds
    them), one + one fold for both weight gradients; the BatchNorms are the kernels of operators/batchnorm.py."""

    @staticmethod
    def forward(ctx, features, w_a, w_b, rb, grad_on, ga, ba, rma, rva, nbta, moma, epsa, relu_a, gb, bb, rmb, rvb, nbtb, momb,
                epsb, relu_b):
        from .._fuse import Ctx, pack
        from ..operators.batchnorm import BatchNormActFunction

        features = features.contiguous()
        cout, cin = w_a.shape[0], w_a.shape[-1]
        wa, wb = w_a.reshape(cout, rb.kvol, cin).contiguous(), w_b.reshape(cout, rb.kvol, cin).contiguous()
        out_a, out_b = _conv_forward_pair(features, wa, wb, rb, w_a, w_b)
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and grad_on:
            rb.prepare_wgrad(cin, cout)
        bna, bnb = Ctx(), Ctx()
        ya = BatchNormActFunction.forward(bna, out_a, None, ga, ba, rma, rva, nbta, moma, epsa, relu_a)
        yb = BatchNormActFunction.forward(bnb, out_b, None, gb, bb, rmb, rvb, nbtb, momb, epsb, relu_b)
        ta, tb = pack(ctx, bna, "bna"), pack(ctx, bnb, "bnb")
        ctx.n_a = len(ta)
        ctx.save_for_backward(features, wa, wb, *ta, *tb)
        ctx.rb, ctx.owners, ctx.wshape = rb, (w_a, w_b), w_a.shape
        ctx.mark_non_differentiable(*[t for t in (rma, rva, nbta, rmb, rvb, nbtb) if t is not None])
        return ya, yb

    @staticmethod
    @once_differentiable
    def backward(ctx, dya, dyb):
        from .._fuse import unpack
        from ..operators.batchnorm import BatchNormActFunction

        saved = ctx.saved_tensors
        features, wa, wb = saved[:3]
        bna = unpack(ctx, saved[3:3 + ctx.n_a], "bna")
        bnb = unpack(ctx, saved[3 + ctx.n_a:], "bnb")
        dxa, _, dga, dba = BatchNormActFunction.backward(bna, dya)[:4]
        dxb, _, dgb, dbb = BatchNormActFunction.backward(bnb, dyb)[:4]
        rb = ctx.rb
        grad_in = gwa = gwb = None
        if ctx.needs_input_grad[0]:
            grad_in = _conv_dgrad_pair(dxa, dxb, wa, wb, rb, *ctx.owners)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gwa, gwb = _conv_wgrad_pair(features, dxa, dxb, rb)
            gwa, gwb = gwa.view(ctx.wshape), gwb.view(ctx.wshape)
        return (grad_in, gwa, gwb, None, None, dga, dba, None, None, None, None, None, None, dgb, dbb, None, None, None, None,
                None, None)


def conv_pair_bn_act(conv_a, bn_a, relu_a, conv_b, bn_b, relu_b, x):
    """(relu?(bn_a(conv_a(x))), relu?(bn_b(conv_b(x)))) -> two SparseConvTensors from ONE autograd node and one launch per
    product, or None when that form does not apply (the caller then runs the two chains one after the other): two bias-free
    strided SparseConv3d of the same shape and geometry, BatchNorms the fused kernels take, a layer the pair launches cover."""
    if not (_CONV_BN_FUSED and isinstance(conv_a, SparseConvolution) and isinstance(conv_b, SparseConvolution)
            and not conv_a.subm and not conv_b.subm and conv_a.bias is None and conv_b.bias is None
            and conv_a.weight.shape == conv_b.weight.shape
            and (conv_a.kernel_size, conv_a.stride, conv_a.padding) == (conv_b.kernel_size, conv_b.stride, conv_b.padding)
            and x.features.is_cuda and x.features.dtype == torch.float32 and x.indices.shape[0] != 0
            and conv_a.weight.requires_grad == conv_b.weight.requires_grad):
        return None
    c = conv_a.out_channels
    for bn in (bn_a, bn_b):
        if not (isinstance(bn, nn.BatchNorm1d) and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None
                and c % 4 == 0 and c <= 1024 and bn.num_features == c and os.environ.get("EFG_FUSED_BN", "1") != "0"):
            return None
    rb, geom = conv_a._rulebook(x)
    cin = conv_a.in_channels
    wa3 = conv_a.weight.reshape(c, rb.kvol, cin)
    if rb.m_out < 2 or not _pair_ok(rb, wa3, wa3, cin, c) or not _wgrad_tiled(cin, c, rb.kvol, rb.m_out, rb.m_in):
        return None
    ya, yb = _ConvPairBnActFunction.apply(
        x.features, conv_a.weight, conv_b.weight, rb, torch.is_grad_enabled(),
        bn_a.weight, bn_a.bias, bn_a.running_mean, bn_a.running_var, bn_a.num_batches_tracked, bn_a.momentum, bn_a.eps, relu_a,
        bn_b.weight, bn_b.bias, bn_b.running_mean, bn_b.running_var, bn_b.num_batches_tracked, bn_b.momentum, bn_b.eps, relu_b)
    out_indices, site_index, out_shape = geom
    mk = lambda f: SparseConvTensor(f, out_indices, out_shape, x.batch_size, indice_dict=x.indice_dict, _site_index=site_index)  # noqa: E731
    return mk(ya), mk(yb)


_CONV_BN_FUSED = os.environ.get("EFG_FUSED_CONV_BN", "1") != "0"


def conv_bn_act(conv, x, bn, relu=False, residual=None):
    """relu?(bn(conv(x)) + residual?) -> SparseConvTensor, or None when the single-node form does not apply (the caller
    then runs the modules one by one): a bias-free SparseConvolution followed by a BatchNorm1d the fused kernels take, GPU."""
    c = conv.out_channels if isinstance(conv, SparseConvolution) else 0
    # (the conditions of operators/batchnorm.py:fusable, on the convolution's output shape)
    if not (_CONV_BN_FUSED and isinstance(conv, SparseConvolution) and conv.bias is None and x.features.is_cuda
            and x.features.dtype == torch.float32 and x.indices.shape[0] != 0 and isinstance(bn, nn.BatchNorm1d)
            and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None and c % 4 == 0
            and c <= 1024 and bn.num_features == c and os.environ.get("EFG_FUSED_BN", "1") != "0"):
        return None
    rb, geom = conv._rulebook(x)
    if rb.m_out < 2:
        return None
    feats = _ConvBnActFunction.apply(x.features, conv.weight, rb, torch.is_grad_enabled(), residual, bn.weight, bn.bias,
                                     bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps, relu)
    if conv.subm:
        return x.replace_feature(feats)
    out_indices, site_index, out_shape = geom
    return SparseConvTensor(feats, out_indices, out_shape, x.batch_size, indice_dict=x.indice_dict, _site_index=site_index)


def _downsample_geometry(x, ks, st, pad):
    """Output sites of a strided SparseConv3d over x: (out_indices [m_out,4] canonical order, SiteIndex, shape)."""
    lib = L.lib()
    dev = x.indices.device
    out_shape = (L.ctypes.c_int * 3)()
    in_shape = L.host_i32(x.spatial_shape, 3)
    oshape_py = [(x.spatial_shape[a] + 2 * pad[a] - ks[a]) // st[a] + 1 for a in range(3)]
    oshp = L.host_i32(oshape_py, 3)
    nbytes = lib.efg_spconv_index_bytes(x.batch_size, oshp)
    ws_bytes = lib.efg_spconv_index_workspace_bytes(x.batch_size, oshp)
    if nbytes == 0:
        raise RuntimeError("efg_hip: " + lib.efg_last_error().decode())
    oindex = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    m_out_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(lib.efg_spconv_index_downsample(L.ptr(x.indices), x.indices.shape[0], x.batch_size, in_shape,
                                            L.host_i32(ks, 3), L.host_i32(st, 3), L.host_i32(pad, 3),
                                            L.ptr(oindex), out_shape, L.ptr(m_out_dev), L.ptr(ws), ws_bytes,
                                            L.stream()))
    m_out = int(m_out_dev.item())  # sizes every downstream tensor of this level: waits for ONE marking kernel
    L.check(lib.efg_spconv_index_rank(L.ptr(oindex), x.batch_size, oshp, None, L.ptr(ws), ws_bytes, L.stream()))
    out_indices = torch.empty((max(m_out, 1), 4), dtype=torch.int32, device=dev)
    L.check(lib.efg_spconv_index_emit(L.ptr(oindex), x.batch_size, oshp, L.ptr(out_indices), L.stream()))
    return out_indices[:m_out], SiteIndex(oindex, None, x.batch_size, oshape_py), oshape_py


class _Sites:
    """What _downsample_geometry reads of a SparseConvTensor."""

    def __init__(self, indices, spatial_shape, batch_size):
        self.indices, self.spatial_shape, self.batch_size = indices, list(spatial_shape), batch_size


def _t3(v):
    return tuple(int(a) for a in v)


def prefetch_downsample_chain(x, chain, branches=()):
    """Output sites of a CHAIN of strided convolutions over x -- chain[i] = (kernel_size, stride, padding) applied to the sites
    chain[i - 1] produced -- and of `branches` = [(level, (ks, st, pad)), ...] applied to the sites of chain level `level`
    (0 = x itself), computed NOW, back to back, on the geometry stream; the convolutions pick them up from x.indice_dict
    (SparseConvolution._rulebook checks that the sites it is given ARE the ones the entry was computed from).

    Why: every level's site count is read back by the host (it sizes the level's tensors), and the read-back waits for
    everything queued on the geometry stream before the level's marking kernel.  Built lazily, level by level, that is the
    previous level's neighbour tables, tile plans and weight-gradient schedules (~300 us of kernels): ~1.2 ms of a step's host
    time spent blocked, which IS step time on a loaded host (step = host issue time + ~2 ms there).  Asked for up front the
    read-backs wait for one marking kernel each, and the plans queue up behind them.  Same kernels, same results; only the
    order of issue on the geometry stream changes (A/B of round 5: profiles/r05c_geom_prefetch.txt)."""
    if _GEO is None or x.indices.shape[0] == 0:
        return
    with _on_geometry_stream(wait=False):
        levels = [_Sites(x.indices, x.spatial_shape, x.batch_size)]
        for spec in chain:
            ks, st, pad = (_t3(v) for v in spec)
            cur = levels[-1]
            out_indices, out_si, oshape = _downsample_geometry(cur, ks, st, pad)
            x.indice_dict[("geom", _t3(cur.spatial_shape), ks, st, pad)] = (cur.indices, out_indices, out_si, oshape)
            if out_indices.shape[0] == 0:
                return
            levels.append(_Sites(out_indices, oshape, x.batch_size))
        for level, spec in branches:
            if level >= len(levels):
                continue
            ks, st, pad = (_t3(v) for v in spec)
            cur = levels[level]
            out_indices, out_si, oshape = _downsample_geometry(cur, ks, st, pad)
            x.indice_dict[("geom", _t3(cur.spatial_shape), ks, st, pad)] = (cur.indices, out_indices, out_si, oshape)


def _build_nbr(si_in, out_indices, m_out, ksize, stride, padding):
    kvol = ksize[0] * ksize[1] * ksize[2]
    nbr = torch.empty((kvol, max(m_out, 1)), dtype=torch.int32, device=out_indices.device)
    if m_out > 0:
        L.check(L.lib().efg_spconv_build_nbr(L.ptr(si_in.index), L.ptr(si_in.perm), si_in.batch_size,
                                             L.host_i32(si_in.spatial_shape, 3), L.ptr(out_indices), m_out,
                                             L.host_i32(ksize, 3), L.host_i32(stride, 3), L.host_i32(padding, 3),
                                             L.ptr(nbr), L.stream()))
    return nbr if m_out > 0 else nbr[:, :0].contiguous()


class SparseModule(nn.Module):
    """Marker base class (spconv.pytorch.SparseModule): modules that take a SparseConvTensor."""


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def run_modules(modules, input):
    """SparseSequential.forward over a list of modules.  A BatchNorm1d (+ the nn.ReLU right after it) on the features
    of a sparse tensor runs as one fused op (operators/batchnorm.py); everything else as in spconv."""
    from ..operators.batchnorm import bn_act, fusable

    i = 0
    while i < len(modules):
        module = modules[i]
        if (isinstance(module, SparseConvolution) and isinstance(input, SparseConvTensor) and i + 1 < len(modules)
                and isinstance(modules[i + 1], nn.BatchNorm1d)):
            # convolution + BatchNorm1d (+ ReLU) as one autograd node (conv_bn_act)
            relu = i + 2 < len(modules) and type(modules[i + 2]) is nn.ReLU
            fused = conv_bn_act(module, input, modules[i + 1], relu=relu)
            if fused is not None:
                input = fused
                i += 3 if relu else 2
                continue
        if is_spconv_module(module):
            input = module(input)
        elif isinstance(input, SparseConvTensor):
            if input.indices.shape[0] != 0:
                if isinstance(module, nn.BatchNorm1d) and fusable(module, input.features):
                    relu = i + 1 < len(modules) and type(modules[i + 1]) is nn.ReLU
                    input = input.replace_feature(bn_act(input.features, module, relu=relu))
                    i += 2 if relu else 1
                    continue
                input = input.replace_feature(module(input.features))
        else:
            input = module(input)
        i += 1
    return input


class SparseSequential(SparseModule):
    """spconv.pytorch.SparseSequential: sparse modules get the tensor, plain nn.Modules (BatchNorm1d,
    ReLU, ...) are applied to `.features` (sparse_net.py:85-95)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], dict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            self.add_module(name, module)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        self.add_module(str(len(self._modules)) if name is None else name, module)

    def forward(self, input):
        return run_modules([m for m in self._modules.values() if m is not None], input)


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, indice_key=None):
        super().__init__()
        assert ndim == 3 and groups == 1
        assert _triple(dilation) == (1, 1, 1), "dilation is not used by the EFG backbones"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.subm, self.indice_key = subm, indice_key
        if subm:
            assert self.stride == (1, 1, 1) and all(k % 2 == 1 for k in self.kernel_size)
        # spconv 2.x parameter layout: [Cout, kd, kh, kw, Cin]
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight.view(self.out_channels, -1), a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return "{}, {}, kernel_size={}, stride={}, padding={}, subm={}, indice_key={}".format(
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding, self.subm,
            self.indice_key)

    def _rulebook(self, x):
        ks, st = self.kernel_size, self.stride
        if self.subm:
            pad = tuple(k // 2 for k in ks)  # SubM ignores `padding`: the window is always centred
            # (the spatial shape is part of the key: a strided convolution hands its input's dict on to its output, and a
            # block that reuses one indice_key on both sides of it -- the reference's strided bottleneck block does,
            # sparse_net.py:198-213 -- must not pick up the table of the other resolution)
            key = ("subm", self.indice_key, ks, tuple(int(v) for v in x.spatial_shape)) if self.indice_key is not None else None
            if key is not None and key in x.indice_dict:
                return x.indice_dict[key], None
            m = x.indices.shape[0]
            with _on_geometry_stream() as main:
                si = x.site_index()
                nbr = _build_nbr(si, x.indices, m, ks, (1, 1, 1), pad)
                _hand_over(main, nbr, si.index, si.perm)
            rb = Rulebook(nbr, m, m, nbr.shape[0], True)
            if key is not None:
                x.indice_dict[key] = rb
            return rb, None
        pad = self.padding
        key = ("conv", ks, st, pad)
        if key in x._conv_cache:  # main and shortcut convs of a block share one geometry
            return x._conv_cache[key]
        with _on_geometry_stream() as main:
            si = x.site_index()
            pre = x.indice_dict.get(("geom", _t3(x.spatial_shape), _t3(ks), _t3(st), _t3(pad)))
            if pre is not None and (pre[0] is x.indices or (pre[0].data_ptr() == x.indices.data_ptr() and pre[0].shape == x.indices.shape)):
                # asked for up front (prefetch_downsample_chain), from THESE sites
                _, out_indices, out_site_index, oshape_py = pre
            else:
                out_indices, out_site_index, oshape_py = _downsample_geometry(x, ks, st, pad)
            m_out = out_indices.shape[0]
            nbr = _build_nbr(si, out_indices, m_out, ks, st, pad)
            _hand_over(main, nbr, out_indices, out_site_index.index, si.index, si.perm)
        rb = Rulebook(nbr, x.indices.shape[0], m_out, nbr.shape[0], False, in_indices=x.indices)
        geom = (out_indices, out_site_index, oshape_py)
        x._conv_cache[key] = (rb, geom)
        return rb, geom

    def forward(self, x):
        assert isinstance(x, SparseConvTensor)
        rb, geom = self._rulebook(x)
        feats = _SparseConvFunction.apply(x.features, self.weight, self.bias, rb, torch.is_grad_enabled())
        if self.subm:
            return x.replace_feature(feats)
        out_indices, site_index, out_shape = geom
        return SparseConvTensor(feats, out_indices, out_shape, x.batch_size, indice_dict=x.indice_dict,
                                _site_index=site_index)


class SubMConv3d(SparseConvolution):
    """spconv.pytorch.SubMConv3d(in, out, kernel_size, stride=1, padding=0, ..., bias=True, indice_key=None)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None):
        super().__init__(3, in_channels, out_channels, kernel_size, 1, padding, dilation, groups, bias, True,
                         indice_key)


class SparseConv3d(SparseConvolution):
    """spconv.pytorch.SparseConv3d(in, out, kernel_size, stride=1, padding=0, ..., bias=True, indice_key=None)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, False,
                         indice_key)
