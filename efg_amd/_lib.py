"""ctypes binding of libefg_hip.so (C ABI declared in include/efg_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails this module
raises.  Tensors are passed as raw device pointers plus the current HIP stream.
"""
import ctypes
import os

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libefg_hip.so")
_lib = None

c_void_p, c_int, c_int64, c_size_t, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t,
                                               ctypes.c_float)

# name -> (restype, argtypes); pointers are passed as c_void_p
_SIGS = {
    "efg_last_error": (ctypes.c_char_p, []),
    "efg_version": (ctypes.c_char_p, []),
    "efg_capture_stream_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "efg_capture_stream_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "efg_dynamic_voxelize_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "efg_hard_voxelize_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "efg_hard_voxelize_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                      c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "efg_hard_voxelize_debug_timeline": (c_size_t, [c_void_p]),
    "efg_scatter_workspace_bytes": (c_size_t, [c_int64, c_int, c_void_p]),
    "efg_scatter_index": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                  c_void_p]),
    "efg_scatter_reduce_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "efg_scatter_reduce_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "efg_scatter_backward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                         c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "efg_spconv_index_bytes": (c_size_t, [c_int, c_void_p]),
    "efg_spconv_index_workspace_bytes": (c_size_t, [c_int, c_void_p]),
    "efg_spconv_index_from_indices": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_size_t, c_void_p]),
    "efg_spconv_index_downsample": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "efg_spconv_index_rank": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "efg_spconv_index_emit": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "efg_spconv_build_nbr": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "efg_spconv_build_rnbr": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "efg_spconv_packed_weight_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "efg_spconv_pack_weight_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "efg_spconv_pack_weights_multi": (c_int, [c_void_p, c_int, c_void_p]),
    "efg_spconv_forward_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64,
                                       c_void_p, c_void_p]),
    "efg_spconv_dgrad_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p,
                                     c_void_p, c_void_p]),
    "efg_spconv_tile_plan_bytes": (c_size_t, [c_int64, c_int]),
    "efg_spconv_tile_plan": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "efg_spconv_forward_tiled_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                             c_int64, c_int, c_void_p, c_void_p]),
    "efg_spconv_tiled_pair_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                          c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "efg_spconv_small_ok": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p]),
    "efg_spconv_tile_bf16x3_ok": (c_int, [c_int, c_int, c_int, c_int64, c_int64]),
    "efg_spconv_streamk_fallbacks": (c_int, [c_void_p, c_int]),
    "efg_ticket_ring_errors": (c_int, [c_void_p, c_int]),
    "efg_spconv_parity_order": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "efg_spconv_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "efg_spconv_wgrad_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "efg_spconv_wgrad_tiled_ok": (c_int, [c_int, c_int, c_int]),
    "efg_spconv_wgrad_tiled_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "efg_spconv_wgrad_sched_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "efg_spconv_wgrad_sched": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "efg_spconv_wgrad_tiled_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "efg_spconv_wgrad_tiled_pair_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "efg_sparse_to_dense_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "efg_dense_to_sparse_f32": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "efg_sparse_to_bev_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "efg_bev_to_sparse_f32": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "efg_msda_forward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_void_p, c_void_p]),
    "efg_msda_backward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "efg_box_attn_fused_forward_f32": (c_int, [c_void_p] * 7 + [c_int] * 8 + [c_void_p, c_void_p]),
    "efg_box_attn_fused_forward_strided_f32": (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_int, c_void_p] + [c_int] * 8 + [c_void_p, c_void_p]),
    "efg_box_attn_fused_backward_strided_f32": (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "efg_box_attn_fused_backward_workspace_bytes": (c_size_t, [c_int] * 6),
    "efg_box_attn_fused_backward_f32": (c_int, [c_void_p] * 8 + [c_int] * 8 + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "efg_boxes_bev_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "efg_nms_workspace_bytes": (c_size_t, [c_int]),
    "efg_topk_unsorted_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "efg_box_refine_forward_f32": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    "efg_box_refine_backward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p]),
    "efg_add_layernorm_forward_f32": (c_int, [c_void_p] * 4 + [c_float, c_int64, c_int] + [c_void_p] * 5),
    "efg_add_layernorm_backward_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "efg_add_layernorm_backward_f32": (c_int, [c_void_p] * 5 + [c_int64, c_int] + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "efg_points_transform_filter_workspace_bytes": (c_size_t, [c_int64]),
    "efg_points_transform_filter_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_size_t, c_void_p]),
    "efg_points_gather_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "efg_match_cost_f32": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_float] * 6 + [c_void_p, c_void_p]),
    "efg_focal_loss_forward_f32": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_float, c_void_p, c_void_p,
                                           c_void_p]),
    "efg_focal_loss_backward_f32": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_float, c_void_p,
                                            c_void_p, c_void_p, c_void_p]),
    "efg_box_loss_forward_f32": (c_int, [c_void_p] * 6 + [c_int64] + [c_int] * 4 + [c_void_p] * 3),
    "efg_box_loss_backward_f32": (c_int, [c_void_p] * 6 + [c_int64] + [c_int] * 4 + [c_void_p] * 4),
    "efg_bn_workspace_bytes": (c_size_t, [c_int]),
    "efg_bn_forward_f32": (c_int, [c_void_p] * 7 + [c_float, c_float, c_int64, c_int, c_int] + [c_void_p] * 4 +
                           [c_size_t, c_void_p]),
    "efg_bn_backward_f32": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int] + [c_void_p] * 5 + [c_size_t, c_void_p]),
    "efg_lsap_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "efg_nms_f32": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "efg_spconv_tile_shape": (c_int, [c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "efg_cylinder_select_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p, c_int64, c_void_p,
                                        c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "efg_attention_fwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int,
                                      c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "efg_attention_bwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "efg_attention_long_fwd_f32": (c_int, [c_void_p, c_int64, c_int64] * 3 + [c_void_p, c_int, c_int64, c_int, c_int, c_float,
                                                                             c_void_p, c_void_p, c_void_p]),
    "efg_attention_long_bwd_f32": (c_int, [c_void_p, c_int64, c_int64] * 3 + [c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                                             c_int64, c_int, c_int, c_float, c_void_p, c_void_p,
                                                                             c_void_p, c_void_p, c_void_p]),
    "efg_relu_bwd_colsum_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "efg_nms_segmented_f32": (c_int, [c_void_p, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                      c_void_p]),
    "efg_gn_workspace_bytes": (c_size_t, [c_int, c_int]),
    "efg_gn_forward_f32": (c_int, [c_void_p] * 3 + [c_float, c_int, c_int64, c_int, c_int] + [c_void_p] * 4 +
                           [c_size_t, c_void_p]),
    "efg_gn_backward_f32": (c_int, [c_void_p] * 5 + [c_int, c_int64, c_int, c_int] + [c_void_p] * 4 +
                            [c_size_t, c_void_p]),
    "efg_gemm_bf16x3_pack_bytes": (c_size_t, [c_int, c_int]),
    "efg_gemm_bf16x3_pack_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "efg_gemm_bf16x3_pack_linear_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "efg_gemm_bf16x3_f32": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64,
                                    c_void_p]),
    "efg_gemm_bf16x3_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "efg_gemm_bf16x3_wgrad_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p,
                                          c_size_t, c_void_p]),
    "efg_colsum_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "efg_colsum_f32": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)


def lib_path():
    return _LIB_PATH


def _warn_if_stale():
    """A library older than its sources measures (and tests) yesterday's kernels: say so, loudly, once."""
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    try:
        built = os.path.getmtime(_LIB_PATH)
        newer = [f for f in os.listdir(csrc) if os.path.getmtime(os.path.join(csrc, f)) > built + 1.0]
    except OSError:
        return
    if newer:
        import warnings

        warnings.warn("efg_amd: %s is OLDER than %s -- rebuild with `python -m efg_amd.build`"
                      % (os.path.basename(_LIB_PATH), ", ".join(sorted(newer)[:4])), RuntimeWarning, stacklevel=3)


def lib():
    """Load libefg_hip.so (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                "efg_amd: %s is missing. Build it with `python -m efg_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback on the product path." % _LIB_PATH)
        _warn_if_stale()
        # EFG_HIP_LIB_AB=<file in efg_amd/lib/>: an A/B build of the same sources (scripts/build_ab.sh compiles one
        # translation unit with other -D switches) -- same C ABI, for same-box kernel comparisons only
        ab = os.environ.get("EFG_HIP_LIB_AB")
        path = os.path.join(os.path.dirname(_LIB_PATH), os.path.basename(ab)) if ab else _LIB_PATH
        if ab and not os.path.exists(path):
            raise RuntimeError("efg_amd: EFG_HIP_LIB_AB=%s not found in %s" % (ab, os.path.dirname(_LIB_PATH)))
        _lib = ctypes.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("efg_hip: " + lib().efg_last_error().decode())


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "efg_hip ops need contiguous tensors"
    return t.data_ptr()


def stream():
    """Raw handle of torch's current stream on the current device (the ~150 launches per step through this library
    each ask for it; `torch.cuda.current_stream()` builds a Stream object every time, ~4 us)."""
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return raw(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "efg_amd ops run on MI355X only (got a %s tensor); there is no CPU fallback" % t.device.type)


def host_f32(vals, n):
    vals = [float(v) for v in vals]
    assert len(vals) == n
    return (ctypes.c_float * n)(*vals)


def host_i32(vals, n=None):
    vals = [int(v) for v in vals]
    assert n is None or len(vals) == n
    return (ctypes.c_int * len(vals))(*vals)


def host_i64(vals):
    vals = [int(v) for v in vals]
    return (ctypes.c_int64 * len(vals))(*vals)
