"""Split-precision (bf16 x 3) products for the A/B arm of the bench -- csrc/gemm_bf16x3.hip through the C ABI.

Never the default: `EFG_GEMM_ARM=bf16x3` swaps it in for the forward and the data-gradient product of the encoder-sized
`nn.Linear` layers (operators/linear.py) -- forward, data gradient and (via `wgrad`) the weight gradient; everything
else stays exact fp32."""
import torch

from .. import _lib


def pack(w, k, n, stride_k, stride_n):
    """Split B(kk, nn) = w.flatten()[kk * stride_k + nn * stride_n] into the MFMA lane order (device buffer)."""
    lib = _lib.lib()
    out = torch.empty(lib.efg_gemm_bf16x3_pack_bytes(k, n), dtype=torch.uint8, device=w.device)
    _lib.check(lib.efg_gemm_bf16x3_pack_f32(_lib.ptr(w), stride_k, stride_n, k, n, _lib.ptr(out), _lib.stream()))
    return out


def pack_linear(weight, transposed):
    """weight [out, in] of an nn.Linear.  transposed=False: B = W^T [in, out] (y = x W^T); True: B = W [out, in]
    (dx = dy W)."""
    w = weight.contiguous()
    o, i = w.shape
    return pack(w, i, o, 1, i) if not transposed else pack(w, o, i, i, 1)


def pack_linear_both(weight):
    """(B = W^T for y = x W^T, B = W for dx = dy W) of an nn.Linear weight [out, in], one launch."""
    w = weight.contiguous()
    o, i = w.shape
    lib = _lib.lib()
    fwd = torch.empty(lib.efg_gemm_bf16x3_pack_bytes(i, o), dtype=torch.uint8, device=w.device)
    dgr = torch.empty(lib.efg_gemm_bf16x3_pack_bytes(o, i), dtype=torch.uint8, device=w.device)
    _lib.check(lib.efg_gemm_bf16x3_pack_linear_f32(_lib.ptr(w), o, i, _lib.ptr(fwd), _lib.ptr(dgr), _lib.stream()))
    return fwd, dgr


def gemm(a, packed, n, bias=None, relu=False):
    """a [m, k] fp32 (rows contiguous) x packed B [k, n] -> [m, n] fp32."""
    assert a.dim() == 2 and a.dtype == torch.float32 and a.stride(1) == 1
    m, k = a.shape
    c = torch.empty((m, n), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().efg_gemm_bf16x3_f32(a.data_ptr(), m, k, a.stride(0), _lib.ptr(packed), n,
                                              _lib.ptr(bias) if bias is not None else None, 1 if relu else 0,
                                              _lib.ptr(c), n, _lib.stream()))
    return c


def wgrad(g, x):
    """g [m, n] (grad_output), x [m, k] (input), fp32 rows contiguous -> g^T x [n, k] (an nn.Linear's weight gradient)."""
    assert g.dim() == 2 and x.dim() == 2 and g.shape[0] == x.shape[0] and g.stride(1) == 1 and x.stride(1) == 1
    m, n = g.shape
    k = x.shape[1]
    lib = _lib.lib()
    out = torch.empty((n, k), dtype=torch.float32, device=g.device)
    ws_bytes = lib.efg_gemm_bf16x3_wgrad_workspace_bytes(m, n, k)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=g.device)
    _lib.check(lib.efg_gemm_bf16x3_wgrad_f32(g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), m, n, k, _lib.ptr(out),
                                             _lib.ptr(ws), ws_bytes, _lib.stream()))
    return out
