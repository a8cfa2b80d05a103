"""Thin tensor-level wrappers over the C ABI (``include/mtn.h``).

Tensors are NHWC ``[B, H, W, C]`` or token-major ``[M, C]``; a channel slice of a wider buffer
is passed as a view (``buf[..., 64:128]``): only the last dim must be dense, the row stride
travels as ``ld``.  Nothing here computes on the host and nothing falls back to torch ops.
"""
import torch

from . import lib

F32, BF16 = 0, 1
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dtype_code(t):
    return _DT[t.dtype]


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def _rows(t):
    """(ptr, ld) of a [..., C] tensor whose leading dims form one regular row sequence."""
    assert t.stride(-1) == 1, "channel axis must be dense"
    ld = t.stride(-2) if t.dim() >= 2 else t.shape[-1]
    n = 1
    for d in range(t.dim() - 2, -1, -1):     # leading dims must be nested multiples of ld
        assert t.shape[d] == 1 or t.stride(d) == ld * n, f"irregular row layout {t.shape} {t.stride()}"
        n *= t.shape[d]
    return t.data_ptr(), ld


def _p(t):
    return 0 if t is None else t.data_ptr()


def conv_stats_tiles(M, cout):
    return lib.call("mtn_conv_stats_tiles", M, cout)


def conv_fwd(x, w, y, stats, ksize, stride):
    """x [B,H,W,Cin] -> y [B,OH,OW,Cout] raw conv output; stats [tiles,2,Cout] f32 or None."""
    B, H, W, Cin = x.shape
    Cout = y.shape[-1]
    xp, ldx = _rows(x)
    yp, ldy = _rows(y)
    lib.call("mtn_conv_fwd", xp, ldx, w.data_ptr(), yp, ldy, _p(stats), dtype_code(x),
             B, H, W, Cin, Cout, ksize, stride, _stream(x))


def conv_dgrad(dy, wt, dx, res, ksize, stride):
    """dy [B,OH,OW,Cout] -> dx [B,H,W,Cin] (+ res, which may alias dx)."""
    B, H, W, Cin = dx.shape
    Cout = dy.shape[-1]
    dyp, lddy = _rows(dy)
    dxp, lddx = _rows(dx)
    rp, ldr = _rows(res) if res is not None else (0, 0)
    lib.call("mtn_conv_dgrad", dyp, lddy, wt.data_ptr(), dxp, lddx, rp, ldr, dtype_code(dy),
             B, H, W, Cin, Cout, ksize, stride, _stream(dy))


def linear_fwd(x, w, out, bias=None, act=ACT_NONE, aux_out=None, res=None):
    """out = act(x @ w.T + bias) + res; x [M,K] rows, w [N,K]."""
    K = x.shape[-1]
    N = out.shape[-1]
    M = x.numel() // K
    xp, ldx = _rows(x)
    op, ldo = _rows(out)
    ap, lda = _rows(aux_out) if aux_out is not None else (0, 0)
    rp, ldr = _rows(res) if res is not None else (0, 0)
    lib.call("mtn_linear_fwd", xp, ldx, w.data_ptr(), _p(bias), act, ap, lda, rp, ldr, op, ldo,
             dtype_code(x), M, N, K, _stream(x))
