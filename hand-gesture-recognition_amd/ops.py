"""Thin tensor-level wrappers over the C ABI (``include/mtn.h``).

Tensors are NHWC ``[B, H, W, C]`` or token-major ``[M, C]``; a channel slice of a wider buffer
is passed as a view (``buf[..., 64:128]``): only the last dim must be dense, the row stride
travels as ``ld``.  Nothing here computes on the host and nothing falls back to torch ops.
"""
import torch

from . import lib

F32, BF16, F32X3 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2

# What a torch.float32 tensor holds: F32 = plain fp32 (exact-f32 MFMA) or F32X3 = split storage (bf16 hi / lo halves per
# 32-channel block, three bf16 MFMAs per product; include/mtn.h MTN_DTYPE_F32X3).  It cannot be read off the tensor, so every
# wrapper takes it per call (``mode=``), like the C ABI takes its dtype argument: an execution plan passes its own
# (``with_mode``), two plans of different modes can run from different threads.  ``_f32_code`` is only the DEFAULT for direct
# callers that pass nothing (kernel tests select it with set_f32_matmul); the package itself never relies on it.
_f32_code = F32


def set_f32_matmul(code):
    global _f32_code
    if code not in (F32, F32X3):
        raise ValueError(f"fp32 matmul mode {code}")
    _f32_code = code


def code_of(dtype, mode=None):
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float32:
        if mode is None:
            return _f32_code
        if mode not in (F32, F32X3):
            raise ValueError(f"fp32 mode {mode}")
        return mode
    raise TypeError(f"unsupported storage dtype {dtype}")


def dtype_code(t, mode=None):
    return code_of(t.dtype, mode)


class with_mode:
    """The wrappers of this module with ``mode`` bound: ``o = ops.with_mode(ops.F32X3); o.conv_fwd(...)`` calls
    ``ops.conv_fwd(..., mode=ops.F32X3)``.  Names are resolved on the module at CALL time (a wrapper that instruments or
    replaces ``ops.conv_fwd`` -- bench.py's roofline brackets, tests -- is honoured and receives ``mode=`` like the function
    it stands in for); names that take no mode (constants, size queries without a dtype) pass through."""

    def __init__(self, mode):
        import sys
        if mode not in (F32, F32X3):
            raise ValueError(f"fp32 mode {mode}")
        self.mode = mode
        self._mod = sys.modules[__name__]

    def __getattr__(self, name):
        fn = getattr(self._mod, name)
        if name in _MODED:
            import functools
            return functools.partial(fn, mode=self.mode)
        return fn


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


def conv_stats_tiles(M, cout, K):
    """rows of the partial-statistics slab conv_fwd writes (K = ksize*ksize*Cin)."""
    return lib.call("mtn_conv_stats_tiles", M, cout, K)


KERNEL_FAMILIES = {0: "igemm_kernel", 1: "igemm_kernel", 2: "igemm_kernel", 3: "pw_kernel", 4: "dgrad2_kernel", 5: "win3_kernel"}
KERNEL_VARIANTS = {0: "streamed tiles", 1: "window", 2: "image tile", 3: "persistent pointwise", 4: "stride-2 data gradient", 5: "win3"}


def conv_kernel_family(dtype, dgrad, B, H, W, Cin, Cout, ksize, stride, has_res=False, has_bn=False, mode=None):
    """MTN_KERNEL_* code (include/mtn.h) of the kernel that serves this conv / Linear-shaped product under the launch rules of
    the loaded library (no launch): dgrad=False forward (a Linear is B = rows, H = W = 1, ksize = stride = 1), True data
    gradient given the geometry of the FORWARD conv.  Profilers and bench.py label their per-family figures with it."""
    code = lib.call("mtn_conv_kernel_family", code_of(dtype, mode), int(bool(dgrad)), B, H, W, Cin, Cout, ksize, stride,
                    int(bool(has_res)), int(bool(has_bn)))
    if code < 0:
        raise RuntimeError(f"mtn_conv_kernel_family: {lib.get().mtn_last_error().decode()}")
    return code


def conv_fwd_stats_tiles(dtype, B, H, W, Cin, Cout, ksize, stride, mode=None):
    """rows of the partial-statistics slab conv_fwd writes for this layer geometry."""
    return lib.call("mtn_conv_fwd_stats_tiles", code_of(dtype, mode), B, H, W, Cin, Cout, ksize, stride)


def conv_fwd(x, w, y, stats, ksize, stride, stats_tiles=None, mode=None):
    """x [B,H,W,Cin] -> y [B,OH,OW,Cout] raw conv output; stats [>=tiles,2,Cout] f32 or None
    (tiles = conv_fwd_stats_tiles(...): the slab is allocated with slab_rows(tiles) rows)."""
    B, H, W, Cin = x.shape
    Cout = y.shape[-1]
    xp, ldx = _rows(x)
    yp, ldy = _rows(y)
    tiles = 0
    if stats is not None:
        # the row count the slab was SIZED for (a plan passes what it allocated; the library refuses
        # the launch if its tile choice no longer matches)
        tiles = conv_fwd_stats_tiles(x.dtype, B, H, W, Cin, Cout, ksize, stride, mode=mode) if stats_tiles is None else int(stats_tiles)
        if stats.dtype != torch.float32 or stats.numel() < tiles * 2 * Cout:
            raise ValueError(f"conv_fwd: stats slab too small for {tiles} row tiles")
    lib.call("mtn_conv_fwd", xp, ldx, w.data_ptr(), yp, ldy, _p(stats), tiles, dtype_code(x, mode),
             B, H, W, Cin, Cout, ksize, stride, _stream(x))


def conv_bias_act_fwd(x, w, bias, y, act, ksize, stride, res=None, mode=None):
    """Inference form of a Conv unit with folded BatchNorm: y = act(conv(x) + bias [+ res]) in one launch
    (w = weights packed with the BatchNorm scale, bias f32 [Cout] = its shift)."""
    B, H, W, Cin = x.shape
    Cout = y.shape[-1]
    if bias.dtype != torch.float32 or bias.numel() != Cout or not bias.is_contiguous():
        raise ValueError("conv_bias_act_fwd: bias must be a contiguous f32 [Cout] tensor")
    xp, ldx = _rows(x)
    yp, ldy = _rows(y)
    rp, ldr = _rows(res) if res is not None else (0, 0)
    lib.call("mtn_conv_bias_act_fwd", xp, ldx, w.data_ptr(), bias.data_ptr(), act, rp, ldr, yp, ldy, dtype_code(x, mode),
             B, H, W, Cin, Cout, ksize, stride, _stream(x))


def conv_dgrad_bn_tiles(dtype, B, H, W, Cin, Cout, ksize, stride, mode=None):
    return lib.call("mtn_conv_dgrad_bn_tiles", code_of(dtype, mode), B, H, W, Cin, Cout, ksize, stride)


# The hand-over record between conv_dgrad(bn=...) and the bn_act_bwd(partial_rows=...) that consumes its rows (include/mtn.h
# mtn_bn_handover): the library fills and checks it but keeps nothing between the calls -- it lives here, one per host thread
# (forward runs on the main thread, backward on autograd's device thread; the two calls of a hand-over are consecutive steps of
# one backward pass on one thread).
import ctypes as _C
import threading as _threading

_handover = _threading.local()


def _handover_record():
    rec = getattr(_handover, "rec", None)
    if rec is None:
        rec = _handover.rec = lib.BnHandover()
    return rec


def conv_dgrad(dy, wt, dx, res, ksize, stride, bn=None, mode=None):
    """dy [B,OH,OW,Cout] -> dx [B,H,W,Cin] (+ res, which may alias dx).

    ``bn`` = (y, res, scale_shift, mean_rstd, act, partial, tiles) of the Conv unit whose output gradient dx is: the
    launch also writes that unit's BatchNorm-backward partial sums (``bn_act_bwd(..., partial_rows=tiles)``)."""
    B, H, W, Cin = dx.shape
    Cout = dy.shape[-1]
    dyp, lddy = _rows(dy)
    dxp, lddx = _rows(dx)
    rp, ldr = _rows(res) if res is not None else (0, 0)
    if bn is None:
        lib.call("mtn_conv_dgrad", dyp, lddy, wt.data_ptr(), dxp, lddx, rp, ldr, dtype_code(dy, mode),
                 B, H, W, Cin, Cout, ksize, stride, _stream(dy))
        return
    y, bres, ss, mr, act, partial, tiles = bn
    if tuple(y.shape) != tuple(dx.shape) or y.dtype != dx.dtype or (bres is not None and tuple(bres.shape) != tuple(dx.shape)):
        raise ValueError("conv_dgrad: the BatchNorm unit's raw output / residual must have the shape and dtype of dx")
    if partial.dtype != torch.float32 or partial.numel() < slab_rows(tiles) * 2 * Cin:
        raise ValueError(f"conv_dgrad: BatchNorm partial slab too small for {tiles} row tiles")
    yp, ldy = _rows(y)
    brp, ldbr = _rows(bres) if bres is not None else (0, 0)
    lib.call("mtn_conv_dgrad_bn", dyp, lddy, wt.data_ptr(), dxp, lddx, rp, ldr, yp, ldy, brp, ldbr, ss.data_ptr(),
             mr.data_ptr(), act, partial.data_ptr(), tiles, _C.addressof(_handover_record()), dtype_code(dy, mode), B, H, W, Cin,
             Cout, ksize, stride, _stream(dy))


def linear_fwd(x, w, out, bias=None, act=ACT_NONE, aux_out=None, res=None, mode=None):
    """out = act(x @ w.T + bias) + res; x [M,K] rows, w [N,K]."""
    K = x.shape[-1]
    N = out.shape[-1]
    M = x.numel() // K
    xp, ldx = _rows(x)
    op, ldo = _rows(out)
    ap, lda = _rows(aux_out) if aux_out is not None else (0, 0)
    rp, ldr = _rows(res) if res is not None else (0, 0)
    lib.call("mtn_linear_fwd", xp, ldx, w.data_ptr(), _p(bias), act, ap, lda, rp, ldr, op, ldo,
             dtype_code(x, mode), M, N, K, _stream(x))


def linear_gelu_bwd_tiles(M, N, K):
    return lib.call("mtn_linear_gelu_bwd_tiles", M, N, K)


def linear_gelu_bwd(dh, wt, a, da, partial, tiles, dbias, mode=None):
    """da = (dh @ wt.T) * gelu'(a); dbias = column sums of da (f32 [N], overwritten).  dh [M,K], wt [N,K], a / da [M,N]."""
    K, N = dh.shape[-1], da.shape[-1]
    M = dh.numel() // K
    if a.shape != da.shape or a.dtype != da.dtype or dh.dtype != da.dtype:
        raise ValueError("linear_gelu_bwd: a and da must have one shape and dtype")
    if partial.dtype != torch.float32 or partial.numel() < slab_rows(tiles) * 2 * N:
        raise ValueError("linear_gelu_bwd: partial slab size")
    if dbias is not None and (dbias.numel() != N or dbias.dtype != torch.float32):
        raise ValueError("linear_gelu_bwd: dbias size")
    hp, ldh = _rows(dh)
    ap, lda = _rows(a)
    dp, ldd = _rows(da)
    lib.call("mtn_linear_gelu_bwd", hp, ldh, wt.data_ptr(), ap, lda, dp, ldd, partial.data_ptr(), tiles, _p(dbias),
             dtype_code(dh, mode), M, N, K, _stream(dh))


def conv_wgrad_workspace_bytes(dtype, B, H, W, Cin, Cout, ksize, stride, mode=None):
    return lib.call("mtn_conv_wgrad_workspace_bytes", code_of(dtype, mode), B, H, W, Cin, Cout, ksize, stride)


def conv_wgrad(x, dy, dw, workspace, ksize, stride, mode=None):
    """x [B,H,W,Cin], dy [B,OH,OW,Cout] -> dw f32 [Cout,Cin,k,k] (overwritten)."""
    B, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    xp, ldx = _rows(x)
    dyp, lddy = _rows(dy)
    assert dw.dtype == torch.float32 and dw.is_contiguous()
    lib.call("mtn_conv_wgrad", xp, ldx, dyp, lddy, dw.data_ptr(), workspace.data_ptr(),
             workspace.numel() * workspace.element_size(), dtype_code(x, mode), B, H, W, Cin, Cout,
             ksize, stride, _stream(x))


def linear_wgrad_workspace_bytes(dtype, M, N, K, mode=None):
    return lib.call("mtn_linear_wgrad_workspace_bytes", code_of(dtype, mode), M, N, K)


def linear_wgrad(x, dy, dw, workspace, mode=None):
    """dw[N,K] = dy[M,N]^T @ x[M,K] (f32, overwritten)."""
    K, N = x.shape[-1], dy.shape[-1]
    M = x.numel() // K
    xp, ldx = _rows(x)
    dyp, lddy = _rows(dy)
    lib.call("mtn_linear_wgrad", xp, ldx, dyp, lddy, dw.data_ptr(), workspace.data_ptr(),
             workspace.numel() * workspace.element_size(), dtype_code(x, mode), M, N, K, _stream(x))


def wgrad_group_items(problems):
    """[(x [M,K], dy [M,N], dw f32 [N,K]), ...] -> the mtn_wgrad_item array mtn_linear_wgrad_group takes.  The tensors' storage must
    stay where it is for as long as the array is used (a plan builds it once over its preallocated buffers)."""
    arr = (lib.WgradItem * len(problems))()
    M = None
    for it, (x, dy, dw) in zip(arr, problems):
        K, N = x.shape[-1], dy.shape[-1]
        m = x.numel() // K
        if M is None:
            M = m
        if m != M or dy.numel() // N != M or tuple(dw.shape) != (N, K) or dw.dtype != torch.float32 or not dw.is_contiguous() or dy.dtype != x.dtype:
            raise ValueError("wgrad_group_items: problems must share the row count; dw must be a contiguous f32 [N, K]")
        (it.x, it.ldx), (it.dy, it.lddy) = _rows(x), _rows(dy)
        it.dw, it.N, it.K = dw.data_ptr(), N, K
    return arr, M


def conv_wgrad_group_items(problems):
    """[(x [B,H,W,Cin] (a channel slice is fine), dy [B,OH,OW,Cout], dw f32 [Cout,Cin,k,k]), ...] of ONE geometry -> the
    mtn_wgrad_item array of mtn_conv_wgrad_group (same lifetime rule as wgrad_group_items)."""
    arr = (lib.WgradItem * len(problems))()
    for it, (x, dy, dw) in zip(arr, problems):
        if dw.dtype != torch.float32 or not dw.is_contiguous() or dy.dtype != x.dtype or x.shape != problems[0][0].shape or dy.shape != problems[0][1].shape:
            raise ValueError("conv_wgrad_group_items: one geometry per group; dw must be a contiguous f32 [Cout, Cin, k, k]")
        (it.x, it.ldx), (it.dy, it.lddy) = _rows(x), _rows(dy)
        it.dw, it.N, it.K = dw.data_ptr(), dy.shape[-1], x.shape[-1]
    return arr


def conv_wgrad_group_workspace_bytes(dtype, count, B, H, W, Cin, Cout, ksize, stride, mode=None):
    return lib.call("mtn_conv_wgrad_group_workspace_bytes", code_of(dtype, mode), count, B, H, W, Cin, Cout, ksize, stride)


def conv_wgrad_group(items, workspace, dtype, B, H, W, Cin, Cout, ksize, stride, mode=None):
    """the weight gradients of every problem of `items` (conv_wgrad_group_items; x [B,H,W,Cin]) in one launch + one split sum"""
    lib.call("mtn_conv_wgrad_group", _C.addressof(items), len(items), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
             code_of(dtype, mode), B, H, W, Cin, Cout, ksize, stride, _stream(workspace))


def linear_wgrad_group_workspace_bytes(dtype, items, M, mode=None):
    """`items`: the array of wgrad_group_items, or just the problems' [(N, K), ...] (sizing a workspace before the buffers exist)"""
    if isinstance(items, (list, tuple)):
        arr = (lib.WgradItem * len(items))()
        for it, (N, K) in zip(arr, items):
            it.N, it.K = N, K
        items = arr
    return lib.call("mtn_linear_wgrad_group_workspace_bytes", code_of(dtype, mode), M, _C.addressof(items), len(items))


def linear_wgrad_group(items, M, workspace, dtype, mode=None):
    """dw_i[N_i,K_i] = dy_i[M,N_i]^T @ x_i[M,K_i] for every problem of `items` (wgrad_group_items) in one launch + one split sum."""
    lib.call("mtn_linear_wgrad_group", _C.addressof(items), len(items), M, workspace.data_ptr(),
             workspace.numel() * workspace.element_size(), code_of(dtype, mode), _stream(workspace))


def slab_rows(rows):
    return lib.call("mtn_slab_rows", rows)


def bn_finalize(stats, tiles, count, gamma, beta, running_mean, running_var, nbt, scale_shift,
                mean_rstd, momentum=0.1, eps=1e-5):
    C = gamma.numel()
    lib.call("mtn_bn_finalize", stats.data_ptr(), tiles, C, count, gamma.data_ptr(), beta.data_ptr(),
             _p(running_mean), _p(running_var), _p(nbt), momentum, eps, scale_shift.data_ptr(),
             mean_rstd.data_ptr(), _stream(stats))


def bn_eval_scale_shift(gamma, beta, running_mean, running_var, scale_shift, eps=1e-5):
    lib.call("mtn_bn_eval_scale_shift", gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(),
             running_var.data_ptr(), eps, scale_shift.data_ptr(), gamma.numel(), _stream(gamma))


def bn_act_fwd(y, scale_shift, out, act, res=None, mode=None):
    C = y.shape[-1]
    M = y.numel() // C
    yp, ldy = _rows(y)
    op, ldo = _rows(out)
    rp, ldr = _rows(res) if res is not None else (0, 0)
    lib.call("mtn_bn_act_fwd", yp, ldy, scale_shift.data_ptr(), rp, ldr, op, ldo, dtype_code(y, mode),
             M, C, act, _stream(y))


def bn_bwd_partial_rows(M):
    return lib.call("mtn_bn_bwd_partial_rows", M)


def bn_act_bwd(dout, y, res, scale_shift, mean_rstd, partial, coef, dgamma, dbeta, dy, act,
               dres=None, extra=None, partial_rows=0, mode=None):
    C = y.shape[-1]
    M = y.numel() // C
    dop, lddo = _rows(dout)
    yp, ldy = _rows(y)
    rp, ldr = _rows(res) if res is not None else (0, 0)
    dyp, lddy = _rows(dy)
    drp, lddr = _rows(dres) if dres is not None else (0, 0)
    ep, lde = _rows(extra) if extra is not None else (0, 0)
    lib.call("mtn_bn_act_bwd", dop, lddo, yp, ldy, rp, ldr, scale_shift.data_ptr(),
             mean_rstd.data_ptr(), partial.data_ptr(), coef.data_ptr(), dgamma.data_ptr(),
             dbeta.data_ptr(), dyp, lddy, drp, lddr, ep, lde, partial_rows, _C.addressof(_handover_record()) if partial_rows > 0 else 0,
             dtype_code(y, mode), M, C, act, _stream(y))


def conv1_stats_tiles(B, H, W):
    return lib.call("mtn_conv1_stats_tiles", B, H, W)


def conv1_fwd(x_nchw, w, y, stats, mode=None):
    """x f32 [B,3,H,W] (NCHW, as the dataloader delivers it) -> y [B,H/2,W/2,64] raw."""
    B, _, H, W = x_nchw.shape
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and y.is_contiguous()
    lib.call("mtn_conv1_fwd", x_nchw.data_ptr(), w.data_ptr(), y.data_ptr(), _p(stats),
             dtype_code(y, mode), B, H, W, _stream(y))


def conv1_wgrad_partial_rows(B, H, W):
    return lib.call("mtn_conv1_wgrad_partial_rows", B, H, W)


def conv1_wgrad(x_nchw, dy, dw, partial, mode=None):
    B, _, H, W = x_nchw.shape
    assert dy.is_contiguous() and dw.is_contiguous()
    lib.call("mtn_conv1_wgrad", x_nchw.data_ptr(), dy.data_ptr(), dw.data_ptr(), partial.data_ptr(),
             dtype_code(dy, mode), B, H, W, _stream(dy))


def conv1_stats(x_nchw, w, stats, mode=None):
    """fused stem, pass 1: BN partial statistics of conv(x) (nothing else is stored)."""
    B, _, H, W = x_nchw.shape
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and stats.dtype == torch.float32
    assert stats.numel() >= conv1_stats_tiles(B, H, W) * 128
    lib.call("mtn_conv1_stats", x_nchw.data_ptr(), w.data_ptr(), stats.data_ptr(), dtype_code(w, mode), B, H, W, _stream(stats))


def conv1_bn_act(x_nchw, w, scale_shift, out, act, mode=None):
    """fused stem, pass 2: out [B,H/2,W/2,64] = act(conv(x) * scale + shift)."""
    B, _, H, W = x_nchw.shape
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and out.is_contiguous()
    assert tuple(out.shape) == (B, H // 2, W // 2, 64) and out.dtype == w.dtype
    lib.call("mtn_conv1_bn_act", x_nchw.data_ptr(), w.data_ptr(), scale_shift.data_ptr(), out.data_ptr(), act,
             dtype_code(out, mode), B, H, W, _stream(out))


def conv1_bwd_partial_shape(B, H, W):
    return lib.call("mtn_conv1_bwd_partial_rows", B, H, W), lib.call("mtn_conv1_bwd_partial_width")


def conv1_bwd(x_nchw, w, dout, scale_shift, mean_rstd, partial, dgamma, dbeta, dw, act, mode=None):
    """fused stem backward: dgamma, dbeta [64] and dw f32 [64,3,3,3] from (x, d activation) in one pass."""
    B, _, H, W = x_nchw.shape
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and dout.is_contiguous() and dw.is_contiguous()
    assert tuple(dout.shape) == (B, H // 2, W // 2, 64) and dout.dtype == w.dtype
    rows, width = conv1_bwd_partial_shape(B, H, W)
    assert partial.dtype == torch.float32 and partial.numel() >= rows * width
    lib.call("mtn_conv1_bwd", x_nchw.data_ptr(), w.data_ptr(), dout.data_ptr(), scale_shift.data_ptr(),
             mean_rstd.data_ptr(), partial.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dw.data_ptr(), act,
             dtype_code(dout, mode), B, H, W, _stream(dout))


def conv1_dx(x_nchw, w_oihw, dout, scale_shift, mean_rstd, dgamma, dbeta, dx, act, mode=None):
    """gradient of the image through the stem (after conv1_bwd of the same pass): dx f32 NCHW, overwritten."""
    B, _, H, W = x_nchw.shape
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and dout.is_contiguous()
    assert dx.dtype == torch.float32 and dx.is_contiguous() and tuple(dx.shape) == tuple(x_nchw.shape)
    assert w_oihw.dtype == torch.float32 and w_oihw.is_contiguous() and tuple(w_oihw.shape) == (64, 3, 3, 3)
    assert tuple(dout.shape) == (B, H // 2, W // 2, 64)
    lib.call("mtn_conv1_dx", x_nchw.data_ptr(), w_oihw.data_ptr(), dout.data_ptr(), scale_shift.data_ptr(),
             mean_rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dx.data_ptr(), act, dtype_code(dout, mode),
             B, H, W, _stream(dout))


# ---- ViT encoder --------------------------------------------------------------------------------
def proj_tokens_fwd(x, w, pos, cls, tok, mode=None):
    """tok[b, 0] = cls ; tok[b, 1 + p] = (x[b] @ w.T)[p] + pos[p]: the 1x1 projection with the token assembly in its epilogue
    (x [B,H,W,Cin], w packed [Cout, Cin], pos f32 [H*W, Cout], cls f32 [Cout] (any shape), tok [B, H*W + 1, Cout])."""
    B, H, W, Cin = x.shape
    Cout = tok.shape[-1]
    if tuple(tok.shape) != (B, H * W + 1, Cout) or not tok.is_contiguous() or tok.dtype != x.dtype:
        raise ValueError(f"proj_tokens_fwd: tok must be a contiguous [{B}, {H * W + 1}, Cout] tensor of x's dtype")
    if pos.dtype != torch.float32 or tuple(pos.shape) != (H * W, Cout) or not pos.is_contiguous() or cls.dtype != torch.float32 or cls.numel() != Cout:
        raise ValueError("proj_tokens_fwd: pos must be f32 [H*W, Cout], cls f32 with Cout elements")
    xp, ldx = _rows(x)
    lib.call("mtn_proj_tokens_fwd", xp, ldx, w.data_ptr(), pos.data_ptr(), cls.data_ptr(), tok.data_ptr(), dtype_code(x, mode),
             B, H, W, Cin, Cout, _stream(x))


def tokens_fwd(feat, pos, cls, tok, mode=None):
    """feat [B*P,256] (+pos f32 [P,256]) and cls f32 [256] -> tok [B,P+1,256]."""
    B, P1, _ = tok.shape
    lib.call("mtn_tokens_fwd", feat.data_ptr(), pos.data_ptr(), cls.data_ptr(), tok.data_ptr(),
             dtype_code(tok, mode), B, P1 - 1, _stream(tok))


def tokens_bwd(dtok, dfeat, dcls, mode=None):
    B, P1, _ = dtok.shape
    lib.call("mtn_tokens_bwd", dtok.data_ptr(), dfeat.data_ptr(), dcls.data_ptr(), dtype_code(dtok, mode),
             B, P1 - 1, _stream(dtok))


def layernorm_fwd(x, gamma, beta, y, mean_rstd, eps=1e-5, mode=None):
    xp, ldx = _rows(x)
    yp, ldy = _rows(y)
    M = x.numel() // x.shape[-1]
    lib.call("mtn_layernorm_fwd", xp, ldx, gamma.data_ptr(), beta.data_ptr(), yp, ldy, _p(mean_rstd),
             dtype_code(x, mode), M, eps, _stream(x))


def rowslab_rows(M):
    return lib.call("mtn_rowslab_rows", M)


def layernorm_bwd(dy, x, mean_rstd, gamma, dx, partial, dgamma_dbeta, dres=None, mode=None):
    dyp, lddy = _rows(dy)
    xp, ldx = _rows(x)
    dxp, lddx = _rows(dx)
    rp, ldr = _rows(dres) if dres is not None else (0, 0)
    M = x.numel() // x.shape[-1]
    lib.call("mtn_layernorm_bwd", dyp, lddy, xp, ldx, mean_rstd.data_ptr(), gamma.data_ptr(), rp, ldr,
             dxp, lddx, partial.data_ptr(), _p(dgamma_dbeta), dtype_code(x, mode), M, _stream(x))


def gelu_bwd(dh, a, da, mode=None):
    assert dh.is_contiguous() and a.is_contiguous() and da.is_contiguous()
    lib.call("mtn_gelu_bwd", dh.data_ptr(), a.data_ptr(), da.data_ptr(), dtype_code(a, mode), a.numel(), _stream(a))


def bias_grad(dy, partial, dbias, mode=None):
    dyp, lddy = _rows(dy)
    M = dy.numel() // dy.shape[-1]
    lib.call("mtn_bias_grad", dyp, lddy, partial.data_ptr(), _p(dbias), dtype_code(dy, mode), M, _stream(dy))


def rowslab_producer_rows(M):
    return lib.call("mtn_rowslab_producer_rows", M)


class ColsumPlan:
    """Deferred parameter-gradient folds (include/mtn.h: mtn_colsum_grouped): the producers were called with a null
    gradient pointer and left their partial sums in their slabs; ``run()`` folds ALL of them with two launches.
    jobs: list of (slab f32 tensor, producer rows, W, ld, out f32 tensor)."""

    def __init__(self, jobs):
        import numpy as np
        rpb, direct = lib.call("mtn_colsum_rows_per_block"), lib.call("mtn_colsum_direct_rows")
        rec = np.dtype([("inp", "<u8"), ("out", "<u8"), ("rows", "<i4"), ("W", "<i4"), ("ld", "<i4"), ("rpb", "<i4")])
        l1, l2 = [], []
        for slab, rows, W, ld, out in jobs:
            # (``out`` may be the first tensor of an adjacent (weight, bias) pair in the flat gradient buffer: W spans both)
            if slab.dtype != torch.float32 or out.dtype != torch.float32:
                raise ValueError("ColsumPlan: f32 slabs / outputs only")
            if rows > direct:
                R = (rows + rpb - 1) // rpb
                if R > direct:
                    raise ValueError(f"ColsumPlan: {rows} rows need more than two fold levels")
                if slab.numel() < rows * ld + R * W:
                    raise ValueError("ColsumPlan: the slab has no scratch tail for its folded rows (allocate slab_rows(...) rows)")
                scratch = slab.data_ptr() + 4 * rows * ld
                l1.append((slab.data_ptr(), scratch, rows, W, ld, rpb))
                l2.append((scratch, out.data_ptr(), R, W, W, R))
            else:
                l2.append((slab.data_ptr(), out.data_ptr(), rows, W, ld, rows))
        dev = jobs[0][0].device
        self.keep = [j[0] for j in jobs] + [j[4] for j in jobs]          # the tables hold raw pointers into these
        mk = lambda recs: torch.from_numpy(np.array(recs, dtype=rec).view(np.uint8).copy()).to(dev) if recs else None
        self.t1, self.t2 = mk(l1), mk(l2)
        self.n1, self.n2 = len(l1), len(l2)
        self.w1 = max((r[3] for r in l1), default=0)
        self.rb1 = max(((r[2] + rpb - 1) // rpb for r in l1), default=0)
        self.w2 = max(r[3] for r in l2)
        self.dev = dev

    def run(self):
        lib.call("mtn_colsum_grouped", _p(self.t1), self.n1, self.w1, self.rb1, self.t2.data_ptr(), self.n2, self.w2,
                 _stream(self.t2))


def attention_fwd(qkv, out, lse, attn, B, N, H, D, scale, mode=None):
    lib.call("mtn_attention_fwd", qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), _p(attn),
             dtype_code(qkv, mode), B, N, H, D, scale, _stream(qkv))


def attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H, D, scale, mode=None):
    lib.call("mtn_attention_bwd", qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
             delta.data_ptr(), dqkv.data_ptr(), dtype_code(qkv, mode), B, N, H, D, scale, _stream(qkv))


def cls_head_fwd(tok, ln_w, ln_b, W, bias, logits, xhat, mean_rstd, eps=1e-5, mode=None):
    B, C = logits.shape
    lib.call("mtn_cls_head_fwd", tok.data_ptr(), tok.stride(0), ln_w.data_ptr(), ln_b.data_ptr(),
             W.data_ptr(), bias.data_ptr(), logits.data_ptr(), xhat.data_ptr(), mean_rstd.data_ptr(),
             dtype_code(tok, mode), B, C, eps, _stream(tok))


def cls_head_bwd(dlogits, W, ln_w, ln_b, xhat, mean_rstd, dtok, dW, dbias, dln_w_b, partial, mode=None):
    B, C = dlogits.shape
    lib.call("mtn_cls_head_bwd", dlogits.data_ptr(), W.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(),
             xhat.data_ptr(), mean_rstd.data_ptr(), dtok.data_ptr(), dtok.stride(0), dW.data_ptr(),
             dbias.data_ptr(), dln_w_b.data_ptr(), partial.data_ptr(), dtype_code(dtok, mode), B, C, _stream(dtok))


def pose_head_fwd(tok, W, bias, hm, h, w, mode=None):
    B, J = hm.shape[:2]
    lib.call("mtn_pose_head_fwd", tok.data_ptr(), W.data_ptr(), bias.data_ptr(), hm.data_ptr(),
             dtype_code(tok, mode), B, h, w, J, _stream(tok))


def pose_slab_floats(B, h, w, J):
    return lib.call("mtn_pose_slab_floats", B, h, w, J)


def pose_head_bwd(tok, W, dhm, dtok, dW_dbias, slab, h, w, mode=None):
    """dW_dbias: f32 [J*256 + J] = weight gradient followed by the bias gradient; slab: f32 [pose_slab_floats(B, h, w, J)]."""
    B, J = dhm.shape[:2]
    lib.call("mtn_pose_head_bwd", tok.data_ptr(), W.data_ptr(), dhm.data_ptr(), dtok.data_ptr(),
             dW_dbias.data_ptr(), slab.data_ptr(), dtype_code(tok, mode), B, h, w, J, _stream(tok))


def loss_partial_len(n):
    return lib.call("mtn_loss_partial_len", n)


def _check(t, name, dtype, shape):
    """The C ABI takes raw pointers: a wrong dtype / layout is an out-of-bounds device access,
    so it is refused here (TypeError / ValueError, not an assert that -O strips)."""
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")


def loss_fwd_bwd(logits, label, hm, target, target_weight, dlogits, dhm, partial, losses, ce_weight=0.001):
    """losses[3] = (0.001*CE + JMSE, 0.001*CE, JMSE); fills dlogits / dhm when given.
    label int64 [B] (a value outside [0, C) is ignored like torch's ignore_index), target f32 of
    hm's shape, target_weight f32 with B*J elements ([B,J,1] or [B,J])."""
    B, C = logits.shape
    J = hm.shape[1]
    HW = hm.shape[2] * hm.shape[3]
    _check(logits, "logits", torch.float32, (B, C))
    _check(hm, "heatmaps", torch.float32, (B, J, hm.shape[2], hm.shape[3]))
    _check(label, "label", torch.int64, (B,))
    _check(target, "target", torch.float32, hm.shape)
    if target_weight.numel() != B * J:
        raise ValueError(f"target_weight: expected {B * J} elements ([B,J,1]), got shape {tuple(target_weight.shape)}")
    _check(target_weight, "target_weight", torch.float32, target_weight.shape)
    for name, t, ref in (("dlogits", dlogits, logits), ("dhm", dhm, hm)):
        if t is not None:
            _check(t, name, torch.float32, ref.shape)
    if partial.dtype != torch.float32 or partial.numel() < loss_partial_len(hm.numel()) or losses.numel() < 3:
        raise ValueError("loss workspace too small")
    devs = {t.device for t in (logits, label, hm, target, target_weight, partial, losses)}
    if len(devs) != 1:
        raise ValueError(f"loss operands live on different devices: {sorted(map(str, devs))}")
    lib.call("mtn_loss_fwd_bwd", logits.data_ptr(), label.data_ptr(), hm.data_ptr(), target.data_ptr(),
             target_weight.data_ptr(), _p(dlogits), _p(dhm), partial.data_ptr(), losses.data_ptr(),
             B, C, J, HW, ce_weight, _stream(hm))


def adamw_step(p, g, m, v, state):
    lib.call("mtn_adamw_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
             state.data_ptr(), _stream(p))


def x3_convert(src, dst, to_split):
    """split storage <-> plain f32 (include/mtn.h mtn_x3_convert): src / dst f32 [..., C] row tensors of one shape; the split
    side holds, per 32-channel block, [32 bf16 hi][32 bf16 lo]."""
    if src.dtype != torch.float32 or dst.dtype != torch.float32 or src.shape != dst.shape:
        raise ValueError("x3_convert: two f32 tensors of one shape")
    C = src.shape[-1]
    sp, lds = _rows(src)
    dp, ldd = _rows(dst)
    lib.call("mtn_x3_convert", sp, lds, dp, ldd, 1 if to_split else 0, src.numel() // C, C, _stream(src))
    return dst


def x3_split(t, out=None):
    """plain f32 -> split storage (new tensor unless ``out``)"""
    return x3_convert(t, torch.empty_like(t) if out is None else out, True)


def x3_join(t, out=None):
    """split storage -> plain f32"""
    return x3_convert(t, torch.empty_like(t) if out is None else out, False)


def pack_weights(src, dst, desc, ntensors, max_elems, scale=None, mode=None):
    """desc int32 [ntensors][7] (include/mtn.h); ``scale``: f32 buffer the rows' scale_off index into."""
    if desc.dtype != torch.int32 or desc.numel() != ntensors * 7:
        raise ValueError("pack_weights: desc must be int32 [ntensors][7]")
    lib.call("mtn_pack_weights", src.data_ptr(), _p(scale), dst.data_ptr(), desc.data_ptr(), ntensors, max_elems,
             dtype_code(dst, mode), _stream(dst))


def _moded_names():
    import inspect
    import sys
    return frozenset(name for name, fn in vars(sys.modules[__name__]).items()
                     if inspect.isfunction(fn) and "mode" in inspect.signature(fn).parameters)


_MODED = _moded_names()      # the wrappers that take ``mode`` (fixed at import: a later stand-in for one of them still gets it)
