"""igemm conv forward / dgrad / linear against torch fp32 ops (kernel-level parity)."""
import pytest
import torch
import torch.nn.functional as F

from backend import GEMM_DTYPES as DTYPES, A, Fl, be, presplit_x3, rel_err, tol, x3_active  # noqa: F401
from backend import dbg  # noqa: E402
from hand_gesture_recognition_amd import ops


def _rand(shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).to(dtype)


def _wfmt(w, dtype):
    """kernel weight operand: T-typed, and under the x3 parametrisation in split storage like the activations"""
    w = w.contiguous().to(dtype)
    return presplit_x3(w.reshape(w.shape[0], -1)).reshape(w.shape) if (dtype == torch.float32 and x3_active()) else w


def pack_krsc(w_oihw, dtype):       # [O,I,kh,kw] -> [O,kh,kw,I]
    return _wfmt(w_oihw.permute(0, 2, 3, 1), dtype)


def pack_krsc_plain(w_oihw, dtype):   # the stem's operand: plain T in every mode (its kernels multiply in exact f32)
    return w_oihw.permute(0, 2, 3, 1).contiguous().to(dtype)


def pack_crsk(w_oihw, dtype):       # dgrad operand [I,kh,kw,O]
    return _wfmt(w_oihw.permute(1, 2, 3, 0), dtype)


@pytest.fixture(params=[1, 2, 3, 4, 5], ids=["lds1", "lds2", "dma2", "dma3", "win"])
def stages(request, be):
    """all igemm pipelines: register-staged (1 or 2 LDS buffers), LDS-DMA (2 or 3 buffers) and the
    window-resident mode of the stride-1 3x3 layers (other layers fall back to dma2)"""
    from hand_gesture_recognition_amd import lib
    dbg(3, request.param)
    dbg(9, 150)       # every stride-1 3x3 shape goes through the window path
    dbg(15, 0)        # stride-2 data gradients through igemm's parity-major rows (dgrad2 has its own test)
    yield request.param
    dbg(3, 5)
    dbg(9, 0)
    dbg(15, 1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    # B, H, W, Cin, Cout, k, s
    (2, 6, 6, 64, 64, 3, 1),
    (1, 8, 8, 64, 128, 3, 2),
    (3, 5, 7, 128, 64, 1, 1),
    (1, 12, 12, 64, 256, 3, 1),
    (3, 5, 7, 128, 128, 3, 1),
    (2, 9, 6, 256, 64, 3, 1),
])
def test_conv_fwd_and_stats(be, stages, dtype, cfg):
    B, H, W, Cin, Cout, k, s = cfg
    x = _rand((B, H, W, Cin), dtype, 1)
    w = _rand((Cout, Cin, k, k), torch.float32, 2) / (Cin * k * k) ** 0.5
    wq = w.to(dtype).float()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wq, None, s, k // 2).permute(0, 2, 3, 1)
    OH, OW = ref.shape[1:3]
    M = B * OH * OW
    xd, wd = A(x).to(be.device), pack_krsc(w, dtype).to(be.device)
    y = torch.full((B, OH, OW, Cout), float("nan"), dtype=dtype, device=be.device)
    tiles = ops.conv_stats_tiles(M, Cout, k * k * Cin)
    stats = torch.full((tiles, 2, Cout), float("nan"), dtype=torch.float32, device=be.device)
    ops.conv_fwd(xd, wd, y, stats, k, s)
    be.sync()
    assert rel_err(Fl(y), ref) < tol(dtype)
    yf = Fl(y).reshape(M, Cout)
    st = stats.cpu().sum(0)
    assert rel_err(st[0], yf.sum(0)) < 1e-4 or float((st[0] - yf.sum(0)).abs().max()) < 1e-3
    assert rel_err(st[1], (yf * yf).sum(0)) < 1e-4


@pytest.fixture
def image_tiles(be):
    """force the image tiles (whole reduction axis resident in LDS) on shapes the CPU executor can run"""
    from hand_gesture_recognition_amd import lib
    dbg(16, 3)
    yield
    dbg(16, 1)


@pytest.mark.parametrize("cfg", [
    # B, H, W, Cin, Cout   (3x3 stride 1, bf16 only: Cin / 64 in {1, 2, 4} resident slabs)
    (2, 12, 12, 64, 256),        # two tiles = two images
    (3, 5, 7, 128, 256),         # one partial tile, three images inside it
    (1, 13, 12, 256, 512),       # four resident slabs, two column tiles, a ragged second row tile
])
def test_conv_image_tiles(be, image_tiles, cfg):
    """igemm_kernel<144, 256, 1, 8, window>: forward with statistics, data gradient with residual and
    BatchNorm-backward sums, against torch."""
    dtype = torch.bfloat16
    B, H, W, Cin, Cout = cfg
    d = be.device
    x = _rand((B, H, W, Cin), dtype, 71)
    w = _rand((Cout, Cin, 3, 3), torch.float32, 72) / (Cin * 9) ** 0.5
    wq = w.to(dtype).float()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wq, None, 1, 1).permute(0, 2, 3, 1)
    M = B * H * W
    tiles = ops.conv_fwd_stats_tiles(dtype, B, H, W, Cin, Cout, 3, 1)
    assert tiles == (M + 143) // 144
    y = torch.full((B, H, W, Cout), float("nan"), dtype=dtype, device=d)
    stats = torch.full((tiles, 2, Cout), float("nan"), device=d)
    ops.conv_fwd(A(x).to(d), pack_krsc(w, dtype).to(d), y, stats, 3, 1)
    be.sync()
    assert rel_err(Fl(y), ref) < tol(dtype)
    yf = Fl(y).reshape(M, Cout)
    assert rel_err(stats.cpu().sum(0)[0], yf.sum(0)) < 1e-3 and rel_err(stats.cpu().sum(0)[1], (yf * yf).sum(0)) < 1e-3
    # data gradient of a Cout -> Cin... here the transposed roles: dy has Cout2 = Cin channels in, dx 256-wide out
    if Cout == 256:
        w2 = _rand((Cin, Cout, 3, 3), torch.float32, 73) / (Cin * 9) ** 0.5         # conv Cout -> Cin; its dgrad: N = Cout, K = 9 Cin
        xz = torch.zeros((B, Cout, H, W), requires_grad=True)
        yr = F.conv2d(xz, w2.to(dtype).float(), None, 1, 1)
        dy = _rand((B, H, W, Cin), dtype, 74)
        res = _rand((B, H, W, Cout), dtype, 75)
        yr.backward(dy.float().permute(0, 3, 1, 2))
        dref = xz.grad.permute(0, 2, 3, 1) + res.float()
        dx = res.clone().to(d)
        ops.conv_dgrad(A(dy).to(d), pack_crsk(w2, dtype).to(d), dx, dx, 3, 1)
        be.sync()
        assert rel_err(Fl(dx), dref) < tol(dtype)


@pytest.mark.x3
@pytest.mark.parametrize("cfg", [
    # B, H, W, Cin, Cout   (split storage: 32-channel blocks; Cin = 256 = eight blocks runs as two passes of four)
    (2, 12, 12, 64, 256),        # two resident blocks, one pass
    (1, 13, 12, 128, 256),       # four blocks, one pass, a ragged second row tile
    (2, 12, 12, 256, 256),       # eight blocks: two passes over the window
    (1, 5, 7, 256, 512),         # two passes, one partial tile, two column tiles
])
def test_conv_image_tiles_split_storage(be, image_tiles, cfg):
    """igemm_kernel<f32x3_t, 144, 256, 1, 8, window>: forward with statistics and data gradient with a residual, against torch"""
    dtype = torch.float32
    B, H, W, Cin, Cout = cfg
    d = be.device
    x = _rand((B, H, W, Cin), dtype, 71)
    w = _rand((Cout, Cin, 3, 3), torch.float32, 72) / (Cin * 9) ** 0.5
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, None, 1, 1).permute(0, 2, 3, 1)
    M = B * H * W
    tiles = ops.conv_fwd_stats_tiles(dtype, B, H, W, Cin, Cout, 3, 1)
    assert tiles == (M + 143) // 144
    y = torch.full((B, H, W, Cout), float("nan"), dtype=dtype, device=d)
    stats = torch.full((tiles, 2, Cout), float("nan"), device=d)
    ops.conv_fwd(A(x).to(d), pack_krsc(w, dtype).to(d), y, stats, 3, 1)
    be.sync()
    assert rel_err(Fl(y), ref) < 5e-5
    yf = Fl(y).reshape(M, Cout)
    assert rel_err(stats.cpu().sum(0)[0], yf.sum(0)) < 1e-3 and rel_err(stats.cpu().sum(0)[1], (yf * yf).sum(0)) < 1e-3
    if Cout == 256:
        w2 = _rand((Cin, Cout, 3, 3), torch.float32, 73) / (Cin * 9) ** 0.5
        xz = torch.zeros((B, Cout, H, W), requires_grad=True)
        yr = F.conv2d(xz, w2, None, 1, 1)
        dy = _rand((B, H, W, Cin), dtype, 74)
        res = _rand((B, H, W, Cout), dtype, 75)
        yr.backward(dy.permute(0, 3, 1, 2))
        dref = xz.grad.permute(0, 2, 3, 1) + res
        dx = A(res).clone().to(d)
        ops.conv_dgrad(A(dy).to(d), pack_crsk(w2, dtype).to(d), dx, dx, 3, 1)
        be.sync()
        assert rel_err(Fl(dx), dref) < 5e-5


def test_conv_fwd_refuses_a_stats_slab_sized_for_another_tile(be):
    """A debug key that changes the tile after the plan sized its slab must fail the launch, not
    write a different number of statistics rows than bn_finalize will read."""
    from hand_gesture_recognition_amd import lib
    B, H, W, Cin, Cout, k, s = 2, 12, 12, 64, 128, 3, 1
    x = _rand((B, H, W, Cin), torch.float32, 5).to(be.device)
    w = pack_krsc(_rand((Cout, Cin, k, k), torch.float32, 6), torch.float32).to(be.device)
    y = torch.empty((B, H, W, Cout), device=be.device)
    M = B * H * W
    tiles = ops.conv_stats_tiles(M, Cout, k * k * Cin)              # default tile: 64 rows at this M
    stats = torch.zeros((max(tiles, (M + 63) // 64), 2, Cout), device=be.device)
    dbg(6, 1)                                  # force 128-row tiles
    try:
        assert ops.conv_stats_tiles(M, Cout, k * k * Cin) != tiles
        with pytest.raises(RuntimeError, match="statistics slab"):
            ops.conv_fwd(x, w, y, stats, k, s, stats_tiles=tiles)
    finally:
        dbg(6, 0)
    ops.conv_fwd(x, w, y, stats, k, s, stats_tiles=tiles)
    be.sync()


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_fwd_channel_slices(be, dtype):
    """reads a channel slice of a wider buffer and writes into a slice of a concat buffer."""
    B, H, W = 1, 4, 4
    wide = _rand((B, H, W, 128), dtype, 3)
    w = _rand((64, 64, 3, 3), torch.float32, 4) / 24.0
    ref = F.conv2d(wide[..., 64:].float().permute(0, 3, 1, 2), w.to(dtype).float(), None, 1, 1).permute(0, 2, 3, 1)
    cat = torch.zeros((B, H, W, 256), dtype=dtype, device=be.device)
    wided = A(wide).to(be.device)
    ops.conv_fwd(wided[..., 64:], pack_krsc(w, dtype).to(be.device), cat[..., 128:192], None, 3, 1)
    be.sync()
    assert rel_err(Fl(cat[..., 128:192]), ref) < tol(dtype)
    assert float(cat[..., :128].float().abs().max()) == 0 and float(cat[..., 192:].float().abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 12, 12, 64, 128, 3, 1), (1, 16, 16, 128, 256, 1, 1), (2, 12, 12, 64, 128, 3, 2),
                                 (2, 12, 12, 128, 128, 3, 1), (3, 5, 7, 128, 256, 3, 1)])      # 128 channels, 3x3 stride 1: the tile's window form (bf16)
def test_conv_fwd_forced_128x128_tile(be, dtype, cfg):
    """the 128x128 tile (8 waves) is only chosen for M >= 8192: force it on shapes the CPU executor can run"""
    from hand_gesture_recognition_amd import lib
    B, H, W, Cin, Cout, k, s = cfg
    x = _rand((B, H, W, Cin), dtype, 31)
    w = _rand((Cout, Cin, k, k), torch.float32, 32) / (Cin * k * k) ** 0.5
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(dtype).float(), None, s, k // 2).permute(0, 2, 3, 1)
    OH, OW = ref.shape[1:3]
    M = B * OH * OW
    win128 = (dtype == torch.bfloat16 or x3_active()) and k == 3 and s == 1 and Cin == 128        # (round 5: this tile's window form)
    dx = dref = None
    dbg(6, 1)
    try:
        assert (ops.conv_kernel_family(dtype, False, B, H, W, Cin, Cout, k, s) == 1) == win128
        y = torch.full((B, OH, OW, Cout), float("nan"), dtype=dtype, device=be.device)
        stats = torch.full((ops.conv_stats_tiles(M, Cout, k * k * Cin), 2, Cout), float("nan"), device=be.device)
        ops.conv_fwd(A(x).to(be.device), pack_krsc(w, dtype).to(be.device), y, stats, k, s)
        be.sync()
        if k == 3 and s == 1 and Cout == 128:
            # the data gradient of a 128 -> Cin layer through the same tile (gather source = dy with 128 channels)
            w2 = _rand((Cout, Cin, 3, 3), torch.float32, 33) / (Cin * 9) ** 0.5      # conv Cin -> Cout; dgrad: N = Cin, K = 9 Cout
            xz = torch.zeros((B, Cin, H, W), requires_grad=True)
            dyv = _rand((B, H, W, Cout), dtype, 34)
            F.conv2d(xz, w2.to(dtype).float(), None, 1, 1).backward(dyv.float().permute(0, 3, 1, 2))
            dref = xz.grad.permute(0, 2, 3, 1)
            dx = torch.full((B, H, W, Cin), float("nan"), dtype=dtype, device=be.device)
            ops.conv_dgrad(A(dyv).to(be.device), pack_crsk(w2, dtype).to(be.device), dx, None, 3, 1)
            be.sync()
    finally:
        dbg(6, 0)
    assert stats.shape[0] == (M + 127) // 128
    assert rel_err(Fl(y), ref) < tol(dtype)
    yf = Fl(y).reshape(M, Cout)
    assert rel_err(stats.cpu().sum(0)[0], yf.sum(0)) < 1e-3
    if dx is not None:
        assert rel_err(Fl(dx), dref) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    (2, 6, 6, 64, 64, 3, 1),
    (1, 8, 8, 64, 128, 3, 2),
    (1, 10, 6, 64, 64, 3, 2),
    (2, 4, 4, 128, 64, 1, 1),
    (3, 5, 7, 128, 128, 3, 1),
    (2, 9, 6, 64, 256, 3, 1),
    (2, 16, 16, 64, 64, 3, 2),        # 4 x 128 rows: the parity classes' tiles are interleaved
])
def test_conv_dgrad(be, stages, dtype, cfg):
    B, H, W, Cin, Cout, k, s = cfg
    w = _rand((Cout, Cin, k, k), torch.float32, 5) / (Cout * k * k) ** 0.5
    wq = w.to(dtype).float()
    x = torch.zeros((B, Cin, H, W), requires_grad=True)
    yref = F.conv2d(x, wq, None, s, k // 2)
    dy = _rand(tuple(yref.permute(0, 2, 3, 1).shape), dtype, 6)
    res = _rand((B, H, W, Cin), dtype, 7)
    yref.backward(dy.float().permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1) + res.float()
    dx = A(res).clone().to(be.device)            # residual aliases the output
    ops.conv_dgrad(A(dy).to(be.device), pack_crsk(w, dtype).to(be.device), dx, dx, k, s)
    be.sync()
    assert rel_err(Fl(dx), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("with_res", [False, True])
@pytest.mark.parametrize("cfg", [
    # B, H, W, Cin, Cout        (3x3 stride 2; dgrad2 needs Cout % 128 == 0 in bf16, % 64 in fp32)
    (1, 8, 8, 64, 128),          # one tile, one channel slab
    (1, 10, 6, 64, 128),         # odd dy height / width (5 x 3)
    (3, 12, 20, 128, 256),       # two row tiles (180 dy pixels), two column tiles, two channel slabs (bf16) / four (fp32)
    (2, 2, 2, 64, 128),          # 1 x 1 dy: every neighbour is padding
    (5, 16, 16, 64, 128),        # rows cross image borders inside a tile
    (1, 8, 192, 64, 128),        # 96-column dy (384x384 images): 92 KB window, one workgroup per CU
])
def test_conv_dgrad_stride2_fused(be, dtype, with_res, cfg):
    """dgrad2.hip (all four output parities per workgroup) against autograd, writing into a channel
    slice of a wider gradient buffer and reading dy from one."""
    B, H, W, Cin, Cout = cfg
    w = _rand((Cout, Cin, 3, 3), torch.float32, 15) / (Cout * 9) ** 0.5
    wq = w.to(dtype).float()
    x = torch.zeros((B, Cin, H, W), requires_grad=True)
    yref = F.conv2d(x, wq, None, 2, 1)
    dy_wide = _rand(tuple(yref.permute(0, 2, 3, 1).shape[:3]) + (2 * Cout,), dtype, 16)
    dy = dy_wide[..., Cout:]
    res = _rand((B, H, W, Cin), dtype, 17)
    yref.backward(dy.float().permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1) + (res.float() if with_res else 0)
    wide = torch.full((B, H, W, 2 * Cin), 7.0, dtype=dtype, device=be.device)
    dx = wide[..., :Cin]
    if with_res:
        dx.copy_(A(res).to(be.device))
    ops.conv_dgrad(A(dy_wide).to(be.device)[..., Cout:], pack_crsk(w, dtype).to(be.device), dx, dx if with_res else None, 3, 2)
    be.sync()
    assert rel_err(Fl(dx), ref) < tol(dtype)
    assert float((wide[..., Cin:].float() - 7.0).abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_bias_gelu_residual(be, stages, dtype):
    M, K, N = 70, 256, 128
    x = _rand((M, K), dtype, 8)
    w = _rand((N, K), torch.float32, 9) / 16.0
    b = _rand((N,), torch.float32, 10)
    r = _rand((M, N), dtype, 11)
    pre = F.linear(x.float(), w.to(dtype).float(), b)
    ref = F.gelu(pre) + r.float()
    out = torch.empty((M, N), dtype=dtype, device=be.device)
    aux = torch.empty((M, N), dtype=dtype, device=be.device)
    ops.linear_fwd(A(x).to(be.device), _wfmt(w, dtype).to(be.device), out, bias=b.to(be.device),
                   act=ops.ACT_GELU, aux_out=aux, res=A(r).to(be.device))
    be.sync()
    assert rel_err(Fl(aux), pre) < tol(dtype)
    assert rel_err(Fl(out), ref) < tol(dtype)


# tr16: 0 = scalar column reads (bf16 only: the switch does nothing for 4-byte elements), 1 = ds_read_tr16 (register staged), 2 = LDS-DMA (default)
@pytest.mark.parametrize("dtype,tr16", [pytest.param(d.values[0], t, id=f"{t}-{d.id}", marks=d.marks) for d in DTYPES for t in (0, 1, 2)
                                        if not (d.values[0] == torch.float32 and t == 0)])
@pytest.mark.parametrize("cfg", [
    (2, 6, 6, 64, 64, 3, 1),
    (1, 8, 8, 64, 128, 3, 2),
    (3, 5, 7, 128, 128, 1, 1),
    (5, 4, 4, 128, 128, 3, 1),
    (1, 20, 30, 64, 64, 1, 1),       # >= 8 pixel splits: exercises the XCD-group split order
])
def test_conv_wgrad(be, dtype, tr16, cfg):
    from hand_gesture_recognition_amd import lib
    dbg(5, 1 if tr16 == 2 else 0)
    B, H, W, Cin, Cout, k, s = cfg
    x = _rand((B, H, W, Cin), dtype, 12)
    w = torch.zeros((Cout, Cin, k, k), requires_grad=True)
    yref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, s, k // 2)
    dy = _rand(tuple(yref.permute(0, 2, 3, 1).shape), dtype, 13)
    yref.backward(dy.float().permute(0, 3, 1, 2))
    dbg(1, min(tr16, 1))
    try:
        ws = torch.empty(ops.conv_wgrad_workspace_bytes(dtype, B, H, W, Cin, Cout, k, s) // 4,
                         dtype=torch.float32, device=be.device)
        dw = torch.full((Cout, Cin, k, k), float("nan"), device=be.device)
        ops.conv_wgrad(A(x).to(be.device), A(dy).to(be.device), dw, ws, k, s)
        be.sync()
    finally:
        dbg(1, 1)
        dbg(5, 1)
    assert rel_err(dw, w.grad) < tol(dtype, 1e-4, 1e-2)


@pytest.mark.parametrize("cfg", [(2, 6, 6, 64, 64, 3, 1), (3, 5, 7, 128, 128, 1, 1), (1, 8, 8, 64, 128, 3, 2), (1, 20, 30, 64, 64, 1, 1)])
def test_conv_wgrad_bf16_slab(be, cfg):
    """The split-K partial tiles of the bf16 mode are stored as bf16 and summed in f32 (debug key 28 = 0: f32 tiles): within the
    mode's own tolerance of torch, and within bf16 rounding of the f32-slab result."""
    B, H, W, Cin, Cout, k, s = cfg
    dtype = torch.bfloat16
    x = _rand((B, H, W, Cin), dtype, 12)
    w = torch.zeros((Cout, Cin, k, k), requires_grad=True)
    yref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, s, k // 2)
    dy = _rand(tuple(yref.permute(0, 2, 3, 1).shape), dtype, 13)
    yref.backward(dy.float().permute(0, 3, 1, 2))
    got = []
    for key in (0, 1):
        dbg(28, key)
        try:
            ws = torch.empty(ops.conv_wgrad_workspace_bytes(dtype, B, H, W, Cin, Cout, k, s) // 4, device=be.device)
            dw = torch.full((Cout, Cin, k, k), float("nan"), device=be.device)
            ops.conv_wgrad(x.to(be.device), dy.to(be.device), dw, ws, k, s)
            be.sync()
        finally:
            dbg(28, 1)
        assert rel_err(dw, w.grad) < 1e-2
        got.append(dw.cpu())
    assert rel_err(got[1], got[0]) < 4e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    # B, H, W, Cout, stride, Cin   (3x3)
    (1, 8, 8, 128, 2, 64),               # conv2's shape class: three kernel-row tiles
    (3, 10, 6, 256, 2, 64),              # odd output map (5 x 3), two column tiles
    (2, 16, 16, 128, 2, 64),             # several pixel steps per workgroup
    (20, 12, 12, 128, 2, 64),            # >= 8 pixel splits: the XCD-group split order
    (2, 12, 12, 256, 2, 128),            # round 5: the wider stride-2 layers (down1 / down2) as 64-channel slabs of kernel-row tiles
    (1, 6, 10, 128, 2, 256),
])
def test_conv_wgrad_kernel_row_tiles(be, dtype, cfg):
    """wgrad_row3_kernel (the three taps of one kernel row per workgroup) against torch and, bit for
    bit in the split-free case..., against the per-tap tiles it replaces."""
    from hand_gesture_recognition_amd import lib
    B, H, W, Cout, s, Cin = cfg
    k = 3
    x = _rand((B, H, W, Cin), dtype, 21)
    w = torch.zeros((Cout, Cin, k, k), requires_grad=True)
    yref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, s, 1)
    dy = _rand(tuple(yref.permute(0, 2, 3, 1).shape), dtype, 22)
    yref.backward(dy.float().permute(0, 3, 1, 2))
    outs = []
    for row3 in (2, 0):
        dbg(18, row3)
        dbg(28, 0)            # (two kernels are compared to 1e-5: f32 partial tiles for both)
        try:
            assert ops.conv_wgrad_workspace_bytes(dtype, B, H, W, Cin, Cout, k, s) > 0
            ws = torch.empty(ops.conv_wgrad_workspace_bytes(dtype, B, H, W, Cin, Cout, k, s) // 4, dtype=torch.float32, device=be.device)
            dw = torch.full((Cout, Cin, k, k), float("nan"), device=be.device)
            ops.conv_wgrad(A(x).to(be.device), A(dy).to(be.device), dw, ws, k, s)
            be.sync()
        finally:
            dbg(18, 2)
            dbg(28, 1)
        assert rel_err(dw, w.grad) < tol(dtype, 1e-4, 1e-2)
        outs.append(dw.clone())
    assert rel_err(outs[0], outs[1]) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    (3, 12, 20, 64, 64, 0),          # non-square, one tile
    (2, 7, 5, 128, 64, 0),           # odd sizes, two input-channel tiles
    (3, 12, 20, 64, 128, 2),         # two workgroups per tile only: several window steps per workgroup
    (1, 24, 24, 64, 64, 1),          # one workgroup walks the whole (padded) pixel axis
])
def test_conv_wgrad_window_kernel(be, dtype, cfg):
    """3x3 / stride 1 through the nine-taps-from-one-window kernel (padded slot coordinates): against
    torch, and against the per-tap tiles it replaces."""
    from hand_gesture_recognition_amd import lib
    B, H, W, Cin, Cout, slots = cfg
    x = _rand((B, H, W, Cin), dtype, 61)
    w = torch.zeros((Cout, Cin, 3, 3), requires_grad=True)
    yref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, 1, 1)
    dy = _rand((B, H, W, Cout), dtype, 62)
    yref.backward(dy.float().permute(0, 3, 1, 2))
    res = []
    for win in (1, 0):
        dbg(13, win)
        dbg(10, slots if win else 0)
        dbg(28, 0)            # (two kernels are compared to 1e-5: f32 partial tiles for both)
        try:
            ws = torch.empty(ops.conv_wgrad_workspace_bytes(dtype, B, H, W, Cin, Cout, 3, 1) // 4, device=be.device)
            dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=be.device)
            ops.conv_wgrad(A(x).to(be.device), A(dy).to(be.device), dw, ws, 3, 1)
            be.sync()
            res.append(dw.cpu())
        finally:
            dbg(13, 1)
            dbg(10, 0)
            dbg(28, 1)
    assert rel_err(res[0], w.grad) < tol(dtype, 1e-4, 1e-2)
    assert rel_err(res[0], res[1]) < tol(dtype, 1e-5, 1e-5)       # same products, f32 sums in another order


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_wgrad_large_m(be, dtype):
    M, K, N = 300, 256, 64
    x, dy = _rand((M, K), dtype, 14), _rand((M, N), dtype, 15)
    ref = dy.float().t() @ x.float()
    ws = torch.empty(ops.linear_wgrad_workspace_bytes(dtype, M, N, K) // 4, device=be.device)
    dw = torch.empty((N, K), device=be.device)
    ops.linear_wgrad(A(x).to(be.device), A(dy).to(be.device), dw, ws)
    be.sync()
    assert rel_err(dw, ref) < tol(dtype, 1e-4, 1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    (300, [(768, 256), (256, 256), (512, 256), (256, 512)], 0),          # the four Linears of a Transformer layer, ragged row count
    (70, [(128, 128), (256, 128), (128, 384)], 0),                       # fewer rows than one split step would like
    (1100, [(256, 256), (128, 256)] * 3, 0),                             # several splits per tile
    (1100, [(256, 256), (128, 256)] * 3, 24),                            # a slot budget below the tile count: one split, several rounds
    (300, [(256, 256), (64, 256), (256, 128)], 0),                       # a 64-wide side: the library runs the problems one by one
    (200, [(128, 128)] * 18, 0),                                         # more than 16 problems: two launches
])
def test_linear_wgrad_group(be, dtype, cfg):
    """mtn_linear_wgrad_group: several Linear weight gradients over the same rows in one launch + one split sum -- against
    torch, and against the one-launch-per-problem path (debug key 30 = 0); operands are slices of wider buffers (row pitch >
    row length) as the plan's qkv / hidden buffers are."""
    M, shapes, slots = cfg
    d = be.device
    probs, refs = [], []
    for i, (N, K) in enumerate(shapes):
        xw, dyw = _rand((M, K + 128), dtype, 100 + i), _rand((M, N + 64), dtype, 200 + i)
        x, dy = A(xw).to(d)[:, 64:64 + K], A(dyw).to(d)[:, 32:32 + N]                 # (multiples of 32 channels: valid split-storage slices)
        refs.append(dyw[:, 32:32 + N].float().t() @ xw[:, 64:64 + K].float())
        probs.append((x, dy, torch.full((N, K), float("nan"), device=d)))
    got = []
    for key in (1, 0):
        dbg(30, key)
        dbg(10, slots)
        dbg(28, 0)               # (f32 partial tiles: the two paths then differ by summation order only)
        try:
            for _, _, dw in probs:
                dw.fill_(float("nan"))
            items, m = ops.wgrad_group_items(probs)
            assert m == M
            nbytes = ops.linear_wgrad_group_workspace_bytes(dtype, items, M)
            assert nbytes > 0
            ws = torch.empty(nbytes // 4 + 1, device=d)
            ops.linear_wgrad_group(items, M, ws, dtype)
            be.sync()
            got.append([dw.cpu().clone() for _, _, dw in probs])
        finally:
            dbg(30, 1)
            dbg(10, 0)
            dbg(28, 1)
    for a, b, ref in zip(got[0], got[1], refs):
        assert rel_err(a, ref) < tol(dtype, 1e-4, 1e-2)
        assert rel_err(a, b) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    # B, H, W, Cin, Cout, k, problems, slot budget (debug key 10; 0 = default)
    (2, 12, 12, 64, 64, 3, 4, 0),            # a stage's four 3x3 layers: one tile each
    (1, 24, 24, 128, 64, 3, 3, 0),           # two input-channel tiles
    (3, 5, 7, 64, 128, 3, 2, 0),             # odd map, two output tiles
    (2, 12, 20, 64, 64, 3, 4, 16),           # four splits per problem: several window steps per workgroup
    (2, 6, 6, 64, 64, 1, 3, 0),              # a geometry the window kernel does not serve: one launch per problem inside the call
])
def test_conv_wgrad_group(be, dtype, cfg):
    """mtn_conv_wgrad_group: the weight gradients of several convolutions of one geometry in one launch + one split sum -- against
    torch, and against one launch per problem (debug key 30 = 0).  The inputs are channel slices of a wider buffer, as the
    concat slices the model's residual blocks read are."""
    B, H, W, Cin, Cout, k, count, slots = cfg
    d = be.device
    probs, refs = [], []
    for i in range(count):
        xw = _rand((B, H, W, Cin + 64), dtype, 300 + i)
        dy = _rand((B, H, W, Cout), dtype, 400 + i)
        w = torch.zeros((Cout, Cin, k, k), requires_grad=True)
        F.conv2d(xw[..., 32:32 + Cin].float().permute(0, 3, 1, 2), w, None, 1, k // 2).backward(dy.float().permute(0, 3, 1, 2))
        refs.append(w.grad)
        probs.append((A(xw).to(d)[..., 32:32 + Cin], A(dy).to(d), torch.full((Cout, Cin, k, k), float("nan"), device=d)))
    got = []
    for key in (1, 0):
        dbg(30, key)
        dbg(10, slots)
        dbg(28, 0)               # (f32 partial tiles: the two paths then differ by summation order only)
        try:
            for _, _, dw in probs:
                dw.fill_(float("nan"))
            items = ops.conv_wgrad_group_items(probs)
            nbytes = ops.conv_wgrad_group_workspace_bytes(dtype, count, B, H, W, Cin, Cout, k, 1)
            assert nbytes > 0
            ws = torch.empty(nbytes // 4 + 1, device=d)
            ops.conv_wgrad_group(items, ws, dtype, B, H, W, Cin, Cout, k, 1)
            be.sync()
            got.append([dw.cpu().clone() for _, _, dw in probs])
        finally:
            dbg(30, 1)
            dbg(10, 0)
            dbg(28, 1)
    for a, b, ref in zip(got[0], got[1], refs):
        assert rel_err(a, ref) < tol(dtype, 1e-4, 1e-2)
        assert rel_err(a, b) < 1e-5


def test_linear_wgrad_rows_beyond_2_16(be):
    """BASELINE config 4 (384x384: 577 tokens x B=128 = 73,856 rows): the row count squared exceeds
    2^32, which the pixel walk of the convolution path cannot index -- Linear must not go through it."""
    if be.name == "emu":
        pytest.skip("73,856-row operand: GPU only")
    M, K, N = 73856, 256, 64
    x, dy = _rand((M, K), torch.bfloat16, 16), _rand((M, N), torch.bfloat16, 17)
    ref = dy.float().t() @ x.float()
    ws = torch.empty(ops.linear_wgrad_workspace_bytes(torch.bfloat16, M, N, K) // 4, device=be.device)
    dw = torch.empty((N, K), device=be.device)
    ops.linear_wgrad(A(x).to(be.device), A(dy).to(be.device), dw, ws)
    be.sync()
    assert rel_err(dw, ref) < 1e-2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 16, 16), (3, 12, 20), (5, 32, 32)])
def test_conv1_fwd_wgrad(be, dtype, cfg):
    B, H, W = cfg
    x = _rand((B, 3, H, W), torch.float32, 20)
    w = torch.nn.Parameter(_rand((64, 3, 3, 3), torch.float32, 21) / 27 ** 0.5)
    wq = w.detach().to(dtype).float().requires_grad_(True)
    ref = F.conv2d(x, wq, None, 2, 1)
    dy = _rand((B, H // 2, W // 2, 64), dtype, 22)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    d = be.device
    y = torch.empty((B, H // 2, W // 2, 64), dtype=dtype, device=d)
    tiles = ops.conv1_stats_tiles(B, H, W)
    stats = torch.full((ops.slab_rows(tiles), 2, 64), float("nan"), device=d)
    xd = x.to(d)
    ops.conv1_fwd(xd, pack_krsc_plain(w.detach(), dtype).to(d), y, stats)
    be.sync()
    assert rel_err(y.float(), ref.permute(0, 2, 3, 1)) < tol(dtype, 1e-4, 1e-2)
    yf = y.float().cpu().reshape(-1, 64)
    st = stats[:tiles].cpu().sum(0)
    assert float((st[0] - yf.sum(0)).abs().max()) < 1e-3 * max(1.0, float(yf.sum(0).abs().max()))
    assert rel_err(st[1], (yf * yf).sum(0)) < 1e-4
    part = torch.empty(ops.conv1_wgrad_partial_rows(B, H, W), 64 * 27, device=d)
    dw = torch.empty((64, 3, 3, 3), device=d)
    ops.conv1_wgrad(xd, dy.to(d), dw, part)
    be.sync()
    assert rel_err(dw, wq.grad) < tol(dtype, 1e-4, 1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 16, 16, 0), (1, 12, 20, 0), (3, 8, 8, 0), (3, 32, 32, 2)])
def test_conv1_fused_stem(be, dtype, cfg):
    """stats-only pass, conv+BN+SiLU pass and the one-pass backward (recomputed conv, closed-form BN
    backward, weight gradient from the pixel contraction) against torch autograd of SiLU(BN(conv(x)))."""
    from hand_gesture_recognition_amd import lib
    B, H, W, grid_cap = cfg
    d = be.device
    dbg(11, grid_cap)      # > 0: fewer workgroups than tiles -> the tile loop + tap prefetch run
    try:
        _conv1_fused_case(be, dtype, B, H, W)
    finally:
        dbg(11, 0)


def _conv1_fused_case(be, dtype, B, H, W):
    d = be.device
    x = _rand((B, 3, H, W), torch.float32, 51)
    w = (_rand((64, 3, 3, 3), torch.float32, 52) / 27 ** 0.5)
    gamma, beta = 1 + 0.2 * _rand((64,), torch.float32, 53), 0.1 * _rand((64,), torch.float32, 54)
    # reference on the operands as the kernels see them (bf16 mode rounds x patches and w)
    xq, wq = x.to(dtype).float(), w.to(dtype).float()
    wr, gr, br = wq.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    raw = F.conv2d(xq, wr, None, 2, 1)
    z = F.batch_norm(raw, None, None, gr, br, True, 0.1, 1e-5)
    out_ref = F.silu(z)
    dout = _rand((B, H // 2, W // 2, 64), dtype, 55)
    out_ref.backward(dout.float().permute(0, 3, 1, 2))
    M = B * (H // 2) * (W // 2)
    xd, wd = x.to(d), pack_krsc_plain(w, dtype).to(d)
    tiles = ops.conv1_stats_tiles(B, H, W)
    stats = torch.full((ops.slab_rows(tiles), 2, 64), float("nan"), device=d)
    ops.conv1_stats(xd, wd, stats)
    ss, mr = torch.empty(2, 64, device=d), torch.empty(2, 64, device=d)
    ops.bn_finalize(stats, tiles, M, gamma.to(d), beta.to(d), None, None, None, ss, mr)
    out = torch.full((B, H // 2, W // 2, 64), float("nan"), dtype=dtype, device=d)
    ops.conv1_bn_act(xd, wd, ss, out, ops.ACT_SILU)
    be.sync()
    rawf = raw.detach().permute(0, 2, 3, 1).reshape(M, 64)
    assert rel_err(stats[:tiles].sum(0)[0], rawf.sum(0)) < 1e-3 or float((stats[:tiles].cpu().sum(0)[0] - rawf.sum(0)).abs().max()) < 1e-3
    assert rel_err(mr[0], rawf.mean(0)) < 1e-4 or float((mr[0].cpu() - rawf.mean(0)).abs().max()) < 1e-5
    assert rel_err(Fl(out), out_ref.permute(0, 2, 3, 1)) < tol(dtype, 1e-4, 2e-2)
    rows, width = ops.conv1_bwd_partial_shape(B, H, W)
    part = torch.full((rows, width), float("nan"), device=d)
    dg, db, dw = torch.empty(64, device=d), torch.empty(64, device=d), torch.empty(64, 3, 3, 3, device=d)
    ops.conv1_bwd(xd, wd, A(dout).to(d), ss, mr, part, dg, db, dw, ops.ACT_SILU)
    be.sync()
    assert rel_err(dg, gr.grad) < tol(dtype, 2e-4, 2e-2) and rel_err(db, br.grad) < tol(dtype, 2e-4, 2e-2)
    assert rel_err(dw, wr.grad) < tol(dtype, 5e-4, 3e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_weights_layouts(be, dtype):
    """mtn_pack_weights: OIHW fp32 -> [O][kh][kw][I] (mode 0) and [I][kh][kw][O] (mode 1, tiled transpose) in T."""
    shapes = [(64, 3, 3, 3), (128, 64, 3, 3), (96, 200, 1, 1), (256, 128, 1, 1), (70, 33, 3, 3)]
    x3 = dtype == torch.float32 and x3_active()

    def want_of(w, mode, plain):
        if plain or not x3:
            return (w.permute(0, 2, 3, 1) if mode == 0 else w.permute(1, 2, 3, 0)).contiguous().to(dtype).flatten()
        return (pack_krsc if mode == 0 else pack_crsk)(w, dtype).flatten()
    ws = [_rand(s, torch.float32, 60 + i) for i, s in enumerate(shapes)]
    flat = torch.cat([w.flatten() for w in ws])
    desc, off, src = [], 0, 0
    for w in ws:
        O, I, kh, kw = w.shape
        for mode in (0, 1):
            # split-bf16 mode: packed rows are pre-split per 32-k block; rows of another length stay plain fp32 (mode + 2)
            plain = x3 and ((I if mode == 0 else O) * kh * kw) % 32 != 0
            desc.append([src, off, O, I, kh * kw, mode + 2 * plain, -1])
            off = (off + w.numel() + 7) & ~7
        src += w.numel()
    dst = torch.full((off,), float("nan"), dtype=dtype, device=be.device)
    dd = torch.tensor(desc, dtype=torch.int32, device=be.device)
    ops.pack_weights(flat.to(be.device), dst, dd, len(desc), max(w.numel() for w in ws))
    be.sync()
    for j, w in enumerate(ws):
        for mode in (0, 1):
            o = desc[2 * j + mode][1]
            want = want_of(w, mode, desc[2 * j + mode][5] >= 2)
            assert torch.equal(dst[o:o + w.numel()].cpu().view(torch.int32), want.view(torch.int32)), (j, mode)
    # folded form (inference): the forward layout of tensors 1 and 3 multiplied by a per-output-channel factor
    scale = 0.5 + torch.rand(400, generator=torch.Generator().manual_seed(9))
    for j, so in ((1, 7), (3, 140)):
        desc[2 * j][6] = so
    dst2 = torch.full((off,), float("nan"), dtype=dtype, device=be.device)
    ops.pack_weights(flat.to(be.device), dst2, torch.tensor(desc, dtype=torch.int32, device=be.device), len(desc),
                     max(w.numel() for w in ws), scale=scale.to(be.device))
    be.sync()
    for j, w in enumerate(ws):
        for mode in (0, 1):
            o = desc[2 * j + mode][1]
            so = desc[2 * j + mode][6]
            wf = w * scale[so:so + w.shape[0]].view(-1, 1, 1, 1) if (so >= 0 and mode == 0) else w
            want = want_of(wf, mode, desc[2 * j + mode][5] >= 2)
            assert torch.equal(dst2[o:o + w.numel()].cpu().view(torch.int32), want.view(torch.int32)), (j, mode, "folded")


@pytest.mark.parametrize("stages,tile", [(7, 1), (7, 4), (5, 5), (5, 6), (8, 1)])
@pytest.mark.parametrize("cfg", [(2, 64, 64, 192, 128, 1, 1), (2, 64, 64, 64, 128, 3, 1), (2, 64, 64, 64, 256, 1, 1)])
def test_conv_fwd_pipeline_experiments(be, cfg, stages, tile):
    """the K-loop experiments kept behind debug keys stay correct: key 3 = 7, the ping-pong loop (half the waves
    multiply while the other half issues the DMA of the tile two steps ahead) on 128x128 and 256x128 (key 6 = 4)
    tiles; key 6 = 5 / 6, the 16-wave 256x128 tile with two / three buffers; key 3 = 8, weight fragments from global memory
    into registers.  K loops of 1, 3 and 9 steps, with statistics"""
    from hand_gesture_recognition_amd import lib
    dtype = torch.bfloat16
    B, H, W, Cin, Cout, k, s = cfg
    x = _rand((B, H, W, Cin), dtype, 41)
    w = _rand((Cout, Cin, k, k), torch.float32, 42) / (Cin * k * k) ** 0.5
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(dtype).float(), None, s, k // 2).permute(0, 2, 3, 1)
    M = B * H * W
    dbg(3, stages)
    dbg(6, tile)
    try:
        y = torch.full((B, H, W, Cout), float("nan"), dtype=dtype, device=be.device)
        stats = torch.full((ops.conv_fwd_stats_tiles(dtype, B, H, W, Cin, Cout, k, s), 2, Cout), float("nan"), device=be.device)
        ops.conv_fwd(A(x).to(be.device), pack_krsc(w, dtype).to(be.device), y, stats, k, s)
        be.sync()
    finally:
        dbg(3, 5)
        dbg(6, 0)
    assert stats.shape[0] == (M + (255 if tile >= 4 else 127)) // (256 if tile >= 4 else 128)
    assert rel_err(Fl(y), ref) < tol(dtype)
    yf = Fl(y).reshape(M, Cout)
    assert rel_err(stats.cpu().sum(0)[0], yf.sum(0)) < 1e-3


@pytest.fixture
def pw_forced(be):
    """the persistent pointwise kernel (pw.hip) on launches of a few tiles"""
    from hand_gesture_recognition_amd import lib
    dbg(22, 2)
    yield
    dbg(22, 1)


@pytest.mark.parametrize("cfg", [
    # M, K, N     (bf16; M % 64 == 0; the ring is 3 slots deep at K <= 256, 2 at K = 512)
    (64, 128, 128),            # one tile, one walker
    (64 * 9, 256, 128),        # nine tiles on eight walkers: one walker takes two
    (64 * 29, 128, 256),       # two n-tiles; walkers with 3 and 4 tiles (the steady-state step runs)
    (64 * 21, 512, 128),       # two-slot ring
    (64 * 50, 256, 256),       # 16 walkers per n-tile, up to 4 tiles each
])
def test_pointwise_persistent_kernel(be, pw_forced, cfg):
    """pw_kernel: 1x1 convolution forward with statistics (one slab row per workgroup), Linear with bias + GELU + pre-activation
    copy, and the plain data-gradient form, against torch; the statistics slab has exactly mtn_conv_fwd_stats_tiles rows."""
    from hand_gesture_recognition_amd import lib
    M, K, N = cfg
    dtype, d = torch.bfloat16, be.device
    x = _rand((1, 1, M, K), dtype, 91)
    w = _rand((N, K, 1, 1), torch.float32, 92) / K ** 0.5
    wq = w.to(dtype).float()
    ref = (x.float().reshape(M, K) @ wq.reshape(N, K).t())
    rows = ops.conv_fwd_stats_tiles(dtype, 1, 1, M, K, N, 1, 1)
    dbg(22, 0)
    rows_igemm = ops.conv_fwd_stats_tiles(dtype, 1, 1, M, K, N, 1, 1)
    dbg(22, 2)
    assert rows % 8 == 0 and rows <= 256 and (rows != rows_igemm or M <= 64 * 8)
    y = torch.full((1, 1, M, N), float("nan"), dtype=dtype, device=d)
    stats = torch.full((ops.slab_rows(rows), 2, N), float("nan"), device=d)
    ops.conv_fwd(x.to(d), pack_krsc(w, dtype).to(d), y, stats, 1, 1, stats_tiles=rows)
    be.sync()
    yf = y.float().cpu().reshape(M, N)
    assert rel_err(yf, ref) < tol(dtype)
    st = stats[:rows].cpu().sum(0)
    assert rel_err(st[0], yf.sum(0)) < 1e-3 or float((st[0] - yf.sum(0)).abs().max()) < 1e-2
    assert rel_err(st[1], (yf * yf).sum(0)) < 1e-3
    # the same result as igemm.hip's tile for tile (same products, same f32 accumulation order over k)
    dbg(22, 0)
    y2 = torch.empty_like(y)
    ops.conv_fwd(x.to(d), pack_krsc(w, dtype).to(d), y2, None, 1, 1)
    be.sync()
    dbg(22, 2)
    assert torch.equal(y.cpu().view(torch.int16), y2.cpu().view(torch.int16))
    # Linear: bias + GELU + pre-activation copy, output into a slice of a wider buffer
    b = _rand((N,), torch.float32, 93)
    pre = ref + b
    wide = torch.full((M, 2 * N), 7.0, dtype=dtype, device=d)
    aux = torch.full((M, N), float("nan"), dtype=dtype, device=d)
    ops.linear_fwd(x.reshape(M, K).to(d), _wfmt(w.reshape(N, K), dtype).to(d), wide[:, N:], bias=b.to(d), act=ops.ACT_GELU, aux_out=aux)
    be.sync()
    assert rel_err(aux.float(), pre) < tol(dtype)
    assert rel_err(wide[:, N:].float(), F.gelu(pre)) < tol(dtype)
    assert float((wide[:, :N].float() - 7.0).abs().max()) == 0


@pytest.mark.x3
@pytest.mark.parametrize("cfg", [(64 * 9, 128), (64 * 29, 256)])
def test_pointwise_persistent_kernel_split_storage(be, pw_forced, cfg):
    """pw_kernel<f32x3_t> (K = 128: four 32-channel slabs per row): conv forward with statistics against torch"""
    M, N = cfg
    K, dtype, d = 128, torch.float32, be.device
    x = _rand((1, 1, M, K), dtype, 95)
    w = _rand((N, K, 1, 1), torch.float32, 96) / K ** 0.5
    ref = x.reshape(M, K) @ w.reshape(N, K).t()
    rows = ops.conv_fwd_stats_tiles(dtype, 1, 1, M, K, N, 1, 1)
    y = torch.full((1, 1, M, N), float("nan"), dtype=dtype, device=d)
    stats = torch.full((ops.slab_rows(rows), 2, N), float("nan"), device=d)
    ops.conv_fwd(A(x).to(d), pack_krsc(w, dtype).to(d), y, stats, 1, 1, stats_tiles=rows)
    be.sync()
    yf = Fl(y).reshape(M, N)
    assert rel_err(yf, ref) < 5e-5
    st = stats[:rows].cpu().sum(0)
    assert rel_err(st[0], yf.sum(0)) < 1e-3 or float((st[0] - yf.sum(0)).abs().max()) < 1e-2
    assert rel_err(st[1], (yf * yf).sum(0)) < 1e-3


@pytest.fixture
def win3_forced(be):
    """the persistent 64 -> 64 channel 3x3 kernel (win3.hip) on launches of a few tiles"""
    from hand_gesture_recognition_amd import lib
    try:
        dbg(23, 2)
    except RuntimeError:
        pytest.skip("win3.hip is a measured-and-rejected experiment: not in the product library (-DMTN_EXPERIMENTS builds only)")
    yield
    dbg(23, 0)


@pytest.mark.parametrize("cfg", [
    # B, H, W     (64 -> 64 channels, bf16; B*H*W % 128 == 0)
    (2, 8, 8),               # one tile: both images and every border inside it
    (1, 16, 24),             # three tiles, halo rows reach into the neighbouring tiles
    (5, 16, 16),             # ten tiles
    (3, 32, 48),             # 36 tiles on 36 walkers
    (2, 64, 96),             # the six-piece window (W = 96: 384x384 images), two-slot ring
])
def test_conv3x3_persistent_kernel(be, win3_forced, cfg):
    """win3_kernel: forward with statistics (one slab row per workgroup), plain data gradient, data gradient with residual and
    BatchNorm-backward sums -- bit for bit against igemm.hip's window mode (same products, same accumulation order over taps
    and channels) and against torch."""
    from hand_gesture_recognition_amd import lib
    B, H, W = cfg
    C, dtype, d = 64, torch.bfloat16, be.device
    M = B * H * W
    x = _rand((B, H, W, C), dtype, 101)
    w = _rand((C, C, 3, 3), torch.float32, 102) / (C * 9) ** 0.5
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(dtype).float(), None, 1, 1).permute(0, 2, 3, 1)

    def both(fn):
        outs = []
        for key in (2, 0):
            dbg(23, key)
            outs.append(fn())
        dbg(23, 2)
        return outs

    def fwd():
        rows = ops.conv_fwd_stats_tiles(dtype, B, H, W, C, C, 3, 1)
        y = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=d)
        stats = torch.full((ops.slab_rows(rows), 2, C), float("nan"), device=d)
        ops.conv_fwd(x.to(d), pack_krsc(w, dtype).to(d), y, stats, 3, 1, stats_tiles=rows)
        be.sync()
        return y.cpu(), stats[:rows].cpu().sum(0), rows
    (y, st, rows), (y0, st0, rows0) = both(fwd)
    assert rows == min(M // 128, 256)
    assert rel_err(y.float(), ref) < tol(dtype)
    assert torch.equal(y.view(torch.int16), y0.view(torch.int16))
    yf = y.float().reshape(M, C)
    assert rel_err(st[0], yf.sum(0)) < 1e-3 or float((st[0] - yf.sum(0)).abs().max()) < 1e-2
    assert rel_err(st[1], (yf * yf).sum(0)) < 1e-3

    # data gradient: plain; with residual (aliasing the output) + BatchNorm-backward sums of the consuming unit
    dy = _rand((B, H, W, C), dtype, 103)
    res = _rand((B, H, W, C), dtype, 104)
    yb = _rand((B, H, W, C), dtype, 105, ) * 2 + 0.5
    bres = _rand((B, H, W, C), dtype, 106)
    xz = torch.zeros((B, C, H, W), requires_grad=True)
    F.conv2d(xz, w.to(dtype).float(), None, 1, 1).backward(dy.float().permute(0, 3, 1, 2))
    dref = xz.grad.permute(0, 2, 3, 1)
    gamma, beta = 1 + 0.2 * _rand((C,), torch.float32, 107), 0.1 * _rand((C,), torch.float32, 108)
    ybf = yb.float().reshape(M, C)
    mean, var = ybf.mean(0), ybf.var(0, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    ss = torch.stack([gamma * rstd, beta - mean * gamma * rstd]).contiguous().to(d)
    mr = torch.stack([mean, rstd]).contiguous().to(d)

    def dgrad_plain():
        dx = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=d)
        ops.conv_dgrad(dy.to(d), pack_crsk(w, dtype).to(d), dx, None, 3, 1)
        be.sync()
        return dx.cpu()
    dx, dx0 = both(dgrad_plain)
    assert rel_err(dx.float(), dref) < tol(dtype)
    assert torch.equal(dx.view(torch.int16), dx0.view(torch.int16))

    def dgrad_bn():
        tiles = ops.conv_dgrad_bn_tiles(dtype, B, H, W, C, C, 3, 1)
        part = torch.full((ops.slab_rows(tiles), 2, C), float("nan"), device=d)
        dxr = res.clone().to(d)
        ops.conv_dgrad(dy.to(d), pack_crsk(w, dtype).to(d), dxr, dxr, 3, 1,
                       bn=(yb.to(d), bres.to(d), ss, mr, ops.ACT_SILU, part, tiles))
        be.sync()
        return dxr.cpu(), part[:tiles].cpu().sum(0)
    (dxr, pt), (dxr0, pt0) = both(dgrad_bn)
    assert rel_err(dxr.float(), dref + res.float()) < tol(dtype)
    assert torch.equal(dxr.view(torch.int16), dxr0.view(torch.int16))
    scale = float(pt0.abs().max()) + 1e-6
    assert float((pt - pt0).abs().max()) < 2e-4 * scale, (pt, pt0)


@pytest.mark.parametrize("dtype", [pytest.param(torch.bfloat16, id="bf16"), pytest.param(torch.float32, id="x3", marks=pytest.mark.x3)])
@pytest.mark.parametrize("cfg", [(2, 12, 12), (1, 48, 20), (3, 5, 7), (9, 8, 8)])
def test_conv_window_256x64_tile(be, dtype, cfg):
    """Round 5: 64 -> 64 channel 3x3 stride-1 layers on the 256 x 64 window tile with eight waves (debug key 8 = 7 lifts its size
    floor: the model reaches it at >= 262,144 rows).  Forward with statistics, data gradient with a residual, and the data
    gradient's BatchNorm hand-over against the 128 x 64 tile's results."""
    B, H, W = cfg
    C, d = 64, be.device
    x = _rand((B, H, W, C), dtype, 61)
    w = _rand((C, C, 3, 3), torch.float32, 62) / (C * 9) ** 0.5
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(dtype).float(), None, 1, 1).permute(0, 2, 3, 1)
    M = B * H * W
    xz = torch.zeros((B, C, H, W), requires_grad=True)
    dyv, res = _rand((B, H, W, C), dtype, 63), _rand((B, H, W, C), dtype, 64)
    F.conv2d(xz, w.to(dtype).float(), None, 1, 1).backward(dyv.float().permute(0, 3, 1, 2))
    dref = xz.grad.permute(0, 2, 3, 1) + res.float()
    raw = A(_rand((B, H, W, C), dtype, 65) * 1.5 + 0.2).to(d)
    ss = torch.stack([torch.rand(C) + 0.5, torch.randn(C) * 0.2]).to(d)
    mr = torch.stack([torch.randn(C) * 0.1, torch.rand(C) + 0.5]).to(d)
    got = {}
    for key in (0, 7):
        dbg(8, key)
        try:
            assert ops.conv_kernel_family(dtype, False, B, H, W, C, C, 3, 1) == 1
            tiles = ops.conv_fwd_stats_tiles(dtype, B, H, W, C, C, 3, 1)
            assert tiles == (M + 255) // 256 if key == 7 else tiles == (M + 127) // 128
            y = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=d)
            stats = torch.full((tiles, 2, C), float("nan"), device=d)
            ops.conv_fwd(A(x).to(d), pack_krsc(w, dtype).to(d), y, stats, 3, 1, stats_tiles=tiles)
            dx = A(res).clone().to(d)
            ops.conv_dgrad(A(dyv).to(d), pack_crsk(w, dtype).to(d), dx, dx, 3, 1)
            dtiles = ops.conv_dgrad_bn_tiles(dtype, B, H, W, C, C, 3, 1)
            part = torch.full((ops.slab_rows(dtiles), 2, C), float("nan"), device=d)
            dx2 = torch.empty((B, H, W, C), dtype=dtype, device=d)
            ops.conv_dgrad(A(dyv).to(d), pack_crsk(w, dtype).to(d), dx2, None, 3, 1, bn=(raw, None, ss, mr, ops.ACT_SILU, part, dtiles))
            be.sync()
            got[key] = (y.cpu(), stats.cpu().sum(0), dx.cpu(), dx2.cpu(), part[:dtiles].cpu().sum(0))
        finally:
            dbg(8, 0)
    y, st, dx, dx2, ps = got[7]
    assert rel_err(Fl(y), ref) < tol(dtype) and rel_err(Fl(dx), dref) < tol(dtype)
    yf = Fl(y).reshape(M, C)
    assert rel_err(st[0], yf.sum(0)) < 1e-3 and rel_err(st[1], (yf * yf).sum(0)) < 1e-3
    # the same values as the 128 x 64 tile writes (same products, same K order), sums in another tile grouping
    if dtype == torch.bfloat16:
        assert torch.equal(y, got[0][0]) and torch.equal(dx, got[0][2]) and torch.equal(dx2, got[0][3])
    else:       # (split storage: the 128 x 64 tile may keep both 32-channel slabs resident -- another K order of the f32 accumulation)
        assert rel_err(Fl(y), Fl(got[0][0])) < 3e-5 and rel_err(Fl(dx), Fl(got[0][2])) < 3e-5 and rel_err(Fl(dx2), Fl(got[0][3])) < 3e-5
    assert rel_err(ps, got[0][4]) < 1e-4 and rel_err(st, got[0][1]) < 1e-4


@pytest.mark.parametrize("dtype", [DTYPES[1], DTYPES[2]])
@pytest.mark.parametrize("cfg", [(2, 6, 6, 64, 64), (1, 12, 12, 64, 128), (3, 5, 7, 128, 64), (2, 24, 24, 64, 64), (1, 4, 50, 64, 64),
                                 (1, 6, 26, 64, 64)])
def test_conv_wgrad_window_kernel_step_sizes(be, dtype, cfg):
    """wgrad3_kernel with the wide steps of round 5 (bf16: 256 slots for maps up to 48 wide; split storage: 128 slots for maps
    up to 24 wide) and with the narrow ones (debug key 29 = 0: 128 / 64; a 50-wide bf16 map or a 26-wide split-storage map takes
    the narrow step either way): both against torch, f32 partial tiles so that the two agree to summation order."""
    B, H, W, Cin, Cout = cfg
    d = be.device
    x = _rand((B, H, W, Cin), dtype, 12)
    w = torch.zeros((Cout, Cin, 3, 3), requires_grad=True)
    yref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, 1, 1)
    dy = _rand(tuple(yref.permute(0, 2, 3, 1).shape), dtype, 13)
    yref.backward(dy.float().permute(0, 3, 1, 2))
    got = []
    for key in (1, 0):
        dbg(29, key)
        dbg(28, 0)
        try:
            assert ops.conv_wgrad_workspace_bytes(dtype, B, H, W, Cin, Cout, 3, 1) > 0
            ws = torch.empty(ops.conv_wgrad_workspace_bytes(dtype, B, H, W, Cin, Cout, 3, 1) // 4, dtype=torch.float32, device=d)
            dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=d)
            ops.conv_wgrad(A(x).to(d), A(dy).to(d), dw, ws, 3, 1)
            be.sync()
            got.append(dw.cpu())
        finally:
            dbg(29, 1)
            dbg(28, 1)
    t = 1e-5 if dtype == torch.bfloat16 else 1e-4
    assert rel_err(got[0], w.grad) < t and rel_err(got[1], w.grad) < t and rel_err(got[0], got[1]) < 1e-5
