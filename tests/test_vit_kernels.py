"""ViT-side kernels (tokens, LayerNorm, attention, GELU', heads, loss, AdamW, pack) vs torch."""
import math

import pytest
import torch
import torch.nn.functional as F

from backend import DTYPES, GEMM_DTYPES, A, Fl, be, dbg, rel_err, tol, x3_active  # noqa: F401
from hand_gesture_recognition_amd import ops


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_tokens(be, dtype):
    B, P = 3, 9
    d = be.device
    feat, pos, cls = _rand((B * P, 256), dtype, 1), _rand((P, 256), torch.float32, 2), _rand((256,), torch.float32, 3)
    tok = torch.empty((B, P + 1, 256), dtype=dtype, device=d)
    ops.tokens_fwd(A(feat).to(d), pos.to(d), cls.to(d), tok)
    ref = torch.cat([cls.expand(B, 1, 256), feat.float().reshape(B, P, 256) + pos], 1)
    be.sync()
    assert rel_err(Fl(tok), ref) < tol(dtype, 1e-6, 1e-2)
    dtok = _rand((B, P + 1, 256), dtype, 4)
    dfeat = torch.empty((B * P, 256), dtype=dtype, device=d)
    dcls = torch.empty(256, device=d)
    ops.tokens_bwd(A(dtok).to(d), dfeat, dcls)
    be.sync()
    assert rel_err(Fl(dfeat), dtok[:, 1:].float().reshape(B * P, 256)) < tol(dtype, 1e-6, 1e-6)
    assert rel_err(dcls, dtok[:, 0].float().sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", GEMM_DTYPES)
@pytest.mark.parametrize("cfg", [(3, 3, 3, 128, 256), (2, 12, 12, 512, 256), (5, 1, 1, 64, 256), (1, 6, 4, 64, 256), (70, 12, 12, 64, 256)])
def test_projection_writes_the_tokens(be, dtype, cfg):
    """SURVEY K5 (mtn_proj_tokens_fwd): the 1x1 projection with + pos, the token-major rows and the class token in its epilogue
    against torch, and against the two-launch form (projection into a feature tensor, then tokens_fwd): identical for f32
    storage, one rounding apart for bf16 (the fused form rounds conv + pos once).  Shapes: images smaller / larger than a
    64-row tile, a one-patch image (P = 1), a non-square map, and > 8192 rows (the 8-wave 128x128 tile)."""
    B, H, W, Cin, Cout = cfg
    if be.name == "emu" and B * H * W > 600:
        pytest.skip("the 128x128-tile case is for the GPU")
    d, P = be.device, H * W
    x = _rand((B, H, W, Cin), dtype, 31)
    w = _rand((Cout, Cin), torch.float32, 32) / Cin ** 0.5
    pos, cls = _rand((P, Cout), torch.float32, 33), _rand((1, 1, Cout), torch.float32, 34)
    from test_conv_kernels import _wfmt
    wd = _wfmt(w, dtype).to(d)
    ref = torch.cat([cls.expand(B, 1, Cout), (x.float().reshape(B, P, Cin) @ w.to(dtype).float().t()) + pos], 1)
    tok = torch.full((B, P + 1, Cout), float("nan"), dtype=dtype, device=d)
    ops.proj_tokens_fwd(A(x).to(d), wd, pos.to(d), cls.to(d), tok)
    be.sync()
    assert rel_err(Fl(tok), ref) < tol(dtype, 1e-5, 1e-2)
    feat = torch.empty((B, H, W, Cout), dtype=dtype, device=d)
    ops.conv_fwd(A(x).to(d), wd, feat, None, 1, 1)
    tok2 = torch.empty_like(tok)
    ops.tokens_fwd(feat.view(B * P, Cout), pos.to(d), cls.to(d), tok2)
    be.sync()
    if dtype == torch.float32 and not x3_active():
        assert torch.equal(tok.cpu(), tok2.cpu())
    else:
        assert rel_err(Fl(tok), Fl(tok2)) < tol(dtype, 1e-5, 1e-2)
    with pytest.raises(ValueError):
        ops.proj_tokens_fwd(A(x).to(d), wd, pos.to(d)[:-1], cls.to(d), tok)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [7, 200])
def test_layernorm(be, dtype, M):
    d = be.device
    x = _rand((M, 256), dtype, 5, 2.0) + 0.3
    g, b = 1 + 0.2 * _rand((256,), torch.float32, 6), 0.1 * _rand((256,), torch.float32, 7)
    dy, dres = _rand((M, 256), dtype, 8), _rand((M, 256), dtype, 9)
    xr, gr, br = x.float().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yref = F.layer_norm(xr, (256,), gr, br, 1e-5)
    yref.backward(dy.float())
    y = torch.empty((M, 256), dtype=dtype, device=d)
    mr = torch.empty((M, 2), device=d)
    xd, gd, bd = A(x).to(d), g.to(d), b.to(d)
    ops.layernorm_fwd(xd, gd, bd, y, mr)
    be.sync()
    assert rel_err(Fl(y), yref) < tol(dtype, 1e-5, 1e-2)
    part = torch.empty((ops.rowslab_rows(M), 512), device=d)
    dgb = torch.empty((2, 256), device=d)
    dx = A(dres).clone().to(d)
    ops.layernorm_bwd(A(dy).to(d), xd, mr, gd, dx, part, dgb, dres=dx)
    be.sync()
    assert rel_err(Fl(dx), xr.grad + dres.float()) < tol(dtype, 1e-4, 2e-2)
    assert rel_err(dgb[0], gr.grad) < tol(dtype, 1e-4, 1e-2)
    assert rel_err(dgb[1], br.grad) < tol(dtype, 1e-4, 1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gelu_bwd_and_bias_grad(be, dtype):
    d = be.device
    a, dh = _rand((50, 256), dtype, 10), _rand((50, 256), dtype, 11)
    ar = a.float().requires_grad_(True)
    F.gelu(ar).backward(dh.float())
    da = torch.empty_like(a, device=d)
    ops.gelu_bwd(A(dh).to(d), A(a).to(d), da)
    be.sync()
    assert rel_err(Fl(da), ar.grad) < tol(dtype, 1e-5, 1e-2)
    part = torch.empty((ops.rowslab_rows(50), 256), device=d)
    db = torch.empty(256, device=d)
    ops.bias_grad(A(dh).to(d), part, db)
    be.sync()
    assert rel_err(db, dh.float().sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(70, 256, 256), (300, 128, 256), (40000, 256, 256)])
def test_linear_gelu_bwd(be, dtype, cfg):
    """da = (dh @ wt.T) * gelu'(a) and dbias = column sums of da, against the three separate steps."""
    M, N, K = cfg
    if be.name == "emu" and M > 1000:
        pytest.skip("large case (several hundred slab rows, folded twice) is for the GPU")
    d = be.device
    dh, a = _rand((M, K), dtype, 20), _rand((M, N), dtype, 21)
    wt = (_rand((N, K), torch.float32, 22) / K ** 0.5).to(dtype)
    ar = a.float().requires_grad_(True)
    F.gelu(ar).backward(torch.ones_like(ar))
    pre = dh.float() @ wt.float().t()
    ref = pre * ar.grad
    da = torch.full((M, N), float("nan"), dtype=dtype, device=d)
    tiles = ops.linear_gelu_bwd_tiles(M, N, K)
    part = torch.full((ops.slab_rows(tiles), 2, N), float("nan"), device=d)
    wide = torch.full((3 * N,), 7.0, device=d)          # the bias gradient is a slice of the flat gradient buffer
    ops.linear_gelu_bwd(A(dh).to(d), A(wt).to(d), A(a).to(d), da, part, tiles, wide[N:2 * N])
    be.sync()
    assert rel_err(Fl(da), ref) < tol(dtype, 1e-5, 1e-2)
    assert rel_err(wide[N:2 * N], Fl(da).sum(0)) < 1e-5 or float((wide[N:2 * N] - Fl(da).sum(0)).abs().max()) < 1e-3
    assert float((wide[:N] - 7).abs().max()) == 0 and float((wide[2 * N:] - 7).abs().max()) == 0


def _attn_ref(qkv, B, N, H, D):
    q, k, v = (t.reshape(B, N, H, D).permute(0, 2, 1, 3) for t in qkv.reshape(B, N, 3 * H * D).chunk(3, -1))
    attn = torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1)
    return (attn @ v).permute(0, 2, 1, 3).reshape(B * N, H * D), attn


@pytest.mark.parametrize("dtype", GEMM_DTYPES)
@pytest.mark.parametrize("cfg", [(2, 5, 2), (1, 37, 8), (8, 5, 2), (1, 145, 2), (1, 200, 1), (1, 577, 1), (1, 608, 1)])
def test_attention_fwd_bwd(be, dtype, cfg, request):
    B, N, H = cfg
    D = 32
    if be.name == "emu" and N > 150:
        pytest.skip("long sequences on the GPU only")
    x3 = request.node.get_closest_marker("x3") is not None        # split-bf16 products: ~1e-5
    d = be.device
    qkv = _rand((B * N, 3 * H * D), dtype, 12)
    dout = _rand((B * N, H * D), dtype, 13)
    qr = qkv.float().requires_grad_(True)
    oref, aref = _attn_ref(qr, B, N, H, D)
    oref.backward(dout.float())
    qd = A(qkv).to(d)
    out = torch.full((B * N, H * D), float("nan"), dtype=dtype, device=d)
    lse = torch.empty((B * H, N), device=d)
    attn = torch.full((B, H, N, N), float("nan"), device=d)
    ops.attention_fwd(qd, out, lse, attn, B, N, H, D, D ** -0.5)
    be.sync()
    assert rel_err(Fl(out), oref) < tol(dtype, 5e-5 if x3 else 1e-5, 2e-2)
    assert rel_err(attn, aref) < tol(dtype, 5e-5 if x3 else 1e-5, 2e-2)
    delta = torch.empty((B * H, N), device=d)
    dqkv = torch.full((B * N, 3 * H * D), float("nan"), dtype=dtype, device=d)
    ops.attention_bwd(qd, out, A(dout).to(d), lse, delta, dqkv, B, N, H, D, D ** -0.5)
    be.sync()
    assert rel_err(Fl(dqkv), qr.grad) < tol(dtype, 1e-4 if x3 else 2e-5, 3e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_cls_head(be, dtype):
    B, N, C = 5, 4, 19
    d = be.device
    tok = _rand((B, N, 256), dtype, 14, 2.0)
    lw, lb = 1 + 0.2 * _rand((256,), torch.float32, 15), 0.1 * _rand((256,), torch.float32, 16)
    W, bias = _rand((C, 256), torch.float32, 17) / 16, 0.1 * _rand((C,), torch.float32, 18)
    dl = _rand((B, C), torch.float32, 19)
    tr = tok.float().requires_grad_(True)
    ps = [t.clone().requires_grad_(True) for t in (lw, lb, W, bias)]
    ref = F.linear(F.layer_norm(tr[:, 0], (256,), ps[0], ps[1], 1e-5), ps[2], ps[3])
    ref.backward(dl)
    td = A(tok).to(d)
    logits = torch.empty((B, C), device=d)
    xhat, mr = torch.empty((B, 256), device=d), torch.empty((B, 2), device=d)
    lwd, lbd, Wd, bd = lw.to(d), lb.to(d), W.to(d), bias.to(d)
    ops.cls_head_fwd(td, lwd, lbd, Wd, bd, logits, xhat, mr)
    be.sync()
    assert rel_err(logits, ref) < 1e-4
    dtok = torch.zeros((B, N, 256), dtype=dtype, device=d)
    dW, db, dlnwb = torch.empty((C, 256), device=d), torch.empty(C, device=d), torch.empty((2, 256), device=d)
    part = torch.empty((ops.slab_rows(B), 512), device=d)
    ops.cls_head_bwd(dl.to(d), Wd, lwd, lbd, xhat, mr, dtok, dW, db, dlnwb, part)
    be.sync()
    assert rel_err(Fl(dtok), tr.grad) < tol(dtype, 1e-4, 1e-2)
    assert rel_err(dW, ps[2].grad) < 1e-4 and rel_err(db, ps[3].grad) < 1e-4
    assert rel_err(dlnwb[0], ps[0].grad) < 1e-4 and rel_err(dlnwb[1], ps[1].grad) < 1e-4


_POSE_CFGS = [(2, 2, 2, 21), (1, 3, 5, 21), (8, 1, 2, 21), (2, 12, 12, 21), (1, 2, 3, 17),
              (2, 6, 6, 21), (1, 3, 6, 17), (1, 24, 24, 21), (3, 5, 12, 21),   # widths 6 / 12 / 24: the row-streaming kernels
              (1, 8, 6, 21), (2, 12, 6, 17), (1, 7, 6, 21), (3, 13, 12, 21),    # ... cut into 2 / 4 bands per image
              (1, 2, 26, 17), (1, 1, 28, 21)]   # maps wider than 24: the general (round-1) kernels, the only ones that take them


@pytest.mark.parametrize("dtype", GEMM_DTYPES)      # (x3: the pose head takes PLAIN f32 tokens and multiplies split-bf16)
# debug key 25: 0 = bands by batch size, 1 = one workgroup per image -- a choice only where the banded kernel applies
@pytest.mark.parametrize("cfg,bands", [pytest.param(c, b, id=f"cfg{i}-{b}") for i, c in enumerate(_POSE_CFGS) for b in (0, 1)
                                       if b == 0 or (c[2] in (6, 12, 24) and c[1] >= 6)])
def test_pose_head(be, dtype, cfg, bands):
    B, h, w, J = cfg
    if be.name == "emu" and h * w > 80:
        pytest.skip("full-size map on the GPU only")
    dbg(25, bands)
    d = be.device
    tok = _rand((B, h * w + 1, 256), dtype, 20)
    W, bias = _rand((J, 256), torch.float32, 21) / 16, 0.1 * _rand((J,), torch.float32, 22)
    dhm = _rand((B, J, 4 * h, 4 * w), torch.float32, 23)
    tr = tok.float().requires_grad_(True)
    Wr, br = W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    feat = tr[:, 1:].reshape(B, h, w, 256).permute(0, 3, 1, 2)
    up = F.interpolate(feat, scale_factor=(4, 4), mode="bilinear", align_corners=True)
    ref = F.conv2d(F.relu(up), Wr.reshape(J, 256, 1, 1), br)
    ref.backward(dhm)
    td, Wd, bd = tok.to(d), W.to(d), bias.to(d)
    hm = torch.empty((B, J, 4 * h, 4 * w), device=d)
    ops.pose_head_fwd(td, Wd, bd, hm, h, w)
    be.sync()
    assert rel_err(hm, ref) < tol(dtype, 1e-4, 1e-2)
    dtok = torch.zeros((B, h * w + 1, 256), dtype=dtype, device=d)
    dwb = torch.empty(J * 257, device=d)
    slab = torch.empty(ops.pose_slab_floats(B, h, w, J), device=d)
    ops.pose_head_bwd(td, Wd, dhm.to(d), dtok, dwb, slab, h, w)
    be.sync()
    dbg(25, 0)
    dW, db = dwb[:J * 256].view(J, 256), dwb[J * 256:]
    got, want = dtok.float()[:, 1:].cpu(), tr.grad[:, 1:]
    if dtype == torch.float32:
        assert rel_err(got, want) < 1e-4
    else:
        # bf16 storage: norm-wise 1e-2.  The ReLU mask is the sign of the interpolated feature, which
        # the kernel sums in a different order than ATen (x taps on MFMA after the y mix): a feature
        # within ~1e-7 of zero may land on the other side, which moves single elements by O(|G|);
        # allow 0.02 % of such elements instead of a max-norm bound.
        assert float((got - want).norm() / want.norm()) < 1e-2
        bad = ((got - want).abs() > 1e-2 * want.abs().max()).sum().item()
        assert bad <= max(4, got.numel() // 5000), bad      # each flipped pixel touches up to 4 token cells
    # dW contracts T-typed operands on MFMA (bf16 mode rounds d and relu(u) like every wgrad); dbias is f32
    assert rel_err(dW, Wr.grad) < tol(dtype, 1e-4, 1e-2) and rel_err(db, br.grad) < 1e-4


def test_loss(be):
    import mtn_oracle as O
    B, C, J, S = 6, 19, 21, 8
    d = be.device
    logits, hm = _rand((B, C), torch.float32, 24), _rand((B, J, S, S), torch.float32, 25)
    label = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(26))
    tgt = torch.rand((B, J, S, S), generator=torch.Generator().manual_seed(27))
    tw = (torch.rand((B, J, 1), generator=torch.Generator().manual_seed(28)) > 0.3).float()
    lr, hr = logits.clone().requires_grad_(True), hm.clone().requires_grad_(True)
    tot, cl, jl = O.total_loss(lr, hr, label, tgt, tw)
    tot.backward()
    dl, dh = torch.empty((B, C), device=d), torch.empty((B, J, S, S), device=d)
    part = torch.empty(ops.loss_partial_len(B * J * S * S), device=d)
    losses = torch.empty(3, device=d)
    ops.loss_fwd_bwd(logits.to(d), label.to(d), hm.to(d), tgt.to(d), tw.to(d), dl, dh, part, losses)
    be.sync()
    assert rel_err(losses, torch.stack([tot, cl, jl]).detach()) < 1e-5
    assert rel_err(dl, lr.grad) < 1e-5 and rel_err(dh, hr.grad) < 1e-5


def test_loss_rejects_bad_inputs_and_ignores_out_of_range_labels(be):
    """The C ABI takes raw pointers: int32 labels / float64 targets / wrong shapes must be refused
    on the host, and a label outside [0, C) is ignored like torch's ignore_index (never indexed)."""
    B, C, J, S = 5, 19, 21, 8
    d = be.device
    logits, hm = _rand((B, C), torch.float32, 40).to(d), _rand((B, J, S, S), torch.float32, 41).to(d)
    tgt = torch.rand((B, J, S, S), generator=torch.Generator().manual_seed(42)).to(d)
    tw = torch.ones((B, J, 1), device=d)
    label = torch.tensor([3, -100, 18, 19, 0], device=d)            # -100 and 19 are out of range
    dl, dh = torch.empty((B, C), device=d), torch.empty((B, J, S, S), device=d)
    part = torch.empty(ops.loss_partial_len(B * J * S * S), device=d)
    losses = torch.empty(3, device=d)
    with pytest.raises(TypeError):
        ops.loss_fwd_bwd(logits, label.int(), hm, tgt, tw, dl, dh, part, losses)
    with pytest.raises(TypeError):
        ops.loss_fwd_bwd(logits, label, hm, tgt.double(), tw, dl, dh, part, losses)
    with pytest.raises(ValueError):
        ops.loss_fwd_bwd(logits, label[:-1], hm, tgt, tw, dl, dh, part, losses)
    with pytest.raises(ValueError):
        ops.loss_fwd_bwd(logits, label, hm, tgt[:, :-1], tw, dl, dh, part, losses)
    with pytest.raises(ValueError):
        ops.loss_fwd_bwd(logits, label, hm, tgt, tw[:, :-1], dl, dh, part, losses)
    ops.loss_fwd_bwd(logits, label, hm, tgt, tw, dl, dh, part, losses, 1.0)
    be.sync()
    lr = logits.cpu().clone().requires_grad_(True)
    lab = label.cpu().clone()
    lab[lab >= C] = -100
    ref = F.cross_entropy(lr, lab, ignore_index=-100)
    ref.backward()
    assert abs(float(losses[1]) - float(ref)) < 1e-5 * abs(float(ref))
    assert rel_err(dl, lr.grad) < 1e-5
    assert float(dl[1].abs().max()) == 0.0 and float(dl[3].abs().max()) == 0.0


def test_adamw_and_pack(be):
    import mtn_oracle as O
    d = be.device
    n = 1000
    p, g = _rand((n,), torch.float32, 29), _rand((n,), torch.float32, 30)
    sd, st = {"w": p.clone()}, {}
    pd, gd = p.clone().to(d), g.clone().to(d)
    m, v = torch.zeros(n, device=d), torch.zeros(n, device=d)
    state = torch.zeros(16, device=d)
    state[:5] = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0.01])
    state[8] = 1.0
    for _ in range(3):
        O.adamw_step(sd, {"w": g}, st, 1e-3)
        ops.adamw_step(pd, gd, m, v, state)
    be.sync()
    assert rel_err(pd, sd["w"]) < 1e-5
    # pack: OIHW -> [O][tap][I] and [I][tap][O]
    w = _rand((8, 4, 3, 3), torch.float32, 31).to(d)
    desc = torch.tensor([[0, 0, 8, 4, 9, 0, -1], [0, 288, 8, 4, 9, 1, -1]], dtype=torch.int32, device=d)
    for dt in (torch.float32, torch.bfloat16):
        dst = torch.empty(576, dtype=dt, device=d)
        ops.pack_weights(w, dst, desc, 2, 288)
        be.sync()
        assert torch.equal(dst[:288].reshape(8, 3, 3, 4), w.permute(0, 2, 3, 1).to(dt))
        assert torch.equal(dst[288:].reshape(4, 3, 3, 8), w.permute(1, 2, 3, 0).to(dt))
