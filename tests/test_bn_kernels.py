"""BatchNorm(train)+SiLU(+residual) forward/backward kernels against torch autograd."""
import pytest
import torch
import torch.nn.functional as F

from backend import DTYPES, GEMM_DTYPES, A, Fl, be, rel_err, tol  # noqa: F401
from backend import dbg  # noqa: E402
from hand_gesture_recognition_amd import ops


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(300, 64, True, True), (70, 128, True, False), (520, 64, False, False),
                                 (40000, 64, True, True)])
def test_bn_act_fwd_bwd(be, dtype, cfg):
    M, C, act, with_res = cfg
    if be.name == "emu" and M > 1000:
        pytest.skip("large case is for the GPU")
    y = _rand((M, C), dtype, 1, 2.0) + 0.5
    res = _rand((M, C), dtype, 2) if with_res else None
    gamma = 1 + 0.2 * _rand((C,), torch.float32, 3)
    beta = 0.1 * _rand((C,), torch.float32, 4)
    dout = _rand((M, C), dtype, 5)
    extra = _rand((M, C), dtype, 6) if with_res else None
    # ---- torch reference (fp32 math on the stored values)
    yr = y.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = res.float().requires_grad_(True) if with_res else None
    rm, rv = torch.zeros(C), torch.ones(C)
    z = F.batch_norm(yr.t().reshape(1, C, M), rm, rv, gr, br, True, 0.1, 1e-5).reshape(C, M).t()
    pre = z + rr if with_res else z
    out_ref = F.silu(pre) if act else pre
    out_ref.backward(dout.float())
    # ---- kernels: stats slab as the conv epilogue would write it (tiles of 128 rows)
    tiles = (M + 127) // 128
    stats = torch.zeros(ops.slab_rows(tiles), 2, C)
    yf = y.float()
    for t in range(tiles):
        blk = yf[t * 128:(t + 1) * 128]
        stats[t, 0], stats[t, 1] = blk.sum(0), (blk * blk).sum(0)
    d = be.device
    stats = stats.to(d)
    ss, mr = torch.empty(2, C, device=d), torch.empty(2, C, device=d)
    rmd, rvd = torch.zeros(C, device=d), torch.ones(C, device=d)
    nbt = torch.zeros((), dtype=torch.int64, device=d)
    yd, gd, bd = A(y).to(d), gamma.to(d), beta.to(d)
    resd = A(res).to(d) if with_res else None
    ops.bn_finalize(stats, tiles, M, gd, bd, rmd, rvd, nbt, ss, mr)
    out = torch.empty((M, C), dtype=dtype, device=d)
    a = ops.ACT_SILU if act else ops.ACT_NONE
    ops.bn_act_fwd(yd, ss, out, a, resd)
    be.sync()
    assert rel_err(Fl(out), out_ref) < tol(dtype, 1e-4, 2e-2)
    assert rel_err(rmd, rm) < 1e-4 and rel_err(rvd, rv) < 1e-4 and int(nbt) == 1
    # ---- backward
    part = torch.empty(ops.bn_bwd_partial_rows(M), 2, C, device=d)
    coef = torch.empty(2, C, device=d)
    dg, db = torch.empty(C, device=d), torch.empty(C, device=d)
    dy = torch.empty((M, C), dtype=dtype, device=d)
    dres = A(extra).clone().to(d) if with_res else None
    ops.bn_act_bwd(A(dout).to(d), yd, resd, ss, mr, part, coef, dg, db, dy, a, dres, dres)
    be.sync()
    assert rel_err(Fl(dy), yr.grad) < tol(dtype, 2e-4, 3e-2)
    assert rel_err(dg, gr.grad) < tol(dtype, 2e-4, 2e-2)
    assert rel_err(db, br.grad) < tol(dtype, 2e-4, 2e-2)
    if with_res:
        assert rel_err(Fl(dres), rr.grad + extra.float()) < tol(dtype, 1e-4, 2e-2)


@pytest.mark.parametrize("R,C", [(700, 64), (4608, 128), (129, 72)])
def test_bn_finalize_many_tile_rows(be, R, C):
    """slabs of any row count (stage-1 layers write 4608 tile rows) reduce exactly (double sums)"""
    if be.name == "emu" and R > 1000:
        pytest.skip("large case is for the GPU")
    g = torch.Generator().manual_seed(R)
    count = R * 128
    s = torch.randn(R, C, generator=g) * 10 + 3.0
    q = torch.rand(R, C, generator=g) * 400 + 200 + s * s / 128
    stats = torch.zeros(ops.slab_rows(R), 2, C)          # producer rows + the library's folding scratch
    stats[:R] = torch.stack([s, q], 1)
    d = be.device
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ss, mr = torch.empty(2, C, device=d), torch.empty(2, C, device=d)
    rm, rv = torch.zeros(C, device=d), torch.ones(C, device=d)
    ops.bn_finalize(stats.to(d), R, count, gamma.to(d), beta.to(d), rm, rv, None, ss, mr)
    be.sync()
    mean = s.double().sum(0) / count
    var = (q.double().sum(0) / count - mean * mean).clamp_min(0)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    assert rel_err(mr[0], mean) < 1e-6 and rel_err(mr[1], rstd) < 1e-6
    assert rel_err(ss[0], gamma.double() * rstd) < 1e-6
    assert rel_err(ss[1], beta.double() - mean * gamma.double() * rstd) < 1e-5
    assert rel_err(rv, 0.9 + 0.1 * var * count / (count - 1)) < 1e-6


@pytest.mark.experiments
@pytest.mark.parametrize("R,C", [(700, 64), (257, 512), (4608, 128), (18432, 64), (1000, 72)])
def test_bn_fold_finalize_one_launch_is_the_two_launch_result(be, R, C):
    """Round 4, measured and not adopted (profiles/r04_tail_ab.txt; -DMTN_EXPERIMENTS builds, debug key 26 = 1): rowblock_sum +
    bn_finalize as ONE launch (the block that takes the last ticket of its channel group finalizes; bn.hip).  Same sums in
    the same order -> BIT-identical to the two-launch path, on every repetition (the ticket is left at zero)."""
    if be.name == "emu" and R > 1000:
        pytest.skip("large case is for the GPU")
    g = torch.Generator().manual_seed(R + C)
    d = be.device
    count = R * 128
    stats = torch.zeros(ops.slab_rows(R), 2, C)
    stats[:R] = torch.randn(R, 2, C, generator=g) * 10 + 3.0
    stats[:R, 1] = stats[:R, 1].abs() * 40 + 200
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(d), torch.randn(C, generator=g).to(d)
    got = {}
    for key, reps in ((0, 1), (1, 3)):
        dbg(26, key)
        for rep in range(reps):
            ss, mr = torch.full((2, C), -7.0, device=d), torch.full((2, C), -7.0, device=d)
            rm, rv = torch.zeros(C, device=d), torch.ones(C, device=d)
            nbt = torch.zeros(1, dtype=torch.int64, device=d)
            sd = stats.to(d)
            sd[R:] = float("nan")                       # the folding scratch is write-before-read
            ops.bn_finalize(sd, R, count, gamma, beta, rm, rv, nbt, ss, mr)
            be.sync()
            got[(key, rep)] = [t.cpu().clone() for t in (ss, mr, rm, rv, nbt)]
    dbg(26, 0)
    for k, v in got.items():
        for a, b in zip(v, got[(0, 0)]):
            assert torch.equal(a, b), k
    assert int(got[(1, 0)][4]) == 1 and torch.isfinite(got[(1, 0)][0]).all()


@pytest.mark.experiments
@pytest.mark.parametrize("M,C", [(40000, 64), (17000, 256)])
def test_bn_bwd_fold_finalize_one_launch_is_the_two_launch_result(be, M, C):
    g = torch.Generator().manual_seed(M)
    d = be.device
    dout, y = torch.randn(M, C, generator=g).bfloat16().to(d), torch.randn(M, C, generator=g).bfloat16().to(d)
    ss = torch.stack([torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)]).to(d)
    mr = torch.stack([torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5]).to(d)
    rows = min((M + 63) // 64, 2048)                    # bn_bwd_reduce blocks = producer rows of the slab
    assert rows > 256, rows                             # (a fold level exists)
    got = {}
    for key, reps in ((0, 1), (1, 2)):
        dbg(26, key)
        for rep in range(reps):
            part = torch.full((ops.bn_bwd_partial_rows(M), 2, C), float("nan"), device=d)
            coef, dg, db = torch.empty(2, C, device=d), torch.empty(C, device=d), torch.empty(C, device=d)
            dy = torch.empty((M, C), dtype=torch.bfloat16, device=d)
            ops.bn_act_bwd(dout, y, None, ss, mr, part, coef, dg, db, dy, 1)
            be.sync()
            got[(key, rep)] = [t.cpu().clone() for t in (coef, dg, db, dy)]
    dbg(26, 0)
    for k, v in got.items():
        for a, b in zip(v, got[(0, 0)]):
            assert torch.equal(a, b), k
    assert torch.isfinite(got[(1, 0)][3].float()).all()


def test_bias_grad_many_slab_rows(be):
    """the generic column sum over a slab with hundreds / thousands of rows"""
    M = 5000 if be.name == "emu" else 37120
    x = _rand((M, 256), torch.float32, 9)
    d = be.device
    part = torch.empty(ops.rowslab_rows(M), 256, device=d)
    db = torch.empty(256, device=d)
    ops.bias_grad(x.to(d), part, db)
    be.sync()
    assert rel_err(db, x.double().sum(0)) < 1e-5


_HANDOVER_CFGS = [
    # B, H, W, Cin, Cout, k, s, act, dgrad residual, BatchNorm-unit residual
    (2, 6, 6, 64, 64, 3, 1, True, False, False),
    (3, 5, 7, 128, 128, 1, 1, True, True, True),        # the c1 -> first-block c2 hand-over (both residuals)
    (2, 9, 6, 64, 256, 3, 1, False, False, False),      # unit without activation
    (1, 8, 8, 64, 128, 3, 2, True, False, False),       # stride 2 (dgrad2 / parity-major rows)
    (3, 12, 20, 128, 256, 3, 2, True, True, False),     # stride 2, several tiles
    (2, 12, 12, 256, 64, 1, 1, True, False, True),
]


@pytest.mark.parametrize("dtype", DTYPES)
# (fused_s2 = 0 -- igemm's parity-major rows instead of dgrad2 -- only exists for the stride-2 shapes)
@pytest.mark.parametrize("fused_s2,cfg", [pytest.param(1, c, id=f"dgrad2-cfg{i}") for i, c in enumerate(_HANDOVER_CFGS)] +
                         [pytest.param(0, c, id=f"igemm-cfg{i}") for i, c in enumerate(_HANDOVER_CFGS) if c[6] == 2])
def test_conv_dgrad_writes_the_bn_backward_sums(be, dtype, fused_s2, cfg):
    """mtn_conv_dgrad_bn: dx equals the plain data gradient, its slab sums to (sum dz, sum dz*xhat) of
    the unit consuming dx, and bn_act_bwd(partial_rows=tiles) equals the two-pass backward."""
    from hand_gesture_recognition_amd import lib
    B, H, W, Cin, Cout, k, s, act, with_res, with_bres = cfg
    d = be.device
    w = _rand((Cout, Cin, k, k), torch.float32, 41) / (Cout * k * k) ** 0.5
    wt = A(w.permute(1, 2, 3, 0).contiguous().to(dtype)).to(d)
    OH, OW = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    dy = A(_rand((B, OH, OW, Cout), dtype, 42)).to(d)
    res = A(_rand((B, H, W, Cin), dtype, 43)).to(d) if with_res else None
    y = A(_rand((B, H, W, Cin), dtype, 44, 2.0) + 0.5).to(d)                 # raw conv output of the consuming unit
    bres = A(_rand((B, H, W, Cin), dtype, 45)).to(d) if with_bres else None
    gamma = (1 + 0.2 * _rand((Cin,), torch.float32, 46)).to(d)
    beta = (0.1 * _rand((Cin,), torch.float32, 47)).to(d)
    M = B * H * W
    yf = Fl(y).reshape(M, Cin).to(d)
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    ss = torch.stack([gamma * rstd, beta - mean * gamma * rstd]).contiguous()
    mr = torch.stack([mean, rstd]).contiguous()
    a = ops.ACT_SILU if act else ops.ACT_NONE
    dbg(15, fused_s2)
    try:
        dx_ref = torch.empty((B, H, W, Cin), dtype=dtype, device=d)
        ops.conv_dgrad(dy, wt, dx_ref, res, k, s)
        tiles = ops.conv_dgrad_bn_tiles(dtype, B, H, W, Cin, Cout, k, s)
        part = torch.full((ops.slab_rows(tiles), 2, Cin), float("nan"), device=d)
        dx = torch.empty_like(dx_ref)
        ops.conv_dgrad(dy, wt, dx, res, k, s, bn=(y, bres, ss, mr, a, part, tiles))
        be.sync()
    finally:
        dbg(15, 1)
    assert torch.equal(dx, dx_ref)
    g = Fl(dx).reshape(M, Cin).to(d)
    pre = yf * ss[0] + ss[1] + (Fl(bres).reshape(M, Cin).to(d) if with_bres else 0)
    sg = torch.sigmoid(pre)
    dz = g * (sg * (1 + pre * (1 - sg))) if act else g
    xh = (yf - mean) * rstd
    got = part[:tiles].sum(0)
    assert rel_err(got[0], dz.sum(0)) < 2e-4 or float((got[0] - dz.sum(0)).abs().max()) < 2e-3
    assert rel_err(got[1], (dz * xh).sum(0)) < 2e-4 or float((got[1] - (dz * xh).sum(0)).abs().max()) < 2e-3
    # the BatchNorm backward from these sums == the backward that reduces them itself
    outs = []
    for rows, slab in ((tiles, part), (0, torch.empty(ops.bn_bwd_partial_rows(M), 2, Cin, device=d))):
        coef = torch.empty(2, Cin, device=d)
        dg, db = torch.empty(Cin, device=d), torch.empty(Cin, device=d)
        dyo = torch.empty((B, H, W, Cin), dtype=dtype, device=d)
        dres = torch.empty_like(dyo) if with_bres else None
        ops.bn_act_bwd(dx, y, bres, ss, mr, slab, coef, dg, db, dyo, a, dres=dres, partial_rows=rows)
        be.sync()
        outs.append((dg.clone(), db.clone(), Fl(dyo), None if dres is None else Fl(dres)))
    (dg1, db1, dy1, dr1), (dg0, db0, dy0, dr0) = outs
    scale = float(dg0.abs().max()) + float(db0.abs().max()) + 1e-6
    assert float((dg1 - dg0).abs().max()) < 1e-4 * scale and float((db1 - db0).abs().max()) < 1e-4 * scale
    assert rel_err(dy1, dy0) < tol(dtype, 1e-5, 1e-2)
    if with_bres:
        assert torch.equal(dr1, dr0)


def test_bn_act_bwd_refuses_a_handover_it_cannot_verify(be):
    """mtn_bn_act_bwd(partial_rows > 0) trusts rows a preceding mtn_conv_dgrad_bn wrote.  The library
    records that launch (slab, rows, channels) and refuses a count / slab that does not match it, a hand-over
    consumed twice, and one that no launch produced."""
    B, H, W, Cin, Cout, k, s = 2, 6, 6, 64, 64, 3, 1
    d, dtype = be.device, torch.float32
    wt = (_rand((Cout, Cin, k, k), dtype, 1) / 24).permute(1, 2, 3, 0).contiguous().to(d)
    dy, y = _rand((B, H, W, Cout), dtype, 2).to(d), _rand((B, H, W, Cin), dtype, 3).to(d)
    ss, mr = torch.ones(2, Cin, device=d), torch.ones(2, Cin, device=d)
    tiles = ops.conv_dgrad_bn_tiles(dtype, B, H, W, Cin, Cout, k, s)
    part = torch.zeros(ops.slab_rows(tiles) + 8, 2, Cin, device=d)
    dx = torch.empty((B, H, W, Cin), device=d)
    coef, dg, db, dyo = torch.empty(2, Cin, device=d), torch.empty(Cin, device=d), torch.empty(Cin, device=d), torch.empty_like(dx)

    def produce():
        ops.conv_dgrad(dy, wt, dx, None, k, s, bn=(y, None, ss, mr, ops.ACT_SILU, part, tiles))

    def consume(rows, slab=part):
        ops.bn_act_bwd(dx, y, None, ss, mr, slab, coef, dg, db, dyo, ops.ACT_SILU, partial_rows=rows)

    produce()
    with pytest.raises(RuntimeError, match="do not match"):
        consume(tiles + 1)                            # another row count than the producer wrote
    with pytest.raises(RuntimeError, match="no mtn_conv_dgrad_bn"):
        consume(tiles)                                # the refused call consumed the record
    produce()
    with pytest.raises(RuntimeError, match="another slab"):
        consume(tiles, torch.zeros_like(part))
    produce()
    consume(tiles)                                    # the matching call works ...
    with pytest.raises(RuntimeError, match="no mtn_conv_dgrad_bn"):
        consume(tiles)                                # ... once
    consume(0)                                        # the two-pass form needs no hand-over
    be.sync()


def test_conv_dgrad_bn_refuses_bad_operands(be):
    """Argument errors of the fused hand-over entry fail the call (no launch): a slab sized for another tile
    count, a missing BatchNorm operand, a residual on a unit without activation."""
    dtype = torch.float32
    d = be.device
    B, H, W, Cin, Cout, k, s = 2, 6, 6, 64, 64, 3, 1
    dy = _rand((B, H, W, Cout), dtype, 1).to(d)
    wt = _rand((Cin, k, k, Cout), dtype, 2).to(d)
    dx = torch.empty((B, H, W, Cin), dtype=dtype, device=d)
    y = _rand((B, H, W, Cin), dtype, 3).to(d)
    ss, mr = torch.ones(2, Cin, device=d), torch.ones(2, Cin, device=d)
    tiles = ops.conv_dgrad_bn_tiles(dtype, B, H, W, Cin, Cout, k, s)
    part = torch.empty(ops.slab_rows(tiles + 3), 2, Cin, device=d)
    with pytest.raises(RuntimeError, match="row tiles"):
        ops.conv_dgrad(dy, wt, dx, None, k, s, bn=(y, None, ss, mr, ops.ACT_SILU, part, tiles + 3))
    with pytest.raises(RuntimeError, match="BatchNorm"):
        ops.conv_dgrad(dy, wt, dx, None, k, s, bn=(y, y, ss, mr, ops.ACT_NONE, part, tiles))      # residual needs SiLU
    with pytest.raises(ValueError, match="shape"):
        ops.conv_dgrad(dy, wt, dx, None, k, s, bn=(y[..., :32], None, ss, mr, ops.ACT_SILU, part, tiles))
    with pytest.raises(ValueError, match="slab"):
        ops.conv_dgrad(dy, wt, dx, None, k, s, bn=(y, None, ss, mr, ops.ACT_SILU, part.flatten()[:10], tiles))
    ops.conv_dgrad(dy, wt, dx, None, k, s, bn=(y, None, ss, mr, ops.ACT_SILU, part, tiles))          # and the good call goes through
    be.sync()
