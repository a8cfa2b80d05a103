"""BatchNorm(train)+SiLU(+residual) forward/backward kernels against torch autograd."""
import pytest
import torch
import torch.nn.functional as F

from backend import DTYPES, be, rel_err, tol  # noqa: F401
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
    yd, gd, bd = y.to(d), gamma.to(d), beta.to(d)
    resd = res.to(d) if with_res else None
    ops.bn_finalize(stats, tiles, M, gd, bd, rmd, rvd, nbt, ss, mr)
    out = torch.empty((M, C), dtype=dtype, device=d)
    a = ops.ACT_SILU if act else ops.ACT_NONE
    ops.bn_act_fwd(yd, ss, out, a, resd)
    be.sync()
    assert rel_err(out.float(), out_ref) < tol(dtype, 1e-4, 2e-2)
    assert rel_err(rmd, rm) < 1e-4 and rel_err(rvd, rv) < 1e-4 and int(nbt) == 1
    # ---- backward
    part = torch.empty(ops.bn_bwd_partial_rows(M), 2, C, device=d)
    coef = torch.empty(2, C, device=d)
    dg, db = torch.empty(C, device=d), torch.empty(C, device=d)
    dy = torch.empty((M, C), dtype=dtype, device=d)
    dres = extra.clone().to(d) if with_res else None
    ops.bn_act_bwd(dout.to(d), yd, resd, ss, mr, part, coef, dg, db, dy, a, dres, dres)
    be.sync()
    assert rel_err(dy.float(), yr.grad) < tol(dtype, 2e-4, 3e-2)
    assert rel_err(dg, gr.grad) < tol(dtype, 2e-4, 2e-2)
    assert rel_err(db, br.grad) < tol(dtype, 2e-4, 2e-2)
    if with_res:
        assert rel_err(dres.float(), rr.grad + extra.float()) < tol(dtype, 1e-4, 2e-2)


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
