"""End-to-end parity of the HIP MultiTaskNet against the oracle / the reference's golden vectors.

fp32 validation mode is held to 1e-3 (max-norm relative) as BASELINE.json's north_star states, outputs AND
gradients; the split-bf16 mode "bf16x3" (fp32 storage, three bf16 MFMAs per product) to 1e-3 on outputs, losses
and BatchNorm statistics and -- at B=4, where the reference's own fp32 gradients are 8.2e-4 away from a float64
evaluation of the same network (profiles/r03_grad_errors_f64_b4.txt) -- to 2.5e-3 on gradients (1e-3 at the full
B=256: tests/test_full_size_parity.py); bf16 throughput mode to the separately stated 5e-2 on outputs
(the reference itself drifts 0.4-2.5e-2 when run in bf16) and per-family gradient bounds
from the committed per-tensor table (profiles/r03_grad_errors_f64_b4.txt: backbone 8.3e-2, ViT 7.1e-3 worst).
"""
import os

import numpy as np
import pytest
import torch

import mtn_oracle as O
from backend import Backend, rel_err
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet


def _setup(be, name, golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    B, S, seed = (int(v) for v in g["meta"])
    sd = O.synthetic_state_dict(21, 19, seed=seed)
    batch = O.synthetic_batch(B, S, seed=42 + seed)
    model = MultiTaskNet(21, 19, [S, S], compute_dtype=dtype)
    model.load_state_dict(sd, strict=True)
    model.to(be.device)
    model.train()
    return g, sd, batch, model


def _run(be, name, golden_dir, dtype, tol_out, tol_grad):
    g, sd, batch, model = _setup(be, name, golden_dir, dtype)
    img, label, target, tw = (t.to(be.device) for t in batch)
    cls_out, hmap, attn = model(img)
    # the reference's loss recipe (train.py:63-64,75) in torch ops on our outputs -> our backward
    total, cl, jl = O.total_loss(cls_out, hmap, label, target, tw)
    total.backward()
    be.sync()
    assert rel_err(cls_out, torch.from_numpy(g["cls_out"])) < tol_out
    assert rel_err(hmap, torch.from_numpy(g["hmap_out"])) < tol_out
    assert rel_err(attn[:, :, 0, :], torch.from_numpy(g["attn_row0"])) < tol_out
    assert rel_err(attn[0, 0], torch.from_numpy(g["attn_b0h0"])) < tol_out
    assert float((attn.sum(-1) - 1).abs().max()) < 1e-3
    assert abs(float(total) - g["losses"][0]) / abs(g["losses"][0]) < tol_out
    # BatchNorm running statistics after one train-mode forward
    msd = model.state_dict()
    bn = torch.cat([msd[str(k)].flatten().cpu() for k in g["bn_keys"]])
    assert rel_err(bn, torch.from_numpy(g["bn_running"])) < tol_out
    assert all(int(msd[k]) == 1 for k in msd if k.endswith("num_batches_tracked"))
    # gradients: every parameter against the oracle (which is pinned to the reference)
    (_, _, _), _, grads = O.forward_backward({k: v.clone() for k, v in sd.items()}, batch)
    # tol_grad: one bound, or {"backbone": conv weights + BatchNorm affine + projection, "vit": everything else}
    bounds = tol_grad if isinstance(tol_grad, dict) else {"backbone": tol_grad, "vit": tol_grad}
    worst = {"backbone": ("", 0.0), "vit": ("", 0.0)}
    for k, p in model.named_parameters():
        e = rel_err(p.grad, grads[k])
        f = "backbone" if k.startswith(("encoder", "proj")) else "vit"
        if e > worst[f][1]:
            worst[f] = (k, e)
    for f, (k, e) in worst.items():
        assert e < bounds[f], (f, k, e)
    tol_grad = max(bounds.values())
    for k in g.files:
        if k.startswith("grad."):
            assert rel_err(dict(model.named_parameters())[k[5:]].grad, torch.from_numpy(g[k])) < tol_grad, k
    return model


def test_tiny_fp32_emu(golden_dir):
    """Kernel logic end to end on the CPU fiber executor (32x32 images, fp32 validation mode)."""
    _run(Backend("emu"), "tiny_b3_s32", golden_dir, torch.float32, 1e-3, 1e-3)


def test_tiny_bf16_emu(golden_dir):
    _run(Backend("emu"), "tiny_b3_s32", golden_dir, torch.bfloat16, 5e-2, 0.25)


def test_tiny_bf16x3_emu(golden_dir):
    """Split-bf16 products through the whole plan on the CPU executor.  Outputs at 1e-3; the gradients of this
    B=3, 32x32 case (BatchNorm over 12 samples in stage 3) amplify the 2^-17 product error to ~1e-2."""
    _run(Backend("emu"), "tiny_b3_s32", golden_dir, "bf16x3", 1e-3, 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, "bf16x3", torch.bfloat16], ids=["f32", "x3", "bf16"])
def test_grouped_weight_gradients_match_the_per_layer_plan(golden_dir, monkeypatch, dtype):
    """The ViT's sixteen Linear weight gradients in one grouped launch at the end of its backward (the default plan) against
    the plan that runs them one by one where autograd would (MTN_GROUP_WGRAD=0): every gradient of the model, same inputs."""
    be = Backend("emu")
    grads = []
    for env in ("1", "0"):
        monkeypatch.setenv("MTN_GROUP_WGRAD", env)
        g, sd, batch, model = _setup(be, "tiny_b3_s32", golden_dir, dtype)
        img, label, target, tw = (t.to(be.device) for t in batch)
        cls_out, hmap, _ = model(img)
        assert model.engine(img.shape[0]).group_wgrad == (env == "1")
        O.total_loss(cls_out, hmap, label, target, tw)[0].backward()
        be.sync()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    for k in grads[0]:
        # (the same products; the weight gradients are summed over another split of the rows: f32 sums of f32 partial tiles --
        #  bf16 tiles in the bf16 mode, whose split count differs between the two plans)
        assert rel_err(grads[0][k], grads[1][k]) < (2e-2 if dtype == torch.bfloat16 else 1e-5), k


@pytest.mark.gpu
def test_tiny_bf16_gpu(golden_dir):
    _run(Backend("hip"), "tiny_b3_s32", golden_dir, torch.bfloat16, 5e-2, 0.3)


def test_state_dict_contract():
    """180 keys, strict load both ways, Lightning-style 'model.' prefix stripped (export.py:34-40)."""
    model = MultiTaskNet(21, 19, [64, 64])
    sd = O.synthetic_state_dict(21, 19, seed=3)
    assert list(model.state_dict().keys()) == list(sd.keys())
    assert [tuple(v.shape) for v in model.state_dict().values()] == [tuple(v.shape) for v in sd.values()]
    ckpt = {"model." + k: v for k, v in sd.items()}
    model.load_state_dict({k[len("model."):]: v for k, v in ckpt.items()}, strict=True)
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert sum(p.numel() for p in model.parameters()) == 7_409_000
    assert len(list(model.parameters())) == 114
    assert "pos_embedding" not in "".join(model.state_dict().keys())
    assert torch.equal(model.decoder.pos_embedding, O.pos_emb_sincos_2d(4, 4, 256))


def test_default_init_matches_reference_order():
    """Same module construction order => same seeded default init as the reference (train.py:21)."""
    torch.manual_seed(42)
    a = MultiTaskNet(21, 19, [32, 32])
    torch.manual_seed(42)
    b = MultiTaskNet(21, 19, [32, 32])
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
    w = a.encoder.conv1.conv.weight
    assert float(w.abs().max()) <= 1.0 / (27 ** 0.5) + 1e-6      # kaiming-uniform(a=sqrt 5) bound


def test_cpu_without_extension_fails_loudly():
    from hand_gesture_recognition_amd import lib
    lib.use_library(lib.HIP_LIB) if os.path.exists(lib.HIP_LIB) else None
    if not os.path.exists(lib.HIP_LIB):
        pytest.skip("HIP library not built")
    model = MultiTaskNet(21, 19, [32, 32])
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 32, 32))


@pytest.mark.gpu
def test_config1_fp32_gpu(golden_dir):
    """BASELINE config 1 (hagrid.yaml, 192x192, B=4) in fp32 validation mode: 1e-3."""
    _run(Backend("hip"), "config1_b4_s192", golden_dir, torch.float32, 1e-3, 1e-3)


BF16_GRAD_BOUNDS = {"backbone": 0.15, "vit": 1.5e-2}     # observed at config 1: 8.3e-2 / 7.1e-3 (per-tensor table in profiles/)


@pytest.mark.gpu
def test_config1_bf16_gpu(golden_dir):
    _run(Backend("hip"), "config1_b4_s192", golden_dir, torch.bfloat16, 5e-2, BF16_GRAD_BOUNDS)


@pytest.mark.gpu
def test_config1_bf16x3_gpu(golden_dir):
    """BASELINE config 1 in the split-bf16 mode: outputs, losses, BN statistics at the north-star 1e-3 (observed
    2e-5); gradients at 2.5e-3 (observed 1.6e-3 worst; the fp32 reference itself: 8.2e-4 from float64)."""
    _run(Backend("hip"), "config1_b4_s192", golden_dir, "bf16x3", 1e-3, 2.5e-3)


@pytest.mark.gpu
def test_small_bf16x3_gpu(golden_dir):
    _run(Backend("hip"), "small_b2_s64", golden_dir, "bf16x3", 1e-3, 1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small_b2_s64", "tiny_b3_s32"])
def test_small_fp32_gpu(golden_dir, name):
    _run(Backend("hip"), name, golden_dir, torch.float32, 1e-3, 1e-3)


# gradient noise floor of a mode at config-1 sizes, relative to the largest gradient element of the tensor: the gates of the
# one-step gradient comparison (_run: fp32 1e-3, bf16x3 2.5e-3; bf16 by family, profiles/r03_grad_errors_f64_b4.txt)
GRAD_NOISE = {torch.float32: 1e-3, "bf16x3": 2.5e-3, torch.bfloat16: 0.15}


def _train_steps(be, name, golden_dir, dtype, nsteps, tol_w, use_graph=False):
    from hand_gesture_recognition_amd.train import TrainStep
    g, sd, batch, model = _setup(be, name, golden_dir, dtype)
    dev_batch = tuple(t.to(be.device) for t in batch)
    step = TrainStep(model, batch[0].shape[0], lr=float(g["lr"]), use_graph=use_graph)
    sd_o, st = {k: v.clone() for k, v in sd.items()}, {}
    ref_grads = []
    for i in range(nsteps):
        losses = step(*dev_batch)
        (_, _, _), ref_losses, grads = O.train_step(sd_o, batch, st, float(g["lr"]))
        ref_grads.append({k: v.detach().clone() for k, v in grads.items()})
        be.sync()
        assert rel_err(losses, torch.stack(ref_losses)) < tol_w, (i, losses, ref_losses)
    # Adam normalises every element by |g|: its first update is -lr * sign(g) whatever |g| is, so an element whose gradient
    # is below the arithmetic's noise may get the opposite sign from ANY other summation order.  That is the ONLY licence
    # for an element to differ from the oracle, and it is checked as such: an element whose weight is off by more than a
    # tenth of a step must have had |g_oracle| below the mode's gradient noise floor (in one of the steps); every other
    # element is within that tenth of a step by construction, and the tensor restricted to those has the oracle's norm to
    # 1e-4 (round 3 widened this norm bound to 5e-4 for small tensors to make room for "one flipped element" without
    # checking that a flip was what happened; flips are now identified, licensed one by one, and excluded from the norm).
    lr = float(g["lr"])
    noise = GRAD_NOISE[dtype]
    bad = tot = 0
    for k, p in model.named_parameters():
        ours, ref = p.detach().cpu(), sd_o[k]
        d = (ours - ref).abs()
        off = d > 0.1 * lr * nsteps
        bad += int(off.sum())
        tot += d.numel()
        if bool(off.any()):
            # smallest |g| / max|g| the element saw over the oracle's steps (later steps start from weights that already
            # differ by the earlier licensed flips: twice the floor for them)
            small = torch.zeros_like(off)
            for t, gr in enumerate(ref_grads):
                ga = gr[k].abs()
                small |= ga <= (1 if t == 0 else 2) * noise * float(ga.max())
            assert bool((small | ~off).all()), (k, int((off & ~small).sum()), "elements off by > 0.1 lr although their gradient was above the noise floor")
        keep = ~off
        n_ours, n_ref = float((ours * keep).double().norm()), float((ref * keep).double().norm())
        assert abs(n_ours / max(n_ref, 1e-30) - 1) < 1e-4, (k, n_ours, n_ref)
    assert bad / tot < (5e-3 if dtype != torch.bfloat16 else 0.2), (bad, tot)
    if nsteps == 1 and dtype != torch.bfloat16:
        names = [str(n) for n in g["param_names"]]
        wn = torch.tensor([float(dict(model.named_parameters())[k].double().norm()) for k in names])
        assert rel_err(wn, torch.from_numpy(g["step_weight_norms"])) < tol_w
    return model


def test_train_step_fp32_emu(golden_dir):
    """forward + loss + backward + fused AdamW vs the oracle's train step (weights after the step)."""
    _train_steps(Backend("emu"), "tiny_b3_s32", golden_dir, torch.float32, 1, 1e-3)


@pytest.mark.gpu
def test_train_steps_fp32_gpu(golden_dir):
    _train_steps(Backend("hip"), "config1_b4_s192", golden_dir, torch.float32, 1, 1e-3)
    _train_steps(Backend("hip"), "small_b2_s64", golden_dir, torch.float32, 3, 2e-3)


@pytest.mark.gpu
def test_train_steps_bf16x3_gpu(golden_dir):
    """the fused step (AdamW included) in the split-bf16 mode against the oracle's train step: losses of three
    consecutive steps at 1e-3 / 2e-3, weights by norm and by the per-element budget; also replayed from a hipGraph"""
    _train_steps(Backend("hip"), "config1_b4_s192", golden_dir, "bf16x3", 1, 1e-3)
    _train_steps(Backend("hip"), "small_b2_s64", golden_dir, "bf16x3", 3, 2e-3)
    _train_steps(Backend("hip"), "small_b2_s64", golden_dir, "bf16x3", 3, 2e-3, use_graph=True)


@pytest.mark.gpu
def test_train_step_bf16_gpu(golden_dir):
    """the same licence in the throughput mode: its losses are within its own tolerance, and every weight that leaves the
    oracle's by more than a tenth of a step had a gradient below bf16's noise floor (profiles/r03_grad_errors_f64_b4.txt)"""
    _train_steps(Backend("hip"), "config1_b4_s192", golden_dir, torch.bfloat16, 1, 3e-2)


@pytest.mark.gpu
def test_train_steps_graph_gpu(golden_dir):
    """the hipGraph replay path gives the same step as the eager one"""
    _train_steps(Backend("hip"), "small_b2_s64", golden_dir, torch.float32, 3, 2e-3, use_graph=True)


def _eval_case(be, dtype, tol):
    """model.eval() with non-trivial running statistics against the oracle in eval mode; the forward must not run
    a single BatchNorm pass (the statistics are folded into the packed weights, the shift is the conv's bias)."""
    from hand_gesture_recognition_amd import ops
    sd = O.synthetic_state_dict(21, 19, seed=2)
    g = torch.Generator().manual_seed(5)
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = 0.3 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
    S = 32 if be.name == "emu" else 192
    batch = O.synthetic_batch(2 if be.name == "emu" else 4, S, seed=11)
    ref = O.multitasknet_forward(batch[0], {k: v.clone() for k, v in sd.items()}, training=False)
    model = MultiTaskNet(21, 19, [S, S], compute_dtype=dtype)
    model.load_state_dict(sd, strict=True)
    model.to(be.device).eval()
    saved = (ops.bn_act_fwd, ops.bn_finalize)

    def no_bn(*a, **k):
        raise AssertionError("a BatchNorm pass ran in eval mode")
    ops.bn_act_fwd = ops.bn_finalize = no_bn
    try:
        with torch.no_grad():
            cls_out, hmap, attn = model(batch[0].to(be.device))
        be.sync()
    finally:
        ops.bn_act_fwd, ops.bn_finalize = saved
    assert rel_err(cls_out, ref[0]) < tol and rel_err(hmap, ref[1]) < tol and rel_err(attn, ref[2]) < tol
    msd = model.state_dict()
    assert all(int(v) == 0 for k, v in msd.items() if k.endswith("num_batches_tracked"))
    assert all(torch.equal(msd[k].cpu(), sd[k]) for k in sd if "running" in k)          # no state is mutated
    # a training forward right after re-packs the unfolded weights
    model.train()
    c2 = model(batch[0].to(be.device))[0]
    r2 = O.multitasknet_forward(batch[0], {k: v.clone() for k, v in sd.items()}, training=True)[0]
    assert rel_err(c2, r2) < max(tol, 2e-2 if dtype == torch.bfloat16 else 0)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), ("bf16x3", 1e-3), (torch.bfloat16, 3e-2)], ids=["f32", "x3", "bf16"])
def test_eval_mode_folded_batchnorm_emu(dtype, tol):
    _eval_case(Backend("emu"), dtype, tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), ("bf16x3", 1e-3), (torch.bfloat16, 3e-2)], ids=["f32", "x3", "bf16"])
def test_eval_mode_folded_batchnorm_gpu(dtype, tol):
    _eval_case(Backend("hip"), dtype, tol)


@pytest.mark.gpu
def test_eval_mode_gpu(golden_dir):
    """model.eval(): BatchNorm uses running statistics (export.py:41), no state is mutated."""
    be = Backend("hip")
    g, sd, batch, model = _setup(be, "small_b2_s64", golden_dir, torch.float32)
    model.eval()
    with torch.no_grad():
        cls_out, hmap, attn = model(batch[0].to(be.device))
    ref = O.multitasknet_forward(batch[0], {k: v.clone() for k, v in sd.items()}, training=False)
    assert rel_err(cls_out, ref[0]) < 1e-3 and rel_err(hmap, ref[1]) < 1e-3 and rel_err(attn, ref[2]) < 1e-3
    assert all(int(v) == 0 for k, v in model.state_dict().items() if k.endswith("num_batches_tracked"))


def test_nonsquare_image_emu():
    """The model (unlike the reference's train.py:191-192 assert) accepts H != W: 16x32 -> 1x2 tokens."""
    be = Backend("emu")
    sd = O.synthetic_state_dict(21, 19, seed=6)
    model = MultiTaskNet(21, 19, [16, 32], compute_dtype=torch.float32)
    model.load_state_dict(sd, strict=True)
    model.train()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 3, 16, 32, generator=g)
    cls_out, hmap, attn = model(x)
    (cls_out.sum() * 0.1 + (hmap ** 2).mean()).backward()
    sdo = {k: v.clone() for k, v in sd.items()}
    for k in O.param_keys():
        sdo[k].requires_grad_(True)
    rc, rh, ra = O.multitasknet_forward(x, sdo, training=True)
    (rc.sum() * 0.1 + (rh ** 2).mean()).backward()
    assert hmap.shape == (2, 21, 4, 8) and attn.shape == (2, 8, 3, 3)
    assert rel_err(cls_out, rc) < 1e-3 and rel_err(hmap, rh) < 1e-3 and rel_err(attn, ra) < 1e-3
    worst = max((rel_err(p.grad, sdo[k].grad), k) for k, p in model.named_parameters())
    assert worst[0] < 1e-3, worst


@pytest.mark.gpu
def test_train_cli_gpu():
    """the reference's flag surface end to end: python -m hand_gesture_recognition_amd.train ..."""
    from hand_gesture_recognition_amd import lib, train
    lib.use_library(lib.HIP_LIB)
    args = train.build_parser().parse_args(["--batch_size", "4", "--epochs", "2", "--steps_per_epoch", "2",
                                            "--image_size", "64", "64", "--lr", "1e-3", "--lr_step", "1", "5"])
    model = train.run(args, {"num_joints": 21, "num_classes": 19})
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_fused_bn_backward_handover_emu(golden_dir, monkeypatch):
    """15 of the 21 BatchNorm units get their backward sums from the data gradient that produces
    their output gradient (engine._dgrad); the parameter gradients equal the two-pass backward's."""
    from hand_gesture_recognition_amd import ops
    be = Backend("emu")
    grads = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("MTN_FUSE_BN_BWD", fuse)
        g, sd, batch, model = _setup(be, "tiny_b3_s32", golden_dir, torch.float32)
        calls = []
        real = ops.conv_dgrad
        monkeypatch.setattr(ops, "conv_dgrad", lambda *a, bn=None, **k: (calls.append(bn is not None), real(*a, bn=bn, **k))[1])
        img, label, target, tw = (t.to(be.device) for t in batch)
        cls_out, hmap, _ = model(img)
        O.total_loss(cls_out, hmap, label, target, tw)[0].backward()
        monkeypatch.setattr(ops, "conv_dgrad", real)
        assert sum(calls) == (15 if fuse == "1" else 0), calls
        assert all(u.pre_rows == 0 for u in model.engine(img.shape[0]).units)
        grads[fuse] = {k: p.grad.clone() for k, p in model.named_parameters()}
    for k in grads["1"]:
        assert rel_err(grads["1"][k], grads["0"][k]) < 2e-5, k
