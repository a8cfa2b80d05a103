"""BASELINE.json configs 2, 4 and 5 at their FULL sizes against the oracle.

The paths that only exist at full size (XCD tile remap, several rounds of 256 image tiles, 512-slot split-K
weight gradients, the one-class tap shortcut of the stride-2 data gradients) meet the CPU oracle here:
  config 2  MultiTaskNet 192x192, B=256: outputs, losses and all 114 parameter gradients.  Truth = the oracle
            in float64; the fp32 oracle (= the reference's own arithmetic) is itself 1.4e-4 away from it on the
            worst gradient tensor, so 1e-3 is a meaningful bar for both fp32-storage modes (exact-f32 MFMA and
            split-bf16 "bf16x3"); the bf16 throughput mode is held to per-family bounds taken from
            profiles/r03_grad_errors_f64_b256.txt (tools/grad_error_table.py).
  config 4  MultiTaskNet 384x384 (577 tokens), B=128: truth = the oracle in float64 as for config 2 (round 5: the fp32
            oracle's own distance from it is measured in the same fixture and printed, not assumed).
  config 5  GELAN backbone only, B=512, synthetic upstream gradient, fp32 oracle.
Host memory: the float64 oracle at B=256 keeps ~40 GB of autograd state, config 4 ~80 GB in float64, config 5 ~40 GB in fp32.
"""
import gc

import pytest
import torch

import mtn_oracle as O
from backend import Backend, rel_err
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet

pytestmark = pytest.mark.gpu

MODES = {"fp32": torch.float32, "bf16x3": "bf16x3", "bf16": torch.bfloat16}


def family(name):
    if name.endswith("bn.weight") or name.endswith("bn.bias") or name.startswith("encoder") or name.startswith("proj"):
        return "backbone"                       # conv weights and BatchNorm affine parameters (+ the 1x1 projection)
    return "vit"                                # Linear / LayerNorm / bias / cls_token / heads


# max-norm relative error per tensor allowed against the oracle: (outputs, backbone gradients, ViT gradients)
# fp32 / bf16x3: the north-star 1e-3 throughout.  bf16 (the mode bench.py times for configs 2, 4 and 5): the stated 5e-2 on
# outputs; gradients by family from the committed tables (profiles/r03_grad_errors_f64_b256.txt at B=256: backbone 7.6e-2
# worst, ViT 6.1e-3 worst; configs 4 / 5: profiles/r04_full_size_parity_v1.log).
BOUNDS = {"fp32": (1e-3, 1e-3, 1e-3), "bf16x3": (1e-3, 1e-3, 1e-3), "bf16": (5e-2, 1.2e-1, 1.2e-2)}


def _threads():
    import os
    torch.set_num_threads(min(16, os.cpu_count() or 1))


def _check(model, outs, losses, ref_outs, ref_losses, ref_grads, mode, prefix="", bounds=None):
    b_out, b_bb, b_vit = bounds or BOUNDS[mode]
    names = ("logits", "heatmaps", "attnmap")
    for n, o, r in zip(names, outs, ref_outs):
        e = rel_err(o, r)
        assert e < b_out, (mode, n, e)
    for n, l, r in zip(("total", "class", "joints"), losses, ref_losses):
        e = abs(float(l.detach()) - float(r)) / abs(float(r))
        assert e < b_out, (mode, n + " loss", e)
    worst = {"backbone": (0.0, ""), "vit": (0.0, "")}
    for k, p in model.named_parameters():
        rk = prefix + k
        if rk not in ref_grads:
            continue
        e = rel_err(p.grad if p.grad is not None else p._mtn_grad, ref_grads[rk])
        f = family(k)
        if e > worst[f][0]:
            worst[f] = (e, k)
    print(f"[{mode}] worst gradient tensors: backbone {worst['backbone'][0]:.2e} ({worst['backbone'][1]}), "
          f"vit {worst['vit'][0]:.2e} ({worst['vit'][1]})")
    assert worst["backbone"][0] < b_bb, (mode, worst["backbone"])
    assert worst["vit"][0] < b_vit, (mode, worst["vit"])


def _run_model(sd, batch, size, mode, dev):
    m = MultiTaskNet(21, 19, [size, size], compute_dtype=MODES[mode])
    m.load_state_dict(sd, strict=True)
    m.to(dev).train()
    img, label, target, tw = (t.to(dev) for t in batch)
    c, h, at = m(img)
    tot, cl, jl = O.total_loss(c, h, label, target, tw)
    tot.backward()
    torch.cuda.synchronize()
    return m, (c, h, at), (tot, cl, jl)


@pytest.fixture(scope="module")
def truth_config2():
    """float64 oracle at B=256, 192x192 (one evaluation shared by the three modes)."""
    _threads()
    sd = O.synthetic_state_dict(21, 19, seed=0)
    batch = O.synthetic_batch(256, 192, seed=42)
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    b64 = tuple(t.double() if t.dtype.is_floating_point else t for t in batch)
    outs, losses, grads = O.forward_backward(sd64, b64)
    del sd64, b64
    gc.collect()
    return sd, batch, outs, losses, grads


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16"])
def test_config2_b256_vs_float64_oracle(truth_config2, mode):
    be = Backend("hip")
    sd, batch, outs, losses, grads = truth_config2
    m, o, l = _run_model(sd, batch, 192, mode, be.device)
    _check(m, o, l, outs, losses, grads, mode)
    del m
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def truth_config4():
    """float64 oracle at B=128, 384x384 (577 tokens), and how far the fp32 oracle -- the reference's own arithmetic -- is
    from it on the worst gradient tensor of each family (printed: the noise floor the 1e-3 gradient gates sit on)."""
    _threads()
    sd = O.synthetic_state_dict(21, 19, seed=0)
    batch = O.synthetic_batch(128, 384, seed=42)
    _, _, g32 = O.forward_backward({k: v.clone() for k, v in sd.items()}, batch)
    gc.collect()
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    b64 = tuple(t.double() if t.dtype.is_floating_point else t for t in batch)
    outs, losses, grads = O.forward_backward(sd64, b64)
    del sd64, b64
    worst = {"backbone": (0.0, ""), "vit": (0.0, "")}
    for k, g in g32.items():
        e = rel_err(g, grads[k])
        if e > worst[family(k)][0]:
            worst[family(k)] = (e, k)
    print(f"[fp32 oracle vs float64 oracle, config 4] worst gradient tensors: backbone {worst['backbone'][0]:.2e} "
          f"({worst['backbone'][1]}), vit {worst['vit'][0]:.2e} ({worst['vit'][1]})")
    del g32
    gc.collect()
    return sd, batch, outs, losses, grads


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16"])
def test_config4_b128_384_vs_oracle(truth_config4, mode):
    """Both fp32-storage modes at 1e-3 on outputs, losses AND gradients against the float64 truth (round 4 compared with the
    fp32 oracle and had widened the split-bf16 gradient bound to 1.5e-3: restored)."""
    be = Backend("hip")
    sd, batch, outs, losses, grads = truth_config4
    m, o, l = _run_model(sd, batch, 384, mode, be.device)
    _check(m, o, l, outs, losses, grads, mode)
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16"])
def test_config5_gelan_b512_vs_oracle(mode):
    """GELAN backbone alone at B=512 with a synthetic upstream gradient: features and every encoder gradient."""
    be = Backend("hip")
    _threads()
    B = 512
    sd = O.synthetic_state_dict(21, 19, seed=4)
    x = O.synthetic_batch(B, 192, seed=9)[0]
    up = torch.randn(B, 512, 12, 12, generator=torch.Generator().manual_seed(1))
    sdo = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items() if k.startswith("encoder")}
    ref = O.gelan_net(x, sdo, training=True)
    ref.backward(up)
    ref = ref.detach()
    grads = {k: v.grad for k, v in sdo.items() if v.requires_grad}
    model = MultiTaskNet(21, 19, [192, 192], compute_dtype=MODES[mode])
    model.load_state_dict(sd, strict=True)
    model.to(be.device).train()
    eng = model.engine(B)
    eng.pack_weights()
    feats = eng.backbone_forward(x.to(be.device), training=True)
    e = rel_err(eng.features_plain(feats).float().permute(0, 3, 1, 2), ref)      # (split storage is joined at this boundary)
    print(f"[{mode}] config 5 features: {e:.2e}")
    assert e < BOUNDS[mode][0], (mode, "features", e)
    gbuf = eng.stages[-1]["g_out"]
    eng.set_plain(gbuf, up.permute(0, 2, 3, 1).to(be.device))
    eng.backbone_backward(gbuf)
    torch.cuda.synchronize()
    worst = max((rel_err(p._mtn_grad, grads["encoder." + k]), k) for k, p in model.encoder.named_parameters())
    print(f"[{mode}] config 5 worst encoder gradient: {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < BOUNDS[mode][1], (mode, worst)
