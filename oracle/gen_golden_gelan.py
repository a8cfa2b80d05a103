"""Golden vectors for SURVEY section 8f row 4: GELANNet('large') and a non-square MultiTaskNet.

Run HERE (the build container), where /root/reference is importable; the .npz files travel.
  * gelanl_b2_s64.npz      reference GELANNet('large') (model/gelan.py:148-151), B=2, 64x64, train mode:
                           features, parameter-gradient norms + samples for a seeded upstream gradient,
                           BatchNorm running statistics after the forward.
  * nonsquare_b2_64x96.npz reference MultiTaskNet(21, 19, [64, 96]): outputs, losses, gradient norms.
The oracle restatement is asserted against the reference on the same inputs (worst relative
difference printed, must be <= 1e-5) before anything is written.
The fourth stage (gelan.py:162-163, 174-175) is commented out in the reference: there is nothing to
import for it, the oracle's restatement of it is unpinned (stated in DESIGN.md).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
import mtn_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def gelan_large():
    from model.gelan import GELANNet
    torch.manual_seed(0)
    sd = O.synthetic_state_dict(21, 19, seed=5, gelan_type="large")
    enc_sd = {k[len("encoder."):]: v.clone() for k, v in sd.items() if k.startswith("encoder.")}
    ref = GELANNet("large")
    ref.load_state_dict(enc_sd, strict=True)
    ref.train()
    x = O.synthetic_batch(2, 64, seed=11)[0]
    up = torch.randn(2, 512, 4, 4, generator=torch.Generator().manual_seed(3))
    feats = ref(x)
    feats.backward(up)
    grads = {"encoder." + k: p.grad.detach() for k, p in ref.named_parameters()}
    after = {"encoder." + k: v.detach().clone() for k, v in ref.state_dict().items()}
    # oracle on the same inputs
    sdo = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()
           if k.startswith("encoder.")}
    fo = O.gelan_net(x, sdo, training=True)
    fo.backward(up)
    worst = max([rel(fo, feats)] + [rel(sdo[k].grad, g) for k, g in grads.items()] +
                [rel(sdo[k].detach(), after[k]) for k in after if "running" in k])
    print(f"[gelanl] oracle vs reference: worst rel = {worst:.3e}")
    assert worst <= 1e-5
    names = list(grads)
    g = {"meta": np.array([2, 64, 5, 11, 3], np.int64), "features": feats.detach().numpy(),
         "param_names": np.array(names), "grad_norms": np.array([float(grads[k].double().norm()) for k in names])}
    for k in ("encoder.conv1.conv.weight", "encoder.cspelan1.cv2.1.cv1.conv.weight", "encoder.cspelan2.cv3.1.cv2.bn.weight",
              "encoder.cspelan1.cv3.1.cv2.conv.weight", "encoder.cspelan3.cv4.bn.bias"):
        g["grad." + k] = grads[k].numpy()
    bn = [k for k in after if k.endswith(("running_mean", "running_var"))]
    g["bn_keys"] = np.array(bn)
    g["bn_running"] = np.concatenate([after[k].numpy().ravel() for k in bn])
    np.savez_compressed(os.path.join(OUT, "gelanl_b2_s64.npz"), **g)


def nonsquare():
    from model.multitasknet import MultiTaskNet
    torch.manual_seed(0)
    H, W = 64, 96
    sd = O.synthetic_state_dict(21, 19, seed=6)
    ref = MultiTaskNet(21, 19, [H, W])
    ref.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    ref.train()
    g = torch.Generator().manual_seed(17)
    img = torch.randn(2, 3, H, W, generator=g)
    label = torch.randint(0, 19, (2,), generator=g)
    target = torch.rand(2, 21, H // 4, W // 4, generator=g)
    tw = (torch.rand(2, 21, 1, generator=g) > 0.2).float()
    cls_out, hm, attn = ref(img)
    tot, cl, jl = O.total_loss(cls_out, hm, label, target, tw)
    tot.backward()
    grads = {k: p.grad.detach() for k, p in ref.named_parameters()}
    sdo = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    (co, ho, ao) = O.multitasknet_forward(img, sdo, training=True)
    to, _, _ = O.total_loss(co, ho, label, target, tw)
    to.backward()
    worst = max([rel(co, cls_out), rel(ho, hm), rel(ao, attn)] + [rel(sdo[k].grad, v) for k, v in grads.items()])
    print(f"[nonsquare] oracle vs reference: worst rel = {worst:.3e}")
    assert worst <= 1e-5
    names = list(grads)
    out = {"meta": np.array([2, H, W, 6, 17], np.int64), "img": img.numpy(), "label": label.numpy(),
           "target": target.numpy(), "tw": tw.numpy(), "cls_out": cls_out.detach().numpy(), "hmap_out": hm.detach().numpy(),
           "attn_row0": attn[:, :, 0, :].detach().numpy(), "losses": np.array([float(tot), float(cl), float(jl)]),
           "param_names": np.array(names), "grad_norms": np.array([float(grads[k].double().norm()) for k in names]),
           "grad.proj.weight": grads["proj.weight"].numpy(),
           "grad.decoder.simple_decoder.1.weight": grads["decoder.simple_decoder.1.weight"].numpy(),
           "grad.encoder.cspelan2.cv4.conv.weight": grads["encoder.cspelan2.cv4.conv.weight"].numpy()}
    np.savez_compressed(os.path.join(OUT, "nonsquare_b2_64x96.npz"), **out)


if __name__ == "__main__":
    gelan_large()
    nonsquare()
