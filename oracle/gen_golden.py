"""Generate tests/golden/*.npz from the REFERENCE ITSELF (build container only).

Imports ``/root/reference/model`` + ``libs/loss.py`` on CPU, loads the deterministic
synthetic ``state_dict`` of ``oracle/mtn_oracle.py`` with ``strict=True`` (which also pins
the 180-key layout), runs the reference forward / loss / backward / AdamW step and

  1. checks the oracle restatement against it (asserts <= 1e-5 relative), and
  2. writes the golden vectors that ``tests/`` compare against on any box.

The reference source never leaves this container: only the vectors (data) are committed.
Usage:  python oracle/gen_golden.py            (writes tests/golden/)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import mtn_oracle as O  # noqa: E402

REF = "/root/reference"


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def run_reference(sd, batch, image_size, lr):
    sys.path.insert(0, REF)
    from model.multitasknet import MultiTaskNet          # noqa: E402
    from libs.loss import JointsMSELoss, ClassificationLoss  # noqa: E402

    img, label, target, tw = batch
    model = MultiTaskNet(21, 19, [image_size, image_size])
    model.load_state_dict(sd, strict=True)
    model.train()
    inter = {}

    def hook(name):
        def f(_m, _i, o):
            inter[name] = (o[0] if isinstance(o, tuple) else o).detach().clone()
        return f

    for name in ("encoder.conv1", "encoder.conv2", "encoder.cspelan1", "encoder.down1",
                 "encoder.cspelan2", "encoder.cspelan3", "proj",
                 "decoder.transformer.layers.0.0", "decoder.transformer.layers.0.1",
                 "decoder.transformer"):
        model.get_submodule(name).register_forward_hook(hook(name))

    cls_out, hmap, attn = model(img)
    cl = ClassificationLoss()(cls_out, label) * 0.001       # train.py:63
    jl = JointsMSELoss(use_target_weight=True)(hmap, target, tw)  # train.py:64
    tot = cl + jl                                           # train.py:75
    opt = torch.optim.AdamW(model.parameters(), lr)         # train.py:50-51
    opt.zero_grad()
    tot.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    sd_after_fwd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    opt.step()
    sd_after_step = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return dict(cls_out=cls_out.detach(), hmap=hmap.detach(), attn=attn.detach(),
                losses=(tot.detach(), cl.detach(), jl.detach()), grads=grads, inter=inter,
                sd_after_fwd=sd_after_fwd, sd_after_step=sd_after_step,
                param_names=[k for k, _ in model.named_parameters()])


def generate(name, batch_size, image_size, seed, lr=1e-3):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sd = O.synthetic_state_dict(21, 19, seed=seed)
    batch = O.synthetic_batch(batch_size, image_size, seed=42 + seed)
    ref = run_reference({k: v.clone() for k, v in sd.items()}, batch, image_size, lr)

    # ---- pin the oracle against the reference ------------------------------------------
    sd_o = {k: v.clone() for k, v in sd.items()}
    state = {}
    (cls_o, hm_o, attn_o), (tot_o, cl_o, jl_o), grads_o = O.train_step(sd_o, batch, state, lr)
    assert O.param_keys() == ref["param_names"], "parameter order mismatch"
    checks = {
        "cls_out": rel(cls_o, ref["cls_out"]), "hmap": rel(hm_o, ref["hmap"]),
        "attn": rel(attn_o, ref["attn"]), "total_loss": rel(tot_o, ref["losses"][0]),
    }
    for k in ref["grads"]:
        checks["grad." + k] = rel(grads_o[k], ref["grads"][k])
    for k, v in ref["sd_after_step"].items():
        if v.dtype.is_floating_point:
            checks["step." + k] = rel(sd_o[k], v)
        else:
            assert int(sd_o[k]) == int(v), k
    worst = max(checks.items(), key=lambda kv: kv[1])
    print(f"[{name}] oracle vs reference: worst rel = {worst[1]:.3e} at {worst[0]}")
    assert worst[1] <= 1e-5, worst

    # ---- golden vectors -----------------------------------------------------------------
    g = {}
    g["meta"] = np.array([batch_size, image_size, seed], np.int64)
    g["lr"] = np.float64(lr)
    g["torch_version"] = np.array(torch.__version__)
    g["cls_out"] = ref["cls_out"].numpy()
    g["hmap_out"] = ref["hmap"].numpy()
    g["attn_row0"] = ref["attn"][:, :, 0, :].numpy()
    g["attn_b0h0"] = ref["attn"][0, 0].numpy()
    g["attn_rowsum_maxdev"] = np.float64((ref["attn"].sum(-1) - 1).abs().max())
    g["losses"] = np.array([float(x) for x in ref["losses"]], np.float64)
    names = ref["param_names"]
    g["param_names"] = np.array(names)
    g["grad_norms"] = np.array([float(ref["grads"][k].double().norm()) for k in names])
    for k in ("decoder.cls_token", "decoder.mlp_head.1.weight", "decoder.mlp_head.1.bias",
              "decoder.mlp_head.0.weight", "encoder.conv1.bn.weight", "encoder.conv1.bn.bias",
              "encoder.conv1.conv.weight", "encoder.cspelan1.cv2.0.cv1.conv.weight",
              "decoder.transformer.layers.3.0.to_out.weight",
              "decoder.transformer.layers.0.0.norm.weight",
              "decoder.transformer.layers.0.1.net.1.bias",
              "decoder.simple_decoder.1.weight", "decoder.simple_decoder.1.bias",
              "encoder.cspelan3.cv4.bn.weight", "proj.weight"):
        g["grad." + k] = ref["grads"][k].numpy()
    bn_keys = [k for k in ref["sd_after_fwd"] if k.endswith(("running_mean", "running_var"))]
    g["bn_keys"] = np.array(bn_keys)
    g["bn_running"] = np.concatenate([ref["sd_after_fwd"][k].numpy().ravel() for k in bn_keys])
    g["nbt"] = np.array([int(v) for k, v in ref["sd_after_fwd"].items()
                         if k.endswith("num_batches_tracked")], np.int64)
    g["step_weight_norms"] = np.array([float(ref["sd_after_step"][k].double().norm()) for k in names])
    g["step.decoder.cls_token"] = ref["sd_after_step"]["decoder.cls_token"].numpy()
    g["step.encoder.conv1.conv.weight"] = ref["sd_after_step"]["encoder.conv1.conv.weight"].numpy()
    for k, v in ref["inter"].items():
        flat = v.reshape(v.shape[0], -1)
        g["inter." + k + ".sample"] = flat[:, :: max(1, flat.shape[1] // 4096)][:, :4096].numpy()
        g["inter." + k + ".stats"] = np.array([float(v.double().mean()), float(v.double().std()),
                                               float(v.double().abs().max())])
    out = os.path.join(ROOT, "tests", "golden", name + ".npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, **g)
    print(f"[{name}] wrote {out} ({os.path.getsize(out) / 1e6:.2f} MB)")


if __name__ == "__main__":
    # config 1 of BASELINE.json: hagrid.yaml (21 joints, 19 classes), 192x192, B=4
    generate("config1_b4_s192", 4, 192, seed=0)
    # a small full-path case the CPU kernel-logic tests can afford
    generate("small_b2_s64", 2, 64, seed=1)
    generate("tiny_b3_s32", 3, 32, seed=2)
