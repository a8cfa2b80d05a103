"""Per-tensor error table of one forward + backward against the CPU oracle (the checker, not the product):
python tools/grad_error_table.py [--batch 4 --size 192 --modes fp32,bf16x3,bf16] > profiles/rNN_grad_errors_*.txt

Prints, per compute mode: max-norm relative error of logits / heat-maps / attention map / losses, then one row
per parameter tensor (114) with its gradient error; `family` groups rows for the per-family bounds of
tests/test_model_parity.py.  With --truth f64 the comparison is against the oracle run in float64 and the
table gets a column `ref_fp32` = the fp32 oracle (the reference's own arithmetic) against that truth: the noise
floor a 1e-3 gate on fp32 gradients has to be read against."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def family(name):
    if name.endswith("bn.weight") or name.endswith("bn.bias"):
        return "bn"
    if name.startswith("encoder.conv1"):
        return "stem.conv"
    if name.startswith("encoder") or name.startswith("proj"):
        return "conv"
    if "norm" in name or ".net.0." in name or "mlp_head.0" in name:
        return "layernorm"
    if name.endswith("bias"):
        return "bias"
    if "cls_token" in name:
        return "cls_token"
    return "linear"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=192)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--modes", default="fp32,bf16x3,bf16")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--truth", default="f32", choices=["f32", "f64"])
    a = ap.parse_args()
    import mtn_oracle as O
    from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
    torch.set_num_threads(a.threads)
    dev = torch.device("cuda:0")
    sd = O.synthetic_state_dict(21, 19, seed=a.seed)
    batch = O.synthetic_batch(a.batch, a.size, seed=42 + a.seed)
    (oc, oh, oa), (ot, ocl, ojl), og = O.forward_backward({k: v.clone() for k, v in sd.items()}, batch)
    table = {}
    if a.truth == "f64":
        sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
        b64 = tuple(t.double() if t.dtype.is_floating_point else t for t in batch)
        (dc, dh, da), (dt_, dcl, djl), dg = O.forward_backward(sd64, b64)
        print(f"# B={a.batch}, {a.size}x{a.size}, synthetic weights seed {a.seed}; errors = max|x - truth| / max|truth| per tensor, "
              "truth = oracle/mtn_oracle.py in float64")
        print(f"{'ref_fp32':8s} outputs: logits {rel(oc, dc):.2e}  heatmaps {rel(oh, dh):.2e}  attnmap {rel(oa, da):.2e}  "
              f"total loss {abs(float(ot) - float(dt_)) / abs(float(dt_)):.2e}   (the fp32 oracle = the reference's arithmetic, against the truth)")
        table["ref_fp32"] = {k: rel(og[k], dg[k]) for k in og}
        (oc, oh, oa), (ot, ocl, ojl), og = (dc, dh, da), (dt_, dcl, djl), dg
    else:
        print(f"# B={a.batch}, {a.size}x{a.size}, synthetic weights seed {a.seed}; errors = max|x - oracle| / max|oracle| per tensor")
    for mode in a.modes.split(","):
        dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "bf16x3": "bf16x3"}[mode]
        m = MultiTaskNet(21, 19, [a.size, a.size], compute_dtype=dt)
        m.load_state_dict(sd, strict=True)
        m.to(dev).train()
        img, label, target, tw = (t.to(dev) for t in batch)
        c, h, at = m(img)
        tot, cl, jl = O.total_loss(c, h, label, target, tw)
        tot.backward()
        torch.cuda.synchronize()
        print(f"{mode:7s} outputs: logits {rel(c, oc):.2e}  heatmaps {rel(h, oh):.2e}  attnmap {rel(at, oa):.2e}  "
              f"total loss {abs(float(tot) - float(ot)) / abs(float(ot)):.2e}  class {abs(float(cl) - float(ocl)) / abs(float(ocl)):.2e}  "
              f"joints {abs(float(jl) - float(ojl)) / abs(float(ojl)):.2e}")
        table[mode] = {k: rel(p.grad, og[k]) for k, p in m.named_parameters()}
        del m
        torch.cuda.empty_cache()
    modes = list(table)
    print(f"\n{'parameter':58s} {'family':10s} " + " ".join(f"{m:>9s}" for m in modes))
    fam = {}
    for k in table[modes[-1]]:
        f = family(k)
        print(f"{k:58s} {f:10s} " + " ".join(f"{table[m][k]:9.2e}" for m in modes))
        for m in modes:
            fam.setdefault(f, {}).setdefault(m, []).append(table[m][k])
    print(f"\n{'family (worst tensor)':58s} {'tensors':10s} " + " ".join(f"{m:>9s}" for m in modes))
    for f, d in fam.items():
        print(f"{f:58s} {len(d[modes[0]]):<10d} " + " ".join(f"{max(d[m]):9.2e}" for m in modes))
    print(f"{'ALL':58s} {len(table[modes[0]]):<10d} " + " ".join(f"{max(table[m].values()):9.2e}" for m in modes))


if __name__ == "__main__":
    main()
