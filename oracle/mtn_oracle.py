"""CPU oracle for the MultiTaskNet hot path -- TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (PyTorch-CPU fp32 functional ops) of the
reference's MultiTaskNet forward / loss / train-step arithmetic.  It is the
checker for the HIP path, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``hand-gesture-recognition_amd/`` imports it.

Pinning: the reference holds no tests / golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, imported on CPU in the build container by ``oracle/gen_golden.py``;
the resulting vectors are committed under ``tests/golden/`` and
``tests/test_oracle.py`` re-checks this file against them.  ``GELANNet('large')``
and the non-square case are pinned the same way (``oracle/gen_golden_gelan.py``).
The fourth GELAN stage is commented out in the reference (model/gelan.py:162-163,
174-175): there is nothing to run it against -- parity unpinned for that one
configuration (the blocks it is made of are the pinned ones).

Every function cites the reference file:line (relative to the reference
repository root) it follows.  The arithmetic lives in PyTorch ATen, the
reference's own third-party dependency (docker/Dockerfile:1-4 pins torch 2.0.1);
the op semantics used here are stable across torch 2.0 -> 2.10.

All tensors are NCHW / [B, N, C] fp32 exactly like the reference.  Weights come
in as a ``state_dict``-shaped mapping with the reference's 180 keys.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ----------------------------------------------------------------------------
# hyper-parameters fixed by model/multitasknet.py:12-22
# ----------------------------------------------------------------------------
DIM = 256
DEPTH = 4
HEADS = 8
HEAD_DIM = 32
MLP_DIM = 256
BN_EPS = 1e-5          # nn.BatchNorm2d default, model/gelan.py:46
BN_MOMENTUM = 0.1      # nn.BatchNorm2d default, model/gelan.py:46
LN_EPS = 1e-5          # nn.LayerNorm default, model/transformer.py:33,54,114

# (name, c_in, c_out) of the stride-2 3x3 stem/downsample Convs, model/gelan.py:155-160
STAGES = (
    ("cspelan1", 128, 128, 128, 64),    # model/gelan.py:157
    ("cspelan2", 256, 256, 256, 128),   # model/gelan.py:159
    ("cspelan3", 512, 512, 512, 256),   # model/gelan.py:161
)


# ----------------------------------------------------------------------------
# deterministic synthetic weights / inputs (no RNG-order dependence)
# ----------------------------------------------------------------------------
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def _key_hash(name: str, seed: int) -> int:
    h = 1469598103934665603 ^ (seed & 0xFFFFFFFF)
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_uniform(name: str, shape: Sequence[int], seed: int = 0) -> np.ndarray:
    """U[-1,1) float32 array, a closed-form function of (name, seed, flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(_key_hash(name, seed))
        bits = _splitmix64(idx)
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)   # [0,1)
    return (2.0 * u - 1.0).astype(np.float32).reshape(shape)


def hash_normal(name: str, shape: Sequence[int], seed: int = 0) -> np.ndarray:
    """~N(0,1) float32 via Box-Muller on two hashed uniforms."""
    u1 = (hash_uniform(name + "#a", shape, seed).astype(np.float64) + 1.0) * 0.5
    u2 = (hash_uniform(name + "#b", shape, seed).astype(np.float64) + 1.0) * 0.5
    u1 = np.clip(u1, 1e-7, 1.0)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


GELAN_LAYERS = {"small": [1, 1, 1, 1], "large": [2, 2, 2, 2]}      # model/gelan.py:148-151


def state_dict_spec(num_joints: int, num_classes: int, gelan_type: str = "small", gelan_stages: int = 3):
    """The reference's state_dict layout: ordered list of (key, shape, kind).

    ``gelan_type`` / ``gelan_stages`` select the other rows of the reference's spec table
    (model/gelan.py:148-151) and the fourth stage it keeps commented out (:162-163); the defaults
    are what ``MultiTaskNet`` hard-codes (model/multitasknet.py:12).

    kind in {conv, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, lin_w, lin_b, ln_w, ln_b, cls}.
    Order == ``MultiTaskNet.state_dict()`` order [SURVEY.md 8(b)];
    model/gelan.py:153-161, model/multitasknet.py:12-22, model/transformer.py:101-127.
    """
    spec = []

    def conv_block(prefix, c1, c2, k):
        spec.append((prefix + ".conv.weight", (c2, c1, k, k), "conv"))
        spec.append((prefix + ".bn.weight", (c2,), "bn_w"))
        spec.append((prefix + ".bn.bias", (c2,), "bn_b"))
        spec.append((prefix + ".bn.running_mean", (c2,), "bn_rm"))
        spec.append((prefix + ".bn.running_var", (c2,), "bn_rv"))
        spec.append((prefix + ".bn.num_batches_tracked", (), "bn_nbt"))

    conv_block("encoder.conv1", 3, 64, 3)
    conv_block("encoder.conv2", 64, 128, 3)

    layers = GELAN_LAYERS[gelan_type]

    def gelan(prefix, c_in, c_out, c_hid1, c_hid2, nblocks):      # model/gelan.py:124-135
        conv_block(prefix + ".cv1", c_in, c_hid1, 1)
        for br, cin in (("cv2", c_hid1 // 2), ("cv3", c_hid2)):
            for i in range(nblocks):
                conv_block(f"{prefix}.{br}.{i}.cv1", cin if i == 0 else c_hid2, c_hid2, 3)
                conv_block(f"{prefix}.{br}.{i}.cv2", c_hid2, c_hid2, 3)
        conv_block(prefix + ".cv4", c_hid1 + 2 * c_hid2, c_out, 1)

    gelan("encoder.cspelan1", 128, 128, 128, 64, layers[0])
    conv_block("encoder.down1", 128, 256, 3)
    gelan("encoder.cspelan2", 256, 256, 256, 128, layers[1])
    conv_block("encoder.down2", 256, 512, 3)
    gelan("encoder.cspelan3", 512, 512, 512, 256, layers[2])
    if gelan_stages == 4:
        conv_block("encoder.down3", 512, 512, 3)
        gelan("encoder.cspelan4", 512, 512, 512, 256, layers[3])

    spec.append(("proj.weight", (DIM, 512, 1, 1), "conv"))
    spec.append(("decoder.cls_token", (1, 1, DIM), "cls"))
    inner = HEADS * HEAD_DIM
    for l in range(DEPTH):
        p = f"decoder.transformer.layers.{l}"
        spec.append((p + ".0.norm.weight", (DIM,), "ln_w"))
        spec.append((p + ".0.norm.bias", (DIM,), "ln_b"))
        spec.append((p + ".0.to_qkv.weight", (3 * inner, DIM), "lin_w"))
        spec.append((p + ".0.to_out.weight", (DIM, inner), "lin_w"))
        spec.append((p + ".1.net.0.weight", (DIM,), "ln_w"))
        spec.append((p + ".1.net.0.bias", (DIM,), "ln_b"))
        spec.append((p + ".1.net.1.weight", (MLP_DIM, DIM), "lin_w"))
        spec.append((p + ".1.net.1.bias", (MLP_DIM,), "lin_b"))
        spec.append((p + ".1.net.4.weight", (DIM, MLP_DIM), "lin_w"))
        spec.append((p + ".1.net.4.bias", (DIM,), "lin_b"))
    spec.append(("decoder.mlp_head.0.weight", (DIM,), "ln_w"))
    spec.append(("decoder.mlp_head.0.bias", (DIM,), "ln_b"))
    spec.append(("decoder.mlp_head.1.weight", (num_classes, DIM), "lin_w"))
    spec.append(("decoder.mlp_head.1.bias", (num_classes,), "lin_b"))
    spec.append(("decoder.simple_decoder.1.weight", (num_joints, DIM, 1, 1), "conv"))
    spec.append(("decoder.simple_decoder.1.bias", (num_joints,), "lin_b"))
    return spec


def synthetic_state_dict(num_joints: int = 21, num_classes: int = 19, seed: int = 0,
                         gelan_type: str = "small", gelan_stages: int = 3) -> Dict[str, Tensor]:
    """Deterministic weights scaled to PyTorch default-init ranges.

    conv / linear weights ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (kaiming-uniform a=sqrt(5));
    BN/LN affine are perturbed around (1, 0) so that gamma/beta gradients and the
    affine arithmetic are exercised; running stats start at (0, 1) like nn.BatchNorm2d.
    """
    sd = {}
    for key, shape, kind in state_dict_spec(num_joints, num_classes, gelan_type, gelan_stages):
        if kind in ("conv", "lin_w"):
            fan_in = int(np.prod(shape[1:]))
            w = hash_uniform(key, shape, seed) / math.sqrt(fan_in)
        elif kind == "lin_b":
            w = hash_uniform(key, shape, seed) * 0.05
        elif kind in ("bn_w", "ln_w"):
            w = 1.0 + 0.2 * hash_uniform(key, shape, seed)
        elif kind in ("bn_b", "ln_b"):
            w = 0.1 * hash_uniform(key, shape, seed)
        elif kind == "bn_rm":
            w = np.zeros(shape, np.float32)
        elif kind == "bn_rv":
            w = np.ones(shape, np.float32)
        elif kind == "bn_nbt":
            sd[key] = torch.zeros((), dtype=torch.int64)
            continue
        elif kind == "cls":
            w = hash_normal(key, shape, seed)
        else:
            raise KeyError(kind)
        sd[key] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return sd


def synthetic_batch(batch: int, image_size: int, num_joints: int = 21, num_classes: int = 19,
                    seed: int = 42, gaussian_targets: bool = True):
    """Synthetic stand-in for the dataloader contract libs/load.py:46-50,76-94,148-206.

    img ~ N(0,1) [B,3,S,S]; label int64 [B]; target [B,J,S/4,S/4] Gaussians (sigma=2,
    peak 1) at hashed centres; target_weight [B,J,1] in {0,1} (~70 % ones).
    """
    S, hm = image_size, image_size // 4
    img = hash_normal("batch.img", (batch, 3, S, S), seed)
    lab = ((hash_uniform("batch.label", (batch,), seed).astype(np.float64) + 1.0) * 0.5 * num_classes)
    lab = np.clip(lab.astype(np.int64), 0, num_classes - 1)
    tw = (hash_uniform("batch.tw", (batch, num_joints, 1), seed) < 0.4).astype(np.float32)
    if gaussian_targets:
        cx = (hash_uniform("batch.cx", (batch, num_joints), seed) + 1.0) * 0.5 * (hm - 1)
        cy = (hash_uniform("batch.cy", (batch, num_joints), seed) + 1.0) * 0.5 * (hm - 1)
        ys, xs = np.meshgrid(np.arange(hm, dtype=np.float32), np.arange(hm, dtype=np.float32),
                             indexing="ij")
        dx = xs[None, None] - np.round(cx)[..., None, None]
        dy = ys[None, None] - np.round(cy)[..., None, None]
        tgt = np.exp(-(dx * dx + dy * dy) / (2.0 * 2.0 ** 2)).astype(np.float32)
        tgt[(np.abs(dx) > 6) | (np.abs(dy) > 6)] = 0.0   # 13x13 patch, libs/load.py:166-204
    else:
        tgt = (hash_uniform("batch.tgt", (batch, num_joints, hm, hm), seed) + 1.0) * 0.5
    return (torch.from_numpy(img), torch.from_numpy(lab), torch.from_numpy(tgt.astype(np.float32)),
            torch.from_numpy(tw))


# ----------------------------------------------------------------------------
# GELAN backbone  (model/gelan.py)
# ----------------------------------------------------------------------------
def autopad(k: int) -> int:
    """model/gelan.py:5-15 with d=1, p=None: 'same' padding k//2."""
    return k // 2


def conv_bn_act(x: Tensor, sd, prefix: str, k: int, s: int, act: bool, training: bool) -> Tensor:
    """``Conv.forward``: act(bn(conv(x)))  -- model/gelan.py:37-56.

    Conv2d bias=False groups=1 dilation=1 (model/gelan.py:37-45); BatchNorm2d eps 1e-5
    momentum 0.1 (model/gelan.py:46): train mode normalises with the batch mean / biased
    variance and updates running stats with the unbiased variance (in place on ``sd``).
    """
    y = F.conv2d(x, sd[prefix + ".conv.weight"], None, stride=s, padding=autopad(k))
    y = F.batch_norm(y, sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"],
                     sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"],
                     training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
    if training:
        sd[prefix + ".bn.num_batches_tracked"] += 1
    return F.silu(y) if act else y


def res_basic_block(x: Tensor, sd, prefix: str, training: bool) -> Tensor:
    """``ResBasicBlock.forward`` with c1 == c2 (no downsample) -- model/gelan.py:78-87."""
    out = x + conv_bn_act(conv_bn_act(x, sd, prefix + ".cv1", 3, 1, True, training),
                          sd, prefix + ".cv2", 3, 1, False, training)
    return F.silu(out)


def gelan_block(x: Tensor, sd, prefix: str, training: bool) -> Tensor:
    """``GELANBlock.forward`` -- model/gelan.py:137-142; the number of residual blocks per branch
    (``nn.Sequential`` cv2 / cv3, :127-133) is read off the state dict."""
    y = list(conv_bn_act(x, sd, prefix + ".cv1", 1, 1, True, training).chunk(2, 1))
    for br in ("cv2", "cv3"):
        t, i = y[-1], 0
        while f"{prefix}.{br}.{i}.cv1.conv.weight" in sd:
            t = res_basic_block(t, sd, f"{prefix}.{br}.{i}", training)
            i += 1
        y.append(t)
    return conv_bn_act(torch.cat(y, 1), sd, prefix + ".cv4", 1, 1, True, training)


def gelan_net(x: Tensor, sd, training: bool, prefix: str = "encoder") -> Tensor:
    """``GELANNet.forward`` -- model/gelan.py:165-176 (plus the commented-out fourth stage,
    :174-175, when the state dict holds ``down3`` / ``cspelan4``)."""
    x = conv_bn_act(x, sd, prefix + ".conv1", 3, 2, True, training)
    x = conv_bn_act(x, sd, prefix + ".conv2", 3, 2, True, training)
    x = gelan_block(x, sd, prefix + ".cspelan1", training)
    x = conv_bn_act(x, sd, prefix + ".down1", 3, 2, True, training)
    x = gelan_block(x, sd, prefix + ".cspelan2", training)
    x = conv_bn_act(x, sd, prefix + ".down2", 3, 2, True, training)
    x = gelan_block(x, sd, prefix + ".cspelan3", training)
    if prefix + ".down3.conv.weight" in sd:
        x = conv_bn_act(x, sd, prefix + ".down3", 3, 2, True, training)
        x = gelan_block(x, sd, prefix + ".cspelan4", training)
    return x


# ----------------------------------------------------------------------------
# ViT decoder  (model/transformer.py)
# ----------------------------------------------------------------------------
def pos_emb_sincos_2d(h: int, w: int, dim: int, temperature: float = 10000.0) -> Tensor:
    """model/transformer.py:9-26.  NB omega_k = 1/temperature**k with NO division by
    dim/4-1 (lines 17-18): for large k, temperature**k overflows fp32 to inf -> omega=0."""
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    omega = torch.arange(dim // 4, dtype=torch.float32)
    omega = 1.0 / (temperature ** omega)
    y = y.flatten()[:, None] * omega[None, :]
    x = x.flatten()[:, None] * omega[None, :]
    return torch.cat((x.sin(), x.cos(), y.sin(), y.cos()), dim=1).type(torch.float32)


def attention(x: Tensor, sd, p: str) -> Tuple[Tensor, Tensor]:
    """``Attention.forward`` -- model/transformer.py:62-77 (qkv and out projections bias-free)."""
    b, n, _ = x.shape
    xn = F.layer_norm(x, (DIM,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], LN_EPS)
    qkv = F.linear(xn, sd[p + ".to_qkv.weight"]).chunk(3, dim=-1)
    q, k, v = (t.reshape(b, n, HEADS, HEAD_DIM).permute(0, 2, 1, 3) for t in qkv)
    dots = torch.matmul(q, k.transpose(-1, -2)) * (HEAD_DIM ** -0.5)
    attn = torch.softmax(dots, dim=-1)
    out = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(b, n, HEADS * HEAD_DIM)
    return F.linear(out, sd[p + ".to_out.weight"]), attn


def feed_forward(x: Tensor, sd, p: str) -> Tensor:
    """``FeedForward.forward`` -- model/transformer.py:32-42 (exact erf GELU, dropout p=0)."""
    h = F.layer_norm(x, (DIM,), sd[p + ".net.0.weight"], sd[p + ".net.0.bias"], LN_EPS)
    h = F.gelu(F.linear(h, sd[p + ".net.1.weight"], sd[p + ".net.1.bias"]))
    return F.linear(h, sd[p + ".net.4.weight"], sd[p + ".net.4.bias"])


def vit(features: Tensor, sd, prefix: str = "decoder") -> Tuple[Tensor, Tensor, Tensor]:
    """``ViT.forward`` -- model/transformer.py:129-152."""
    b, c, h, w = features.shape
    x = features.permute(0, 2, 3, 1).reshape(b, h * w, c)                 # :132
    x = x + pos_emb_sincos_2d(h, w, c)                                    # :135
    cls = sd[prefix + ".cls_token"].expand(b, -1, -1)                     # :137
    x = torch.cat([cls, x], dim=1)                                        # :139
    attn = None
    for l in range(DEPTH):                                                # :90-94
        p = f"{prefix}.transformer.layers.{l}"
        msg, attn = attention(x, sd, p + ".0")
        x = msg + x
        x = feed_forward(x, sd, p + ".1") + x
    cls_feat, hmap_feat = x[:, 0], x[:, 1:]                               # :142
    cls_out = F.linear(F.layer_norm(cls_feat, (DIM,), sd[prefix + ".mlp_head.0.weight"],
                                    sd[prefix + ".mlp_head.0.bias"], LN_EPS),
                       sd[prefix + ".mlp_head.1.weight"], sd[prefix + ".mlp_head.1.bias"])
    hm = hmap_feat.reshape(b, h, w, c).permute(0, 3, 1, 2)                # :146
    hm = F.interpolate(hm, scale_factor=(4, 4), mode="bilinear", align_corners=True)  # :148
    hm = F.conv2d(F.relu(hm), sd[prefix + ".simple_decoder.1.weight"],
                  sd[prefix + ".simple_decoder.1.bias"])                  # :118-127,150
    return cls_out, hm, attn


def multitasknet_forward(x: Tensor, sd, training: bool = True):
    """``MultiTaskNet.forward`` -- model/multitasknet.py:24-29."""
    feats = gelan_net(x, sd, training)
    feats = F.conv2d(feats, sd["proj.weight"])                            # multitasknet.py:13,26
    return vit(feats, sd)


# ----------------------------------------------------------------------------
# losses and train step  (libs/loss.py, train.py)
# ----------------------------------------------------------------------------
def joints_mse_loss(output: Tensor, target: Tensor, target_weight: Tensor) -> Tensor:
    """``JointsMSELoss(use_target_weight=True)`` -- libs/loss.py:10-30 (the 21-joint loop)."""
    b, j = output.shape[:2]
    pred = output.reshape(b, j, -1)
    gt = target.reshape(b, j, -1)
    loss = 0
    for i in range(j):
        w = target_weight[:, i]
        loss = loss + 0.5 * F.mse_loss(pred[:, i] * w, gt[:, i] * w, reduction="mean")
    return loss / j


def classification_loss(output: Tensor, target: Tensor) -> Tensor:
    """``ClassificationLoss`` -- libs/loss.py:33-40."""
    return F.cross_entropy(output, target, reduction="mean")


def total_loss(cls_out, hmap, label, target, target_weight):
    """train.py:63-64,75: total = 0.001 * CE + JointsMSE."""
    cl = classification_loss(cls_out, label) * 0.001
    jl = joints_mse_loss(hmap, target, target_weight)
    return cl + jl, cl, jl


PARAM_KINDS = ("conv", "bn_w", "bn_b", "lin_w", "lin_b", "ln_w", "ln_b", "cls")


def param_keys(num_joints: int = 21, num_classes: int = 19):
    """``MultiTaskNet.parameters()`` order (114 tensors)."""
    return [k for k, _, kind in state_dict_spec(num_joints, num_classes) if kind in PARAM_KINDS]


def forward_backward(sd, batch, training: bool = True):
    """One forward + loss + backward on the oracle; returns outputs, losses and grads."""
    img, label, target, tw = batch
    keys = [k for k in sd if sd[k].dtype.is_floating_point and not k.endswith(("running_mean",
                                                                              "running_var"))]
    for k in keys:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    cls_out, hmap, attn = multitasknet_forward(img, sd, training)
    tot, cl, jl = total_loss(cls_out, hmap, label, target, tw)
    tot.backward()
    grads = {k: sd[k].grad.detach().clone() for k in keys}
    for k in keys:
        sd[k].requires_grad_(False)
    return (cls_out.detach(), hmap.detach(), attn.detach()), (tot.detach(), cl.detach(), jl.detach()), grads


def adamw_step(sd, grads, state, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
               weight_decay: float = 0.01):
    """torch.optim.AdamW defaults as used by train.py:50-51 (decoupled decay on EVERY tensor)."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    for k, g in grads.items():
        p = sd[k]
        m = state.setdefault("m." + k, torch.zeros_like(p))
        v = state.setdefault("v." + k, torch.zeros_like(p))
        p.mul_(1.0 - lr * weight_decay)
        m.mul_(b1).add_(g, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        bc1 = 1.0 - b1 ** t
        bc2 = 1.0 - b2 ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def train_step(sd, batch, state, lr: float = 1e-3):
    """Forward + 0.001*CE + JMSE + backward + AdamW, the step BASELINE.json's metric times."""
    outs, losses, grads = forward_backward(sd, batch, training=True)
    adamw_step(sd, grads, state, lr)
    return outs, losses, grads
