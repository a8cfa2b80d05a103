"""The oracle (oracle/mtn_oracle.py) against the golden vectors that oracle/gen_golden.py
produced from the reference itself.  CPU only."""
import os

import numpy as np
import pytest
import torch

import mtn_oracle as O


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("name", ["config1_b4_s192", "small_b2_s64", "tiny_b3_s32"])
def test_oracle_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    B, S, seed = (int(v) for v in g["meta"])
    torch.set_num_threads(8)
    sd = O.synthetic_state_dict(21, 19, seed=seed)
    batch = O.synthetic_batch(B, S, seed=42 + seed)
    state = {}
    (cls_o, hm_o, attn_o), losses, grads = O.train_step(sd, batch, state, float(g["lr"]))
    tol = 2e-5   # same ATen ops -> normally bit-identical; leave room for thread-count effects
    assert _rel(cls_o, g["cls_out"]) < tol
    assert _rel(hm_o, g["hmap_out"]) < tol
    assert _rel(attn_o[:, :, 0, :], g["attn_row0"]) < tol
    assert _rel(attn_o[0, 0], g["attn_b0h0"]) < tol
    assert _rel([float(x) for x in losses], g["losses"]) < tol
    names = [str(n) for n in g["param_names"]]
    assert names == O.param_keys()
    gn = np.array([float(grads[k].double().norm()) for k in names])
    assert _rel(gn, g["grad_norms"]) < 1e-4
    for k in g.files:
        if k.startswith("grad."):
            assert _rel(grads[k[5:]], g[k]) < 1e-4, k
    bn = np.concatenate([sd[str(k)].numpy().ravel() for k in g["bn_keys"]])
    assert _rel(bn, g["bn_running"]) < tol
    wn = np.array([float(sd[k].double().norm()) for k in names])
    assert _rel(wn, g["step_weight_norms"]) < tol
    assert _rel(sd["decoder.cls_token"], g["step.decoder.cls_token"]) < tol


def test_state_dict_layout():
    spec = O.state_dict_spec(21, 19)
    assert len(spec) == 180
    assert len(O.param_keys()) == 114
    sd = O.synthetic_state_dict()
    n = sum(v.numel() for k, v in sd.items() if k in set(O.param_keys()))
    assert n == 7_409_000


def test_joints_loss_closed_form():
    """the 21-joint loop equals 0.5*mean((w*(p-t))^2)."""
    torch.manual_seed(0)
    p, t = torch.randn(3, 21, 8, 8), torch.rand(3, 21, 8, 8)
    w = (torch.rand(3, 21, 1) > 0.3).float()
    ref = O.joints_mse_loss(p, t, w)
    closed = 0.5 * ((w[..., None] * (p - t)) ** 2).mean()
    assert abs(float(ref) - float(closed)) < 1e-7


def test_pos_emb_degenerate_frequencies():
    """omega_k = 1/10000**k overflows to 0 for large k (sin=0, cos=1)."""
    pe = O.pos_emb_sincos_2d(12, 12, 256)
    assert pe.shape == (144, 256)
    assert float(pe[:, 20:64].abs().max()) == 0.0       # sin(x*0)
    assert float((pe[:, 64 + 20:128] - 1).abs().max()) == 0.0   # cos(x*0)
