"""Lightning-compatible checkpoint I/O (reference: train.py:214-221, export.py:34-40)."""
import os
from collections import OrderedDict

import torch

import mtn_oracle as O
from hand_gesture_recognition_amd import checkpoint as C
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet


def test_roundtrip_and_reference_loader(tmp_path):
    sd = O.synthetic_state_dict(21, 19, seed=9)
    a = MultiTaskNet(21, 19, [32, 32])
    a.load_state_dict(sd, strict=True)
    path = C.save_checkpoint(str(tmp_path / "weight" / "best.ckpt"), a, epoch=3, global_step=42)
    # exactly what export.py:34-40 does
    checkpoint = torch.load(path, map_location="cpu", weights_only=False)["state_dict"]
    state_dict = OrderedDict()
    for key in checkpoint.keys():
        state_dict[key.replace("model.", "")] = checkpoint[key]
    b = MultiTaskNet(21, 19, [32, 32])
    b.load_state_dict(state_dict, strict=True)
    assert all(torch.equal(p, q) for p, q in zip(a.state_dict().values(), b.state_dict().values()))
    assert list(checkpoint.keys()) == ["model." + k for k in sd.keys()]
    # a file as the reference's Lightning callback writes it
    ref_file = str(tmp_path / "ref.ckpt")
    torch.save({"epoch": 1, "state_dict": {"model." + k: v for k, v in sd.items()}}, ref_file)
    c = MultiTaskNet(21, 19, [32, 32])
    C.load_checkpoint(ref_file, c)
    assert all(torch.equal(c.state_dict()[k], sd[k]) for k in sd)


def test_best_and_last_policy(tmp_path):
    m = MultiTaskNet(21, 19, [32, 32])
    pol = C.CheckpointPolicy(str(tmp_path / "weight"))
    w1 = pol.step(m, {"val/total_loss": 0.5}, epoch=0)
    w2 = pol.step(m, {"val/total_loss": 0.7}, epoch=1)
    w3 = pol.step(m, {"val/total_loss": 0.3}, epoch=2)
    names = lambda ws: sorted(os.path.basename(w) for w in ws)   # noqa: E731
    assert names(w1) == ["best.ckpt", "last.ckpt"] and names(w2) == ["last.ckpt"] and names(w3) == ["best.ckpt", "last.ckpt"]
    assert pol.best_value == 0.3
    assert torch.load(pol.best_path, weights_only=False)["epoch"] == 2
    assert pol.step(m, {"val/total_loss": 0.1}, epoch=3, is_main=False) == []
