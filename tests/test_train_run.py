"""train.run: the reference's fit loop (train.py:186-240) -- step metrics, a validation pass per epoch and the best / last
checkpoint policy (train.py:214-221) wired around the fused training step."""
import os

import pytest
import torch

from backend import Backend
from hand_gesture_recognition_amd import train
from hand_gesture_recognition_amd.checkpoint import load_checkpoint
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet


def _run(be, tmp_path, dtype, epochs=3):
    args = train.build_parser().parse_args([
        "--batch_size", "2", "--epochs", str(epochs), "--image_size", "32", "32", "--steps_per_epoch", "1", "--val_steps", "1",
        "--lr_step", "2", "--save_dir", str(tmp_path), "--suffix", "t", "--compute_dtype", dtype])
    model = train.run(args, {"num_joints": 21, "num_classes": 19}, device=be.device)
    return args, model, train.run.history


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_run_loop_metrics_validation_checkpoints_emu(tmp_path, dtype):
    be = Backend("emu")
    args, model, hist = _run(be, tmp_path, dtype)
    assert len(hist) == 3
    for log in hist:
        for k in ("train/total_loss", "train/cls_f1score", "train/pose_acc", "val/total_loss", "val/cls_f1score", "val/pose_acc"):
            assert k in log and log[k] == log[k], (k, log)
        assert 0.0 <= log["train/pose_acc"] <= 1.0 and 0.0 <= log["val/cls_f1score"] <= 1.0
        assert abs(log["train/total_loss"] - log["train/class_loss"] - log["train/joints_loss"]) < 1e-5
    wdir = os.path.join(str(tmp_path), "gelans_32x32_t", "weight")                       # train.py:190-193
    assert os.path.exists(os.path.join(wdir, "last.ckpt")) and os.path.exists(os.path.join(wdir, "best.ckpt"))
    # best = the epoch with the smallest val/total_loss, last = the final weights; both load with the reference's loader recipe
    best_epoch = min(range(3), key=lambda e: hist[e]["val/total_loss"])
    ck = torch.load(os.path.join(wdir, "best.ckpt"), weights_only=True)
    assert ck["epoch"] == best_epoch and abs(ck["best_value"] - hist[best_epoch]["val/total_loss"]) < 1e-9
    fresh = MultiTaskNet(21, 19, [32, 32], compute_dtype=torch.float32)
    last = load_checkpoint(os.path.join(wdir, "last.ckpt"), fresh)
    assert last["epoch"] == 2
    # ... and, as after the reference's trainer.test(module, dm, "best"), the returned module holds the BEST weights
    load_checkpoint(os.path.join(wdir, "best.ckpt"), fresh)
    for (k, a), (_, b) in zip(fresh.state_dict().items(), model.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    # the model is back in training mode with unfolded weights after every validation pass
    assert model.training
    # the test epoch (train.py:239-240) ran on the BEST checkpoint: macro F1 in [0, 1], a confusion matrix of the test labels
    f1, cm = train.run.test_result
    assert 0.0 <= f1 <= 1.0 and cm.shape == (19, 19) and int(cm.sum()) == 2
    import numpy as np
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "gelans_32x32_t", "confusion_matrix.npy")), cm)
    assert os.path.getsize(os.path.join(str(tmp_path), "gelans_32x32_t", "confusion_matrix.npy")) > 0


@pytest.mark.gpu
def test_run_loop_gpu(tmp_path):
    be = Backend("hip")
    _, model, hist = _run(be, tmp_path, "bf16", epochs=2)
    assert len(hist) == 2 and all(h["val/total_loss"] == h["val/total_loss"] for h in hist)
