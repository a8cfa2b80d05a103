"""On-device step metrics / target rendering against the metrics oracle and the reference golden."""
import os

import numpy as np
import pytest
import torch

import mtn_metrics_oracle as M
from backend import be, rel_err  # noqa: F401
from hand_gesture_recognition_amd import metrics


def test_metrics_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    out, tgt = g["out"].astype(np.float32), g["tgt"].astype(np.float32)
    preds, maxv = M.get_max_preds(out)
    acc, avg, cnt, _ = M.pose_accuracy(out, tgt)
    assert np.array_equal(preds, g["preds"]) and np.array_equal(maxv, g["maxvals"])
    assert np.allclose(acc, g["acc"]) and cnt == int(g["cnt"]) and abs(avg - float(g["avg"])) < 1e-12


def test_pose_metrics_kernels(be, golden_dir):
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    out = torch.from_numpy(g["out"].astype(np.float32)).to(be.device)
    tgt = torch.from_numpy(g["tgt"].astype(np.float32)).to(be.device)
    preds, maxv = metrics.get_max_preds(out)
    acc, avg, cnt, pred = metrics.pose_accuracy(out, tgt)
    be.sync()
    assert np.array_equal(preds.cpu().numpy(), g["preds"])
    assert np.allclose(maxv.cpu().numpy(), g["maxvals"])
    assert np.allclose(acc.cpu().numpy(), g["acc"], atol=1e-6)
    assert int(cnt) == int(g["cnt"]) and abs(float(avg) - float(g["avg"])) < 1e-6


def test_classification_meter(be):
    from sklearn.metrics import confusion_matrix, f1_score
    gen = torch.Generator().manual_seed(3)
    C = 19
    meter = metrics.ClassificationMeter(C, be.device)
    yp, yt = [], []
    for _ in range(3):
        logits = torch.randn(50, C, generator=gen)
        label = torch.randint(0, 12, (50,), generator=gen)        # some classes never appear as labels
        pred = meter.update(logits.to(be.device), label.to(be.device))
        be.sync()
        assert torch.equal(pred.cpu(), logits.argmax(1))
        yp += pred.cpu().tolist()
        yt += label.tolist()
    f1 = float(meter.f1())
    assert abs(f1 - f1_score(yp, yt, average="macro")) < 1e-6            # argument order of train.py:67
    assert abs(f1 - M.macro_f1_from_confusion(meter.cm.cpu().numpy())) < 1e-6
    assert np.array_equal(meter.cm.cpu().numpy(), confusion_matrix(yp, yt, labels=list(range(C))))


def test_test_epoch_confusion_matrix_orientation(be, tmp_path):
    """the matrix ``train.test`` saves is the reference's ``confusion_matrix(y_pred, y_true)``
    (train.py:181: rows = predicted class, columns = label), pinned with an asymmetric pair -- every sample predicted
    as class 2, labelled 0 or 1."""
    from sklearn.metrics import confusion_matrix
    from hand_gesture_recognition_amd import train as T
    C = 4

    class _Eng:                                        # (the test epoch only needs forward() to leave logits behind)
        def forward(self, img, training, want_attn):
            self.logits = img

    logits = torch.full((6, C), -1.0)
    logits[:, 2] = 3.0
    label = torch.tensor([0, 0, 0, 0, 1, 1])
    model = torch.nn.Identity()
    f1, cm = T.test(model, _Eng(), [(logits.to(be.device), label.to(be.device), None, None)], C, out_dir=str(tmp_path))
    ref = confusion_matrix([2] * 6, label.tolist(), labels=list(range(C)))
    assert np.array_equal(cm, ref) and cm[2, 0] == 4 and cm[2, 1] == 2 and cm[0, 2] == 0
    assert np.array_equal(np.load(str(tmp_path / "confusion_matrix.npy")), ref)


def test_classification_meter_guards_labels(be):
    C = 5
    meter = metrics.ClassificationMeter(C, be.device)
    logits = torch.randn(6, C, generator=torch.Generator().manual_seed(1)).to(be.device)
    with pytest.raises(TypeError):
        meter.update(logits, torch.zeros(6, dtype=torch.int32, device=be.device))
    with pytest.raises(TypeError):
        meter.update(logits, torch.zeros(5, dtype=torch.int64, device=be.device))
    label = torch.tensor([0, 4, -1, 5, 1 << 40, 2], device=be.device)   # three labels outside [0, C)
    meter.update(logits, label)
    be.sync()
    assert int(meter.cm.sum()) == 3


def test_render_targets_match_the_reference_vectors(be, golden_dir):
    """SURVEY 8f row 2 pinned: tests/golden/targets.npz holds inputs and outputs of the REFERENCE's own
    ``HagridDataset.generate_target`` (libs/load.py:148-206; oracle/gen_golden_targets.py compiled that method from the
    reference's file in the build container): borders and corners, both sides of the in / out test, invisible joints,
    v > 0.5, round-half-up of the centre, negative coordinates, non-square maps, sigma 1 / 2 / 3, the 192 x 192 default."""
    g = np.load(os.path.join(golden_dir, "targets.npz"))
    assert len(g["names"]) >= 15
    for name in g["names"]:
        joints, vis = g[f"{name}.joints"], g[f"{name}.vis"]
        iw, ih = (int(v) for v in g[f"{name}.image_size"])             # the reference's image_size[0] goes with x
        sigma = int(g[f"{name}.sigma"])
        rt, rw = g[f"{name}.target"], g[f"{name}.weight"]
        # the restatement the other CPU tests use is the same function
        ot, ow = M.generate_target(joints.astype(np.float64), vis.astype(np.float64)[:, None], (iw, ih), (iw // 4, ih // 4), sigma)
        assert np.array_equal(ot, rt) and np.array_equal(ow, rw), name
        tgt, w = metrics.render_targets(torch.from_numpy(joints)[None].to(be.device), torch.from_numpy(vis)[None].to(be.device),
                                        (ih, iw), sigma)
        be.sync()
        tgt, w = tgt[0].cpu().numpy(), w[0].cpu().numpy()
        assert tgt.shape == rt.shape and np.array_equal(w, rw), name
        assert np.array_equal(tgt != 0, rt != 0), name                 # the same pixels are drawn
        assert np.allclose(tgt, rt, atol=1e-6, rtol=0), (name, float(np.abs(tgt - rt).max()))


def test_render_targets_and_normalize(be):
    rng = np.random.default_rng(5)
    B, J, S = 3, 21, 64
    joints = rng.uniform(-20, S + 20, size=(B, J, 2)).astype(np.float32)
    joints[0, 0] = [0.0, 0.0]
    joints[0, 1] = [S - 1, S - 1]
    joints[0, 2] = [-200.0, 10.0]                                  # Gaussian fully outside -> weight 0
    vis = (rng.uniform(size=(B, J)) > 0.3).astype(np.float32)
    tgt, w = metrics.render_targets(torch.from_numpy(joints).to(be.device), torch.from_numpy(vis).to(be.device), (S, S), 2)
    be.sync()
    for b in range(B):
        rt, rw = M.generate_target(joints[b], vis[b][:, None], (S, S), (S // 4, S // 4), 2)
        assert np.allclose(tgt[b].cpu().numpy(), rt, atol=1e-6)
        assert np.array_equal(w[b].cpu().numpy(), rw)
    img = torch.from_numpy(rng.integers(0, 256, size=(2, 8, 12, 3), dtype=np.uint8))
    out = metrics.normalize_image(img.to(be.device))
    be.sync()
    ref = (img.float() / 255.0 - torch.tensor([0.485, 0.456, 0.406])) / torch.tensor([0.229, 0.224, 0.225])
    assert rel_err(out, ref.permute(0, 3, 1, 2)) < 1e-6
