"""CPU oracle for the step metrics and the heat-map target rendering -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's host-side evaluation code:
  get_max_preds      libs/utils.py:4-32
  calc_dists/dist_acc/pose_accuracy   libs/metrics.py:6-62
  macro F1 as train.py:67-69 computes it: sklearn f1_score(pred, label, average='macro')
  generate_target    libs/load.py:148-206
Pinning: get_max_preds / pose_accuracy are checked against the reference's own functions (they
import with numpy only) by oracle/gen_golden_metrics.py -> tests/golden/metrics.npz.
generate_target lives in libs/load.py, whose module imports cv2 / lightning / torchvision (absent here); the method
itself needs only numpy: oracle/gen_golden_targets.py compiles that one function node from the reference's file (build
container only), checks this restatement returns identical arrays on 15 edge-case sets and writes the reference's
inputs / outputs to tests/golden/targets.npz (round 5: pinned).
Only tests/ may import this file.
"""
import numpy as np


def get_max_preds(batch_heatmaps):
    """libs/utils.py:4-32: argmax location (x, y) per joint, zeroed where the max is <= 0."""
    b, j, _, w = batch_heatmaps.shape
    flat = batch_heatmaps.reshape(b, j, -1)
    idx = np.argmax(flat, 2)
    maxvals = np.amax(flat, 2).reshape(b, j, 1)
    preds = np.zeros((b, j, 2), np.float32)
    preds[:, :, 0] = idx % w
    preds[:, :, 1] = np.floor(idx / w)
    preds *= (maxvals > 0.0).astype(np.float32)
    return preds, maxvals


def pose_accuracy(output, target, thr=0.5):
    """libs/metrics.py:31-62 (+ calc_dists :6-18, dist_acc :21-28)."""
    num_joints = output.shape[1]
    pred, _ = get_max_preds(output)
    tgt, _ = get_max_preds(target)
    h, w = output.shape[2], output.shape[3]
    norm = np.ones((pred.shape[0], 2)) * np.array([h, w]) / 10
    dists = np.zeros((num_joints, pred.shape[0]))
    for n in range(pred.shape[0]):
        for c in range(num_joints):
            if tgt[n, c, 0] > 1 and tgt[n, c, 1] > 1:
                dists[c, n] = np.linalg.norm(pred[n, c].astype(np.float32) / norm[n] - tgt[n, c].astype(np.float32) / norm[n])
            else:
                dists[c, n] = -1
    acc = np.zeros(num_joints + 1)
    avg, cnt = 0.0, 0
    for i in range(num_joints):
        valid = dists[i] != -1
        acc[i + 1] = (dists[i][valid] < thr).sum() / valid.sum() if valid.sum() > 0 else -1
        if acc[i + 1] >= 0:
            avg += acc[i + 1]
            cnt += 1
    avg = avg / cnt if cnt else 0
    if cnt:
        acc[0] = avg
    return acc, avg, cnt, pred


def macro_f1_from_confusion(cm):
    """Macro F1 over the labels present in y_true or y_pred (sklearn semantics used at train.py:67-69);
    cm[p][t] counts samples predicted p with label t.  F1 is symmetric in (pred, true)."""
    cm = np.asarray(cm, np.float64)
    tp = np.diag(cm)
    fp = cm.sum(1) - tp
    fn = cm.sum(0) - tp
    present = (cm.sum(0) + cm.sum(1)) > 0
    den = 2 * tp + fp + fn
    f1 = np.where(den > 0, 2 * tp / np.maximum(den, 1), 0.0)
    return float(f1[present].mean()) if present.any() else 0.0


def generate_target(joints, joints_vis, image_size, heatmap_size, sigma=2):
    """libs/load.py:148-206 for one sample: joints [J,>=2] in image pixels, joints_vis [J,>=1]."""
    nj = joints.shape[0]
    target = np.zeros((nj, heatmap_size[1], heatmap_size[0]), np.float32)
    weight = np.ones((nj, 1), np.float32)
    weight[:, 0] = joints_vis[:, 0]
    tmp = sigma * 3
    for j in range(nj):
        stride = [i / h for i, h in zip(image_size, heatmap_size)]
        mu_x = int(joints[j][0] / stride[0] + 0.5)
        mu_y = int(joints[j][1] / stride[1] + 0.5)
        ul = [int(mu_x - tmp), int(mu_y - tmp)]
        br = [int(mu_x + tmp + 1), int(mu_y + tmp + 1)]
        if ul[0] >= heatmap_size[0] or ul[1] >= heatmap_size[1] or br[0] < 0 or br[1] < 0:
            weight[j] = 0
            continue
        size = 2 * tmp + 1
        x = np.arange(0, size, 1, np.float32)
        y = x[:, None]
        x0 = y0 = size // 2
        g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
        gx = max(0, -ul[0]), min(br[0], heatmap_size[0]) - ul[0]
        gy = max(0, -ul[1]), min(br[1], heatmap_size[1]) - ul[1]
        ix = max(0, ul[0]), min(br[0], heatmap_size[0])
        iy = max(0, ul[1]), min(br[1], heatmap_size[1])
        if weight[j] > 0.5:
            target[j][iy[0]:iy[1], ix[0]:ix[1]] = g[gy[0]:gy[1], gx[0]:gx[1]]
    return target, weight
