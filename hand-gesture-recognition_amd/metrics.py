"""Step metrics and input-side rendering on the device.

Mirrors ``libs/utils.py:get_max_preds``, ``libs/metrics.py:pose_accuracy`` and the macro-F1 of
``train.py:66-73`` without the per-step ``.cpu()`` synchronisation of the reference, and the
dataloader's target rendering / image normalisation (``libs/load.py:46-50,148-206``).
"""
import torch

from . import lib


def _s(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def get_max_preds(batch_heatmaps):
    """[B,J,H,W] f32 -> (preds [B,J,2] = (x, y) of the maximum, maxvals [B,J,1])."""
    hm = batch_heatmaps.contiguous().float()
    B, J, H, W = hm.shape
    preds = torch.empty(B, J, 2, dtype=torch.float32, device=hm.device)
    maxv = torch.empty(B, J, 1, dtype=torch.float32, device=hm.device)
    lib.call("mtn_heatmap_argmax", hm.data_ptr(), preds.data_ptr(), maxv.data_ptr(), B * J, H, W, _s(hm))
    return preds, maxv


def pose_accuracy(output, target, thr=0.5):
    """(acc [J+1], avg_acc, cnt, pred) like libs/metrics.py:31-62 -- all device tensors, no sync."""
    pred, _ = get_max_preds(output)
    tgt, _ = get_max_preds(target)
    B, J, H, W = output.shape
    acc = torch.empty(J + 2, dtype=torch.float32, device=output.device)
    lib.call("mtn_pose_accuracy", pred.data_ptr(), tgt.data_ptr(), acc.data_ptr(), B, J, H, W, thr, _s(output))
    return acc[:J + 1], acc[0], acc[J + 1], pred


class ClassificationMeter:
    """Running confusion matrix + macro F1 (sklearn semantics of train.py:67-69,176-181)."""

    def __init__(self, num_classes, device):
        self.C = num_classes
        self.cm = torch.zeros(num_classes, num_classes, dtype=torch.int32, device=device)
        self._f1 = torch.zeros(1, dtype=torch.float32, device=device)

    def reset(self):
        self.cm.zero_()

    def update(self, logits, label):
        logits = logits.contiguous().float()
        if logits.dim() != 2 or logits.shape[1] != self.C:
            raise ValueError(f"logits: expected [B,{self.C}], got {tuple(logits.shape)}")
        if label.dtype != torch.int64 or tuple(label.shape) != (logits.shape[0],):
            raise TypeError(f"label: expected int64 [{logits.shape[0]}], got {label.dtype} {tuple(label.shape)}")
        label = label.contiguous()          # a label outside [0, C) is not counted (kernel guard)
        pred = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
        lib.call("mtn_cls_confusion", logits.data_ptr(), label.data_ptr(), pred.data_ptr(),
                 self.cm.data_ptr(), logits.shape[0], self.C, _s(logits))
        return pred

    def f1(self):
        lib.call("mtn_macro_f1", self.cm.data_ptr(), self._f1.data_ptr(), self.C, _s(self.cm))
        return self._f1[0]


def render_targets(joints_xy, vis, image_size, sigma=2):
    """joints [B,J,2] image pixels, vis [B,J] -> (target [B,J,H/4,W/4], target_weight [B,J,1])."""
    B, J, _ = joints_xy.shape
    HH, HW = image_size[0] // 4, image_size[1] // 4
    j = joints_xy.contiguous().float()
    v = vis.contiguous().float()
    tgt = torch.empty(B, J, HH, HW, dtype=torch.float32, device=j.device)
    w = torch.empty(B, J, 1, dtype=torch.float32, device=j.device)
    lib.call("mtn_render_targets", j.data_ptr(), v.data_ptr(), tgt.data_ptr(), w.data_ptr(), B * J, HH, HW,
             image_size[0], image_size[1], sigma, _s(j))
    return tgt, w


def normalize_image(img_u8_hwc):
    """uint8 [B,H,W,3] (cv2 channel order) -> fp32 [B,3,H,W] normalised like the reference transform."""
    assert img_u8_hwc.dtype == torch.uint8 and img_u8_hwc.shape[-1] == 3
    x = img_u8_hwc.contiguous()
    B, H, W, _ = x.shape
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=x.device)
    lib.call("mtn_normalize_image", x.data_ptr(), out.data_ptr(), B, H, W, _s(x))
    return out
