"""Checkpoint I/O compatible with the reference's Lightning files.

The reference saves with ``ModelCheckpoint(dirpath=<save>/weight, filename='best',
monitor='val/total_loss', mode='min', save_top_k=1, save_last=True, save_weights_only=True)``
(train.py:214-221) and loads with ``torch.load(path)['state_dict']`` + stripping the ``model.``
prefix + ``load_state_dict(strict=True)`` (export.py:34-40).  Files written here open with that
loader, and files written by the reference open here.
"""
import os
from collections import OrderedDict

import torch

PREFIX = "model."          # LightningModule attribute name of the network (train.py:33)


def lightning_state_dict(model):
    return OrderedDict((PREFIX + k, v.detach().cpu().clone()) for k, v in model.state_dict().items())


def save_checkpoint(path, model, epoch=0, global_step=0, extra=None):
    """Weights-only Lightning-style checkpoint: {'epoch', 'global_step', 'state_dict': {'model.<key>': tensor}}."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ckpt = {"epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": "2.0.0",
            "state_dict": lightning_state_dict(model)}
    if extra:
        ckpt.update(extra)
    tmp = path + ".tmp"
    torch.save(ckpt, tmp)
    os.replace(tmp, path)
    return path


def load_checkpoint(path, model, map_location="cpu", trusted=False):
    """The loader of export.py:34-40: strict load of a (reference- or self-written) checkpoint.

    Files are unpickled with ``weights_only=True`` (tensors and primitives only -- what a
    Lightning weights checkpoint holds): a third-party file cannot run code here.  A full
    Lightning checkpoint with pickled callbacks / hyper-parameter objects needs ``trusted=True``,
    which is the reference's own ``torch.load`` behaviour (export.py:34) and executes the pickle."""
    ckpt = torch.load(path, map_location=map_location, weights_only=not trusted)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    clean = OrderedDict((k[len(PREFIX):] if k.startswith(PREFIX) else k, v) for k, v in sd.items())
    model.load_state_dict(clean, strict=True)
    return ckpt


class CheckpointPolicy:
    """best (top-1 on a monitored value) + last, as configured at train.py:214-221."""

    def __init__(self, dirpath, filename="best", monitor="val/total_loss", mode="min", save_last=True):
        assert mode in ("min", "max")
        self.dir, self.filename, self.monitor, self.mode, self.save_last = dirpath, filename, monitor, mode, save_last
        self.best_value, self.best_path = None, None

    def _better(self, v):
        return self.best_value is None or (v < self.best_value if self.mode == "min" else v > self.best_value)

    def step(self, model, metrics, epoch, global_step=0, is_main=True):
        """Call at the end of a validation epoch; returns the paths written (rank 0 only under DDP)."""
        written = []
        if not is_main:
            return written
        if self.save_last:
            written.append(save_checkpoint(os.path.join(self.dir, "last.ckpt"), model, epoch, global_step))
        v = float(metrics[self.monitor])
        if self._better(v):
            self.best_value = v
            self.best_path = save_checkpoint(os.path.join(self.dir, self.filename + ".ckpt"), model, epoch, global_step,
                                             {"monitor": self.monitor, "best_value": v})
            written.append(self.best_path)
        return written
