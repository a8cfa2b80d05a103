"""Race screen: the same training step from the same state, repeated; every repetition must give
bit-identical losses, gradients, BatchNorm statistics and updated weights (all reductions are
ordered, no float atomics -- any difference means a data race or an uninitialised read).

usage (GPU box): python tools/soak_determinism.py [--reps 30] [--batch 256] [--size 192]
"""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet  # noqa: E402
from hand_gesture_recognition_amd.train import TrainStep, synthetic_batch  # noqa: E402


def digest(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().contiguous().cpu().view(torch.uint8).numpy().tobytes())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--size", type=int, default=192)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    dt = {"bf16": torch.bfloat16, "fp32": torch.float32, "bf16x3": "bf16x3"}[a.dtype]
    model = MultiTaskNet(21, 19, [a.size, a.size], compute_dtype=dt).to(dev)
    step = TrainStep(model, a.batch, lr=1e-3)
    batch = synthetic_batch(a.batch, (a.size, a.size), 21, 19, dev, 99)
    keep = [model.flat_params, step.m, step.v, step.state] + list(model.buffers())
    saved = [t.clone() for t in keep]
    seen = {}
    for i in range(a.reps):
        with torch.no_grad():
            for t, s in zip(keep, saved):
                t.copy_(s)
        losses = step(*batch)
        torch.cuda.synchronize()
        d = digest(losses, model.flat_grads, model.flat_params, *[b for b in model.buffers() if b.dtype.is_floating_point])
        seen[d] = seen.get(d, 0) + 1
    print(f"{a.reps} repetitions, B={a.batch} {a.size}x{a.size} {a.dtype}: {len(seen)} distinct result(s) {seen}")
    sys.exit(0 if len(seen) == 1 else 1)


if __name__ == "__main__":
    main()
