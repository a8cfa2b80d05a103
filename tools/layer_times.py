"""Per-call GPU time of one training step, grouped by entry point and shape.

Wraps ``lib.call`` with HIP events (recorded on the launch stream) for one eager step at the
BASELINE config-2 size, then prints a table sorted by total time.  Event bracketing serialises
nothing, but the bracket of a short call includes launch gaps: read it as an upper bound.

Usage (GPU box): python tools/layer_times.py [--batch 256] [--top 60]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand_gesture_recognition_amd import lib  # noqa: E402
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet  # noqa: E402
from hand_gesture_recognition_amd.train import TrainStep, synthetic_batch  # noqa: E402

# positions of the integer shape arguments of the GEMM-like entry points
_SHAPE = {
    "mtn_conv_fwd": (7, 14), "mtn_conv_dgrad": (8, 15), "mtn_conv_wgrad": (8, 15),
    "mtn_linear_fwd": (12, 15), "mtn_linear_wgrad": (8, 11),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--top", type=int, default=70)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    model = MultiTaskNet(21, 19, [192, 192], compute_dtype=torch.bfloat16).to(dev)
    step = TrainStep(model, a.batch, lr=1e-3, use_graph=False)
    batch = synthetic_batch(a.batch, (192, 192), 21, 19, dev, 1234)
    for _ in range(3):
        step(*batch)
    torch.cuda.synchronize()

    records = []
    real_call = lib.call

    def timed_call(name, *args):
        if name.endswith(("_bytes", "_rows", "_tiles", "_len", "_version", "_set")):
            return real_call(name, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_call(name, *args)
        e1.record()
        lo_hi = _SHAPE.get(name)
        shape = tuple(args[lo_hi[0]:lo_hi[1]]) if lo_hi else ()
        records.append((name, shape, e0, e1))
        return r

    lib.call = timed_call          # ops.py and engine.py call through the module attribute
    step(*batch)
    torch.cuda.synchronize()
    lib.call = real_call

    agg = collections.OrderedDict()
    for name, shape, e0, e1 in records:
        k = (name, shape)
        t, n = agg.get(k, (0.0, 0))
        agg[k] = (t + e0.elapsed_time(e1) * 1e3, n + 1)
    total = sum(t for t, _ in agg.values())
    print(f"# {len(records)} calls, sum of brackets {total/1e3:.2f} ms")
    for (name, shape), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print(f"{name:22s} {str(shape):44s} x{n:3d} {t:9.1f} us  {t/n:8.1f} us/call  {100*t/total:5.1f}%")


if __name__ == "__main__":
    main()
