"""Timing ablation (results WRONG): the training step without the 16 ViT Linear weight-gradient calls -> what they cost inside the step.
python tools/probes/skip_linear_wgrad.py [--dtype=bf16x3]"""
import os, sys, time, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hand_gesture_recognition_amd import ops
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
from hand_gesture_recognition_amd.train import TrainStep, synthetic_batch
dtype = "bf16x3" if "--dtype=bf16x3" in sys.argv else torch.bfloat16
dev = torch.device("cuda:0")
real = ops.linear_wgrad
def run(skip):
    ops.linear_wgrad = (lambda *a, **k: None) if skip else real
    torch.manual_seed(42)
    model = MultiTaskNet(21, 19, [192, 192], compute_dtype=dtype).to(dev)
    step = TrainStep(model, 256, lr=1e-3)
    batch = synthetic_batch(256, (192, 192), 21, 19, dev, 1234)
    for _ in range(8): step(*batch)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step(*batch)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    return statistics.median(ts)
for i in range(3):
    a, b = run(False), run(True)
    print(f"{dtype}: step {a:.3f} ms, without the ViT Linear weight gradients {b:.3f} ms: {a - b:.3f} ms", flush=True)
