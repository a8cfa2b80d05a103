"""Loss trajectories of the same training run on five engines: the CPU oracle (PyTorch-CPU restatement of the reference)
in float64 -- the truth every other curve is measured against -- and in float32 (the reference's own arithmetic), and the
HIP path in its three modes (exact-f32 MFMA, split-bf16 "bf16x3", bf16).  The distance between the two ORACLE curves
is what rounding at 2^-24 alone grows into over the run (the dynamics amplify any perturbation); a HIP mode whose curve
stays as close to the truth as the fp32 oracle does trains like the reference.  Same synthetic weights, the same four
synthetic batches cycled, AdamW lr 1e-3, train-mode BatchNorm.  Not a parity gate (those are in tests/): a check that the
throughput mode TRAINS like the reference arithmetic over many steps, not only for one.
usage (GPU box): python tools/convergence_check.py [--steps 120] [--batch 16] [--size 192]"""
import argparse, os, sys, time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mtn_oracle as O  # noqa: E402
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet  # noqa: E402
from hand_gesture_recognition_amd.train import TrainStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=120)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=192)
ap.add_argument("--lr", type=float, default=1e-3)
ap.add_argument("--bf16-f32-slabs", action="store_true",
                help="also run the bf16 mode with f32 split-K partial tiles (debug key 28 = 0): ADVICE round 4 asked which of the two "
                     "slab types the lr 1e-3 spike of the bf16 curve belongs to")
ap.add_argument("--no-oracle", action="store_true", help="skip the two CPU oracle runs: deviations are then measured against HIP fp32")
a = ap.parse_args()
if a.bf16_f32_slabs:
    os.environ["MTN_ALLOW_DEBUG_KEYS"] = "1"
dev = torch.device("cuda", 0)
torch.set_num_threads(min(16, os.cpu_count() or 1))
batches = [O.synthetic_batch(a.batch, a.size, seed=100 + i) for i in range(4)]

curves = {}
t0 = time.time()
for name, f64 in (() if a.no_oracle else (("oracle f64 (truth)", True), ("oracle fp32 (CPU)", False))):
    sd = O.synthetic_state_dict(21, 19, seed=7)
    bs = batches
    if f64:
        sd = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
        bs = [tuple(t.double() if t.dtype.is_floating_point else t for t in b) for b in batches]
    st = {}
    curves[name] = []
    for i in range(a.steps):
        _, losses, _ = O.train_step(sd, bs[i % 4], st, a.lr)
        curves[name].append([float(x) for x in losses])
t_cpu = time.time() - t0
from hand_gesture_recognition_amd import lib  # noqa: E402
runs = [("HIP fp32", torch.float32, None), ("HIP bf16x3", "bf16x3", None), ("HIP bf16", torch.bfloat16, None)]
if a.bf16_f32_slabs:
    runs.append(("HIP bf16, f32 slabs", torch.bfloat16, 0))
for name, dt, key28 in runs:
    if key28 is not None:
        lib.call("mtn_debug_set", 28, key28)
    model = MultiTaskNet(21, 19, [a.size, a.size], compute_dtype=dt)
    model.load_state_dict(O.synthetic_state_dict(21, 19, seed=7), strict=True)
    model.to(dev).train()
    step = TrainStep(model, a.batch, lr=a.lr)
    dbs = [tuple(t.to(dev) for t in b) for b in batches]
    curves[name] = []
    for i in range(a.steps):
        curves[name].append([float(x) for x in step(*dbs[i % 4]).cpu()])
    if key28 is not None:
        lib.call("mtn_debug_set", 28, 1)

truth = "HIP fp32" if a.no_oracle else "oracle f64 (truth)"
ref = curves[truth]
print(f"# {a.steps} AdamW steps (lr {a.lr}), B={a.batch}, {a.size}x{a.size}, 4 synthetic batches cycled; both oracle runs on {torch.get_num_threads()} CPU threads: {t_cpu:.0f} s")
print(f"{'step':>5s} " + " ".join(f"{k:>20s}" for k in curves) + "    (total loss = 0.001*CE + JointsMSE)")
for i in sorted(set(list(range(0, a.steps, max(1, a.steps // 12))) + [a.steps - 1])):
    print(f"{i:5d} " + " ".join(f"{curves[k][i][0]:20.6f}" for k in curves))
print(f"# relative deviation of the total loss from {truth}: over the first 25 % / 50 % / 100 % of the run, and the first step it exceeds 1e-3 / 1e-2")
for k in curves:
    if k == truth:
        continue
    d = [abs(c[0] - r[0]) / abs(r[0]) for c, r in zip(curves[k], ref)]
    first = lambda t: next((i for i, v in enumerate(d) if v > t), None)
    n = len(d)
    print(f"# {k:>18s}: {max(d[:max(1, n // 4)]):.2e} / {max(d[:max(1, n // 2)]):.2e} / {max(d):.2e};  first > 1e-3 at step {first(1e-3)}, > 1e-2 at step {first(1e-2)};  "
          f"final {curves[k][-1][0]:.6f} vs {ref[-1][0]:.6f}")
