"""Ceilings before building: the B=256 training step with chosen BatchNorm passes SKIPPED -- wrong
numerics, timing only -- interleaved with the unmodified step in ONE process on one box.

  python tools/ablate_step.py [--dtype=bf16|bf16x3] [--rounds=3]

Groups (unit tags of engine.py):
  t       bn_act_fwd of the six c1 units (output `t` of a residual block's first conv: read only by its c2)
  single  bn_act_fwd of the other single-consumer activations: conv2 (-> s1.cv1), s*.cv4 (-> down / proj), s*.down (-> next cv1)
  multi   bn_act_fwd of the multi-consumer ones: s*.cv1 (concat halves), c2 units (residual add + SiLU into the concat)
  bwd_t / bwd_single / bwd_multi   the whole BatchNorm backward (reduce + finalize + apply) of the same groups
What a skipped pass leaves behind is the buffer's previous contents (a valid activation of an earlier step), so the
consumers read the same number of bytes as before."""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
from hand_gesture_recognition_amd.train import TrainStep, synthetic_batch

opt = {a[2:].split("=")[0]: a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "bf16x3": "bf16x3"}[opt.get("dtype", "bf16")]
B, S, rounds = int(opt.get("batch", 256)), int(opt.get("size", 192)), int(opt.get("rounds", 3))
dev = torch.device("cuda:0")
torch.manual_seed(42)
model = MultiTaskNet(21, 19, [S, S], compute_dtype=dtype).to(dev)
step = TrainStep(model, B, lr=1e-3)
eng = step.eng
batch = synthetic_batch(B, (S, S), 21, 19, dev, 1234)
tags = [u.tag for u in eng.units if u.bn and u.tag != "conv1"]
grp = {"t": [t for t in tags if t.endswith(".c1")],
       "single": [t for t in tags if t == "conv2" or t.endswith(".cv4") or t.endswith(".down")],
       "multi": [t for t in tags if t.endswith(".cv1") or t.endswith(".c2")]}
assert sorted(sum(grp.values(), [])) == sorted(tags)
cases = [("baseline", (), ())] + [(k, grp[k], ()) for k in grp] + [("bwd_" + k, (), grp[k]) for k in grp] + \
        [("all_fwd", tags, ()), ("all_bwd", (), tags)]
for _ in range(8):
    step(*batch)
res = {c[0]: [] for c in cases}
for r in range(rounds):
    for name, f, b in cases:
        eng.ablate_fwd, eng.ablate_bwd = frozenset(f), frozenset(b)
        for _ in range(3):
            step(*batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            step(*batch)
        torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 20 * 1e3)
    # (AdamW on garbage gradients drifts the weights: timing does not depend on values, but keep them finite)
    eng.ablate_fwd = eng.ablate_bwd = frozenset()
base = statistics.median(res["baseline"])
print(f"# {opt.get('dtype', 'bf16')} B={B} {S}x{S}: median ms/step over {rounds} interleaved rounds of 20 steps (skipped passes: wrong numerics)")
for name, f, b in cases:
    m = statistics.median(res[name])
    print(f"{name:12s} {len(f) + len(b):3d} units  {m:8.3f} ms  {m - base:+.3f}", flush=True)
