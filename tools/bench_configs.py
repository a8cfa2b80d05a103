"""BASELINE.json configs 4 and 5 on one MI355X (reporting lines, not the driver's bench):
  config 4: MultiTaskNet 384x384 (N = 577 tokens), B=128, bf16 train step -> img/s + attention kernels' TFLOP/s
  config 5: GELAN backbone only, 192x192, B=512, bf16 forward+backward with a synthetic upstream
            gradient -> img/s (HBM GB/s comes from the rocprofv3 PMC passes of this command)
usage: python tools/bench_configs.py [4|5]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand_gesture_recognition_amd import ops  # noqa: E402
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet  # noqa: E402
from hand_gesture_recognition_amd.train import TrainStep, synthetic_batch  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def config4():
    dev = torch.device("cuda:0")
    B, S = 128, 384
    torch.manual_seed(42)
    model = MultiTaskNet(21, 19, [S, S], compute_dtype=torch.bfloat16).to(dev)
    step = TrainStep(model, B, lr=1e-3)
    batch = synthetic_batch(B, (S, S), 21, 19, dev, 1234)
    t = timed(lambda: step(*batch), 10, 3)
    # attention kernels alone at this shape (per layer): fwd 4*B*H*N^2*D flops, bwd 10*B*H*N^2*D
    N, H, D = 577, 8, 32
    eng = step.eng
    L = eng.layers[0]
    tf = timed(lambda: ops.attention_fwd(L["qkvb"], L["ao"], L["lse"], None, B, N, H, D, D ** -0.5), 20, 3)
    tb = timed(lambda: ops.attention_bwd(L["qkvb"], L["ao"], eng.t_a, L["lse"], eng.delta, eng.t_qkv, B, N, H, D, D ** -0.5), 20, 3)
    ffl, bfl = 4.0 * B * H * N * N * D, 10.0 * B * H * N * N * D
    print(json.dumps({"config": "4: MultiTaskNet 384x384 B=128 bf16 train step", "images_per_sec": round(B / t, 1),
                      "ms_per_step": round(t * 1e3, 3), "algorithmic_tflops": round(B / t * 55.4e9 / 1e12, 1),
                      "attention_fwd_us": round(tf * 1e6, 1), "attention_fwd_tflops": round(ffl / tf / 1e12, 1),
                      "attention_bwd_us": round(tb * 1e6, 1), "attention_bwd_tflops": round(bfl / tb / 1e12, 1),
                      "attention_mfma_frac_of_2.5PF": round(max(ffl / tf, bfl / tb) / 2.5e15, 4)}))


def config5():
    dev = torch.device("cuda:0")
    B, S = 512, 192
    torch.manual_seed(42)
    model = MultiTaskNet(21, 19, [S, S], compute_dtype=torch.bfloat16).to(dev)
    eng = model.engine(B)
    x = torch.randn(B, 3, S, S, device=dev)
    up = torch.randn(B, 12, 12, 512, device=dev).to(torch.bfloat16)
    gbuf = eng.stages[-1]["g_out"]

    def step():
        eng.pack_weights()
        eng.backbone_forward(x, training=True)
        gbuf.copy_(up)
        eng.backbone_backward(gbuf)

    t = timed(step, 10, 3)
    print(json.dumps({"config": "5: GELANNet('small') only 192x192 B=512 bf16 forward+backward (synthetic upstream grad)",
                      "images_per_sec": round(B / t, 1), "ms_per_step": round(t * 1e3, 3),
                      "algorithmic_tflops": round(B / t * 11.27e9 / 1e12, 1)}))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    if which in ("4", "both"):
        config4()
    if which in ("5", "both"):
        config5()
