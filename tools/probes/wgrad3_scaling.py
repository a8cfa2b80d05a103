"""3x3 stride-1 weight gradients (wgrad3_kernel) at the model's batch and at 4x the batch: what a four-problem grouped launch could gain.
python tools/probes/wgrad3_scaling.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hand_gesture_recognition_amd import ops
dev = torch.device("cuda:0")
def t(B, H, C, reps=20):
    x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16); dy = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    dw = torch.empty(C, C, 3, 3, device=dev)
    ws = torch.empty(ops.conv_wgrad_workspace_bytes(torch.bfloat16, B, H, H, C, C, 3, 1) // 4, device=dev)
    for _ in range(3): ops.conv_wgrad(x, dy, dw, ws, 3, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): ops.conv_wgrad(x, dy, dw, ws, 3, 1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"B={B:5d} {H}x{H} C={C}: {us:7.1f} us  {2.0 * B * H * H * C * C * 9 / us / 1e6:6.0f} TFLOP/s", flush=True)
    return us
for H, C in ((48, 64), (24, 128), (12, 256)):
    a = t(256, H, C); b = t(1024, H, C, reps=8)
    print(f"   per 256 images at 4x: {b / 4:.1f} us (-{a - b / 4:.1f} us per launch)")
