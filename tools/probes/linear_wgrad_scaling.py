"""How far the ViT Linear weight gradients (37,120 rows) are from the same kernel on a long reduction axis:
python tools/probes/linear_wgrad_scaling.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hand_gesture_recognition_amd import ops
dev = torch.device("cuda:0")
def t(M, N, K, dtype=torch.bfloat16, reps=30):
    x = torch.randn(M, K, device=dev).to(dtype); dy = torch.randn(M, N, device=dev).to(dtype)
    dw = torch.empty(N, K, device=dev)
    ws = torch.empty(ops.linear_wgrad_workspace_bytes(dtype, M, N, K) // 4, device=dev)
    for _ in range(5): ops.linear_wgrad(x, dy, dw, ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): ops.linear_wgrad(x, dy, dw, ws)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"M={M:7d} N={N:4d} K={K:4d}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  {(M * (N + K) * 2) / us / 1e3:7.1f} GB/s unique", flush=True)
    return us
tot = 0
for N, K in ((768, 256), (256, 256), (512, 256), (256, 512)):
    tot += t(37120, N, K)
print(f"one layer: {tot:.1f} us, four layers {4 * tot:.1f} us")
tot16 = 0
for N, K in ((768, 256), (256, 256), (512, 256), (256, 512)):
    tot16 += t(37120 * 16, N, K, reps=10)
print(f"the same at 16x the rows: {tot16:.1f} us = {tot16 / 16:.1f} us per 37,120 rows -> four layers {4 * tot16 / 16:.1f} us")
