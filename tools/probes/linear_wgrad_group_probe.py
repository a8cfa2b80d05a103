"""The sixteen ViT Linear weight gradients: sixteen launches vs one grouped launch, standalone (operands resident, nothing else running).
python tools/probes/linear_wgrad_group_probe.py [--dtype=bf16x3] [--rows=37120]"""
import os, sys, torch
os.environ.setdefault("MTN_ALLOW_DEBUG_KEYS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hand_gesture_recognition_amd import ops, lib
x3 = "--dtype=bf16x3" in sys.argv
M = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--rows=")), 37120))
dt = torch.float32 if x3 else torch.bfloat16
mode = ops.F32X3 if x3 else None
if x3: ops._f32_code = ops.F32X3
dev = torch.device("cuda:0")
probs = []
for layer in range(4):
    for N, K in ((256, 256), (256, 256), (256, 256), (768, 256)):
        probs.append((torch.randn(M, K, device=dev).to(dt), torch.randn(M, N, device=dev).to(dt), torch.empty(N, K, device=dev)))
items, m = ops.wgrad_group_items(probs)
ws = torch.empty(max(ops.linear_wgrad_group_workspace_bytes(dt, items, M), 1 << 28) // 4, device=dev)
def timed(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
one = timed(lambda: [ops.linear_wgrad(x, dy, dw, ws) for x, dy, dw in probs])
grp = timed(lambda: ops.linear_wgrad_group(items, M, ws, dt))
flops = sum(2.0 * M * x.shape[1] * dy.shape[1] for x, dy, _ in probs)
print(f"{'bf16x3' if x3 else 'bf16'} M={M}: sixteen launches {one:.1f} us ({flops / one / 1e6:.0f} TFLOP/s), grouped {grp:.1f} us ({flops / grp / 1e6:.0f} TFLOP/s)")
for slots in (256, 384, 512, 768, 1024):
    lib.call("mtn_debug_set", 10, slots)
    print(f"  slots {slots}: grouped {timed(lambda: ops.linear_wgrad_group(items, M, ws, dt)):.1f} us")
lib.call("mtn_debug_set", 10, 0)
