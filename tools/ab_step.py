"""A/B of two builds of the library on the full training step, in ONE GPU session (boxes of the pool
differ by several percent): python tools/ab_step.py [path/to/other/libmtn_hip.so] [--graph] [--key K=V ...]
Prints the median / min of several timing windows."""
import sys, os, time, statistics, torch
os.environ.setdefault("MTN_ALLOW_DEBUG_KEYS", "1")     # diagnostic tool: opts in to mtn_debug_set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand_gesture_recognition_amd import lib
args = [a for a in sys.argv[1:] if not a.startswith("--")]
graph = "--graph" in sys.argv
opt = {a[2:].split("=")[0]: a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a and not a.startswith("--key=")}
dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "bf16x3": "bf16x3"}[opt.get("dtype", "bf16")]
B, S = int(opt.get("batch", 256)), int(opt.get("size", 192))
if args: lib.use_library(os.path.abspath(args[0]))
for a in sys.argv[1:]:
    if a.startswith("--key="):
        k, v = a[6:].split("=")
        lib.call("mtn_debug_set", int(k), int(v))
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
from hand_gesture_recognition_amd.train import TrainStep, synthetic_batch
dev = torch.device("cuda:0")
torch.manual_seed(42)
model = MultiTaskNet(21, 19, [S, S], compute_dtype=dtype).to(dev)
step = TrainStep(model, B, lr=1e-3, use_graph=graph)
batch = synthetic_batch(B, (S, S), 21, 19, dev, 1234)
for _ in range(8): step(*batch)
ts = []
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step(*batch)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
print(args or "default", " ".join(a for a in sys.argv[1:] if a.startswith("--")), "graph" if graph else "eager", f"median {statistics.median(ts):.3f} min {min(ts):.3f} ms/step", flush=True)
