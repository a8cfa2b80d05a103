"""Diagnostic: tokens written by the fused projection epilogue vs the two-launch form, both against a float64 product of the
SAME stored backbone features (split-storage mode, config 1 B=4).  python tools/probes/x3_tokens_probe.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mtn_oracle as O
from backend import join_x3
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
dev = torch.device("cuda:0")
sd = O.synthetic_state_dict(21, 19, seed=0)
batch = O.synthetic_batch(4, 192, seed=42)
m = MultiTaskNet(21, 19, [192, 192], compute_dtype="bf16x3"); m.load_state_dict(sd, strict=True); m.to(dev).train()
img = batch[0].to(dev)
res = {}
for fuse in (True, False):
    eng = m.engine(4)
    eng.fuse_tokens = fuse
    with torch.no_grad():
        eng.forward(img, True, True) if False else m(img)
    torch.cuda.synchronize()
    tok = join_x3(eng.x[0].reshape(-1, 256).cpu()).double().reshape(4, -1, 256)
    feats = join_x3(eng.stages[-1]["out"].reshape(-1, eng.stages[-1]["out"].shape[-1]).cpu()).double()
    W = m.proj.weight.detach().cpu().double().reshape(256, -1)
    ref = (feats @ W.t()).reshape(4, -1, 256) + eng.pos.cpu().double().reshape(1, -1, 256)
    e = (tok[:, 1:] - ref).abs().max() / ref.abs().max()
    ecls = (tok[:, 0] - m.decoder.cls_token.detach().cpu().double().reshape(1, 256)).abs().max()
    res[fuse] = tok
    print("fused" if fuse else "two launches", f"tokens vs float64 product of the stored features: {float(e):.3e}   class-token rows: {float(ecls):.3e}", flush=True)
print("fused vs two launches:", float((res[True] - res[False]).abs().max() / res[False].abs().max()))
