"""Phase breakdown of the igemm K loop from in-kernel s_memtime stamps (diagnostic build only:
`touch igemm.hip; make -C hand-gesture-recognition_amd/csrc hip EXTRA=-DMTN_STAMPS`).

Prints, averaged over wave 0 of the first 256 workgroups: cycles per K step spent issuing the
LDS-DMA, in the fragment-read + MFMA body, and waiting at the barrier; the window load; the whole
loop and the epilogue.
"""
import argparse
import ctypes
import os
os.environ.setdefault("MTN_ALLOW_DEBUG_KEYS", "1")     # diagnostic tool: opts in to mtn_debug_set
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand_gesture_recognition_amd import lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stages", type=int, default=3)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--last", type=int, default=0, help="1: stamp the last 256 workgroups (steady state)")
    ap.add_argument("--shape", default="256,24,24,128,128,3,1")
    a = ap.parse_args()
    lib.call("mtn_debug_set", 3, a.stages)
    lib.call("mtn_debug_set", 6, a.tile)
    lib.call("mtn_debug_set", 8, 4 if a.last else 0)
    B, H, W, Cin, Cout, k, s = (int(v) for v in a.shape.split(","))
    dev = torch.device("cuda:0")
    OH, OW = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    y = torch.empty(B, OH, OW, Cout, device=dev, dtype=torch.bfloat16)
    stats = torch.empty(ops.conv_stats_tiles(B * OH * OW, Cout, k * k * Cin), 2, Cout, device=dev)
    for _ in range(5):
        ops.conv_fwd(x, w, y, stats, k, s)
    torch.cuda.synchronize()
    buf = np.zeros((256, 16), dtype=np.uint64)
    lib.call("mtn_debug_read", 1, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    steps = buf[:, 4].astype(np.float64)
    ok = steps > 0
    f = lambda c: float((buf[ok, c].astype(np.float64) / steps[ok]).mean())
    print(f"shape {a.shape} stages {a.stages} tile {a.tile}: K steps/tile {steps[ok].mean():.0f}")
    print(f"  per step: dma issue {f(1):7.0f}  frag+mfma {f(2):7.0f}  barrier {f(3):7.0f}  cycles")
    print(f"  prologue {buf[ok,7].mean():8.0f}  window load {buf[ok,0].mean():8.0f}  loop total {buf[ok,5].mean():8.0f}  epilogue {buf[ok,6].mean():8.0f} cycles")
    print(f"  epilogue: C tile -> LDS {buf[ok,8].mean():7.0f}  row-wise stores {buf[ok,9].mean():7.0f}  statistics {buf[ok,10].mean():7.0f}")


if __name__ == "__main__":
    main()
