"""Micro-benchmarks of individual HIP kernels at BASELINE config-2 sizes (B=256, 192x192).

Usage (GPU box): python tools/bench_kernels.py [--batch 256] [--filter conv]
Prints one line per case: time (HIP events on the launch stream), TFLOP/s, GB/s (algorithmic).
"""
import argparse
import os
os.environ.setdefault("MTN_ALLOW_DEBUG_KEYS", "1")     # diagnostic tool: opts in to mtn_debug_set
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand_gesture_recognition_amd import ops  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--filter", default="")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--stages", type=int, default=5)
    ap.add_argument("--n64rows", type=int, default=128)
    ap.add_argument("--wgrad_dma", type=int, default=1)
    ap.add_argument("--igemm_tile", type=int, default=0)
    ap.add_argument("--wgrad_tile", type=int, default=0)
    ap.add_argument("--igemm_dbg", type=int, default=0)
    ap.add_argument("--wgrad_slots", type=int, default=0)
    ap.add_argument("--conv1_px", type=int, default=0)
    ap.add_argument("--key", action="append", default=[], help="K=V: mtn_debug_set(K, V) before the runs (repeatable)")
    ap.add_argument("--win_limit", type=int, default=0, help="window-mode LDS limit in KB (default 53)")
    ap.add_argument("--lib", default="", help="A/B: load this build of the library instead of libmtn_hip.so")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    from hand_gesture_recognition_amd import lib as _lib
    if a.dtype == "bf16x3":          # fp32 tensors, split-bf16 products (weights: any bits do for timing)
        from hand_gesture_recognition_amd import ops as _ops
        _ops.set_f32_matmul(_ops.F32X3)
    if a.lib:
        _lib.use_library(os.path.abspath(a.lib))
    _lib.call("mtn_debug_set", 3, a.stages)
    _lib.call("mtn_debug_set", 4, a.n64rows)
    _lib.call("mtn_debug_set", 5, a.wgrad_dma)
    _lib.call("mtn_debug_set", 6, a.igemm_tile)
    _lib.call("mtn_debug_set", 7, a.wgrad_tile)
    _lib.call("mtn_debug_set", 8, a.igemm_dbg)
    for kv in a.key:
        k, v = kv.split("=")
        _lib.call("mtn_debug_set", int(k), int(v))
    try:
        _lib.call("mtn_debug_set", 10, a.wgrad_slots)
    except RuntimeError:
        pass
    if a.win_limit:
        _lib.call("mtn_debug_set", 9, a.win_limit)
    dev = torch.device("cuda:0")
    B = a.batch
    esz = 2 if dt == torch.bfloat16 else 4
    convs = [  # name, H, W, Cin, Cout, k, s
        ("conv2", 96, 96, 64, 128, 3, 2),
        ("s1.cv1", 48, 48, 128, 128, 1, 1), ("s1.3x3", 48, 48, 64, 64, 3, 1), ("s1.cv4", 48, 48, 256, 128, 1, 1),
        ("down1", 48, 48, 128, 256, 3, 2),
        ("s2.cv1", 24, 24, 256, 256, 1, 1), ("s2.3x3", 24, 24, 128, 128, 3, 1), ("s2.cv4", 24, 24, 512, 256, 1, 1),
        ("down2", 24, 24, 256, 512, 3, 2),
        ("s3.cv1", 12, 12, 512, 512, 1, 1), ("s3.3x3", 12, 12, 256, 256, 3, 1), ("s3.cv4", 12, 12, 1024, 512, 1, 1),
        ("proj", 12, 12, 512, 256, 1, 1),
    ]
    if not a.filter or a.filter in "conv1":
        H = W = 192
        x = torch.randn(B, 3, H, W, device=dev)
        w = (torch.randn(64, 3, 3, 3, device=dev) / 27 ** 0.5).to(dt)
        tiles = ops.conv1_stats_tiles(B, H, W)
        stats = torch.empty(ops.slab_rows(tiles), 2, 64, device=dev)
        ss, mr = torch.randn(2, 64, device=dev), torch.rand(2, 64, device=dev) + 0.5
        out = torch.empty(B, H // 2, W // 2, 64, device=dev, dtype=dt)
        dout = torch.randn(B, H // 2, W // 2, 64, device=dev).to(dt)
        dg, db, dw = torch.empty(64, device=dev), torch.empty(64, device=dev), torch.empty(64, 3, 3, 3, device=dev)
        xb, ob = x.numel() * 4, out.numel() * esz
        t = timeit(lambda: ops.conv1_stats(x, w, stats))
        print(f"conv1.stats    {t*1e6:9.1f} us  {xb/t/1e9:7.0f} GB/s")
        t = timeit(lambda: ops.conv1_bn_act(x, w, ss, out, ops.ACT_SILU))
        print(f"conv1.bn_act   {t*1e6:9.1f} us  {(xb+ob)/t/1e9:7.0f} GB/s")
        for px in (a.conv1_px and [a.conv1_px] or [1024, 2048, 3072, 4096]):
            try:
                _lib.call("mtn_debug_set", 12, px)
            except RuntimeError:
                pass                                   # an older build without the key
            part = torch.empty(*ops.conv1_bwd_partial_shape(B, H, W), device=dev)
            t = timeit(lambda: ops.conv1_bwd(x, w, dout, ss, mr, part, dg, db, dw, ops.ACT_SILU))
            print(f"conv1.bwd px/wg={px:5d} {t*1e6:9.1f} us  {(xb+ob)/t/1e9:7.0f} GB/s")
        try:
            _lib.call("mtn_debug_set", 12, 0)
        except RuntimeError:
            pass
    for name, H, W, Cin, Cout, k, s in convs:
        if a.filter and a.filter not in "conv_fwd." + name:
            continue
        OH, OW = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        x = torch.randn(B, H, W, Cin, device=dev).to(dt)
        w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).to(dt)
        y = torch.empty(B, OH, OW, Cout, device=dev, dtype=dt)
        M = B * OH * OW
        stats = torch.empty(ops.conv_fwd_stats_tiles(dt, B, H, W, Cin, Cout, k, s), 2, Cout, device=dev)
        t = timeit(lambda: ops.conv_fwd(x, w, y, stats, k, s))
        fl = 2.0 * M * Cout * k * k * Cin
        by = (x.numel() + y.numel() + w.numel()) * esz
        print(f"conv_fwd.{name:8s} M={M:8d} N={Cout:4d} K={k*k*Cin:5d}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.0f} GB/s")
        if k == 3 or name.endswith(("cv1", "cv4")):
            dy = torch.randn(B, OH, OW, Cout, device=dev).to(dt)
            wt = (torch.randn(Cin, k, k, Cout, device=dev) / (Cout * k * k) ** 0.5).to(dt)
            dx = torch.empty(B, H, W, Cin, device=dev, dtype=dt)
            t = timeit(lambda: ops.conv_dgrad(dy, wt, dx, None, k, s))
            print(f"conv_dgrad.{name:6s} M={B*H*W:8d} N={Cin:4d} K={k*k*Cout:5d}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s (useful)  {by/t/1e9:7.0f} GB/s")
    for name, H, W, Cin, Cout, k, s in convs:
        if a.filter and a.filter not in "conv_wgrad." + name:
            continue
        OH, OW = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        M = B * OH * OW
        x = torch.randn(B, H, W, Cin, device=dev).to(dt)
        dy = torch.randn(B, OH, OW, Cout, device=dev).to(dt)
        dw = torch.empty(Cout, Cin, k, k, device=dev)
        wsb = ops.conv_wgrad_workspace_bytes(dt, B, H, W, Cin, Cout, k, s)
        ws = torch.empty(wsb // 4, device=dev)
        t = timeit(lambda: ops.conv_wgrad(x, dy, dw, ws, k, s))
        fl = 2.0 * M * Cout * k * k * Cin
        by = (x.numel() + dy.numel()) * esz
        print(f"conv_wgrad.{name:6s} M={M:8d} N={Cout:4d} K={k*k*Cin:5d}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.0f} GB/s  ws={wsb/1e6:.0f}MB")
    if not a.filter or "bn" in a.filter:
        for M, C in ((B * 96 * 96, 64), (B * 48 * 48, 128), (B * 48 * 48, 256), (B * 12 * 12, 512)):
            y = torch.randn(M, C, device=dev).to(dt)
            out = torch.empty_like(y)
            ss = torch.randn(2, C, device=dev)
            mr = torch.rand(2, C, device=dev) + 0.5
            t = timeit(lambda: ops.bn_act_fwd(y, ss, out, ops.ACT_SILU))
            print(f"bn_act_fwd M={M:8d} C={C:4d} {t*1e6:8.1f} us  {2*M*C*esz/t/1e9:7.0f} GB/s")
            dout = torch.randn(M, C, device=dev).to(dt)
            part = torch.empty(ops.bn_bwd_partial_rows(M), 2, C, device=dev)
            coef, dg, db = torch.empty(2, C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
            dy = torch.empty_like(y)
            t = timeit(lambda: ops.bn_act_bwd(dout, y, None, ss, mr, part, coef, dg, db, dy, ops.ACT_SILU))
            print(f"bn_act_bwd M={M:8d} C={C:4d} {t*1e6:8.1f} us  {5*M*C*esz/t/1e9:7.0f} GB/s (reduce 2 reads + apply 2 reads 1 write)")
    if not a.filter or "pose" in a.filter:
        from hand_gesture_recognition_amd import lib
        h = w = 12
        J = 21
        tok = torch.randn(B, h * w + 1, 256, device=dev).to(dt)
        Wp, bias = torch.randn(J, 256, device=dev) / 16, torch.zeros(J, device=dev)
        hm = torch.empty(B, J, 4 * h, 4 * w, device=dev)
        t = timeit(lambda: ops.pose_head_fwd(tok, Wp, bias, hm, h, w))
        print(f"pose_fwd  {t*1e6:9.1f} us")
        dhm = torch.randn(B, J, 4 * h, 4 * w, device=dev)
        dtok = torch.empty_like(tok)
        dwb = torch.empty(J * 257, device=dev)
        slab = torch.empty(ops.slab_rows(B), J * 257, device=dev)
        # the "MFMA - phase" lines differ from the first only in a build with EXTRA=-DMTN_ABLATE
        for dbg, what in ((0, "MFMA kernel"), (1, "VALU kernel"), (2, "MFMA - interpolation"), (4, "MFMA - mfma phases"),
                          (8, "MFMA - scatter"), (16, "MFMA - prefetch"), (30, "MFMA skeleton (loads + syncs only)")):
            lib.call("mtn_debug_set", 2, dbg)
            t = timeit(lambda: ops.pose_head_bwd(tok, Wp, dhm, dtok, dwb, slab, h, w))
            print(f"pose_bwd {what:36s} {t*1e6:9.1f} us")
        lib.call("mtn_debug_set", 2, 0)
    if not a.filter or "attn" in a.filter:
        N, H, D = 145, 8, 32
        qkv = torch.randn(B * N, 3 * H * D, device=dev).to(dt)
        out, dout = torch.empty(B * N, H * D, device=dev, dtype=dt), torch.randn(B * N, H * D, device=dev).to(dt)
        lse, delta = torch.empty(B * H, N, device=dev), torch.empty(B * H, N, device=dev)
        dqkv = torch.empty_like(qkv)
        t = timeit(lambda: ops.attention_fwd(qkv, out, lse, None, B, N, H, D, D ** -0.5))
        print(f"attn_fwd N={N}  {t*1e6:9.1f} us  {4.0*B*H*N*N*D/t/1e12:7.1f} TF/s")
        t = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H, D, D ** -0.5))
        print(f"attn_bwd N={N}  {t*1e6:9.1f} us  {10.0*B*H*N*N*D/t/1e12:7.1f} TF/s")
    # ViT linears: M = B*145
    M = B * 145
    for name, K, N in (("qkv", 256, 768), ("out", 256, 256), ("dqkv", 768, 256)):
        if a.filter and a.filter not in "linear." + name:
            continue
        if name != "dqkv":
            x = torch.randn(M, K, device=dev).to(dt)
            dy = torch.randn(M, N, device=dev).to(dt)
            dw = torch.empty(N, K, device=dev)
            ws = torch.empty(ops.linear_wgrad_workspace_bytes(dt, M, N, K) // 4, device=dev)
            t = timeit(lambda: ops.linear_wgrad(x, dy, dw, ws))
            print(f"linear_wgrad.{name:4s} M={M:8d} N={N:4d} K={K:5d}  {t*1e6:9.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF/s  ws={ws.numel()*4/1e6:.0f}MB")
        x = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) / 16).to(dt)
        o = torch.empty(M, N, device=dev, dtype=dt)
        t = timeit(lambda: ops.linear_fwd(x, w, o))
        fl = 2.0 * M * N * K
        by = (x.numel() + o.numel() + w.numel()) * esz
        print(f"linear.{name:10s} M={M:8d} N={N:4d} K={K:5d}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.0f} GB/s")


if __name__ == "__main__":
    main()
