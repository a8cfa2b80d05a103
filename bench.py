"""bench.py -- images/sec of one MultiTaskNet training step (BASELINE.json metric).

  python bench.py [--gpus N --steps K --warmup W]       (N > 1: under torch.distributed.run, or plain -- it then starts
                                                         its own N ranks as a child torch.distributed.run)

A "step" = forward + 0.001*CE + JointsMSE + backward (+ RCCL gradient all-reduce) + AdamW on one
synthetic batch already resident in HBM; workload = BASELINE.json configs[1]: MultiTaskNet
(hagrid.yaml: 21 joints, 19 classes) 192x192, B=256 per GPU, bf16 storage / fp32 accumulate.
Prints ONE JSON line on rank 0:
  value / ms_per_step   exactly K steps between two barrier + synchronize brackets (max over ranks);
  timing_windows        4 more windows of K steps each (min / median: boxes of the pool differ by +-4 %);
  roofline              dominant kernel = the MFMA implicit-GEMM template `igemm_kernel`, all instances, timed
                        live with events on the launch stream in one instrumented step after the timed region;
  validation            the SAME step in the two fp32-storage modes that meet the north-star tolerance 1e-3
                        (bf16x3 = split-bf16 products, fp32 = exact-f32 MFMA): images/sec at the same batch, and
                        the measured errors of all three modes against the CPU oracle at BASELINE config 1 (B=4);
  configs               BASELINE configs 4 (384x384, B=128: attention stress) and 5 (GELAN only, B=512: conv HBM
                        stress): images/sec + the roofline figure of the kernel family each one stresses;
  inference             eval() forward with BatchNorm folded: images/sec at B=256, latency at B=1;
  cpu_baseline          the oracle (CPU port of the reference path) on the host cores;
  ranks (N > 1)         rccl_ranks, per-rank ms/step, comm_exposed_ms (time the compute stream waited for RCCL).
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0              # HBM3E spec (same guide: ~6.3 TB/s achievable)
TRAIN_GFLOP_PER_IMG = 13.09        # fwd + dgrad + wgrad @192^2
FWD_GFLOP_PER_IMG = 4.374           # forward @192^2
TRAIN_GFLOP_PER_IMG_384 = 55.4
BACKBONE_GFLOP_PER_IMG = 11.27
BACKBONE_MB_PER_IMG = 3 * 15.3     # conv part of the one-read-one-write byte model, x3 for the train step


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--size", type=int, default=192)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "bf16x3"],
                    help="bf16: bf16 storage + MFMA (BASELINE config 2); fp32: fp32 storage, exact-f32 MFMA; "
                         "bf16x3: fp32 storage, split-bf16 products (three MFMAs) -- both fp32 modes meet 1e-3")
    ap.add_argument("--graph", type=int, default=0,
                    help="replay the step from a hipGraph (measured equal to eager launch: the step is GPU-bound)")
    ap.add_argument("--windows", type=int, default=4, help="extra timing windows of --steps steps after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-validation", action="store_true")
    ap.add_argument("--no-configs", action="store_true")
    return ap.parse_args()


_DT = {"bf16": torch.bfloat16, "fp32": torch.float32, "bf16x3": "bf16x3"}


STEP_GB_ONE_READ_ONE_WRITE = 14.1   # 55 MB per image x 256 (each tensor read once, written once)


def _pmc_file(mode):
    """newest committed rocprofv3 PMC summary of this same command (FETCH_SIZE / WRITE_SIZE collected in separate passes and
    corrected as MI355X_MICROARCH.md prescribes; tools/refresh_profiles.sh) -> (dict, 'profiles/<name>') or (None, None)"""
    import re
    tag = "" if mode == "bf16" else mode + "_"
    cands = [n for n in os.listdir(os.path.join(ROOT, "profiles")) if re.match(r"r\d+_pmc_hbm_traffic_" + tag + r"v\d+\.json$", n)]   # (not the configs' files)
    # newest collection: highest round, then highest version (r02_..._v3 > r02_..._v2 > r01_..._v10)
    cands.sort(key=lambda n: [int(x) for x in re.findall(r"\d+", n)], reverse=True)
    for name in cands:
        return json.load(open(os.path.join(ROOT, "profiles", name))), "profiles/" + name
    return None, None


def igemm_roofline(step, batch, mode="bf16"):
    """Time every conv forward / conv data gradient / ViT Linear launch inside one real training step (HIP events on the
    launch stream around each call; each call is exactly one kernel launch) and report them BY THE KERNEL THAT SERVED THEM
    (ops.conv_kernel_family = the library's own launch rules): the `igemm_kernel` template (the dominant kernel: the object
    itself), `pw_kernel` and `dgrad2_kernel` as families of their own.

    achieved = algorithmic FLOPs of the igemm_kernel launches / their summed durations; `traffic` = measured HBM bytes per
    launch of the SAME population (the igemm_kernel rows of the committed PMC summary except the GELU'-fused instantiation,
    whose calls are not bracketed here)."""
    from hand_gesture_recognition_amd import ops
    rec = []
    dt = step.eng.dt
    fcode = step.eng.f32_code

    def wrap(name, fn, geom):
        def inner(*a, **k):
            info = geom(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                      # current stream == the stream the kernel is launched on
            r = fn(*a, **k)
            e1.record()
            rec.append((name, info, e0, e1))
            return r
        return inner

    def nbytes(*ts):
        return float(sum(t.numel() * t.element_size() for t in ts if t is not None))

    def g_fwd(x, w, y, stats, k, s, stats_tiles=None, **kw):
        B, H, W, Cin = x.shape
        Cout, M = y.shape[-1], y.numel() // y.shape[-1]
        fam = ops.conv_kernel_family(dt, False, B, H, W, Cin, Cout, k, s, mode=fcode)
        return 2.0 * M * Cout * k * k * Cin, nbytes(x, w, y), k, fam

    def g_dgrad(dy, wt, dx, res, k, s, bn=None, **kw):
        Cout, Cin, Mout = dy.shape[-1], dx.shape[-1], dy.numel() // dy.shape[-1]
        B, H, W = dx.shape[0], dx.shape[1], dx.shape[2]
        # (a launch that also writes the BatchNorm-backward sums of the next unit reads that unit's raw output too)
        extra = (bn[0], bn[1]) if bn is not None else ()
        fam = ops.conv_kernel_family(dt, True, B, H, W, Cin, Cout, k, s, has_res=res is not None, has_bn=bn is not None, mode=fcode)
        return 2.0 * Mout * Cout * k * k * Cin, nbytes(dy, wt, dx, res, *extra), k, fam

    def g_lin(x, w, out, bias=None, act=0, aux_out=None, res=None, **kw):
        K, N = x.shape[-1], out.shape[-1]
        M = x.numel() // K
        fam = ops.conv_kernel_family(dt, False, M, 1, 1, K, N, 1, 1, has_res=res is not None, mode=fcode)
        return 2.0 * M * N * K, nbytes(x, w, out, aux_out, res), 1, fam

    def g_proj(x, w, pos, cls, tok, **kw):      # the 1x1 projection with the token assembly in its epilogue: an igemm_kernel launch
        B, H, W, Cin = x.shape
        Cout = tok.shape[-1]
        fam = ops.conv_kernel_family(dt, False, B, H, W, Cin, Cout, 1, 1, mode=fcode)
        return 2.0 * B * H * W * Cout * Cin, nbytes(x, w, tok, pos), 1, (0 if fam == 3 else fam)      # (never the persistent kernel: pw.hip)

    saved = (ops.conv_fwd, ops.conv_dgrad, ops.linear_fwd, ops.proj_tokens_fwd)
    ops.conv_fwd, ops.conv_dgrad, ops.linear_fwd, ops.proj_tokens_fwd = (
        wrap("conv_fwd", saved[0], g_fwd), wrap("conv_dgrad", saved[1], g_dgrad), wrap("linear", saved[2], g_lin),
        wrap("proj_tokens", saved[3], g_proj))
    try:
        step._eager(*batch)
        torch.cuda.synchronize()
    finally:
        ops.conv_fwd, ops.conv_dgrad, ops.linear_fwd, ops.proj_tokens_fwd = saved
    # (flops, algorithmic bytes, kernel size, seconds) per launch, grouped by the kernel source that served it
    by = {}
    for name, (fl, by_, k, famc), e0, e1 in rec:
        by.setdefault(ops.KERNEL_FAMILIES[famc], []).append((fl, by_, k, e0.elapsed_time(e1) * 1e-3, famc))

    def fam_obj(rows, bound, kernel=None):
        f, b, t = sum(r[0] for r in rows), sum(r[1] for r in rows), sum(r[3] for r in rows)
        o = {"bound": bound}
        if kernel:
            o["kernel"] = kernel
        if bound == "mfma":
            o.update({"achieved": round(f / t / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": round(f / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "gbs": round(b / t / 1e9, 1)})
        else:
            o.update({"achieved": round(b / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b / t / 1e9 / HBM_PEAK_GBS, 4),
                      "tflops": round(f / t / 1e12, 1)})
        o.update({"launches_per_step": len(rows), "share_of_step_ms": round(t * 1e3, 3)})
        return o

    ig = by.get("igemm_kernel", [])
    flops, abytes, secs = sum(r[0] for r in ig), sum(r[1] for r in ig), sum(r[3] for r in ig)
    achieved = flops / secs / 1e12
    # the template's two regimes, reported separately as well: 3x3 convolutions (forward and data gradient; MFMA-bound:
    # 288..1152 flop/B) and 1x1 convolutions / Linears (HBM-bound: 64..256 flop/B); then the kernels outside the template
    families = {"conv3x3": fam_obj([r for r in ig if r[2] == 3], "mfma", "igemm_kernel (3x3 taps: streamed tiles, window, image tiles)"),
                "pointwise": fam_obj([r for r in ig if r[2] == 1], "hbm", "igemm_kernel (1x1 taps: convolutions and Linears)")}
    if by.get("pw_kernel"):
        families["pointwise_persistent"] = fam_obj(by["pw_kernel"], "hbm", "pw_kernel")
    if by.get("dgrad2_kernel"):
        families["conv3x3_stride2_dgrad"] = fam_obj(by["dgrad2_kernel"], "hbm", "dgrad2_kernel")
    # HBM bytes per launch of the same population from the committed PMC passes of this same command
    traffic, src, whole = None, None, None
    tname = {"bf16": "bf16_t", "bf16x3": "f32x3_t", "fp32": "float"}[mode]
    pm, src = _pmc_file(mode)
    if pm is not None:
        rows = {k: v for k, v in pm.items() if k.startswith("igemm_kernel<" + tname) and not k.rstrip().endswith(", 4>")}
        n = sum(v["dispatches"] for v in rows.values())
        if n:
            traffic = round(sum(v["hbm_bytes_per_dispatch"] * v["dispatches"] for v in rows.values()) / n)
        steps_profiled = max([v["dispatches"] for k, v in pm.items() if k.startswith("adamw_kernel")] or [0])
        if steps_profiled:
            whole = sum(v["hbm_bytes_per_dispatch"] * v["dispatches"] for v in pm.values()) / steps_profiled
    out = {"bound": "mfma", "kernel": "igemm_kernel<" + tname + ",...> (all instances of the template that one call = one launch brackets: conv forward, "
                      "conv data gradient, ViT Linears; pw_kernel and dgrad2_kernel launches are families of their own below; not "
                      "counted: the 4 fc2 data gradients fused with GELU' / bias sums, whose call brackets three launches)",
            "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": src,
            "launches_per_step": len(ig), "avg_launch_us": round(secs / len(ig) * 1e6, 1),
            "flops_per_launch": round(flops / len(ig)), "algorithmic_bytes_per_launch": round(abytes / len(ig)),
            "traffic_over_algorithmic": round(traffic / (abytes / len(ig)), 3) if traffic else None,
            "share_of_step_ms": round(secs * 1e3, 3), "families": families,
            "whole_step_hbm_bytes_measured": round(whole) if whole else None}
    if mode == "bf16x3":
        # `achieved` counts ALGORITHMIC flops (2 M N K); the split-bf16 product issues three bf16 MFMAs per one of them
        out["mfma_issue_frac"] = round(3 * achieved / MFMA_BF16_PEAK_TFLOPS, 4)
        out["note"] = "achieved / frac are algorithmic flops against the dense bf16 MFMA roof; each product is three bf16 MFMAs (mfma_issue_frac)"
    return out


def _timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def validation(a, dev, oracle_ref):
    """The same training step in the fp32-storage modes (north-star tolerance 1e-3), and every mode's measured
    error against the CPU oracle at BASELINE config 1 (B=4, 192x192; `oracle_ref` comes from the cpu_baseline leg:
    the oracle is the checker here, never the thing measured)."""
    from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
    from hand_gesture_recognition_amd.train import TrainStep, synthetic_batch
    out = {"tolerance": "north_star: logits / heatmaps within 1e-3 (max-norm relative) of the reference's fp32 CPU outputs",
           "batch": a.batch}
    for mode in ("bf16x3", "fp32"):
        torch.manual_seed(42)
        model = MultiTaskNet(21, 19, [a.size, a.size], compute_dtype=_DT[mode]).to(dev)
        step = TrainStep(model, a.batch, lr=1e-3)
        batch = synthetic_batch(a.batch, (a.size, a.size), 21, 19, dev, 1234)
        if mode == "bf16x3":
            # the mode that meets the north-star tolerance is timed with the headline's protocol: --warmup steps, then
            # 1 + --windows windows of --steps steps each, and gets its own roofline object
            ws = [_timed(lambda: step(*batch), a.steps, a.warmup if i == 0 else 0) * 1e3 for i in range(1 + max(0, a.windows))]
            out[mode] = {"images_per_sec": round(a.batch / ws[0] * 1e3, 1), "ms_per_step": round(ws[0], 3), "steps": a.steps, "warmup": a.warmup,
                         "timing_windows": {"ms_per_step": [round(x, 3) for x in ws], "min": round(min(ws), 3),
                                            "median": round(statistics.median(ws), 3),
                                            "images_per_sec_at_median": round(a.batch / statistics.median(ws) * 1e3, 1)},
                         "algorithmic_tflops": round(a.batch / ws[0] * 1e3 * TRAIN_GFLOP_PER_IMG / 1e3, 1) if a.size == 192 else None}
            if not a.no_roofline:
                out[mode]["roofline"] = igemm_roofline(step, batch, mode)
        else:
            t = _timed(lambda: step(*batch), 5, 2)
            out[mode] = {"images_per_sec": round(a.batch / t, 1), "ms_per_step": round(t * 1e3, 3), "steps": 5, "warmup": 2}
        del step, model, batch
        torch.cuda.empty_cache()
    if oracle_ref is not None:
        import mtn_oracle as O
        sd, batch, (oc, oh, oa), (ot, _, _), og = oracle_ref
        errs = {}
        # the same pass in float64 (a few seconds of CPU): the truth both the reference's fp32 arithmetic and ours are measured from
        sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
        _, _, tg = O.forward_backward(sd64, tuple(t.double() if t.dtype.is_floating_point else t for t in batch))
        truth = {"reference_fp32": float(f"{max(_rel(og[k], tg[k]) for k in og):.2e}")}
        for mode in ("bf16", "bf16x3", "fp32"):
            m = MultiTaskNet(21, 19, [192, 192], compute_dtype=_DT[mode])
            m.load_state_dict(sd, strict=True)
            m.to(dev).train()
            img, label, target, tw = (t.to(dev) for t in batch)
            c, h, at = m(img)
            tot, _, _ = O.total_loss(c, h, label, target, tw)
            tot.backward()
            torch.cuda.synchronize()
            errs[mode] = {"logits": float(f"{_rel(c, oc):.2e}"), "heatmaps": float(f"{_rel(h, oh):.2e}"),
                          "attnmap": float(f"{_rel(at, oa):.2e}"),
                          "total_loss": float(f"{abs(float(tot.detach()) - float(ot)) / abs(float(ot)):.2e}"),
                          "worst_param_grad": float(f"{max(_rel(p.grad, og[k]) for k, p in m.named_parameters()):.2e}")}
            truth[mode] = float(f"{max(_rel(p.grad, tg[k]) for k, p in m.named_parameters()):.2e}")
            del m
        out["errors_vs_oracle_config1_b4"] = errs
        out["meets_1e-3_on_outputs"] = {m: bool(max(e["logits"], e["heatmaps"], e["attnmap"], e["total_loss"]) < 1e-3) for m, e in errs.items()}
        # (parameter gradients at B=4 are sums over few pixels of cancelling terms: the reference's own fp32 arithmetic is
        #  8.2e-4 from a float64 run there -- DESIGN.md section 2b; the split-bf16 mode is gated at 2.5e-3 for this size)
        out["meets_1e-3_on_grads"] = {m: bool(e["worst_param_grad"] < 1e-3) for m, e in errs.items()}
        # worst parameter gradient against the float64 evaluation: the reference's fp32 arithmetic is itself ~8e-4 away at B=4, so the
        # figure against the fp32 oracle above moves between ~0.9e-3 and ~1.6e-3 with the summation order of either side; the 1e-3
        # gradient gates run at the full batch against a float64 truth (tests/test_full_size_parity.py: bf16x3 7.5e-4 at B=256)
        out["worst_param_grad_vs_float64_config1_b4"] = truth
    return out


def config4(dev, steps=10, warmup=3):
    """BASELINE config 4: MultiTaskNet 384x384 (N = 577 tokens), B=128, bf16 train step; the kernel family it
    stresses is attention (forward 4 B H N^2 D flops, backward 10 B H N^2 D), timed alone at this shape."""
    from hand_gesture_recognition_amd import ops
    from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
    from hand_gesture_recognition_amd.train import TrainStep, synthetic_batch
    B, S = 128, 384
    torch.manual_seed(42)
    model = MultiTaskNet(21, 19, [S, S], compute_dtype=torch.bfloat16).to(dev)
    step = TrainStep(model, B, lr=1e-3)
    batch = synthetic_batch(B, (S, S), 21, 19, dev, 1234)
    t = _timed(lambda: step(*batch), steps, warmup)
    N, H, D = 577, 8, 32
    eng = step.eng
    L = eng.layers[0]
    tf = _timed(lambda: ops.attention_fwd(L["qkvb"], L["ao"], L["lse"], None, B, N, H, D, D ** -0.5), 20, 3)
    tb = _timed(lambda: ops.attention_bwd(L["qkvb"], L["ao"], eng.t_a, L["lse"], eng.delta, eng.t_qkv, B, N, H, D, D ** -0.5), 20, 3)
    ffl, bfl = 4.0 * B * H * N * N * D, 10.0 * B * H * N * N * D
    ach = (ffl + bfl) / (tf + tb) / 1e12
    return {"workload": "MultiTaskNet 384x384 (577 tokens) B=128 bf16 train step", "images_per_sec": round(B / t, 1),
            "ms_per_step": round(t * 1e3, 3), "algorithmic_tflops": round(B / t * TRAIN_GFLOP_PER_IMG_384 / 1e3, 1),
            "roofline": {"kernel": "attn_fwd_kernel + attn_bwd_dq/dkv_kernel (one layer, timed alone)", "bound": "mfma",
                         "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                         "attention_fwd_us": round(tf * 1e6, 1), "attention_fwd_tflops": round(ffl / tf / 1e12, 1),
                         "attention_bwd_us": round(tb * 1e6, 1), "attention_bwd_tflops": round(bfl / tb / 1e12, 1)}}


def inference(dev):
    """SURVEY 8f row 3 / export.py:41-45: the module in eval() (BatchNorm folded into the packed weights, one launch per
    Conv unit), forward only, 192x192: images/sec at B=256 and latency at B=1 (the detector's frame-at-a-time use)."""
    from hand_gesture_recognition_amd.infer import InferenceSession
    from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
    out = {"workload": "MultiTaskNet 192x192 eval() forward (BN folded), logits + heat-maps + attention map"}
    for name, dt in (("bf16", torch.bfloat16), ("bf16x3", "bf16x3")):
        torch.manual_seed(42)
        model = MultiTaskNet(21, 19, [192, 192], compute_dtype=dt).to(dev).eval()
        r = {}
        with torch.no_grad():
            for B, steps in ((256, 20), (1, 100)):
                x = torch.randn(B, 3, 192, 192, device=dev)
                t = _timed(lambda: model(x), steps, 3)
                if B == 1:
                    r["latency_ms_b1"] = round(t * 1e3, 3)
                    sess = InferenceSession(model, 1, use_graph=True)     # BN folded once + one hipGraph replay per frame
                    r["latency_ms_b1_session_graph"] = round(_timed(lambda: sess(x), 200, 5) * 1e3, 3)
                    del sess
                else:
                    r["images_per_sec_b256"] = round(B / t, 1)
                    r["algorithmic_tflops_b256"] = round(B / t * FWD_GFLOP_PER_IMG / 1e3, 1)
        out[name] = r
        del model
        torch.cuda.empty_cache()
    return out


def config5(dev, steps=10, warmup=3):
    """BASELINE config 5: GELAN backbone only, 192x192, B=512, bf16 forward + backward with a synthetic upstream
    gradient (conv HBM stress): roofline = algorithmic bytes of the conv stack / step time."""
    from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
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

    t = _timed(step, steps, warmup)
    gbs = B * BACKBONE_MB_PER_IMG * 1e6 / t / 1e9
    return {"workload": "GELANNet('small') only, 192x192 B=512 bf16 forward+backward (synthetic upstream gradient)",
            "images_per_sec": round(B / t, 1), "ms_per_step": round(t * 1e3, 3),
            "algorithmic_tflops": round(B / t * BACKBONE_GFLOP_PER_IMG / 1e3, 1),
            "roofline": {"kernel": "whole conv stack (igemm / dgrad2 / wgrad / BatchNorm passes)", "bound": "hbm",
                         "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "algorithmic_mb_per_image": BACKBONE_MB_PER_IMG,
                         "mfma_frac": round(B / t * BACKBONE_GFLOP_PER_IMG / 1e3 / MFMA_BF16_PEAK_TFLOPS, 4)}}


def cpu_baseline(size, want_ref):
    """The oracle (CPU port of the reference path) timed on this box's host cores.

    Threads = the box's CPU share for one GPU (16): measured on the MI355X host, torch-CPU runs this
    model fastest at 16 threads (8: 53, 16: 68, 32: 33, 64: 18, 256: 0.1 img/s -- oversubscription).
    Sample: ~10 s of B=16 training steps of the same workload shape.  With `want_ref` the leg also returns
    the oracle's outputs / losses / gradients at BASELINE config 1 (B=4) for `validation` to check against.
    """
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mtn_oracle as O
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    B = 16
    sd = O.synthetic_state_dict(21, 19, seed=0)
    batch = O.synthetic_batch(B, size, seed=1234, gaussian_targets=False)
    st = {}
    O.train_step(sd, batch, st, 1e-3)                  # warm-up
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 10.0 and n < 200:
        O.train_step(sd, batch, st, 1e-3)
        n += 1
    dt = time.perf_counter() - t0
    # SURVEY.md section 8(d)'s two CPU workloads, its protocol (3 warm-up + 10 timed iterations, median), same thread count:
    # config 1 = B=4 forward + loss (train-mode BatchNorm), and a B=32 full train step
    def med(fn, warm=3, n_t=10):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n_t):
            t1 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t1)
        return statistics.median(ts)
    sd4, b4 = O.synthetic_state_dict(21, 19, seed=0), O.synthetic_batch(4, size, seed=1234, gaussian_targets=False)

    def fwd_loss():
        with torch.no_grad():
            c, h, _ = O.multitasknet_forward(b4[0], sd4, training=True)
            O.total_loss(c, h, b4[1], b4[2], b4[3])
    t_c1 = med(fwd_loss)
    sd32, b32, st32 = O.synthetic_state_dict(21, 19, seed=0), O.synthetic_batch(32, size, seed=1234, gaussian_targets=False), {}
    t_b32 = med(lambda: O.train_step(sd32, b32, st32, 1e-3))
    survey = {"protocol": "SURVEY.md 8(d): 3 warm-up + 10 timed iterations, median",
              "config1_b4_fwd_loss": {"images_per_sec": round(4 / t_c1, 2), "ms": round(t_c1 * 1e3, 1)},
              "b32_train_step": {"images_per_sec": round(32 / t_b32, 2), "ms": round(t_b32 * 1e3, 1)}}
    ref = None
    if want_ref:
        sd1 = O.synthetic_state_dict(21, 19, seed=0)
        b1 = O.synthetic_batch(4, 192, seed=42)
        outs, losses, grads = O.forward_backward({k: v.clone() for k, v in sd1.items()}, b1)
        ref = (sd1, b1, outs, losses, grads)
    return {"value": round(B * n / dt, 2), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"{n} train steps (fwd+loss+bwd+AdamW) at B={B}, {size}x{size}, fp32, oracle/mtn_oracle.py "
                      f"(torch-CPU ops), {threads} threads of {os.cpu_count()} logical CPUs",
            "survey_8d": survey}, ref


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves as a CHILD process
    (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), relay its output and return its exit code.
    The parent never touches the GPU and never exec()s; under torch.distributed.run (WORLD_SIZE set) this is not used."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (this pool's host driver: dmabuf IPC only -- RCCL needs it between the ranks)
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))                      # (before anything initialises the GPU in this process)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # (before the first HIP call of a rank: see self_launch)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box (RCCL refuses two ranks on one device): MTN_BENCH_BACKEND=gloo runs the same N-rank code
    # path with every rank on the card that exists; the line then says backend "gloo" and is not a scaling figure
    backend = os.environ.get("MTN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
    from hand_gesture_recognition_amd.train import TrainStep, synthetic_batch

    torch.manual_seed(42)
    model = MultiTaskNet(21, 19, [a.size, a.size], compute_dtype=_DT[a.dtype]).to(dev)
    step = TrainStep(model, a.batch, lr=1e-3, use_graph=bool(a.graph))
    batch = synthetic_batch(a.batch, (a.size, a.size), 21, 19, dev, 1234 + rank)

    def bracket():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def window():
        """exactly a.steps steps between two barrier + synchronize brackets -> (max over ranks, this rank) seconds"""
        bracket()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            losses = step(*batch)
        bracket()
        mine = time.perf_counter() - t0
        el = torch.tensor([mine], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el), mine, losses

    # proof of what the process group saw: a ones tensor summed over the group (RCCL for N > 1) = the number of ranks
    ones = torch.ones(1, device=dev)
    if world > 1:
        dist.all_reduce(ones)
    ranks_counted = int(ones.item())
    assert ranks_counted == world, f"the all-reduce saw {ranks_counted} ranks, WORLD_SIZE={world}"
    for _ in range(a.warmup):
        step(*batch)
    elapsed, mine, losses = window()                   # THE timed region
    loss_val = [float(x) for x in losses]
    assert all(x == x and abs(x) < 1e6 for x in loss_val), f"non-finite losses {loss_val}"
    # per-rank step time of the timed region, and how much of the all-reduce the backward did not hide
    rank_ms = torch.zeros(world, device=dev, dtype=torch.float64)
    rank_ms[rank] = mine / a.steps * 1e3
    comm_ms = torch.zeros(world, device=dev, dtype=torch.float64)
    if world > 1:
        step.comm_events = []
        for _ in range(min(a.steps, 10)):
            step(*batch)
        torch.cuda.synchronize()
        comm_ms[rank] = step.comm_exposed_ms()
        step.comm_events = None
        dist.all_reduce(rank_ms)
        dist.all_reduce(comm_ms)
    extra = [window()[0] / a.steps * 1e3 for _ in range(max(0, a.windows))]
    # N > 1: the same step WITHOUT partners, in the same processes and through the same bucketed all-reduce path, on a
    # one-rank group per rank -- the N = 1 figure this run's scaling can be checked against directly (the driver computes
    # efficiency from its own N = 1 run; this one shares the box, the clocks and the process with the N-rank figure)
    n1 = None
    if world > 1:
        groups = [dist.new_group([r]) for r in range(world)]          # (collective: every rank creates every group)
        step1 = TrainStep(model, a.batch, lr=1e-3, process_group=groups[rank], overlap_comm="force")
        for _ in range(min(a.warmup, 3)):
            step1(*batch)
        bracket()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step1(*batch)
        bracket()
        el1 = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(el1, op=dist.ReduceOp.MAX)
        n1 = {"ms_per_step_max_over_ranks": round(float(el1) / a.steps * 1e3, 3),
              "images_per_sec_per_gpu": round(a.batch * a.steps / float(el1), 1),
              "how": "TrainStep on a one-rank process group per rank (bucketed all-reduce launched, no partner), all ranks at once"}
        del step1

    roof = None
    if not a.no_roofline:
        roof = igemm_roofline(step, batch, a.dtype)     # every rank runs it (the step holds a collective)
    if rank == 0:
        ips = world * a.batch * a.steps / elapsed
        gflop = TRAIN_GFLOP_PER_IMG if a.size == 192 else (TRAIN_GFLOP_PER_IMG_384 if a.size == 384 else None)
        all_ms = [elapsed / a.steps * 1e3] + extra
        out = {
            "metric": f"images/sec training step, MultiTaskNet {a.size}x{a.size} B={a.batch}", "value": round(ips, 1),
            "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"MultiTaskNet(21 joints, 19 classes) {a.size}x{a.size} train step "
                                   f"(fwd + 0.001*CE + JointsMSE + bwd + AdamW), B={a.batch}/GPU, random-init weights",
                       "global_batch": world * a.batch, "parallelism": f"dp{world}", "hipgraph": bool(a.graph)},
            "algorithmic_tflops": round(ips * gflop / 1e3, 1) if gflop else None,
            "mfma_frac_whole_step": round(ips * gflop / 1e3 / MFMA_BF16_PEAK_TFLOPS / world, 4) if gflop else None,
            # one-read-one-write byte model of SURVEY 8(d) (55 MB per image, bf16) against the HBM roof at this step time
            "hbm_frac_whole_step_algorithmic": round(ips / world * 55e6 / 1e9 / HBM_PEAK_GBS, 4) if (a.size == 192 and a.dtype == "bf16") else None,
            "final_losses": [round(x, 6) for x in loss_val],
            "timing_windows": {"ms_per_step": [round(x, 3) for x in all_ms], "min": round(min(all_ms), 3),
                               "median": round(statistics.median(all_ms), 3),
                               "images_per_sec_at_median": round(world * a.batch / statistics.median(all_ms) * 1e3, 1),
                               "note": "window 0 is the timed region `value` is computed from"},
        }
        if world > 1:
            out["ranks"] = {"rccl_ranks": ranks_counted, "rccl_ranks_how": "all-reduce (sum) of a ones tensor over the group", "backend": dist.get_backend(), "ms_per_step_per_rank": [round(float(x), 3) for x in rank_ms],
                            "comm_exposed_ms_per_rank": [round(float(x), 4) for x in comm_ms],
                            "comm_exposed_ms": round(float(comm_ms.max()), 4),
                            "gradient_bytes_all_reduced": int(model.n_params * 4), "buckets": len(step._buckets),
                            "overlap_with_backward": bool(step.overlap), "n1_same_processes": n1}
        if roof is not None:
            out["roofline"] = roof
            wb = roof.pop("whole_step_hbm_bytes_measured", None)
            if wb and a.size == 192 and a.batch == 256:
                # measured HBM bytes of one step (all kernels of the committed PMC summary) at THIS run's step time, and
                # against the one-read-one-write model: > 1 = passes that re-read / re-write tensors (BatchNorm, split-K slabs)
                out["hbm_whole_step"] = {"measured_gb_per_step": round(wb / 1e9, 2), "source": roof["traffic_source"],
                                         "hbm_frac": round(wb / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                         "traffic_ratio": round(wb / 1e9 / (STEP_GB_ONE_READ_ONE_WRITE * (2 if a.dtype != "bf16" else 1)), 3),
                                         "one_read_one_write_gb": STEP_GB_ONE_READ_ONE_WRITE * (2 if a.dtype != "bf16" else 1)}
    del step, model
    torch.cuda.empty_cache()
    if rank == 0 and world == 1:
        base, ref = (None, None)
        if not a.no_cpu_baseline:
            base, ref = cpu_baseline(a.size, want_ref=not a.no_validation)
        if not a.no_validation:
            out["validation"] = validation(a, dev, ref)
        if not a.no_configs and a.dtype == "bf16":
            out["configs"] = {"4": config4(dev), "5": config5(dev)}
            out["inference"] = inference(dev)
        if base is not None:
            out["cpu_baseline"] = base
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
