"""bench.py -- images/sec of one MultiTaskNet training step (BASELINE.json metric).

  python bench.py [--gpus N --steps K --warmup W]       (N > 1: launched by torch.distributed.run)

A "step" = forward + 0.001*CE + JointsMSE + backward (+ RCCL gradient all-reduce) + AdamW on one
synthetic batch already resident in HBM; workload = BASELINE.json configs[1]: MultiTaskNet
(hagrid.yaml: 21 joints, 19 classes) 192x192, B=256 per GPU, bf16 storage / fp32 accumulate.
Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: the MFMA implicit-GEMM template
`igemm_kernel`, all instances, timed live with events on the launch stream in an extra
instrumented step after the timed region) and `cpu_baseline` (the oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0              # HBM3E spec (same guide: ~6.3 TB/s achievable)
TRAIN_GFLOP_PER_IMG = 13.09        # fwd + dgrad + wgrad @192^2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--size", type=int, default=192)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "bf16x3"],
                    help="bf16: bf16 storage + MFMA (BASELINE config 2); fp32: fp32 storage, exact-f32 MFMA; "
                         "bf16x3: fp32 storage, split-bf16 products (three MFMAs) -- both fp32 modes meet 1e-3")
    ap.add_argument("--graph", type=int, default=0,
                    help="replay the step from a hipGraph (measured equal to eager launch: the step is GPU-bound)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def igemm_roofline(step, batch):
    """Time every launch of the dominant kernel (the `igemm_kernel` template: every convolution
    forward / data gradient and every ViT linear) inside one real training step.

    achieved = algorithmic FLOPs of those launches / their summed durations (HIP events on the
    launch stream around each call; each call is exactly one igemm_kernel launch).
    """
    from hand_gesture_recognition_amd import ops
    rec = []

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

    def g_fwd(x, w, y, stats, k, s, stats_tiles=None):
        Cin, Cout, M = x.shape[-1], y.shape[-1], y.numel() // y.shape[-1]
        return 2.0 * M * Cout * k * k * Cin, nbytes(x, w, y), k

    def g_dgrad(dy, wt, dx, res, k, s, bn=None):
        Cout, Cin, Mout = dy.shape[-1], dx.shape[-1], dy.numel() // dy.shape[-1]
        # (a launch that also writes the BatchNorm-backward sums of the next unit reads that unit's raw output too)
        extra = (bn[0], bn[1]) if bn is not None else ()
        # 3x3 stride-2 data gradients run dgrad2_kernel, not the igemm template: their own family below
        return 2.0 * Mout * Cout * k * k * Cin, nbytes(dy, wt, dx, res, *extra), (32 if (k == 3 and s == 2) else k)

    def g_lin(x, w, out, bias=None, act=0, aux_out=None, res=None):
        K, N = x.shape[-1], out.shape[-1]
        M = x.numel() // K
        return 2.0 * M * N * K, nbytes(x, w, out, aux_out, res), 1

    saved = (ops.conv_fwd, ops.conv_dgrad, ops.linear_fwd)
    ops.conv_fwd, ops.conv_dgrad, ops.linear_fwd = (wrap("conv_fwd", saved[0], g_fwd), wrap("conv_dgrad", saved[1], g_dgrad),
                                                    wrap("linear", saved[2], g_lin))
    try:
        step._eager(*batch)
        torch.cuda.synchronize()
    finally:
        ops.conv_fwd, ops.conv_dgrad, ops.linear_fwd = saved
    all_times = [r[2].elapsed_time(r[3]) * 1e-3 for r in rec]
    d2 = [(r, t) for r, t in zip(rec, all_times) if r[1][2] == 32]
    keep = [(r, t) for r, t in zip(rec, all_times) if r[1][2] != 32]
    rec, times = [r for r, _ in keep], [t for _, t in keep]
    flops = sum(r[1][0] for r in rec)
    abytes = sum(r[1][1] for r in rec)
    secs = sum(times)
    achieved = flops / secs / 1e12
    # the template's two regimes, reported separately as well: 3x3 convolutions (forward and data gradient;
    # MFMA-bound: 288..1152 flop/B) and 1x1 convolutions / Linears (HBM-bound: 64..256 flop/B)
    fam = {}
    for r, t in zip(rec, times):
        k = "conv3x3" if r[1][2] == 3 else "pointwise"
        f = fam.setdefault(k, [0.0, 0.0, 0.0, 0])
        f[0] += r[1][0]; f[1] += r[1][1]; f[2] += t; f[3] += 1
    families = {
        "conv3x3": {"bound": "mfma", "achieved": round(fam["conv3x3"][0] / fam["conv3x3"][2] / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(fam["conv3x3"][0] / fam["conv3x3"][2] / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                    "launches_per_step": fam["conv3x3"][3], "share_of_step_ms": round(fam["conv3x3"][2] * 1e3, 3)},
        "pointwise": {"bound": "hbm", "achieved": round(fam["pointwise"][1] / fam["pointwise"][2] / 1e9, 1), "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": round(fam["pointwise"][1] / fam["pointwise"][2] / 1e9 / HBM_PEAK_GBS, 4),
                      "launches_per_step": fam["pointwise"][3], "share_of_step_ms": round(fam["pointwise"][2] * 1e3, 3)},
    }
    if d2:
        f2, b2, t2 = sum(r[1][0] for r, _ in d2), sum(r[1][1] for r, _ in d2), sum(t for _, t in d2)
        families["conv3x3_stride2_dgrad"] = {"kernel": "dgrad2_kernel", "bound": "hbm", "achieved": round(b2 / t2 / 1e9, 1),
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b2 / t2 / 1e9 / HBM_PEAK_GBS, 4),
                                             "tflops": round(f2 / t2 / 1e12, 1), "launches_per_step": len(d2),
                                             "share_of_step_ms": round(t2 * 1e3, 3)}
    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    # (FETCH_SIZE / WRITE_SIZE collected separately and corrected as MI355X_MICROARCH.md prescribes),
    # averaged over all instances of the template
    traffic, src = None, None
    import re
    cands = [n for n in os.listdir(os.path.join(ROOT, "profiles")) if re.match(r"r\d+_pmc_hbm_traffic.*\.json$", n)]
    # newest collection: highest round, then highest version (r02_..._v3 > r02_..._v2 > r01_..._v10)
    cands.sort(key=lambda n: [int(x) for x in re.findall(r"\d+", n)], reverse=True)
    for name in cands:
        pm = json.load(open(os.path.join(ROOT, "profiles", name)))
        tot = sum(v["hbm_bytes_per_dispatch"] * v["dispatches"] for k, v in pm.items() if k.startswith("igemm_kernel<bf16_t"))
        n = sum(v["dispatches"] for k, v in pm.items() if k.startswith("igemm_kernel<bf16_t"))
        if n:
            traffic, src = round(tot / n), "profiles/" + name
        break
    return {"bound": "mfma", "kernel": "igemm_kernel<bf16_t,...> (all instances: conv fwd + conv dgrad + ViT linears, one launch per call; not counted: "
                      "the 4 fc2 data gradients fused with GELU' / bias sums, whose call brackets three launches)",
            "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": src,
            "launches_per_step": len(rec), "avg_launch_us": round(secs / len(rec) * 1e6, 1),
            "flops_per_launch": round(flops / len(rec)), "algorithmic_bytes_per_launch": round(abytes / len(rec)),
            "share_of_step_ms": round(secs * 1e3, 3), "families": families}


def cpu_baseline(size):
    """The oracle (CPU port of the reference path) timed on this box's host cores.

    Threads = the box's CPU share for one GPU (16): measured on the MI355X host, torch-CPU runs this
    model fastest at 16 threads (8: 53, 16: 68, 32: 33, 64: 18, 256: 0.1 img/s -- oversubscription).
    Sample: ~10 s of B=16 training steps of the same workload shape.
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
    return {"value": round(B * n / dt, 2), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"{n} train steps (fwd+loss+bwd+AdamW) at B={B}, {size}x{size}, fp32, oracle/mtn_oracle.py "
                      f"(torch-CPU ops), {threads} threads of {os.cpu_count()} logical CPUs"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from hand_gesture_recognition_amd.multitasknet import MultiTaskNet
    from hand_gesture_recognition_amd.train import TrainStep, synthetic_batch

    dt = {"bf16": torch.bfloat16, "fp32": torch.float32, "bf16x3": "bf16x3"}[a.dtype]
    torch.manual_seed(42)
    model = MultiTaskNet(21, 19, [a.size, a.size], compute_dtype=dt).to(dev)
    step = TrainStep(model, a.batch, lr=1e-3, use_graph=bool(a.graph))
    batch = synthetic_batch(a.batch, (a.size, a.size), 21, 19, dev, 1234 + rank)

    for _ in range(a.warmup):
        step(*batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step(*batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed)
    loss_val = [float(x) for x in losses]
    assert all(x == x and abs(x) < 1e6 for x in loss_val), f"non-finite losses {loss_val}"

    roof = None
    if not a.no_roofline and a.dtype == "bf16":
        roof = igemm_roofline(step, batch)          # every rank runs it (the step holds a collective)
    if rank == 0:
        ips = world * a.batch * a.steps / elapsed
        out = {
            "metric": f"images/sec training step, MultiTaskNet {a.size}x{a.size} B={a.batch}", "value": round(ips, 1),
            "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"MultiTaskNet(21 joints, 19 classes) {a.size}x{a.size} train step "
                                   f"(fwd + 0.001*CE + JointsMSE + bwd + AdamW), B={a.batch}/GPU, random-init weights",
                       "global_batch": world * a.batch, "parallelism": f"dp{world}", "hipgraph": bool(a.graph)},
            "algorithmic_tflops": round(ips * TRAIN_GFLOP_PER_IMG / 1e3, 1) if a.size == 192 else None,
            "final_losses": [round(x, 6) for x in loss_val],
        }
        if roof is not None:
            out["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.size)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
