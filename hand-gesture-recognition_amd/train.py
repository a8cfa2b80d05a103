"""Training step + flag surface of the reference's ``train.py`` for the MI355X-native MultiTaskNet.

What is mirrored (reference train.py): the loss recipe ``0.001 * CE + JointsMSE`` (:63-64,75),
``AdamW(model.parameters(), lr)`` with PyTorch defaults (:50-51), ``MultiStepLR(lr_step,
lr_factor)`` stepped per epoch (:53-56) and the argparse flags (:244-281).  Lightning, the HaGRID
dataloader, TensorBoard and the CPU metrics are out of scope: the step consumes
tensors that follow the batch contract of libs/load.py (img, label, target, target_weight).

New surface (not in the reference, which is single-device): one process per GPU under
``torchrun``; gradients live in ONE flat fp32 buffer that is all-reduced (RCCL over xGMI, sum) in
five contiguous buckets launched from inside backward as each becomes final (one bucket between
the two graphs under hipGraph replay); the 1/world_size mean is folded into the fused AdamW
kernel.  BatchNorm statistics stay per GPU (no SyncBN), so DDP parity is "mean of per-rank
gradients".
"""
import argparse

import torch
import torch.distributed as dist
from torch import nn

from . import ops
from .multitasknet import MultiTaskNet


# ------------------------------------------------------------------------------------------------
# loss modules with the reference's interface (libs/loss.py)
# ------------------------------------------------------------------------------------------------
class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, hm, label, target, tw, ce_weight):
        logits, hm = logits.contiguous().float(), hm.contiguous().float()
        dl, dh = torch.empty_like(logits), torch.empty_like(hm)
        part = torch.empty(ops.loss_partial_len(hm.numel()), dtype=torch.float32, device=hm.device)
        losses = torch.empty(3, dtype=torch.float32, device=hm.device)
        ops.loss_fwd_bwd(logits, label.contiguous(), hm, target.contiguous().float(),
                         tw.contiguous().float(), dl, dh, part, losses, ce_weight)
        ctx.save_for_backward(dl, dh)
        return losses

    @staticmethod
    def backward(ctx, g):
        dl, dh = ctx.saved_tensors
        # d total = d(ce part) + d(joints part); losses = (total, ce, joints)
        gl = (g[0] + g[1]) * dl
        gh = (g[0] + g[2]) * dh
        return gl, gh, None, None, None, None


class MultiTaskCriterion(nn.Module):
    """(total, class_loss, joints_loss) = (0.001*CE + JMSE, 0.001*CE, JMSE)  -- train.py:63-64,75."""

    def __init__(self, class_weight=0.001):
        super().__init__()
        self.class_weight = class_weight

    def forward(self, pred_label, heatmap, label, target, target_weight):
        losses = _FusedLoss.apply(pred_label, heatmap, label, target, target_weight, self.class_weight)
        return losses[0], losses[1], losses[2]


class JointsMSELoss(nn.Module):
    """libs/loss.py:4-30: 0.5 * mean((w (p - t))^2); with ``use_target_weight=False`` the weights
    are ignored (libs/loss.py:26-28), i.e. w = 1 for every joint."""

    def __init__(self, use_target_weight=True):
        super().__init__()
        self.use_target_weight = bool(use_target_weight)

    def forward(self, output, target, target_weight=None):
        B = output.shape[0]
        if not self.use_target_weight or target_weight is None:
            if self.use_target_weight:
                raise ValueError("JointsMSELoss(use_target_weight=True) needs target_weight [B,J,1]")
            target_weight = torch.ones(B, output.shape[1], 1, device=output.device)
        logits = torch.zeros(B, 1, device=output.device)
        label = torch.zeros(B, dtype=torch.int64, device=output.device)
        return _FusedLoss.apply(logits, output, label, target, target_weight, 0.0)[2]


class ClassificationLoss(nn.Module):
    """libs/loss.py:33-40: CrossEntropyLoss(reduction='mean')."""

    def forward(self, output, target):
        B = output.shape[0]
        z = torch.zeros(B, 1, 1, 1, device=output.device)
        return _FusedLoss.apply(output, z, target, z, torch.zeros(B, 1, 1, device=output.device), 1.0)[1]


# ------------------------------------------------------------------------------------------------
# fused training step
# ------------------------------------------------------------------------------------------------
class TrainStep:
    """forward + loss + backward (+ gradient all-reduce) + AdamW as one device-side sequence.

    No host synchronisation inside ``__call__``: the returned losses tensor is device resident.
    With ``use_graph`` the whole step is captured once into a hipGraph and replayed.
    """

    def __init__(self, model: MultiTaskNet, batch_size, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.01, lr_step=(30, 40), lr_factor=0.1, process_group=None,
                 use_graph=False, want_attn=True, overlap_comm=True):
        self.model, self.B = model, int(batch_size)
        self.eng = model.engine(self.B)
        self.generation = model._generation      # Adam moments are laid out like THIS flat parameter buffer
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        dev = model.flat_params.device
        self.m = torch.zeros_like(model.flat_params)
        self.v = torch.zeros_like(model.flat_params)
        self.state = torch.zeros(16, dtype=torch.float32, device=dev)
        self.state[:5] = torch.tensor([lr, betas[0], betas[1], eps, weight_decay])
        self.state[8] = 1.0 / self.world
        self.base_lr, self.lr_step, self.lr_factor, self.epoch = lr, tuple(lr_step), lr_factor, 0
        self.want_attn = want_attn
        self.use_graph, self._graph, self._static = use_graph, None, None
        if self.world > 1:
            dist.broadcast(model.flat_params, src=0, group=process_group)   # identical replicas
        # Gradient buckets in backward order (head, stages last..first, stem): each is a contiguous
        # slice of the flat gradient buffer and is all-reduced asynchronously the moment the engine
        # reports it final, overlapping the reduction with the rest of the backward pass -- the
        # parameter-heavy stages finish first, the long stage-1 backward hides their traffic.
        self._buckets, self._works = self._bucket_ranges(model), []
        # overlap_comm="force" runs the bucketed path on a one-rank group too (RCCL smoke test)
        self.overlap = bool(overlap_comm) and not use_graph and (self.world > 1 or (
            overlap_comm == "force" and dist.is_available() and dist.is_initialized()))
        # instrumentation (bench.py --gpus N): event pairs around the wait for the gradient all-reduce on
        # the compute stream; their elapsed time is the communication the backward pass did NOT hide
        self.comm_events = None
        model.train()

    # -- schedule (MultiStepLR stepped once per epoch, train.py:53-56) -----------------------------
    def epoch_end(self):
        self.epoch += 1
        lr = self.base_lr * (self.lr_factor ** sum(1 for ms in self.lr_step if self.epoch >= ms))
        self.state[0:1].fill_(lr)
        return lr

    def _fwd_bwd(self, img, label, target, tw):
        e = self.eng
        if self.generation != self.model._generation:
            raise RuntimeError("the module's parameter storage was replaced after this TrainStep was built "
                               "(load_state_dict(assign=True) / .to() / _flatten()): its plan and Adam moments "
                               "address the old buffers; build a new TrainStep")
        e.forward(img, training=True, want_attn=self.want_attn)
        e.loss(label, target, tw, with_grads=True)
        # the bucket hook lives on the (shared) engine only for the duration of THIS backward: a
        # model(img) + loss.backward() on the same plan (validation, debugging) must not fire
        # collectives nobody waits for
        self._works = []
        e.on_grads_ready = self._bucket_ready if self.overlap else None
        try:
            e.backward()
        finally:
            e.on_grads_ready = None

    def _update(self):
        mdl = self.model
        ops.adamw_step(mdl.flat_params, mdl.flat_grads, self.m, self.v, self.state)
        self._weights_moved()

    def _weights_moved(self):
        """The fused AdamW writes the flat parameter buffer through raw pointers (no version bump), and a graph replay packs
        and updates without running any Python: the plan's record of what its packed weight copy holds is void after
        either -- an eval forward with ``refold=False`` must fold again rather than serve this step's unfolded pack."""
        self.eng._packed = None

    @staticmethod
    def _bucket_ranges(model):
        """{tag: (start, end)} over the flat gradient buffer; tags as Engine.on_grads_ready fires them."""
        ranges, off = {}, 0
        for name, p in model.named_parameters():
            part = name.split(".")
            if part[0] == "encoder" and part[1] in ("conv1", "conv2"):
                tag = "stem"
            elif part[0] == "encoder":
                tag = ("stage", int(part[1][-1]) - 1)          # cspelan<k> and the down<k> that follows it
            else:
                tag = "head"
            a, b = ranges.get(tag, (off, off))
            assert b == off, f"gradient bucket {tag} is not contiguous at {name}"
            ranges[tag] = (a, off + p.numel())
            off += p.numel()
        return ranges

    def _bucket_ready(self, tag):
        a, b = self._buckets[tag]
        self._works.append(dist.all_reduce(self.model.flat_grads[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _sync_grads(self):
        """RCCL all-reduce (sum) of the flat gradient buffer; AdamW applies 1/world.  With
        ``overlap_comm`` the buckets were launched during backward and are only waited for here."""
        if self.overlap:
            if len(self._works) != len(self._buckets):
                raise RuntimeError(f"{len(self._works)} of {len(self._buckets)} gradient buckets were reported final")
            for w in self._works:
                w.wait()
            self._works = []
        elif self.world > 1:
            dist.all_reduce(self.model.flat_grads, op=dist.ReduceOp.SUM, group=self.pg)

    def _eager(self, img, label, target, tw):
        self._fwd_bwd(img, label, target, tw)
        if self.comm_events is not None and img.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._sync_grads()
            e1.record()
            self.comm_events.append((e0, e1))
        else:
            self._sync_grads()
        self._update()
        return self.eng.losses

    def comm_exposed_ms(self):
        """Mean time per step the compute stream spent waiting for the gradient all-reduce (needs
        ``comm_events = []`` before the steps and a device synchronisation after them)."""
        ev = self.comm_events or []
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else 0.0

    def _capture(self, img, label, target, tw):
        """Warm up once eagerly (lazy kernel attributes, allocator), capture, then restore every piece
        of training state the warm-up step touched so that capture is invisible to the trajectory."""
        self._static = [t.clone() for t in (img, label, target, tw)]
        mdl = self.model
        keep = [mdl.flat_params, self.m, self.v, self.state] + list(mdl.buffers())
        saved = [t.clone() for t in keep]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._eager(*self._static)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if self.world == 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._fwd_bwd(*self._static)
                self._update()
            self._graph = (g,)
        else:                                   # RCCL all-reduce stays outside the graphs
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self._fwd_bwd(*self._static)
            with torch.cuda.graph(g2):
                self._update()
            self._graph = (g1, g2)
        with torch.no_grad():
            for t, sv in zip(keep, saved):
                t.copy_(sv)
        torch.cuda.synchronize()

    def __call__(self, img, label, target, target_weight):
        if not (self.use_graph and img.is_cuda):
            return self._eager(img, label, target, target_weight)
        if self._graph is None:
            self._capture(img, label, target, target_weight)
        for dst, src in zip(self._static, (img, label, target, target_weight)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self._graph[0].replay()
        if self.world > 1:
            self._sync_grads()
            self._graph[1].replay()
        self._weights_moved()
        return self.eng.losses

    # -- optimizer state for checkpoints ------------------------------------------------------------
    def state_dict(self):
        return {"exp_avg": self.m.clone(), "exp_avg_sq": self.v.clone(), "state": self.state.clone(),
                "epoch": self.epoch}

    def load_state_dict(self, sd):
        self.m.copy_(sd["exp_avg"]); self.v.copy_(sd["exp_avg_sq"]); self.state.copy_(sd["state"])
        self.epoch = int(sd["epoch"])


def synthetic_batch(batch, image_size, num_joints, num_classes, device, seed):
    """Synthetic stand-in for the dataloader contract."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    S = image_size
    img = torch.randn(batch, 3, S[0], S[1], generator=g)
    label = torch.randint(0, num_classes, (batch,), generator=g)
    target = torch.rand(batch, num_joints, S[0] // 4, S[1] // 4, generator=g)
    tw = torch.ones(batch, num_joints, 1)
    return tuple(t.to(device) for t in (img, label, target, tw))


def build_parser():
    """The reference's flag surface (train.py:244-281) plus --devices / --steps for synthetic runs."""
    p = argparse.ArgumentParser(description="MI355X-native MultiTaskNet training (synthetic data)")
    p.add_argument("--data_config", type=str, default=None, help="YAML with num_joints / num_classes")
    p.add_argument("--suffix", type=str, default="run")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--devices", type=int, default=1, help="number of GPUs (launch with torchrun)")
    p.add_argument("--backbone", type=str, default="gelans")
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--epochs", type=int, default=50)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--lr_step", type=int, nargs="+", default=[30, 40])
    p.add_argument("--lr_factor", type=float, default=0.1)
    p.add_argument("--image_size", type=int, nargs=2, default=[192, 192])
    p.add_argument("--sigma", type=int, default=2)
    p.add_argument("--log_dir", type=str, default="logs")
    p.add_argument("--save_dir", type=str, default="output")
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--steps_per_epoch", type=int, default=10, help="synthetic steps per epoch")
    p.add_argument("--val_steps", type=int, default=1, help="synthetic validation batches per epoch (0: no validation, no checkpoints)")
    p.add_argument("--test_steps", type=int, default=1, help="synthetic test batches after the fit (0: no test epoch)")
    p.add_argument("--no_checkpoint", dest="checkpoint", action="store_false", help="skip the best / last checkpoint policy")
    p.add_argument("--compute_dtype", type=str, default="bf16", choices=["bf16", "fp32", "bf16x3"])
    return p


class EpochMeters:
    """The per-step metrics of the reference's training / validation steps (train.py:66-73,90-99,110-121) kept on the device:
    macro F1 from a running confusion matrix and the count-weighted pose accuracy (PCK of the decoded heat-map maxima) --
    no host synchronisation per step; ``result()`` reads them once per epoch."""

    def __init__(self, num_classes, device):
        from .metrics import ClassificationMeter
        self.cls = ClassificationMeter(num_classes, device)
        self.acc_sum = torch.zeros((), dtype=torch.float32, device=device)
        self.cnt = torch.zeros((), dtype=torch.float32, device=device)
        self.loss_sum = torch.zeros(3, dtype=torch.float32, device=device)
        self.steps = 0

    def reset(self):
        self.cls.reset()
        self.acc_sum.zero_(); self.cnt.zero_(); self.loss_sum.zero_()
        self.steps = 0

    def update(self, logits, heatmaps, label, target, losses):
        from .metrics import pose_accuracy
        self.cls.update(logits, label)
        _, avg_acc, cnt, _ = pose_accuracy(heatmaps, target)
        self.acc_sum += avg_acc * cnt                       # train.py:93-94: total_acc += avg_acc * cnt
        self.cnt += cnt
        self.loss_sum += losses
        self.steps += 1

    def result(self, prefix):
        n = max(self.steps, 1)
        tot, cl, jl = (float(x) / n for x in self.loss_sum)
        cnt = float(self.cnt)
        return {f"{prefix}/total_loss": tot, f"{prefix}/class_loss": cl, f"{prefix}/joints_loss": jl,
                f"{prefix}/cls_f1score": float(self.cls.f1()), f"{prefix}/pose_acc": float(self.acc_sum) / cnt if cnt > 0 else 0.0}


def validate(model, eng, batches, meters):
    """The reference's validation epoch (train.py:110-138; Lightning runs it under model.eval() and no_grad): forward with the
    running BatchNorm statistics folded into the weights, the loss recipe without gradients, the same metrics."""
    was_training = model.training
    model.eval()
    meters.reset()
    with torch.no_grad():
        for img, label, target, tw in batches:
            eng.forward(img, training=False, want_attn=False)
            losses = eng.loss(label, target, tw, with_grads=False)
            meters.update(eng.logits, eng.hm, label, target, losses)
    model.train(was_training)
    return meters.result("val")


def test(model, eng, batches, num_classes, out_dir=None, names=None):
    """The reference's test epoch (train.py:133-138,176-188; ``trainer.test(module, dm, "best")``): eval-mode forward over the
    test batches, predicted vs true labels into one confusion matrix on the device, macro F1 printed as the reference prints
    it, the matrix saved under ``out_dir`` (``confusion_matrix.npy``, and the reference's ``confusion_matrix.png`` when
    matplotlib is importable).  Rows = predicted class, columns = true class: the reference calls
    ``confusion_matrix(y_pred, y_true)`` with the arguments in that order (train.py:181)."""
    import os
    from .metrics import ClassificationMeter
    meter = ClassificationMeter(num_classes, batches[0][0].device)
    was_training = model.training
    model.eval()
    with torch.no_grad():
        for img, label, _target, _tw in batches:
            eng.forward(img, training=False, want_attn=False)
            meter.update(eng.logits, label)
    model.train(was_training)
    f1 = float(meter.f1())
    cm = meter.cm.cpu().numpy()                            # the meter counts cm[pred][label] (include/mtn.h: mtn_cls_confusion)
    print("Test F1 Score: {:.4f}".format(f1), flush=True)                                    # train.py:178
    if out_dir is not None:
        import numpy as np
        os.makedirs(out_dir, exist_ok=True)
        np.save(os.path.join(out_dir, "confusion_matrix.npy"), cm)
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            fig, ax = plt.subplots(figsize=(10, 10))
            ax.imshow(cm, cmap="Blues")
            ticks = list(names) if names is not None else list(range(num_classes))
            ax.set_xticks(range(num_classes), ticks, rotation=90)
            ax.set_yticks(range(num_classes), ticks)
            for i in range(num_classes):
                for j in range(num_classes):
                    ax.text(j, i, int(cm[i, j]), ha="center", va="center")
            plt.tight_layout()
            plt.savefig(os.path.join(out_dir, "confusion_matrix.png"))
            plt.close(fig)
        except Exception:                                  # plotting is a convenience of the reference, not a result
            pass
    return f1, cm


def run(args, data_cfg, device=None):
    """The reference's fit loop (train.py:186-240) on synthetic batches: per epoch ``steps_per_epoch`` training steps with
    the step metrics of train.py:66-73 accumulated on the device, MultiStepLR, a validation pass (``val_steps`` batches of
    another seed, eval mode) and the checkpoint policy of train.py:214-221 -- ``<save_dir>/<backbone>_<H>x<W>_<suffix>/weight/
    best.ckpt`` (top-1 on val/total_loss, min) + ``last.ckpt``, weights only, Lightning-compatible -- then the test epoch on
    the best checkpoint (train.py:239-240)."""
    import os
    from .checkpoint import CheckpointPolicy
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.devices != world:
        raise SystemExit(f"--devices {args.devices} needs one process per GPU: launch with `python -m "
                         f"torch.distributed.run --nproc-per-node {args.devices} ...` (WORLD_SIZE is {world})")
    local = int(os.environ.get("LOCAL_RANK", str(args.device)))
    if device is None:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        if world > 1 and not dist.is_initialized():
            dist.init_process_group("nccl")
    torch.manual_seed(42)                                                                 # train.py:21
    dt = {"bf16": torch.bfloat16, "fp32": torch.float32, "bf16x3": "bf16x3"}[args.compute_dtype]
    J, C = data_cfg["num_joints"], data_cfg["num_classes"]
    model = MultiTaskNet(J, C, args.image_size, compute_dtype=dt).to(device)
    step = TrainStep(model, args.batch_size, lr=args.lr, lr_step=args.lr_step, lr_factor=args.lr_factor)
    rank = dist.get_rank() if (world > 1 and dist.is_initialized()) else 0
    batch = synthetic_batch(args.batch_size, args.image_size, J, C, device, 1234 + rank)
    val_batches = [synthetic_batch(args.batch_size, args.image_size, J, C, device, 4321 + 17 * i + rank)
                   for i in range(max(0, getattr(args, "val_steps", 1)))]
    train_m, val_m = EpochMeters(C, device), EpochMeters(C, device)
    save_path = os.path.join(args.save_dir, "{}_{}x{}_{}".format(args.backbone, args.image_size[0], args.image_size[1],
                                                                 args.suffix))           # train.py:190-193
    policy = CheckpointPolicy(os.path.join(save_path, "weight")) if getattr(args, "checkpoint", True) else None
    history, gstep = [], 0
    for epoch in range(args.epochs):
        train_m.reset()                                                                   # train.py:140-142
        for _ in range(args.steps_per_epoch):
            losses = step(*batch)
            train_m.update(step.eng.logits, step.eng.hm, batch[1], batch[2], losses)
            gstep += 1
        lr = step.epoch_end()
        log = train_m.result("train")
        if val_batches:
            log.update(validate(model, step.eng, val_batches, val_m))
            if policy is not None:
                log["checkpoints"] = policy.step(model, log, epoch, gstep, is_main=rank == 0)
        history.append(log)
        if rank == 0:
            msg = " ".join(f"{k} {v:.6f}" for k, v in log.items() if isinstance(v, float))
            print(f"epoch {epoch}: {msg} lr {lr:g}", flush=True)
    run.history, run.test_result = history, None
    test_steps = max(0, getattr(args, "test_steps", 1))
    if test_steps:                                                                        # train.py:239-240
        if world > 1:
            dist.barrier()                                                                # (rank 0 has written best.ckpt)
        best = os.path.join(save_path, "weight", "best.ckpt")
        if policy is not None and os.path.exists(best):
            from .checkpoint import load_checkpoint
            load_checkpoint(best, model)                                                  # trainer.test(..., "best"): every rank
        batches = [synthetic_batch(args.batch_size, args.image_size, J, C, device, 9876 + 13 * i + rank) for i in range(test_steps)]
        run.test_result = test(model, step.eng, batches, C, out_dir=save_path if rank == 0 else None,
                               names=(data_cfg.get("names") or {}).keys() or None)
    return model


def main(argv=None):
    args = build_parser().parse_args(argv)
    cfg = {"num_joints": 21, "num_classes": 19}
    if args.data_config:
        import yaml
        with open(args.data_config) as f:
            y = yaml.safe_load(f)
        cfg = {"num_joints": y["num_joints"], "num_classes": y["num_classes"]}
    run(args, cfg)


if __name__ == "__main__":
    main()
