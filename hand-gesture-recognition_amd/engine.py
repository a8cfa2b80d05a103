"""Static execution plan of the MultiTaskNet hot path: explicit forward and backward over
preallocated HBM buffers, every arithmetic step a HIP kernel reached through ``ops`` (C ABI).

Data layout in HBM
  * activations: NHWC, storage dtype T (bf16 throughput mode / fp32 validation mode);
  * the GELAN concat (model/gelan.py:138,142) is ONE [B,H,W,2C] buffer per stage: cv1 writes
    channels [0,C), the two residual branches write [C,3C/2) and [3C/2,2C), cv4 reads it whole --
    ``chunk`` and ``cat`` cost nothing; gradients use a mirror buffer and accumulate in place;
  * every Conv keeps its raw (pre-BatchNorm) output: it is what BN/SiLU backward needs;
  * tokens [B, N, 256]; qkv [B*N, 768]; attention reads/writes those layouts directly;
  * parameters / gradients: flat fp32 buffers owned by the module (nn.Parameters are views);
    kernels read a repacked copy in T ([O][kh][kw][I] forward, [I][kh][kw][O] data-gradient).
  * all parameter gradients are WRITTEN (not accumulated) once per backward.

Reference op order: model/multitasknet.py:24-29 -> model/gelan.py:165-176 ->
model/transformer.py:129-152; loss libs/loss.py + train.py:63-64,75.
"""
import os

import torch

from . import ops

_ACT = {True: ops.ACT_SILU, False: ops.ACT_NONE}


class _Unit:
    """One Conv2d(+BatchNorm) site and its per-layer buffers."""

    def __init__(self, mod, cin, cout, k, s, hin, win, act=True, bn=True):
        self.mod, self.cin, self.cout, self.k, self.s, self.act, self.bn = mod, cin, cout, k, s, act, bn
        self.hin, self.win = hin, win
        self.hout, self.wout = (hin + 2 * (k // 2) - k) // s + 1, (win + 2 * (k // 2) - k) // s + 1


class Engine:
    def __init__(self, model, batch, dtype):
        from .multitasknet import resolve_compute_dtype
        self.model, self.B = model, int(batch)
        self.dt, self.f32_code = resolve_compute_dtype(dtype)      # torch dtype of the buffers, what an fp32 buffer holds
        # the wrappers with this plan's mode bound (ops.with_mode): what a torch.float32 buffer of the plan holds -- plain fp32
        # or split storage -- travels with every call, like the dtype argument of the C ABI (no process-wide state)
        self.o = ops.with_mode(self.f32_code)
        self.x3 = self.dt == torch.float32 and self.f32_code == ops.F32X3
        self.dev = model.flat_params.device
        self.generation = model._generation       # the flat buffers this plan's pointers were taken from
        H, W = model.image_size
        stride = model.encoder.total_stride
        if H % stride or W % stride:
            raise ValueError(f"image_size must be a multiple of {stride} (total stride of GELANNet)")
        self.H, self.W = H, W
        self.fh, self.fw = H // stride, W // stride
        self.P = self.fh * self.fw
        self.N = self.P + 1
        self.J, self.C = model.num_joints, model.num_classes
        self.heads, self.hd, self.dim = 8, 32, 256
        self._bufs = {}
        # optional callback(tag) fired inside backward() as soon as a contiguous group of parameter
        # gradients is final: "head" (projection + decoder), ("stage", i), "stem" -- in that order
        self.on_grads_ready = None
        # bumped by every pass that overwrites the saved activations (backbone_forward / vit_forward): an autograd node of an
        # older pass compares it with the value it saw and refuses to run backward on another pass's activations
        self.serial = [0, 0]            # [backbone buffers, ViT buffers]
        # data gradients pre-reduce the BatchNorm backward of the unit they feed (MTN_FUSE_BN_BWD=0: two-pass backward)
        self.fuse_bn_bwd = os.environ.get("MTN_FUSE_BN_BWD", "1") != "0"
        self.fuse_gelu_bwd = os.environ.get("MTN_FUSE_GELU_BWD", "1") != "0"     # (0: three launches per FeedForward)
        # LayerNorm / bias gradients of the ViT are leaves: their partial-sum slabs are folded by ONE grouped call at the end
        # of the bucket instead of two launches per slab (MTN_DEFER_LEAF_SUMS=0: fold each one where it is produced)
        self.defer_leaf_sums = os.environ.get("MTN_DEFER_LEAF_SUMS", "1") != "0"
        # the token assembly in the projection's epilogue (MTN_FUSE_TOKENS=0: projection, then the tokens_fwd launch)
        self.fuse_tokens = os.environ.get("MTN_FUSE_TOKENS", "1") != "0"
        # the sixteen Linear weight gradients of the ViT as ONE grouped launch at the end of its backward (mtn_linear_wgrad_group):
        # their output gradients then live in per-layer buffers instead of the shared temporaries (MTN_GROUP_WGRAD=0: one launch each)
        self.group_wgrad = os.environ.get("MTN_GROUP_WGRAD", "1") != "0"
        # ... and the four 3x3 weight gradients of each stage's residual blocks as one grouped launch at the end of the stage's
        # backward (mtn_conv_wgrad_group; MTN_GROUP_WGRAD=vit: the ViT's only)
        self.group_wgrad3 = os.environ.get("MTN_GROUP_WGRAD", "1") not in ("0", "vit")
        self._build()

    # ---------------------------------------------------------------- buffers
    def _new(self, shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.dt, device=self.dev)

    def _f32(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def _build(self):
        B, enc = self.B, self.model.encoder
        H2, W2 = self.H // 2, self.W // 2
        units = []

        def unit(mod, hin, win, bn=True):
            if isinstance(mod, torch.nn.Conv2d):
                u = _Unit(mod, mod.in_channels, mod.out_channels, mod.kernel_size[0], mod.stride[0], hin, win,
                          act=False, bn=False)
            else:
                u = _Unit(mod, mod.c1, mod.c2, mod.k, mod.s, hin, win, act=mod.has_act, bn=bn)
            units.append(u)
            return u

        self.u_conv1 = unit(enc.conv1, self.H, self.W)
        self.u_conv2 = unit(enc.conv2, H2, W2)
        self.u_conv1.tag, self.u_conv2.tag = "conv1", "conv2"
        self.stages = []
        h, w = self.H // 4, self.W // 4
        for si, (blk, down) in enumerate(enc.stage_modules()):
            # st["branches"][j] = the residual blocks of branch j (cv2, cv3), each {"c1", "c2"} units
            st = {"C": blk.c_out, "h": h, "w": w}
            st["cv1"] = unit(blk.cv1, h, w)
            st["branches"] = [[{"c1": unit(rb.cv1, h, w), "c2": unit(rb.cv2, h, w)} for rb in seq]
                              for seq in (blk.cv2, blk.cv3)]
            st["cv4"] = unit(blk.cv4, h, w)
            st["down"] = unit(down, h, w) if down is not None else None
            st["cv1"].tag, st["cv4"].tag = f"s{si + 1}.cv1", f"s{si + 1}.cv4"
            if down is not None:
                st["down"].tag = f"s{si + 1}.down"
            for j, br in enumerate(st["branches"]):
                for i, rb in enumerate(br):
                    rb["c1"].tag, rb["c2"].tag = f"s{si + 1}.b{j}.{i}.c1", f"s{si + 1}.b{j}.{i}.c2"
            self.stages.append(st)
            h, w = h // 2, w // 2
        self.u_proj = unit(self.model.proj, self.fh, self.fw)
        self.u_proj.tag = "proj"
        self.units = units
        # TIMING ABLATIONS (tools/ablate_step.py; wrong numerics, never set by the product): tags of units whose
        # bn_act_fwd / whole BatchNorm backward is skipped to measure what a fusion could save at most
        self.ablate_fwd, self.ablate_bwd = frozenset(), frozenset()

        # ---- packed weights: descriptor table for mtn_pack_weights ---------------------------------
        flat = self.model.flat_params
        desc, off = [], 0

        def add_pack(p, O, I, taps, mode):
            nonlocal off
            if self.x3 and not (mode & 2) and ((O if mode & 1 else I) * taps) % 32:
                raise ValueError(f"split storage packs weight rows in 32-element blocks: {tuple(p.shape)} has rows of "
                                 f"{(O if mode & 1 else I) * taps} (use compute_dtype=torch.float32 for this model variant)")
            src = (p.data_ptr() - flat.data_ptr()) // 4
            desc.append([src, off, O, I, taps, mode, -1])
            o = off
            off += O * I * taps
            off = (off + 7) & ~7           # keep every packed tensor 16-byte aligned
            return o, O * I * taps

        packs = []
        for u in units:
            wgt = u.mod.weight if isinstance(u.mod, torch.nn.Conv2d) else u.mod.conv.weight
            u.weight = wgt
            u.pk_row = len(desc)                     # row of the forward-layout entry in the pack table
            # (mode + 2: plain fp32 also in the split-bf16 mode -- the stem kernels multiply in exact f32)
            u.pk_f = add_pack(wgt, u.cout, u.cin, u.k * u.k, 2 if u is self.u_conv1 else 0)
            u.pk_d = add_pack(wgt, u.cout, u.cin, u.k * u.k, 1) if u is not self.u_conv1 else None
            packs.append(u)
        self.layers = []
        for attn, ff in self.model.decoder.transformer.layers:
            L = {"attn": attn, "ff": ff}
            for key, lin in (("qkv", attn.to_qkv), ("out", attn.to_out), ("fc1", ff.net[1]), ("fc2", ff.net[4])):
                L[key] = lin
                L[key + "_f"] = add_pack(lin.weight, lin.out_features, lin.in_features, 1, 0)
                L[key + "_d"] = add_pack(lin.weight, lin.out_features, lin.in_features, 1, 1)
            self.layers.append(L)
        self.wpack = self._new((off,))
        self.pack_desc = torch.tensor(desc, dtype=torch.int32, device=self.dev)
        # BatchNorm scale / shift of every unit in ONE buffer ([2][C] per unit): in eval() the pack launch folds the
        # scale half into the forward weights (second descriptor table) and the shift half is the conv's bias
        fold_off, foff = {}, 0
        for u in units:
            if u.bn:
                fold_off[id(u)] = foff
                foff += 2 * u.cout
        self.fold = self._f32(max(foff, 4))
        desc_eval = [list(d) for d in desc]
        for u in units:
            if u.bn and u is not self.u_conv1:       # (the stem applies scale / shift itself: conv1_bn_act)
                desc_eval[u.pk_row][6] = fold_off[id(u)]
        self.pack_desc_eval = torch.tensor(desc_eval, dtype=torch.int32, device=self.dev)
        self._fold_off = fold_off
        self.pack_n, self.pack_max = len(desc), max(d[2] * d[3] * d[4] for d in desc)

        def pview(t):
            o, n = t
            return self.wpack[o:o + n]

        for u in units:
            u.wf = pview(u.pk_f)
            u.wd = pview(u.pk_d) if u.pk_d is not None else None
        for L in self.layers:
            for key in ("qkv", "out", "fc1", "fc2"):
                L[key + "_wf"] = pview(L[key + "_f"])
                L[key + "_wd"] = pview(L[key + "_d"])

        # ---- per-unit buffers -------------------------------------------------------------------------
        max_dy = max_part = max_ws = 0
        for u in units:
            M = B * u.hout * u.wout
            u.M = M
            # the stem is fused (conv1.hip): its raw output -- the largest tensor of the network -- is
            # recomputed from the image instead of stored
            u.raw = self._new((B, u.hout, u.wout, u.cout)) if u is not self.u_conv1 else None
            if u.bn:
                tiles = self.o.conv1_stats_tiles(B, self.H, self.W) if u is self.u_conv1 else self.o.conv_fwd_stats_tiles(self.dt, B, u.hin, u.win, u.cin, u.cout, u.k, u.s)
                u.tiles = tiles
                u.stats = self._f32(self.o.slab_rows(tiles), 2, u.cout)
                fo = self._fold_off[id(u)]
                u.ss = self.fold[fo:fo + 2 * u.cout].view(2, u.cout)
                u.mr, u.coef = self._f32(2, u.cout), self._f32(2, u.cout)
                if u is not self.u_conv1:
                    max_part = max(max_part, self.o.bn_bwd_partial_rows(M) * 2 * u.cout)
            u.pre_rows = 0      # > 0: the data gradient that produced this unit's output gradient left that many rows of BN-backward sums in bn_part
            if u is self.u_conv1:
                self.conv1_part = self._f32(*self.o.conv1_bwd_partial_shape(B, self.H, self.W))
            else:
                # rows of BatchNorm-backward sums this unit's data gradient writes for the unit consuming it (conv_dgrad(bn=...))
                u.dgrad_tiles = self.o.conv_dgrad_bn_tiles(self.dt, B, u.hin, u.win, u.cin, u.cout, u.k, u.s)
                max_part = max(max_part, self.o.slab_rows(u.dgrad_tiles) * 2 * u.cin)
                max_dy = max(max_dy, M * u.cout)
                max_ws = max(max_ws, self.o.conv_wgrad_workspace_bytes(self.dt, B, u.hin, u.win, u.cin, u.cout, u.k, u.s))
        Mt = B * self.N
        for n_out, k_in in ((768, 256), (256, 256)):
            max_ws = max(max_ws, self.o.linear_wgrad_workspace_bytes(self.dt, Mt, n_out, k_in))
        if self.group_wgrad:
            shapes = []
            for L in self.layers:
                at, ff = L["attn"], L["ff"]
                shapes += [tuple(w.shape) for w in (ff.net[4].weight, ff.net[1].weight, at.to_out.weight, at.to_qkv.weight)]
            max_ws = max(max_ws, self.o.linear_wgrad_group_workspace_bytes(self.dt, shapes, Mt))
        # the 3x3 convolutions of a stage's residual blocks (one geometry per stage) share ONE grouped weight-gradient launch at the
        # end of the stage's backward: their raw-output gradients stay alive in buffers of their own until then
        for u in self.units:
            u.dy_own, u.wgrad_x = None, None
        for st in self.stages:
            st["wg_units"], st["wg_items"] = [], None
            if not self.group_wgrad3:
                continue
            us = [rb[k] for br in st["branches"] for rb in br for k in ("c1", "c2")]
            u0 = us[0]
            if len(us) >= 2 and all((u.k, u.s, u.cin, u.cout, u.hin, u.win) == (3, 1, u0.cin, u0.cout, u0.hin, u0.win) for u in us):
                st["wg_units"] = us
                max_ws = max(max_ws, self.o.conv_wgrad_group_workspace_bytes(self.dt, len(us), B, u0.hin, u0.win, u0.cin, u0.cout, 3, 1))
                for u in us:
                    u.dy_own = self._new((B, u.hout, u.wout, u.cout))
        self.dyraw = self._new((max_dy,))
        self.bn_part = self._f32(max_part)
        self.wgrad_ws = self._f32(max_ws // 4 + 4)

        # ---- activations and their gradients ----------------------------------------------------------
        u1, u2 = self.u_conv1, self.u_conv2
        self.a1 = self._new((B, u1.hout, u1.wout, 64))
        self.a2 = self._new((B, u2.hout, u2.wout, 128))
        self.g_a1, self.g_a2 = torch.empty_like(self.a1), torch.empty_like(self.a2)
        for st in self.stages:
            C, h, w = st["C"], st["h"], st["w"]
            st["cat"], st["g_cat"] = self._new((B, h, w, 2 * C)), self._new((B, h, w, 2 * C))
            for br in st["branches"]:
                for i, rb in enumerate(br):
                    rb["t"] = self._new((B, h, w, C // 2))                  # cv1 output of the block
                    if i + 1 < len(br):                                     # output of a non-final block
                        rb["o"], rb["g_o"] = self._new((B, h, w, C // 2)), self._new((B, h, w, C // 2))
            st["g_t"] = self._new((B, h, w, C // 2))
            st["out"], st["g_out"] = self._new((B, h, w, C)), self._new((B, h, w, C))
            if st["down"] is not None:
                d = st["down"]
                st["dn"], st["g_dn"] = self._new((B, d.hout, d.wout, d.cout)), self._new((B, d.hout, d.wout, d.cout))
        # ViT
        D = self.dim
        self.pos = self.model.decoder.pos_embedding.to(self.dev, torch.float32).contiguous()
        self.feat = self.u_proj.raw.view(B * self.P, D)           # proj output == token features
        self.g_feat = self._new((B * self.P, D))
        self.x = [self._new((B, self.N, D)) for _ in range(2 * len(self.layers) + 1)]   # residual stream states
        self.G = self._new((B, self.N, D))                        # gradient of the residual stream
        for L in self.layers:
            L["xn"], L["mr1"] = self._new((Mt, D)), self._f32(Mt, 2)
            L["qkvb"], L["lse"] = self._new((Mt, 3 * D)), self._f32(B * self.heads, self.N)
            L["ao"] = self._new((Mt, D))
            L["xn2"], L["mr2"] = self._new((Mt, D)), self._f32(Mt, 2)
            L["a"], L["hg"] = self._new((Mt, D)), self._new((Mt, D))
        self.t_a, self.t_b = self._new((Mt, D)), self._new((Mt, D))     # backward temporaries
        self.t_qkv = self._new((Mt, 3 * D))
        self._wg_items = None
        if self.group_wgrad:
            # every Linear's output gradient stays alive until the grouped weight-gradient launch: the gradient of the residual
            # stream after each of a layer's two LayerNorm backwards (G1: at x1, G0: at x0), d(fc1 output), d(qkv)
            for L in self.layers:
                L["G1"], L["G0"] = self._new((B, self.N, D)), self._new((B, self.N, D))
                L["t_b"], L["t_qkv"] = self._new((Mt, L["fc1"].out_features)), self._new((Mt, 3 * D))
        self.delta = self._f32(B * self.heads, self.N)
        self.row_part = self._f32(self.o.rowslab_rows(Mt), 2 * D)
        mlp = self.layers[0]["fc1"].out_features
        self.gelu_tiles = self.o.linear_gelu_bwd_tiles(Mt, mlp, D)
        self.gelu_part = self._f32(self.o.slab_rows(self.gelu_tiles), 2, mlp)
        self.leaf_plan = None
        if self.defer_leaf_sums and self.fuse_gelu_bwd:
            # one slab per producer (they stay unreduced until the end of the ViT backward) + the table of folds
            prod, jobs = self.o.rowslab_producer_rows(Mt), []
            for L in self.layers:
                at, ff = L["attn"], L["ff"]
                L["p_ln1"], L["p_ln2"] = self._f32(self.o.rowslab_rows(Mt), 2 * D), self._f32(self.o.rowslab_rows(Mt), 2 * D)
                L["p_b2"] = self._f32(self.o.rowslab_rows(Mt), D)
                L["p_gelu"] = self._f32(self.o.slab_rows(self.gelu_tiles), 2, mlp)
                jobs.append((L["p_ln1"], prod, 2 * D, 2 * D, self._pair(at.norm.weight, at.norm.bias)))
                jobs.append((L["p_ln2"], prod, 2 * D, 2 * D, self._pair(ff.net[0].weight, ff.net[0].bias)))
                jobs.append((L["p_b2"], prod, D, D, self._g(ff.net[4].bias)))
                jobs.append((L["p_gelu"], self.gelu_tiles, mlp, 2 * mlp, self._g(ff.net[1].bias)))      # the sum half of [2][mlp] rows
            self.leaf_plan = self.o.ColsumPlan(jobs)
        # the pose head's kernels take plain f32 tokens (include/mtn.h): split storage converts at that boundary
        self.xl_plain = self._f32(B, self.N, D) if self.x3 else None
        self.G_plain = self._f32(B, self.N, D) if self.x3 else None
        self.attn = self._f32(B, self.heads, self.N, self.N)
        self.logits = self._f32(B, self.C)
        self.hm = self._f32(B, self.J, 4 * self.fh, 4 * self.fw)
        self.cls_xhat, self.cls_mr = self._f32(B, D), self._f32(B, 2)
        self.cls_part = self._f32(self.o.slab_rows(B), 2 * D)
        self.pose_slab = self._f32(self.o.pose_slab_floats(B, self.fh, self.fw, self.J))
        self.dlogits, self.dhm = self._f32(B, self.C), torch.empty_like(self.hm)
        self.loss_part = self._f32(self.o.loss_partial_len(self.hm.numel()))
        self.losses = self._f32(3)

    # ---------------------------------------------------------------- helpers
    @staticmethod
    def _g(p):
        return p._mtn_grad

    def check_current(self):
        """A plan caches views of the module's flat parameter / gradient buffers (conv weights, the pack
        table).  load_state_dict(assign=True), .to() and _flatten() replace those buffers: a plan built
        before would read stale weights and write gradients nobody consumes -- refuse instead."""
        if self.generation != self.model._generation:
            raise RuntimeError("this execution plan was built before the module's parameter storage was replaced "
                               "(load_state_dict(assign=True) / .to() / _flatten()); take a new one with "
                               "model.engine(batch) and rebuild the TrainStep")

    def pack_weights(self, folded=False):
        """Refresh the kernels' copy of the weights from the fp32 masters.  ``folded`` (inference): every Conv
        unit's BatchNorm scale -- its u.ss[0], filled by bn_eval_scale_shift before this call -- is multiplied
        into its forward weights."""
        self.check_current()
        self.o.pack_weights(self.model.flat_params, self.wpack, self.pack_desc_eval if folded else self.pack_desc,
                         self.pack_n, self.pack_max, scale=self.fold if folded else None)
        # what the packed copy holds: folded or not, and from which state of the master weights (in-place loads bump the
        # version; the fused optimizer writes through raw pointers, but a training step always re-packs unfolded first)
        self._packed = (bool(folded), self.model.flat_params._version)

    def fold_batchnorm(self):
        """eval(): scale / shift of every BatchNorm from its running statistics, then the packed weights with the
        scale folded in -- after this the forward has no BatchNorm pass at all (export.py:41-45)."""
        for u in self.units:
            if u.bn:
                bn = u.mod.bn
                self.o.bn_eval_scale_shift(bn.weight, bn.bias, bn.running_mean, bn.running_var, u.ss, bn.eps)
        self.pack_weights(folded=True)

    def _dy(self, u):
        if u.dy_own is not None:        # (a unit whose weight gradient waits for its stage's grouped launch)
            return u.dy_own
        return self.dyraw[:u.M * u.cout].view(self.B, u.hout, u.wout, u.cout)

    def _bn_fwd(self, u, out, res=None):
        """train mode: batch statistics from the conv epilogue's partial sums, then normalise + activation"""
        bn = u.mod.bn
        self.o.bn_finalize(u.stats, u.tiles, u.M, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                        bn.num_batches_tracked, u.ss, u.mr, bn.momentum, bn.eps)
        if u.tag in self.ablate_fwd:
            return
        self.o.bn_act_fwd(u.raw, u.ss, out, _ACT[u.act or res is not None], res)

    def _conv(self, u, x, out, training, res=None):
        """out = act(bn(conv(x)) [+ res])   (model/gelan.py:56, 78-87)"""
        if not training:
            # eval(): BatchNorm folded into the weights (pack_weights(folded=True)) and an epilogue bias: one launch
            self.o.conv_bias_act_fwd(x, u.wf, u.ss[1], out, _ACT[u.act or res is not None], u.k, u.s, res=res)
            return
        self.o.conv_fwd(x, u.wf, u.raw, u.stats, u.k, u.s, stats_tiles=u.tiles)
        self._bn_fwd(u, out, res)

    def _dgrad(self, u, dy, g_in, g_in_res=None, nxt=None):
        """g_in = data gradient of unit ``u`` [+ g_in_res].  ``nxt`` = (unit, its residual input or None) when
        g_in is, after this launch, the complete output gradient of that unit: the launch then also writes
        the partial sums of that unit's BatchNorm backward (self.o.conv_dgrad(bn=...)), which its _conv_bwd
        -- always the next one to use bn_part -- consumes instead of running its own reduce pass."""
        if nxt is None or not self.fuse_bn_bwd or not nxt[0].bn:
            self.o.conv_dgrad(dy, u.wd, g_in, g_in_res, u.k, u.s)
            return
        n, nres = nxt
        assert n.cout == u.cin and n.raw.shape == g_in.shape
        self.o.conv_dgrad(dy, u.wd, g_in, g_in_res, u.k, u.s,
                       bn=(n.raw, nres, n.ss, n.mr, _ACT[n.act or nres is not None], self.bn_part, u.dgrad_tiles))
        n.pre_rows = u.dgrad_tiles

    def _conv_bwd(self, u, g_out, x, g_in, res=None, dres=None, g_in_res=None, dres_accumulate=True, nxt=None):
        """Backward of _conv: BN+act backward -> wgrad -> dgrad (g_in = dgrad [+ g_in_res]).
        ``dres`` receives the gradient of the residual operand, added to its old contents unless
        ``dres_accumulate`` is False.  ``nxt``: see _dgrad."""
        bn, dy = u.mod.bn, self._dy(u)
        rows, u.pre_rows = u.pre_rows, 0
        if u.tag not in self.ablate_bwd:
            self.o.bn_act_bwd(g_out, u.raw, res, u.ss, u.mr, self.bn_part, u.coef, self._g(bn.weight), self._g(bn.bias),
                              dy, _ACT[u.act or res is not None], dres=dres, extra=dres if dres_accumulate else None,
                              partial_rows=rows)
        if u.dy_own is None:
            self.o.conv_wgrad(x, dy, self._g(u.weight), self.wgrad_ws, u.k, u.s)
        else:
            u.wgrad_x = x               # its stage's grouped launch (_stage_weight_grads) reads x and dy_own
        if g_in is not None:
            self._dgrad(u, dy, g_in, g_in_res, nxt)

    # ---------------------------------------------------------------- forward
    def features_plain(self, feats):
        """A feature / gradient tensor of this plan as plain numbers: split storage (compute_dtype "bf16x3") is joined into a
        new fp32 tensor, the other modes return the tensor itself."""
        return self.o.x3_join(feats) if self.x3 else feats

    def set_plain(self, dst, src):
        """dst (a buffer of this plan) = src given as plain numbers of dst's shape (split storage: split on the way in)."""
        if self.x3:
            self.o.x3_split(src.to(torch.float32).contiguous(), out=dst)
        else:
            dst.copy_(src)
        return dst

    def backbone_forward(self, x, training=True):
        """GELANNet.forward (model/gelan.py:165-176) on an NCHW fp32 batch; returns NHWC [B,h,w,512] in the plan's storage
        (features_plain() gives plain numbers).  The packed weights must be current: pack_weights() before a training pass,
        fold_batchnorm() before an inference pass (forward() does either)."""
        u1, bn1 = self.u_conv1, self.u_conv1.mod.bn
        self.x_in = x
        self.serial[0] += 1
        if training:
            self.o.conv1_stats(x, u1.wf, u1.stats)
            self.o.bn_finalize(u1.stats, u1.tiles, u1.M, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var,
                            bn1.num_batches_tracked, u1.ss, u1.mr, bn1.momentum, bn1.eps)
        self.o.conv1_bn_act(x, u1.wf, u1.ss, self.a1, _ACT[u1.act])
        self._conv(self.u_conv2, self.a1, self.a2, training)
        cur = self.a2
        for st in self.stages:
            C, cat = st["C"], st["cat"]
            h2 = C // 2
            st["in"] = cur
            self._conv(st["cv1"], cur, cat[..., :C], training)                           # y0 | y1
            for j, br in enumerate(st["branches"]):                                      # y2 = cv2(y1), y3 = cv3(y2)
                xin = cat[..., h2 * (j + 1):h2 * (j + 2)]
                for i, rb in enumerate(br):
                    out = cat[..., h2 * (j + 2):h2 * (j + 3)] if i + 1 == len(br) else rb["o"]
                    self._conv(rb["c1"], xin, rb["t"], training)
                    self._conv(rb["c2"], rb["t"], out, training, res=xin)
                    xin = out
            self._conv(st["cv4"], cat, st["out"], training)
            cur = st["out"]
            if st["down"] is not None:
                self._conv(st["down"], cur, st["dn"], training)
                cur = st["dn"]
        return cur

    def forward(self, x, training=True, want_attn=True, refold=True):
        """MultiTaskNet.forward (model/multitasknet.py:24-29) -> (logits, heatmaps, attnmap).  ``refold=False``
        (eval only): the folded weights of the previous eval forward are still current (infer.InferenceSession)."""
        assert tuple(x.shape) == (self.B, 3, self.H, self.W) and x.dtype == torch.float32 and x.is_contiguous()
        if training:
            self.pack_weights()
        elif refold or getattr(self, "_packed", None) != (True, self.model.flat_params._version):
            # (refold=False is a promise that the folded weights of the previous eval forward are still current; a training
            #  forward at this batch size or an in-place weight load since then broke it: fold again instead of serving a
            #  hybrid of stale folded convolutions and live LayerNorm / head parameters)
            self.fold_batchnorm()
        feats = self.backbone_forward(x, training)
        up = self.u_proj
        if self.fuse_tokens:
            # multitasknet.py:26 + transformer.py:132-139 in one launch: the projection's epilogue adds the position embedding,
            # writes token-major rows and the class token; the projected features are never stored
            self.o.proj_tokens_fwd(feats, up.wf, self.pos, self.model.decoder.cls_token, self.x[0])
            return self.vit_forward(want_attn, tokens_ready=True)
        self.o.conv_fwd(feats, up.wf, up.raw, None, 1, 1)                                   # multitasknet.py:26
        return self.vit_forward(want_attn)

    def vit_forward(self, want_attn=True, tokens_ready=False):
        """ViT.forward (model/transformer.py:129-152) on the projected features in ``self.feat`` ([B*P, 256], the plan's
        storage): -> (logits, heatmaps, attnmap).  The packed weights must be current (pack_weights / fold_batchnorm).
        ``tokens_ready``: forward() already assembled the tokens in the projection's epilogue."""
        dec = self.model.decoder
        self.serial[1] += 1
        if not tokens_ready:
            self.o.tokens_fwd(self.feat, self.pos, dec.cls_token, self.x[0])                # transformer.py:132-139
        B, N, D, Hh = self.B, self.N, self.dim, self.heads
        Mt = B * N
        for i, L in enumerate(self.layers):
            x0, x1, x2 = self.x[2 * i].view(Mt, D), self.x[2 * i + 1].view(Mt, D), self.x[2 * i + 2].view(Mt, D)
            at, ff = L["attn"], L["ff"]
            self.o.layernorm_fwd(x0, at.norm.weight, at.norm.bias, L["xn"], L["mr1"], at.norm.eps)
            self.o.linear_fwd(L["xn"], L["qkv_wf"], L["qkvb"])
            last = i == len(self.layers) - 1
            self.o.attention_fwd(L["qkvb"], L["ao"], L["lse"], self.attn if (last and want_attn) else None,
                              B, N, Hh, self.hd, at.scale)
            self.o.linear_fwd(L["ao"], L["out_wf"], x1, res=x0)                             # x = attn(x) + x
            ln = ff.net[0]
            self.o.layernorm_fwd(x1, ln.weight, ln.bias, L["xn2"], L["mr2"], ln.eps)
            self.o.linear_fwd(L["xn2"], L["fc1_wf"], L["hg"], bias=ff.net[1].bias, act=self.o.ACT_GELU, aux_out=L["a"])
            self.o.linear_fwd(L["hg"], L["fc2_wf"], x2, bias=ff.net[4].bias, res=x1)       # x = ff(x) + x
        xl = self.x[-1]
        head = dec.mlp_head
        self.o.cls_head_fwd(xl, head[0].weight, head[0].bias, head[1].weight, head[1].bias, self.logits,
                         self.cls_xhat, self.cls_mr, head[0].eps)
        pw = dec.simple_decoder[1]
        if self.x3:
            self.o.x3_join(xl, out=self.xl_plain)
        self.o.pose_head_fwd(self.xl_plain if self.x3 else xl, pw.weight, pw.bias, self.hm, self.fh, self.fw)
        return self.logits, self.hm, (self.attn if want_attn else None)

    # ---------------------------------------------------------------- loss
    def loss(self, label, target, target_weight, with_grads=True):
        """total = 0.001*CE + JointsMSE (train.py:63-64,75); fills dlogits/dhm when with_grads.
        The batch tensors follow libs/load.py (label int64 [B], target f32 [B,J,H/4,W/4],
        target_weight f32 [B,J,1]); float tensors of another float dtype are converted, anything
        else (integer labels that are not int64, wrong shapes) raises in self.o.loss_fwd_bwd."""
        if target.dtype != torch.float32 and target.is_floating_point():
            target = target.float()
        if target_weight.dtype != torch.float32 and target_weight.is_floating_point():
            target_weight = target_weight.float()
        self.o.loss_fwd_bwd(self.logits, label.contiguous(), self.hm, target.contiguous(), target_weight.contiguous(),
                         self.dlogits if with_grads else None, self.dhm if with_grads else None,
                         self.loss_part, self.losses)
        return self.losses

    # ---------------------------------------------------------------- backward
    def backbone_backward(self, g_feats):
        """Backward of backbone_forward given the gradient of its NHWC output (written into g buffers)."""
        for st in self.stages:          # (entered directly: no hand-over can be pending for the backbone's units except the
            for u in (st["cv1"], st["down"]):                          # last cv4, which the projection's data gradient feeds)
                if u is not None:
                    u.pre_rows = 0
        g_cur = g_feats
        for st in reversed(self.stages):
            C, cat, gc = st["C"], st["cat"], st["g_cat"]
            h2 = C // 2
            if st["down"] is not None:
                self._conv_bwd(st["down"], g_cur, st["out"], st["g_out"], nxt=(st["cv4"], None))
                g_cur = st["g_out"]
            self._conv_bwd(st["cv4"], g_cur, cat, gc)
            # branches in reverse (cv3: y3 = blocks(y2), then cv2: y2 = blocks(y1)), blocks in reverse.
            # The gradient of a block's input gets the residual path (dres) and the c1 data gradient;
            # for the first block of a branch that input is a slice of the concat, whose gradient
            # already holds cv4's contribution (accumulate); a later block's input is private (overwrite).
            for j in (1, 0):
                br = st["branches"][j]
                for i in reversed(range(len(br))):
                    rb = br[i]
                    g_out = gc[..., h2 * (j + 2):h2 * (j + 3)] if i + 1 == len(br) else rb["g_o"]
                    first = i == 0
                    xin = cat[..., h2 * (j + 1):h2 * (j + 2)] if first else br[i - 1]["o"]
                    g_xin = gc[..., h2 * (j + 1):h2 * (j + 2)] if first else br[i - 1]["g_o"]
                    self._conv_bwd(rb["c2"], g_out, rb["t"], st["g_t"], res=xin, dres=g_xin, dres_accumulate=first,
                                   nxt=(rb["c1"], None))
                    # g_xin is complete after c1's data gradient, except the y1 slice of the concat (branch 0, first
                    # block), which is only half of cv1's output gradient.  Its consumer: the previous block's c2,
                    # or -- for the y2 slice (branch 1, first block) -- the last block of branch 0
                    if not first:
                        prev = br[i - 1]
                        pxin = cat[..., h2 * (j + 1):h2 * (j + 2)] if i == 1 else br[i - 2]["o"]
                        nxt = (prev["c2"], pxin)
                    elif j == 1:
                        b0 = st["branches"][0]
                        nxt = (b0[-1]["c2"], cat[..., h2:2 * h2] if len(b0) == 1 else b0[-2]["o"])
                    else:
                        nxt = None
                    self._conv_bwd(rb["c1"], st["g_t"], xin, g_xin, g_in_res=g_xin, nxt=nxt)
            si = self.stages.index(st)
            g_in = self.g_a2 if si == 0 else self.stages[si - 1]["g_dn"]
            self._conv_bwd(st["cv1"], gc[..., :C], st["in"], g_in,
                           nxt=(self.u_conv2 if si == 0 else self.stages[si - 1]["down"], None))
            g_cur = g_in
            self._stage_weight_grads(st)
            self._ready(("stage", self.stages.index(st)))
        self._conv_bwd(self.u_conv2, self.g_a2, self.a1, self.g_a1)
        u1, bn = self.u_conv1, self.u_conv1.mod.bn
        self.o.conv1_bwd(self.x_in, u1.wf, self.g_a1, u1.ss, u1.mr, self.conv1_part, self._g(bn.weight), self._g(bn.bias),
                      self._g(u1.weight), _ACT[u1.act])
        if getattr(self, "_dx", None) is not None:
            self.o.conv1_dx(self.x_in, u1.weight.detach(), self.g_a1, u1.ss, u1.mr, self._g(bn.weight), self._g(bn.bias), self._dx,
                         _ACT[u1.act])
            self._dx = None
        self._ready("stem")

    def backward(self, dlogits=None, dhm=None, dx=None):
        """Backward of forward() from gradients of (logits, heatmaps); all parameter grads written.  ``dx`` (fp32, shaped like
        the image): also the gradient of the image -- the reference module is plain autograd (multitasknet.py:24-29); the
        training step never asks for it."""
        self._dx = dx
        self.check_current()
        dlogits = self.dlogits if dlogits is None else dlogits
        dhm = self.dhm if dhm is None else dhm
        for u in self.units:
            u.pre_rows = 0          # (a backward that raised half-way must not leave a stale hand-over)
        self.vit_backward(dlogits, dhm)
        up, last = self.u_proj, self.stages[-1]
        gf = self.g_feat.view(self.B, self.fh, self.fw, self.dim)
        self.o.conv_wgrad(last["out"], gf, self._g(up.weight), self.wgrad_ws, 1, 1)
        self._dgrad(up, gf, last["g_out"], None, nxt=(last["cv4"], None))
        self._ready("head")
        self.backbone_backward(last["g_out"])

    def vit_backward(self, dlogits, dhm):
        """Backward of vit_forward: the decoder's parameter gradients, and the gradient of the projected features in
        ``self.g_feat`` ([B*P, 256], the plan's storage)."""
        dec, g = self.model.decoder, self._g
        B, N, D, Hh = self.B, self.N, self.dim, self.heads
        Mt = B * N
        G, Gm = self.G, self.G.view(Mt, D)
        head, pw = dec.mlp_head, dec.simple_decoder[1]
        xl = self.x[-1]
        # (the pose head writes the patch-token rows of G, the classification head the class-token rows: split storage takes
        #  the former through a plain copy that is split into G before the latter lands)
        if self.x3:
            self.o.pose_head_bwd(self.xl_plain, pw.weight, dhm, self.G_plain, self._pair(pw.weight, pw.bias), self.pose_slab, self.fh, self.fw)
            self.o.x3_split(self.G_plain, out=G)
        else:
            self.o.pose_head_bwd(xl, pw.weight, dhm, G, self._pair(pw.weight, pw.bias), self.pose_slab, self.fh, self.fw)
        self.o.cls_head_bwd(dlogits, head[1].weight, head[0].weight, head[0].bias, self.cls_xhat, self.cls_mr, G,
                            g(head[1].weight), g(head[1].bias), self._pair(head[0].weight, head[0].bias), self.cls_part)
        grp = self.group_wgrad
        for i in reversed(range(len(self.layers))):
            L = self.layers[i]
            x0, x1 = self.x[2 * i].view(Mt, D), self.x[2 * i + 1].view(Mt, D)
            at, ff = L["attn"], L["ff"]
            fc1, fc2, ln2 = ff.net[1], ff.net[4], ff.net[0]
            # grouped weight gradients: every output gradient into its own buffer (G1 / G0: the stream's gradient at x1 / x0)
            t_b, t_qkv = (L["t_b"], L["t_qkv"]) if grp else (self.t_b, self.t_qkv)
            G1, G0 = (L["G1"].view(Mt, D), L["G0"].view(Mt, D)) if grp else (Gm, Gm)
            # x2 = fc2(gelu(fc1(LN(x1)))) + x1
            defer = self.leaf_plan is not None
            if defer:
                self.o.bias_grad(Gm, L["p_b2"], None)
            else:
                self.o.bias_grad(Gm, self.row_part, g(fc2.bias))
            if not grp:
                self.o.linear_wgrad(L["hg"], Gm, g(fc2.weight), self.wgrad_ws)
            if self.fuse_gelu_bwd:
                # d a = (G @ fc2.W) * gelu'(a) and fc1's bias gradient (its column sums) in one launch
                self.o.linear_gelu_bwd(Gm, L["fc2_wd"], L["a"], t_b, L["p_gelu"] if defer else self.gelu_part, self.gelu_tiles,
                                    None if defer else g(fc1.bias))
            else:
                self.o.linear_fwd(Gm, L["fc2_wd"], self.t_a)                   # d hgelu
                self.o.gelu_bwd(self.t_a, L["a"], t_b)                         # d a
                self.o.bias_grad(t_b, self.row_part, g(fc1.bias))
            if not grp:
                self.o.linear_wgrad(L["xn2"], t_b, g(fc1.weight), self.wgrad_ws)
            self.o.linear_fwd(t_b, L["fc1_wd"], self.t_a)                      # d xn2
            self.o.layernorm_bwd(self.t_a, x1, L["mr2"], ln2.weight, G1, L["p_ln2"] if defer else self.row_part,
                              None if defer else self._pair(ln2.weight, ln2.bias), dres=Gm)
            # x1 = to_out(attention(LN(x0))) + x0
            if not grp:
                self.o.linear_wgrad(L["ao"], G1, g(at.to_out.weight), self.wgrad_ws)
            self.o.linear_fwd(G1, L["out_wd"], self.t_a)                       # d ao
            self.o.attention_bwd(L["qkvb"], L["ao"], self.t_a, L["lse"], self.delta, t_qkv, B, N, Hh, self.hd, at.scale)
            if not grp:
                self.o.linear_wgrad(L["xn"], t_qkv, g(at.to_qkv.weight), self.wgrad_ws)
            self.o.linear_fwd(t_qkv, L["qkv_wd"], self.t_a)                    # d xn
            self.o.layernorm_bwd(self.t_a, x0, L["mr1"], at.norm.weight, G0, L["p_ln1"] if defer else self.row_part,
                              None if defer else self._pair(at.norm.weight, at.norm.bias), dres=G1)
            Gm = G0
        self.o.tokens_bwd(Gm.view(B, N, D), self.g_feat, g(dec.cls_token))
        if grp:
            self._vit_weight_grads()
        if self.leaf_plan is not None:
            self.leaf_plan.run()                                        # all LayerNorm / bias gradients of the ViT: two launches

    def _stage_weight_grads(self, st):
        """`.weight.grad` of the 3x3 convolutions of a stage's residual blocks (model/gelan.py:69-70) in one grouped launch over the
        inputs and raw-output gradients their _conv_bwd calls left in place."""
        us = st["wg_units"]
        if not us:
            return
        key = self._g(us[0].weight).data_ptr()
        if st["wg_items"] is None or st["wg_items"][0] != key:
            st["wg_items"] = (key, self.o.conv_wgrad_group_items([(u.wgrad_x, u.dy_own, self._g(u.weight)) for u in us]))
        u = us[0]
        self.o.conv_wgrad_group(st["wg_items"][1], self.wgrad_ws, self.dt, self.B, u.hin, u.win, u.cin, u.cout, 3, 1)

    def _vit_weight_grads(self):
        """to_qkv / to_out / fc1 / fc2 `.weight.grad` of every Transformer layer (model/transformer.py:44,58-60) in one grouped launch
        over the forward inputs and the per-layer output gradients vit_backward left in place."""
        key = self._g(self.layers[0]["attn"].to_qkv.weight).data_ptr()
        if self._wg_items is None or self._wg_items[0] != key:
            Mt, D, probs = self.B * self.N, self.dim, []
            for i, L in enumerate(self.layers):
                at, ff = L["attn"], L["ff"]
                top = self.G if i + 1 == len(self.layers) else self.layers[i + 1]["G0"]     # the stream's gradient at this layer's output
                probs += [(L["hg"], top.view(Mt, D), self._g(ff.net[4].weight)), (L["xn2"], L["t_b"], self._g(ff.net[1].weight)),
                          (L["ao"], L["G1"].view(Mt, D), self._g(at.to_out.weight)), (L["xn"], L["t_qkv"], self._g(at.to_qkv.weight))]
            items, M = self.o.wgrad_group_items(probs)
            assert 0 < self.o.linear_wgrad_group_workspace_bytes(self.dt, items, M) <= self.wgrad_ws.numel() * 4
            self._wg_items = (key, items, M)
        _, items, M = self._wg_items
        self.o.linear_wgrad_group(items, M, self.wgrad_ws, self.dt)

    def _ready(self, tag):
        if self.on_grads_ready is not None:
            self.on_grads_ready(tag)

    def _pair(self, w, b):
        """Gradient slot of a (weight, bias) pair that one kernel writes as one block: the two are
        adjacent in the flat gradient buffer because they are consecutive in parameters() order."""
        gw, gb = self._g(w), self._g(b)
        assert gb.data_ptr() == gw.data_ptr() + gw.numel() * 4, "weight/bias grads must be adjacent"
        return gw
