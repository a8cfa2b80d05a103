"""Frame-at-a-time inference on the hot path's kernels (the reference's deployment loop, detect.py:60-110 /
export.py:41-45, runs the module in eval() on one image per call).

``InferenceSession(model, batch_size)`` fixes the batch size, folds every BatchNorm into the packed convolution weights
ONCE (``refresh()`` folds again after the weights changed) and, with ``use_graph=True``, captures the ~55 launches of
the eval forward into one hipGraph that every call replays: at B=1 the eager forward is launch-bound (0.8 ms for
75 launches, 22 of them the per-call fold), the replay is not.

    sess = InferenceSession(model, batch_size=1)
    cls_out, hmap_out, attnmap = sess(frame)      # views of the session's buffers, valid until the next call
"""
import torch

__all__ = ["InferenceSession"]


class InferenceSession:
    def __init__(self, model, batch_size=1, use_graph=True, want_attn=True):
        if model.flat_params.device.type != "cuda" and use_graph:
            raise RuntimeError("InferenceSession(use_graph=True) needs the module on an MI355X ('cuda')")
        model.eval()
        self.model, self.B = model, int(batch_size)
        self.want_attn, self.use_graph = bool(want_attn), bool(use_graph)
        self.eng = model.engine(self.B)
        H, W = model.image_size
        self.x = torch.zeros(self.B, 3, H, W, device=model.flat_params.device, dtype=torch.float32)
        self._graph = None
        self.refresh()

    def refresh(self):
        """Fold the BatchNorms and pack the weights again (after load_state_dict / an optimizer step)."""
        self.eng.check_current()
        with torch.no_grad():
            self.eng.fold_batchnorm()

    def _run(self):
        return self.eng.forward(self.x, training=False, want_attn=self.want_attn, refold=False)

    def _capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                 # one eager pass: lazy kernel attributes, allocator
            self._run()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._run()
        self._graph = g

    @torch.no_grad()
    def __call__(self, x):
        if self.model.training:
            raise RuntimeError("InferenceSession: the module was put back into train(); call model.eval()")
        self.eng.check_current()
        if tuple(x.shape) != tuple(self.x.shape):
            raise ValueError(f"expected {tuple(self.x.shape)}, got {tuple(x.shape)}")
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        if not self.use_graph:
            return self._run()
        if self._graph is None:
            self._capture()
        self._graph.replay()
        e = self.eng
        return e.logits, e.hm, (e.attn if self.want_attn else None)
