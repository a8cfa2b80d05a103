"""infer.InferenceSession: the eval() forward at a fixed batch size with BatchNorm folded once, optionally replayed from
one hipGraph (the reference's frame-at-a-time deployment use: detect.py:60-110, export.py:41-45)."""
import pytest
import torch

import mtn_oracle as O
from backend import Backend, rel_err
from hand_gesture_recognition_amd.infer import InferenceSession
from hand_gesture_recognition_amd.multitasknet import MultiTaskNet


def _model(be, dtype, size=32, seed=5):
    sd = O.synthetic_state_dict(21, 19, seed=seed)
    model = MultiTaskNet(21, 19, [size, size], compute_dtype=dtype)
    model.load_state_dict(sd, strict=True)
    return model.to(be.device), sd


def _case(be, dtype, use_graph, tol):
    model, sd = _model(be, dtype)
    g = torch.Generator().manual_seed(3)
    frames = [torch.randn(2, 3, 32, 32, generator=g) for _ in range(3)]
    sess = InferenceSession(model, 2, use_graph=use_graph)
    for x in frames:                                       # (with use_graph: capture on the first call, replays after)
        out = [t.clone() for t in sess(x.to(be.device))]
        with torch.no_grad():
            ref = model(x.to(be.device))
        for a, b in zip(out, ref):
            assert torch.equal(a, b)                        # the same kernels on the same folded weights
    orc = O.multitasknet_forward(frames[-1], {k: v.clone() for k, v in sd.items()}, training=False)
    for a, b in zip(out, orc):
        assert rel_err(a, b) < tol
    # new weights: the session keeps the OLD folded copy until refresh()
    sd2 = O.synthetic_state_dict(21, 19, seed=9)
    model.load_state_dict(sd2, strict=True)
    stale = [t.clone() for t in sess(frames[0].to(be.device))]
    sess.refresh()
    fresh = [t.clone() for t in sess(frames[0].to(be.device))]
    orc2 = O.multitasknet_forward(frames[0], {k: v.clone() for k, v in sd2.items()}, training=False)
    assert rel_err(fresh[0], orc2[0]) < tol and rel_err(fresh[1], orc2[1]) < tol
    assert rel_err(stale[0], orc2[0]) > 10 * tol
    with pytest.raises(ValueError):
        sess(torch.zeros(3, 3, 32, 32, device=be.device))
    model.train()
    with pytest.raises(RuntimeError, match="eval"):
        sess(frames[0].to(be.device))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), ("bf16x3", 1e-3)], ids=["f32", "x3"])
def test_inference_session_eager_emu(dtype, tol):
    _case(Backend("emu"), dtype, False, tol)


def _train_step_between_eval_forwards(be, use_graph):
    """eval forward, then a training step on the SAME plan (it re-packs unfolded weights and moves them
    through raw pointers: no version bump; under ``use_graph`` without running any Python), then an eval forward that
    promises ``refold=False``: the plan must notice and fold again instead of serving unfolded weights with folded biases."""
    from hand_gesture_recognition_amd.train import TrainStep
    model, _ = _model(be, torch.float32)
    batch = [t.to(be.device) for t in O.synthetic_batch(2, 32, seed=31)]
    eng = model.engine(2)
    step = TrainStep(model, 2, lr=1e-2, use_graph=use_graph)
    for _ in range(2):                                     # (graph: capture on the first call, a pure replay on the second)
        model.eval()
        with torch.no_grad():
            eng.forward(batch[0], training=False)
        model.train()
        step(*batch)
        model.eval()
        with torch.no_grad():
            got = [t.clone() for t in eng.forward(batch[0], training=False, refold=False)]
            ref = [t.clone() for t in eng.forward(batch[0], training=False, refold=True)]
        for a, b in zip(got, ref):
            assert torch.equal(a, b)


def test_eval_after_train_step_refolds_emu():
    _train_step_between_eval_forwards(Backend("emu"), False)


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_eval_after_train_step_refolds_gpu(use_graph):
    _train_step_between_eval_forwards(Backend("hip"), use_graph)


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), ("bf16x3", 1e-3), (torch.bfloat16, 3e-2)], ids=["f32", "x3", "bf16"])
def test_inference_session_gpu(dtype, tol, use_graph):
    _case(Backend("hip"), dtype, use_graph, tol)


@pytest.mark.gpu
def test_inference_session_refuses_a_replaced_parameter_storage():
    be = Backend("hip")
    model, sd = _model(be, torch.bfloat16)
    sess = InferenceSession(model, 1)
    sess(torch.zeros(1, 3, 32, 32, device=be.device))
    model.load_state_dict({k: v.to(be.device) for k, v in sd.items()}, strict=True, assign=True)
    with pytest.raises(RuntimeError, match="parameter storage"):
        sess(torch.zeros(1, 3, 32, 32, device=be.device))
