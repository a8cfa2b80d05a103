"""Backend selection for kernel tests.

``hip``: the product library libmtn_hip.so on cuda:0 (tests carry @pytest.mark.gpu).
``emu``: the same kernel sources compiled for the CPU fiber executor (tests/emu/hipemu.h),
         used by the ``not gpu`` suite to check kernel logic at tiny sizes.
"""
import os
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hand-gesture-recognition_amd", "csrc")
# MTN_EMU_LIB: another build of the CPU executor library, e.g. one compiled with -fsanitize=address,undefined
# (tools/asan_emu.sh); it is used as it is, not rebuilt
EMU_LIB = os.environ.get("MTN_EMU_LIB") or os.path.join(ROOT, "hand-gesture-recognition_amd", "libmtn_emu.so")

_built = {}


def build_emu():
    """make emu under an exclusive file lock: the xdist controller calls this once before it starts its workers
    (tests/conftest.py pytest_configure), so that six workers never run `make` into the same build/*.o at once; a
    worker that still gets here (a run without the controller hook) queues on the lock and finds the build current."""
    import fcntl
    if os.environ.get("MTN_EMU_LIB"):
        return
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    with open(os.path.join(CSRC, "build", ".emu.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            subprocess.run(["make", "-C", CSRC, "-j4", "emu"], check=True, stdout=subprocess.DEVNULL)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def _ensure_emu():
    if "emu" not in _built:
        if os.environ.get("MTN_EMU_BUILT") != "1":      # (set by the controller after its build: workers do not rebuild)
            build_emu()
        _built["emu"] = True


class Backend:
    def __init__(self, name):
        from hand_gesture_recognition_amd import lib
        self.name = name
        if name == "emu":
            _ensure_emu()
            lib.use_library(EMU_LIB)
            self.device = torch.device("cpu")
        else:
            lib.use_library(lib.HIP_LIB)
            assert torch.cuda.is_available(), "hip backend needs a GPU"
            self.device = torch.device("cuda:0")

    def sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize()


BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
DTYPES = [pytest.param(torch.float32, id="f32"), pytest.param(torch.bfloat16, id="bf16")]
# GEMM-shaped kernels also run fp32 tensors in split-bf16 mode (MTN_DTYPE_F32X3): the `x3` marker makes
# conftest's autouse fixture select it in ops for the duration of the test
GEMM_DTYPES = DTYPES + [pytest.param(torch.float32, id="x3", marks=pytest.mark.x3)]


@pytest.fixture(params=BACKENDS)
def be(request):
    return Backend(request.param)


def split_x3(t):
    """Python restatement of split storage (include/mtn.h MTN_DTYPE_F32X3; csrc/mtn_types.h): a plain fp32 tensor [..., C]
    (C % 32 == 0) -> the same shape and dtype holding, per 32-channel block, [32 bf16 hi][32 bf16 lo] in natural channel
    order, hi = bf16_rne(x), lo = bf16_rne(x - hi).  Activations, gradients and packed weights share this layout."""
    t = t.detach().contiguous().float()
    C = t.shape[-1]
    assert C % 32 == 0, C
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    out = torch.cat([hi.reshape(-1, C // 32, 32), lo.reshape(-1, C // 32, 32)], -1).contiguous()   # [rows, blocks, 64] bf16
    return out.view(torch.float32).reshape(t.shape)


def join_x3(t):
    """split storage -> plain fp32 (x = hi + lo)"""
    t = t.detach().contiguous()
    C = t.shape[-1]
    assert t.dtype == torch.float32 and C % 32 == 0, (t.dtype, C)
    b = t.reshape(-1, C // 32, 32).view(torch.bfloat16).reshape(-1, C // 32, 2, 32).float()
    return (b[:, :, 0] + b[:, :, 1]).reshape(t.shape)


presplit_x3 = split_x3          # (packed weights: the same layout since round 4)


def A(t):
    """a plain reference tensor in the storage format the kernels under test expect: split storage under the x3
    parametrisation (fp32 tensors), unchanged otherwise.  Slices at multiples of 32 channels stay valid."""
    return split_x3(t) if (t.dtype == torch.float32 and x3_active()) else t


def Fl(t):
    """a kernel output as plain fp32 numbers on the host"""
    t = t.detach().cpu()
    return join_x3(t) if (t.dtype == torch.float32 and x3_active()) else t.float()


# set by conftest's autouse fixture for the duration of one test: [True] once the test selected an experiment-only variant
_saw_experiment = [False]


def dbg(key, value):
    """mtn_debug_set(key, value); a value that selects a measured-and-rejected kernel variant exists only in -DMTN_EXPERIMENTS
    builds (the CPU executor library).  Parametrisations that do so carry the ``experiments`` marker (tests/experiment_ids.txt,
    kept in step by conftest.py) and are DESELECTED when the library under test has no experiments, so a product-library
    run shows no skips for them; one that is not on the list yet still skips here -- visibly -- rather than fail."""
    from hand_gesture_recognition_amd import lib
    if lib.call("mtn_debug_key_is_experiment", key, value):
        _saw_experiment[0] = True
    try:
        lib.call("mtn_debug_set", key, value)
    except RuntimeError as e:
        if "does not contain" in str(e):
            pytest.skip(f"debug key {key} = {value}: variant not in the product library (-DMTN_EXPERIMENTS builds only)")
        raise


def x3_active():
    from hand_gesture_recognition_amd import ops
    return ops._f32_code == ops.F32X3


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def tol(dtype, f32=1e-4, bf16=2e-2):
    if dtype == torch.float32:           # (split storage keeps ~17 bits of every stored element: 2^-18 rounding per tensor)
        return max(f32, 2e-5) if x3_active() else f32
    return bf16
