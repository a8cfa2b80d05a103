import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the kernel tests drive every tile / pipeline variant through mtn_debug_set; the library refuses those keys
# unless the process opted in before its first call (product processes never do)
os.environ["MTN_ALLOW_DEBUG_KEYS"] = "1"
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_xdist_auto_num_workers(config):
    """`-n auto` (pytest.ini): worker processes only for the CPU suite, i.e. when the marker expression excludes
    the gpu tests; any run that may touch the card stays in ONE process."""
    expr = (config.getoption("markexpr", "") or "").replace(" ", "")
    if "notgpu" not in expr:
        return 0
    return max(1, min(6, (os.cpu_count() or 2) - 2))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "x3: fp32 tensors multiplied in split-bf16 mode (MTN_DTYPE_F32X3)")
    # the CPU kernel-executor library is built ONCE, here in the controller, before any xdist worker exists
    # (workers inherit MTN_EMU_BUILT and skip their own `make`: no concurrent builds into one object directory)
    if not hasattr(config, "workerinput") and os.environ.get("MTN_EMU_BUILT") != "1":
        import backend
        backend.build_emu()
        os.environ["MTN_EMU_BUILT"] = "1"


@pytest.fixture(autouse=True)
def _f32_matmul_mode(request):
    """ops' fp32 matmul mode is process state (it cannot be read off an fp32 tensor): exact-f32 MFMA for
    every test except those marked x3, and never leaked from one test into the next."""
    from hand_gesture_recognition_amd import ops
    ops.set_f32_matmul(ops.F32X3 if request.node.get_closest_marker("x3") is not None else ops.F32)
    yield
    ops.set_f32_matmul(ops.F32)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
