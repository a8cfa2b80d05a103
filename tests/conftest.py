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
    config.addinivalue_line("markers", "experiments: the parametrisation selects a measured-and-rejected kernel variant that only "
                                       "-DMTN_EXPERIMENTS builds contain (deselected on the product library)")
    # the CPU kernel-executor library is built ONCE, here in the controller, before any xdist worker exists
    # (workers inherit MTN_EMU_BUILT and skip their own `make`: no concurrent builds into one object directory)
    if not hasattr(config, "workerinput") and os.environ.get("MTN_EMU_BUILT") != "1":
        import backend
        backend.build_emu()
        os.environ["MTN_EMU_BUILT"] = "1"


# ---- experiment-only parametrisations --------------------------------------------------------------------------------
# tests/experiment_ids.txt lists the test ids (backend token replaced by '*') whose body selects a measured-and-rejected
# kernel variant through backend.dbg(): they get the `experiments` marker at collection and, when the library their backend
# loads was built without -DMTN_EXPERIMENTS (the product libmtn_hip.so), are DESELECTED -- `pytest -m gpu` on the product
# library then reports no skips unless something is really missing.  The list is checked on
# every run: a test whose marker and behaviour disagree fails with the command that regenerates the file
# (MTN_RECORD_EXPERIMENTS=tests/experiment_ids.txt python -m pytest tests -m "not gpu" -n 0 -q).
_EXP_FILE = os.path.join(ROOT, "tests", "experiment_ids.txt")


def _exp_key(nodeid):
    import re
    nodeid = nodeid.replace("\\", "/")
    nodeid = nodeid[nodeid.index("tests/"):] if "tests/" in nodeid else nodeid
    return re.sub(r"(?<=[\[-])(emu|hip)(?=[\]-])", "*", nodeid, count=1)


def _exp_ids():
    if not os.path.exists(_EXP_FILE):
        return set()
    with open(_EXP_FILE) as fh:
        return {ln.strip() for ln in fh if ln.strip() and not ln.startswith("#")}


def _hip_has_experiments():
    import ctypes
    from hand_gesture_recognition_amd import lib
    try:
        dll = ctypes.CDLL(lib.HIP_LIB)
        return bool(dll.mtn_has_experiments())
    except (OSError, AttributeError):
        return True          # (no library / an old one: keep everything, the tests themselves will say what is wrong)


def pytest_collection_modifyitems(config, items):
    ids = _exp_ids()
    recording = bool(os.environ.get("MTN_RECORD_EXPERIMENTS"))
    hip_ex = None
    keep, drop = [], []
    for it in items:
        marked = _exp_key(it.nodeid) in ids or it.get_closest_marker("experiments") is not None
        if marked:
            it.add_marker(pytest.mark.experiments)
        if marked and not recording and it.get_closest_marker("gpu") is not None:
            if hip_ex is None:
                hip_ex = _hip_has_experiments()
            if not hip_ex:
                drop.append(it)
                continue
        keep.append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(autouse=True)
def _experiment_marker_in_step(request):
    import backend
    backend._saw_experiment[0] = False
    yield
    saw, marked = backend._saw_experiment[0], request.node.get_closest_marker("experiments") is not None
    rec = os.environ.get("MTN_RECORD_EXPERIMENTS")
    if rec:
        if saw:
            with open(rec + ".new", "a") as fh:
                fh.write(_exp_key(request.node.nodeid) + "\n")
        return
    rep = getattr(request.node, "_mtn_call_outcome", None)
    if rep == "passed" and saw != marked:
        pytest.fail(f"{_exp_key(request.node.nodeid)}: the test {'selected' if saw else 'did not select'} an experiment-only kernel "
                    f"variant but is {'not ' if not marked else ''}listed in tests/experiment_ids.txt -- regenerate it: "
                    "MTN_RECORD_EXPERIMENTS=tests/experiment_ids.txt python -m pytest tests -m 'not gpu' -n 0 -q", pytrace=False)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    out = yield
    if call.when == "call":
        item._mtn_call_outcome = out.get_result().outcome


def pytest_sessionfinish(session, exitstatus):
    rec = os.environ.get("MTN_RECORD_EXPERIMENTS")
    if rec and not hasattr(session.config, "workerinput") and os.path.exists(rec + ".new"):
        with open(rec + ".new") as fh:
            ids = sorted({ln.strip() for ln in fh if ln.strip()})
        with open(rec, "w") as fh:
            fh.write("# test ids ('*' = the backend) whose body selects an experiment-only kernel variant (backend.dbg):\n"
                     "# generated -- MTN_RECORD_EXPERIMENTS=tests/experiment_ids.txt python -m pytest tests -m 'not gpu' -n 0 -q\n")
            fh.write("\n".join(ids) + "\n")
        os.remove(rec + ".new")


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
