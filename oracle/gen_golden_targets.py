"""Pins the target renderer against the REFERENCE's own ``HagridDataset.generate_target``.

TEST INFRASTRUCTURE, build container only: reads /root/reference (absent on the GPU box) and writes data -- inputs and
expected outputs -- to tests/golden/targets.npz.  ``libs/load.py`` cannot be imported here (cv2 / lightning / torchvision are
not installed: an ordinary ImportError, nothing was denied), but the method's body needs nothing besides numpy: this script
parses the file, compiles only the ``generate_target`` function node of ``HagridDataset`` (libs/load.py:148-206) and calls it
on a stand-in ``self`` carrying the four attributes it reads (``num_joints``, ``heatmap_size``, ``image_size``, ``sigma``;
set as ``__init__`` sets them, libs/load.py:34-37).  No reference source text is stored: only the vectors travel.

It also checks, case by case, that oracle/mtn_metrics_oracle.generate_target (the restatement the CPU tests use) returns
exactly the same arrays, so the restatement is pinned too.

Cases: centres on every border and corner, Gaussians partly and fully outside, the exact in/out
boundary of the ul / br test, invisible joints, a non-square heat-map, fractional coordinates on both sides of the
round-half-up of ``int(x / stride + 0.5)``, negative coordinates (int() truncates towards zero), the reference's
192 x 192 default, an empty joint list (all-zero target and weights), and a seeded random sweep.

Usage (from the repo root):  python oracle/gen_golden_targets.py
"""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_FILE = "/root/reference/libs/load.py"
sys.path.insert(0, HERE)
import mtn_metrics_oracle as M  # noqa: E402


def reference_generate_target():
    """The reference's method as a plain function ``f(self, joints, joints_vis)``, compiled from its own source file."""
    with open(REF_FILE) as fh:
        tree = ast.parse(fh.read(), REF_FILE)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "HagridDataset")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "generate_target")
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, REF_FILE, "exec"), ns)
    return ns["generate_target"], (fn.lineno, fn.end_lineno)


def stand_in(num_joints, image_size, sigma):
    # libs/load.py:34-37: image_size is [W, H] as the method indexes it ([0] with x), heatmap_size = [s // 4 ...]
    return types.SimpleNamespace(num_joints=num_joints, image_size=list(image_size),
                                 heatmap_size=[s // 4 for s in image_size], sigma=sigma)


def cases():
    out = []
    rng = np.random.default_rng(20250105)

    def add(name, joints, vis, image_size, sigma=2):
        joints = np.asarray(joints, np.float64)
        vis = np.asarray(vis, np.float64).reshape(-1, 1)
        out.append((name, joints, vis, tuple(image_size), sigma))

    S = 64                      # heat-map 16 x 16, stride 4, Gaussian patch 13 x 13 (sigma 2)
    border = [[0, 0], [S - 1, 0], [0, S - 1], [S - 1, S - 1], [S / 2, 0], [0, S / 2], [S - 1, S / 2], [S / 2, S - 1],
              [S / 2, S / 2], [2.0, 2.0], [1.9, 1.9], [6.0, 6.0], [5.9, 61.9]]
    add("borders_64", border, np.ones(len(border)), (S, S))
    # ul >= heatmap_size <=> mu - 6 >= 16 <=> mu >= 22; br < 0 <=> mu + 7 < 0 <=> mu <= -8: both sides of each boundary
    edge = [[4 * 21, 8], [4 * 22, 8], [4 * 21.4, 8], [4 * 21.5, 8], [8, 4 * 22], [8, 4 * 21], [-4 * 7, 8], [-4 * 8, 8],
            [-4 * 8.6, 8], [8, -4 * 7], [8, -4 * 9], [-200.0, 10.0], [10.0, 900.0], [-2.1, -2.1], [-1.9, -1.9], [-30.0, -30.0]]
    add("in_out_boundary_64", edge, np.ones(len(edge)), (S, S))
    inv = [[20, 20], [40, 12], [12, 40], [-200, 3]]
    add("invisible_64", inv, [0.0, 1.0, 0.0, 0.0], (S, S))
    add("half_visible_64", inv, [0.5, 0.51, 1.0, 1.0], (S, S))                      # the method tests v > 0.5
    frac = [[x, y] for x in (9.9, 10.0, 10.1, 13.9, 14.0) for y in (29.9, 30.0, 30.1)]
    add("rounding_64", frac, np.ones(len(frac)), (S, S))
    ns = [[0, 0], [95, 31], [47.5, 15.5], [95, 0], [0, 31], [100, 10], [-9, 40], [60, 33.9], [60, 56], [60, 57.9], [60, 58.1]]
    add("nonsquare_96x32", ns, np.ones(len(ns)), (96, 32))
    add("nonsquare_32x96", [[p[1], p[0]] for p in ns], np.ones(len(ns)), (32, 96))
    add("sigma3_64", border[:9], np.ones(9), (S, S), sigma=3)
    add("sigma1_64", border[:9], np.ones(9), (S, S), sigma=1)
    j = rng.uniform(-30, 222, size=(21, 2))
    add("random_192", j, (rng.uniform(size=21) > 0.3).astype(np.float64), (192, 192))
    for k in range(4):
        j = rng.uniform(-20, 84, size=(21, 2))
        add(f"random_64_{k}", j, (rng.uniform(size=21) > 0.3).astype(np.float64), (S, S))
    j3 = np.concatenate([rng.uniform(0, 64, size=(21, 2)), np.zeros((21, 1))], 1)     # [J, 3] like the docstring says
    add("three_columns_64", j3, np.ones((21, 3))[:, :1], (S, S))
    return out


def main():
    ref, lines = reference_generate_target()
    data, names = {}, []
    for name, joints, vis, image_size, sigma in cases():
        me = stand_in(joints.shape[0], image_size, sigma)
        t_ref, w_ref = ref(me, joints.copy(), vis.copy())
        t_own, w_own = M.generate_target(joints.copy(), vis.copy(), image_size, me.heatmap_size, sigma)
        assert t_ref.dtype == np.float32 and t_ref.shape == (joints.shape[0], me.heatmap_size[1], me.heatmap_size[0])
        assert np.array_equal(t_ref, t_own) and np.array_equal(w_ref, w_own), f"restatement differs from the reference: {name}"
        names.append(name)
        data[f"{name}.joints"], data[f"{name}.vis"] = joints[:, :2].astype(np.float32), vis[:, 0].astype(np.float32)
        # (float32 inputs are what the HIP renderer takes; the case is only kept if the reference returns the same arrays for them)
        t32, w32 = ref(me, data[f"{name}.joints"].astype(np.float64), data[f"{name}.vis"].astype(np.float64)[:, None])
        assert np.array_equal(t32, t_ref) and np.array_equal(w32, w_ref), f"case {name} is not float32-stable"
        data[f"{name}.image_size"] = np.asarray(image_size, np.int32)
        data[f"{name}.sigma"] = np.asarray(sigma, np.int32)
        data[f"{name}.target"], data[f"{name}.weight"] = t_ref, w_ref
    # the empty-annotation branch (libs/load.py:159-161): all-zero target, all-zero weights
    me = stand_in(21, (64, 64), 2)
    t0, w0 = ref(me, np.zeros((0, 2)), np.ones((21, 1)))
    assert not t0.any() and not w0.any() and t0.shape == (21, 16, 16) and w0.shape == (21, 1)
    data["names"] = np.asarray(names)
    data["reference_lines"] = np.asarray(lines, np.int32)
    path = os.path.join(ROOT, "tests", "golden", "targets.npz")
    np.savez_compressed(path, **data)
    print(f"{len(names)} cases from {REF_FILE}:{lines[0]}-{lines[1]} -> {path} ({os.path.getsize(path)} bytes); "
          "oracle restatement identical on every case")


if __name__ == "__main__":
    main()
