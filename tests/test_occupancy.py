"""Register budgets the launch heuristics rely on, read from the built gfx950 code object (no GPU).

A kernel that grows past its budget still runs and still passes parity -- it just loses a resident
workgroup per CU (the GELU-backward epilogue once cost every igemm instantiation 16 VGPRs and the
training step 6 %).  These limits turn that into a failing CPU test."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def regs():
    from kernel_regs import kernels
    lib = os.path.join(ROOT, "hand-gesture-recognition_amd", "libmtn_hip.so")
    if not os.path.exists(lib):
        pytest.skip("HIP library not built")
    return kernels(lib)


# (kernel name prefix, max VGPRs, why)
BUDGETS = [
    ("igemm_kernel<bf16_t, 128, 128, 4, 2, 3,", 128, "8 waves x 2 workgroups per CU = 4 waves per SIMD"),
    ("igemm_kernel<bf16_t, 64, 128, 1, 4, 3,", 168, "4 waves x 3 workgroups per CU = 3 waves per SIMD"),
    ("igemm_kernel<bf16_t, 128, 64, 4, 1, 3,", 168, "4 waves x 3 workgroups per CU"),
    ("igemm_kernel<bf16_t, 128, 64, 4, 1, 5, 2>", 168, "window mode: 3 workgroups per CU (53 KB of LDS each)"),
    ("dgrad2_kernel<bf16_t>", 128, "8 waves x 2 workgroups per CU"),
    ("wgrad_dma2_kernel<bf16_t, 128, 128, 4, 2, false>", 128, "8 waves x 2 workgroups per CU"),
    ("wgrad3_kernel<bf16_t,", 256, "8 waves, one workgroup per CU (128 and 256 slots per step)"),
    ("igemm_kernel<bf16_t, 128, 128, 4, 2, 5, 2>", 128, "window mode on the 8-wave tile: 2 workgroups per CU (78 KB of LDS each)"),
    ("igemm_kernel<bf16_t, 256, 64, 8, 1, 5, 2>", 128, "256x64 window tile: 8 waves x 2 workgroups per CU"),
    ("wgrad_row3_kernel<bf16_t>", 256, "4 waves x 2 workgroups per CU (80 KB of LDS each)"),
    ("igemm_kernel<bf16_t, 144, 256, 1, 8, 6, 2>", 256, "image tiles: 8 waves, one workgroup per CU"),
    ("conv1_bwd_kernel<bf16_t, true, true,", 168, "__launch_bounds__(256, 3)"),
]


@pytest.mark.parametrize("prefix,limit,why", BUDGETS, ids=[b[0] for b in BUDGETS])
def test_vgpr_budget(regs, prefix, limit, why):
    hit = {k: v for k, v in regs.items() if k.startswith(prefix)}
    assert hit, f"no kernel named {prefix}*"
    for k, v in hit.items():
        assert v["vgpr"] <= limit, f"{k}: {v['vgpr']} VGPRs > {limit} ({why})"


def test_no_spills_in_the_mfma_kernels(regs):
    """scratch traffic inside a K loop is never acceptable; dgrad2's epilogue is allowed its few dwords"""
    for k, v in regs.items():
        if k.startswith(("igemm_kernel<bf16_t", "wgrad_dma2_kernel<bf16_t", "wgrad_dma_kernel<bf16_t", "wgrad3_kernel<bf16_t", "wgrad_row3_kernel<bf16_t",
                         "wgrad3_kernel<f32x3_t")):
            assert v["spill"] == 0, (k, v)
        if k.startswith("dgrad2_kernel<bf16_t"):
            assert v["spill"] <= 32, (k, v)
