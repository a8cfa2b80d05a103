"""Register / LDS / spill table of every kernel in a built code object (no GPU needed):
python tools/kernel_regs.py [path/to/libmtn_hip.so] [substring]

Reads the AMDGPU metadata notes of the gfx950 bundles embedded in the shared library.  Used by
tests/test_occupancy.py to pin the occupancy the launch heuristics assume (e.g. the 8-wave 128x128
igemm tile needs <= 128 VGPRs for two workgroups per CU)."""
import os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    out = {}
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "lib.so")
        shutil.copy(lib, p)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", p], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, cwd=td)
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(td, f)],
                                   capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name:
                    continue
                get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
                out[dem.replace("(anonymous namespace)::", "").replace("void ", "")] = {
                    "vgpr": get("vgpr_count"), "sgpr": get("sgpr_count"), "spill": get("vgpr_spill_count"),
                    "lds": get("group_segment_fixed_size"), "scratch": get("private_segment_fixed_size"),
                    "wg": get("max_flat_workgroup_size")}
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(here, "hand-gesture-recognition_amd", "libmtn_hip.so")
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for k, v in sorted(kernels(lib).items()):
        if sub in k:
            print(f"{k[:90]:90s} vgpr {v['vgpr']:4d} sgpr {v['sgpr']:4d} spill {v['spill']:4d} scratch {v['scratch']:5d} lds {v['lds']:6d} wg {v['wg']}")
