#!/bin/bash
# kernel-level A/B of two library builds inside the real step, same box:
#   tools/prof_ab.sh [other.so]   -> gpurun_out/profA (other) and gpurun_out/profB (default) rocpd databases
cd "$(dirname "$0")/.."
OTHER=${1:-hand-gesture-recognition_amd/libmtn_base.so}
export TMPDIR=/tmp
rm -rf gpurun_out/profA gpurun_out/profB
rocprofv3 --kernel-trace -d gpurun_out/profA -o run -- python tools/ab_step.py "$OTHER" > gpurun_out/profA.log 2>&1
rocprofv3 --kernel-trace -d gpurun_out/profB -o run -- python tools/ab_step.py > gpurun_out/profB.log 2>&1
python tools/prof_diff.py gpurun_out/profA/run_results.db gpurun_out/profB/run_results.db
