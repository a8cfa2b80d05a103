#!/bin/bash
# Re-collect the evidence under gpurun_out/ on a GPU box: tools/refresh_profiles.sh <tag>   (e.g. r02_v2)
# kernel trace of the bench command -> per-kernel table; FETCH_SIZE / WRITE_SIZE passes (separate runs) -> HBM bytes per launch.
cd "$(dirname "$0")/.."
TAG=${1:-rXX}
export TMPDIR=/tmp
rm -rf gpurun_out/${TAG}_stats gpurun_out/${TAG}_pmcf gpurun_out/${TAG}_pmcw
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -o run -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_stats.log 2>&1 || exit 1
python tools/prof_db.py gpurun_out/${TAG}_stats/run_results.db > gpurun_out/${TAG}_kernel_stats.txt || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmcf -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_pmcf.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmcw -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_pmcw.log 2>&1 || exit 1
python tools/pmc_summary.py gpurun_out/${TAG}_pmcf gpurun_out/${TAG}_pmcw gpurun_out/${TAG}_pmc_hbm_traffic.json > gpurun_out/${TAG}_pmc_summary.txt || exit 1
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-200
