#!/bin/bash
# Re-collect the evidence under gpurun_out/ on a GPU box: tools/refresh_profiles.sh <tag> [step|configs|all]   (e.g. r03_v1)
#   step:    kernel trace of the bench command -> per-kernel table; FETCH_SIZE / WRITE_SIZE passes (separate runs) -> HBM bytes
#            per launch; one pass with the eight SQ counters -> issue / stall / wait / MFMA-busy per kernel; the bench line.
#   x3:      kernel trace + HBM passes of the same command with --dtype bf16x3.
#   configs: the same trace + HBM passes for BASELINE configs 4 (384x384, B=128) and 5 (GELAN only, B=512).
# (rocprofv3 gets the program directly after `--`; counter passes never combine with other trace domains.)
cd "$(dirname "$0")/.."
TAG=${1:-rXX}
WHAT=${2:-all}
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-roofline --no-validation --no-configs --windows 0"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE"
O=gpurun_out/${TAG}
mkdir -p $O
if [ "$WHAT" = step ] || [ "$WHAT" = all ]; then
  rm -rf $O/stats $O/pmcf $O/pmcw $O/pmcsq
  rocprofv3 --kernel-trace --stats -d $O/stats -o run -- python3 bench.py --steps 10 --warmup 3 $Q > $O/stats.log 2>&1 || exit 1
  python tools/prof_db.py $O/stats/run_results.db > $O/kernel_stats_step_b256.txt || exit 1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf -o run -- python3 bench.py --steps 3 --warmup 1 $Q > $O/pmcf.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw -o run -- python3 bench.py --steps 3 --warmup 1 $Q > $O/pmcw.log 2>&1 || exit 1
  python tools/pmc_summary.py $O/pmcf $O/pmcw $O/pmc_hbm_traffic.json > $O/pmc_summary.txt || exit 1
  python tools/per_kernel_roofline.py $O/stats/run_results.db $O/pmc_hbm_traffic.json > $O/per_kernel_hbm.txt || exit 1
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmcsq -o run -- python3 bench.py --steps 2 --warmup 1 $Q > $O/pmcsq.log 2>&1 || exit 1
  python tools/sq_summary.py $O/pmcsq igemm_kernel pw_kernel wgrad dgrad2 attn_ bn_ > $O/sq_counters_step_b256.txt || exit 1
  rm -rf $O/pmcf $O/pmcw $O/pmcsq        # (raw CSVs are tens of MB: only the summaries travel back)
fi
if [ "$WHAT" = x3 ] || [ "$WHAT" = all ]; then
  # the split-bf16 mode (the one that meets the 1e-3 tolerance): kernel table + HBM bytes per launch for its roofline object
  rm -rf $O/x3_stats $O/x3_pmcf $O/x3_pmcw
  rocprofv3 --kernel-trace --stats -d $O/x3_stats -o run -- python3 bench.py --dtype bf16x3 --steps 10 --warmup 3 $Q > $O/x3_stats.log 2>&1 || exit 1
  python tools/prof_db.py $O/x3_stats/run_results.db > $O/kernel_stats_step_b256_bf16x3.txt || exit 1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/x3_pmcf -o run -- python3 bench.py --dtype bf16x3 --steps 3 --warmup 1 $Q > $O/x3_pmcf.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/x3_pmcw -o run -- python3 bench.py --dtype bf16x3 --steps 3 --warmup 1 $Q > $O/x3_pmcw.log 2>&1 || exit 1
  python tools/pmc_summary.py $O/x3_pmcf $O/x3_pmcw $O/pmc_hbm_traffic_bf16x3.json > $O/pmc_summary_bf16x3.txt || exit 1
  python tools/per_kernel_roofline.py $O/x3_stats/run_results.db $O/pmc_hbm_traffic_bf16x3.json > $O/per_kernel_hbm_bf16x3.txt || exit 1
  rm -rf $O/x3_pmcf $O/x3_pmcw
fi
if [ "$WHAT" = configs ] || [ "$WHAT" = all ]; then
  for C in 4 5; do
    rm -rf $O/c${C}_stats $O/c${C}_pmcf $O/c${C}_pmcw
    rocprofv3 --kernel-trace --stats -d $O/c${C}_stats -o run -- python3 tools/bench_configs.py $C > $O/c${C}_stats.log 2>&1 || exit 1
    python tools/prof_db.py $O/c${C}_stats/run_results.db 1 > $O/kernel_stats_config${C}.txt || exit 1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c${C}_pmcf -o run -- python3 tools/bench_configs.py $C > $O/c${C}_pmcf.log 2>&1 || exit 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c${C}_pmcw -o run -- python3 tools/bench_configs.py $C > $O/c${C}_pmcw.log 2>&1 || exit 1
    python tools/pmc_summary.py $O/c${C}_pmcf $O/c${C}_pmcw $O/pmc_hbm_traffic_config${C}.json > $O/pmc_summary_config${C}.txt || exit 1
    python tools/per_kernel_roofline.py $O/c${C}_stats/run_results.db $O/pmc_hbm_traffic_config${C}.json 1 > $O/per_kernel_hbm_config${C}.txt || exit 1
    rm -rf $O/c${C}_pmcf $O/c${C}_pmcw
  done
fi
ls $O
