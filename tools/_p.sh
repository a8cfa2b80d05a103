export TMPDIR=/tmp
rm -rf gpurun_out/profB
rocprofv3 --kernel-trace -d gpurun_out/profB -o run -- python tools/ab_step.py > gpurun_out/profB.log 2>&1
python tools/prof_db.py gpurun_out/profB/run_results.db > gpurun_out/profB.txt
