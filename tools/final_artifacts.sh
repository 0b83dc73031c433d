#!/bin/bash
# end-of-round artifacts: GPU tests, full bench (with cpu_baseline), rocprofv3 kernel trace of the same command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_sdf.py tests/test_gpu_pipeline.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_quick.log 2>&1 || { tail -5 gpurun_out/pytest_quick.log; exit 1; }
tail -n 1 gpurun_out/pytest_quick.log
timeout -k 10 400 python bench.py > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err || { tail -5 gpurun_out/bench_full.err; exit 1; }
tail -n 1 gpurun_out/bench_full.log | cut -c1-600
rm -rf gpurun_out/prof_r02
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02 -o bench -- python3 bench.py --no-cpu-baseline --no-batch > gpurun_out/prof_r02.log 2>&1 || { tail -5 gpurun_out/prof_r02.log; exit 1; }
find gpurun_out/prof_r02 -name "*.db" | head -2
DB=$(find gpurun_out/prof_r02 -name "*.db" | head -1)
python3 tools/prof_summary.py $DB > gpurun_out/prof_r02_kernel_stats.txt && head -12 gpurun_out/prof_r02_kernel_stats.txt | cut -c1-150
grep '^{' gpurun_out/prof_r02.log | tail -1 > gpurun_out/prof_r02_bench_line.json
rm -rf gpurun_out/prof_r02
