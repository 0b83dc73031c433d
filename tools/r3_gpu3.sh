#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b -o b -- python3 tools/batch_lanes.py 1 26 > gpurun_out/r3_batchprof.log 2>&1 || { tail -5 gpurun_out/r3_batchprof.log; exit 1; }
DB=$(find gpurun_out/prof_b -name "*.db" | head -1)
python3 tools/prof_summary.py $DB > gpurun_out/r3_batch_kernel_stats.txt; head -30 gpurun_out/r3_batch_kernel_stats.txt | cut -c1-60,73-130
rm -rf gpurun_out/prof_b
