#!/bin/bash
# round-3 artifacts, part 3: batch configuration (sdf_res 128): wall per view with 1..3 lanes, per-kernel GPU time of a one-lane run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
for l in 1 2 3; do timeout -k 10 200 python3 -u tools/batch_lanes.py $l 2>/dev/null | tee -a $O/batch_lanes.txt || exit 1; done
rm -rf $O/prof_batch
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_batch -o batch -- python3 tools/batch_lanes.py 1 > $O/prof_batch.log 2>&1 || { tail -5 $O/prof_batch.log; exit 1; }
python3 tools/prof_summary.py $(find $O/prof_batch -name "*.db" | head -1) | cut -c1-170 > $O/batch_res128_one_lane_kernel_stats.txt
head -14 $O/batch_res128_one_lane_kernel_stats.txt
rm -rf $O/prof_batch
