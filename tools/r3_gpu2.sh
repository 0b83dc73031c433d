#!/bin/bash
# per-kernel times of marching cubes at sdf_res 1024, walking vs linear classifier
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in walk linear; do
  rm -rf gpurun_out/prof_mc_$v
  DISN_MC_CLASSIFY=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mc_$v -o mc -- python3 tools/mc_bench.py 1024 > gpurun_out/r3_mcprof_$v.log 2>&1 || { tail -5 gpurun_out/r3_mcprof_$v.log; exit 1; }
  DB=$(find gpurun_out/prof_mc_$v -name "*.db" | head -1)
  echo "== $v"
  python3 tools/prof_summary.py $DB > gpurun_out/r3_mc1024_${v}_kernel_stats.txt && grep -E "mc_|tri_sort|kernel " gpurun_out/r3_mc1024_${v}_kernel_stats.txt | cut -c1-170
  rm -rf gpurun_out/prof_mc_$v
done
