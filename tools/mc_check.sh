#!/bin/bash
# marching cubes: parity tests, timing at sdf_res 256/512/1024, per-kernel times at 1024
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/mc; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_mc.py tests/test_gpu_mc_geometry.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest_mc.log 2>&1 || { tail -30 $O/pytest_mc.log; exit 1; }
tail -n 2 $O/pytest_mc.log
timeout -k 10 200 python tools/mc_bench.py 2>/dev/null | tee $O/mc_bench.txt || exit 1
rm -rf $O/prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof -o mc -- python3 tools/mc_bench.py 1024 > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
python3 tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) | cut -c1-170 > $O/mc_res1024_kernel_stats.txt
head -12 $O/mc_res1024_kernel_stats.txt
rm -rf $O/prof
