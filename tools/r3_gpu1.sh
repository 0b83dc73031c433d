#!/bin/bash
# round-3 check: MC tests with the walking classifier, MC timing A/B
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_mc.py tests/test_gpu_mc_geometry.py tests/test_gpu_large.py -m gpu -x -q --timeout 170 --timeout-method thread > gpurun_out/r3_mc_tests.log 2>&1
rc=$?; tail -n 3 gpurun_out/r3_mc_tests.log
[ $rc -eq 0 ] || exit $rc
echo "== walk" ; timeout -k 10 120 python -u tools/mc_bench.py 256 512 1024 2>&1 | grep res || exit 1
echo "== linear" ; DISN_MC_CLASSIFY=linear timeout -k 10 120 python -u tools/mc_bench.py 256 512 1024 2>&1 | grep res || exit 1
