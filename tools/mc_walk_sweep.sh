#!/bin/bash
# timing sweep of the walking classifier's walk length L (DISN_MC_WALK_L) at sdf_res 256 / 512 / 1024
cd $GRAFT_REPO_ROOT
O=gpurun_out/mc; mkdir -p $O; : > $O/walk_sweep.txt
for spec in "256 1 2 3 4" "512 1 2 3 4 5" "1024 2 3 4 5 6 8 10"; do
  set -- $spec; res=$1; shift
  for L in "$@"; do
    echo -n "L $L  " >> $O/walk_sweep.txt
    DISN_MC_WALK_L=$L timeout -k 10 100 python tools/mc_bench.py $res 2>/dev/null | cut -c1-100 >> $O/walk_sweep.txt || exit 1
  done
done
cat $O/walk_sweep.txt
