#!/bin/bash
# encoder split-policy sweep (experiment): per-pass busy time of the encoder's kernels for several (workgroups per CU, minimum stages per split)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/enc; mkdir -p $O
IFS=","; CFGS=${ENC_CFGS:-2 4,3 4,2 6,2 3}
for cfg in $CFGS; do
  IFS=" "
  set -- $cfg
  export DISN_VGG_WG_PER_CU=$1 DISN_VGG_MIN_STAGES=$2
  rm -rf $O/prof
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof -o enc -- python3 tools/prof_encoder.py > $O/log.txt 2>&1 || { tail -3 $O/log.txt; exit 1; }
  python3 tools/prof_calls.py $(find $O/prof -name "*.db" | head -1) 52 > $O/calls_$1_$2.txt
  echo "wg/cu $1 min stages $2: $(python3 - <<PY
import re
rows=[l for l in open('$O/calls_$1_$2.txt') if ' us  +' in l]
# one pass = from the last k_resize_pad4 on
idx=max(i for i,l in enumerate(rows) if 'k_resize_pad4' in l)
p=rows[idx:]
p=[l for l in p if 'reduce_kernel' not in l and 'copyBuffer' not in l]
busy=sum(float(l.split('+')[1].split('us')[0]) for l in p)
conv=sum(float(l.split('+')[1].split('us')[0]) for l in p if 'k_conv3x3' in l)
sm=sum(float(l.split('+')[1].split('us')[0]) for l in p if 'k_sum_splits' in l)
fc=sum(float(l.split('+')[1].split('us')[0]) for l in p if 'k_fc_partial' in l)
print('busy %.0f us (conv %.0f, sums %.0f, fc %.0f)' % (busy, conv, sm, fc))
PY
)"
done
rm -rf $O/prof
