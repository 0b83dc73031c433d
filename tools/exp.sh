#!/bin/bash
# timing experiment: apply a python patch to sdf.hip (argument = file with python code operating on s), build, run bench, restore
cp disn_amd/csrc/sdf.hip /tmp/sdf_keep.hip
python - "$1" <<'PY'
import sys
p='/root/repo/disn_amd/csrc/sdf.hip'
s=open(p).read()
exec(open(sys.argv[1]).read())
open(p,'w').write(s)
PY
make -s -C disn_amd/csrc 2>&1 | grep error
/usr/local/graft/bin/gpurun --timeout 600 -- 'timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"decode_ms\", d[\"roofline\"][\"kernel_ms\"])"' 2>&1 | grep decode_ms
cp /tmp/sdf_keep.hip disn_amd/csrc/sdf.hip
