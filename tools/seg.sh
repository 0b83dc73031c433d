#!/bin/bash
# per-segment cycle profile of sdf_decode: patch in counters (tools/seg_patch.py), run tools/seg_probe.py, restore
cp disn_amd/csrc/sdf.hip /tmp/sdf_keep.hip
python - <<'PY'
p='/root/repo/disn_amd/csrc/sdf.hip'
s=open(p).read()
exec(open(__import__('os').environ.get('SEGPATCH','tools/seg_patch.py')).read())
print("ticks", s.count("TICK("))
open(p,'w').write(s)
PY
make -s -C disn_amd/csrc SKIP_CHECK=1 2>&1 | grep -i error
grep -c scratch_ disn_amd/csrc/.temps/sdf-hip-amdgcn-amd-amdhsa-gfx950.s
/usr/local/graft/bin/gpurun --timeout 600 -- 'timeout -k 10 200 python tools/seg_probe.py' 2>&1 | tail -10
cp /tmp/sdf_keep.hip disn_amd/csrc/sdf.hip; make -s -C disn_amd/csrc 2>&1 | grep -i error
