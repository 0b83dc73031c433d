#!/bin/bash
# per-segment cycle profile of sdf_decode: build a patched COPY of sdf.hip with counters (tools/seg_patch.py)
# into the library, run tools/seg_probe.py on the GPU box, rebuild the clean library.  Run from anywhere.
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$ROOT"
KEEP="$(mktemp)"
cp disn_amd/csrc/sdf.hip "$KEEP"
trap 'cp "$KEEP" disn_amd/csrc/sdf.hip; rm -f "$KEEP"; make -s -C disn_amd/csrc 2>&1 | grep -i error || true' EXIT
python - <<'PY'
import os
p = 'disn_amd/csrc/sdf.hip'
s = open(p).read()
exec(open(os.environ.get('SEGPATCH', 'tools/seg_patch.py')).read())
print("ticks", s.count("TICK("))
open(p, 'w').write(s)
PY
make -s -C disn_amd/csrc SKIP_CHECK=1 2>&1 | grep -i error || true
grep -c scratch_ disn_amd/csrc/.temps/sdf-hip-amdgcn-amd-amdhsa-gfx950.s || true
gpurun --timeout 600 -- "timeout -k 10 200 python ${SEGPROBE:-tools/seg_probe.py}" 2>&1 | tail -10
