#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit counters of the decoder only (each pass its own run; counters only, no tracing)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -o $name -- python3 tools/prof_decode.py 2 > gpurun_out/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/pmc_$name.log; exit 1; }; }
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob('gpurun_out/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        if 'sdf_decode' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    print(f.split('/')[1], {k: '%.4g' % (sum(v) / len(v)) for k, v in acc.items()})
PY
