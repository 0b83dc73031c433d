#!/bin/bash
# PMC passes over the decoder (each pass its own run; counters only, no tracing)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -o $name -- python3 tools/prof_decode.py 2 > gpurun_out/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/pmc_$name.log; exit 1; }; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob('gpurun_out/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        if 'sdf_decode' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    print(f.split('/')[1], {k: '%.4g' % (sum(v) / len(v)) for k, v in acc.items()})
PY
