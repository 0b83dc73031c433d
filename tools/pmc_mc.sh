#!/bin/bash
# PMC passes over the marching-cubes kernels at sdf_res 1024 (each pass its own run; counters only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcmc_$name -o $name -- python3 tools/mc_bench.py 1024 > gpurun_out/pmcmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/pmcmc_$name.log; exit 1; }; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob('gpurun_out/pmcmc_*/**/*counter_collection.csv', recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for n in ('mc_classify_walk', 'mc_vertices', 'mc_tri_records', 'mc_tri_finish'):
            if n in k:
                acc.setdefault((n, r['Counter_Name']), []).append(float(r['Counter_Value']))
    for (n, c), v in sorted(acc.items()):
        print(f.split('/')[1], n, c, '%.4g' % (sum(v) / len(v)))
PY
