#!/bin/bash
# round-3 artifacts, part 2: marching cubes alone (timing, per-kernel trace, PMC at sdf_res 1024), larger grids on one GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
echo "== mc_bench"; timeout -k 10 200 python3 -u tools/mc_bench.py 256 512 1024 2>&1 | grep "res " | tee $O/mc_bench.txt
echo "== mc kernel trace at 1024"
rm -rf $O/prof_mc
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_mc -o mc -- python3 tools/mc_bench.py 1024 > $O/prof_mc.log 2>&1 || { tail -5 $O/prof_mc.log; exit 1; }
python3 tools/prof_summary.py $(find $O/prof_mc -name "*.db" | head -1) | grep -E "kernel |mc_|tri_sort" | cut -c1-170 | tee $O/mc1024_kernel_stats.txt
rm -rf $O/prof_mc
echo "== mc pmc at 1024"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O/pmcmc_$name -o $name -- python3 tools/mc_bench.py 1024 > $O/pmcmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $O/pmcmc_$name.log; exit 1; }; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE || exit 1
python3 - > $O/pmc_mc_res1024.txt <<'PY'
import csv, glob
for f in sorted(glob.glob('gpurun_out/r03/pmcmc_*/**/*counter_collection.csv', recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for n in ('mc_classify_walk', 'mc_vertices', 'mc_triangles'):
            if n in k:
                acc.setdefault((n, r['Counter_Name']), []).append(float(r['Counter_Value']))
    for (n, c), v in sorted(acc.items()):
        print(f.split('/')[2], n, c, '%.4g' % (sum(v) / len(v)))
PY
cat $O/pmc_mc_res1024.txt
rm -rf $O/pmcmc_*/
echo "== res 512 / 1024 on one GPU"
timeout -k 10 300 python bench.py --res 512 --steps 3 --warmup 1 --no-cpu-baseline --no-batch > $O/bench_res512.json 2>/dev/null || exit 1
cut -c1-260 $O/bench_res512.json
timeout -k 10 400 python bench.py --res 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-batch > $O/bench_res1024.json 2>/dev/null || exit 1
cut -c1-260 $O/bench_res1024.json
