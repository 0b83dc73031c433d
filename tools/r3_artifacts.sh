#!/bin/bash
# round-3 artifacts: full bench line, rocprofv3 kernel trace of the same command, PMC passes (decoder, marching cubes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
step() { echo "== $1"; }
step pmc-decode
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -o $name -- python3 tools/prof_decode.py 2 256 mixed4 > $O/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $O/pmc_$name.log; exit 1; }; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS &&
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE || exit 1
python3 - > $O/pmc_sdf_decode_mx4.txt <<'PY'
import csv, glob
for f in sorted(glob.glob('gpurun_out/r03/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'sdf_decode' in k:
            name = 'local' if 'ILb1' in k or '<true' in k else 'global'
            acc.setdefault((name, r['Counter_Name']), []).append(float(r['Counter_Value']))
    for (n, c), v in sorted(acc.items()):
        print(f.split('/')[2], 'sdf_decode_mx<%s,true>' % ('true' if n == 'local' else 'false'), c, '%.5g' % (sum(v) / len(v)), 'launches', len(v))
PY
cat $O/pmc_sdf_decode_mx4.txt
rm -rf $O/pmc_*/
python3 - <<'PY'
import json, re
v = {}
for line in open('gpurun_out/r03/pmc_sdf_decode_mx4.txt'):
    f = line.split()
    if f[2] in ('FETCH_SIZE', 'WRITE_SIZE', 'TCC_HIT_sum', 'TCC_REQ_sum'):
        v[(f[2], 'local' if '<true' in f[1] else 'global')] = float(f[3])
fg, fl, wg, wl = v[('FETCH_SIZE', 'global')], v[('FETCH_SIZE', 'local')], v[('WRITE_SIZE', 'global')], v[('WRITE_SIZE', 'local')]
total = int(((fg + fl) * 2 + wg + wl) * 1024)
hit = [100.0 * v[('TCC_HIT_sum', k)] / v[('TCC_REQ_sum', k)] for k in ('local', 'global')]
note = ("round 3, all-mixed decoder (DISN_DECODER_MIXED4), one launch per stream (tools/r3_artifacts.sh, profiles/r03_pmc_sdf_decode_mx4.txt): "
        "FETCH_SIZE %.3g KB (global stream) + %.3g KB (local stream), each x2 (gfx950 counts 128-B requests as 64 B), WRITE_SIZE %.3g + %.3g KB "
        "-> %.2f GB of L2<->fabric traffic per grid (both launches), L2 hit rate %.2f-%.2f %%.  Algorithmic HBM bytes: 110 MB (weights 4.3 + Pmap 38 + "
        "output 68); the rest is the intermediate g (written by the global launch, read back by the local one), partial-line writes of the "
        "brick-ordered output and pixel-map lines that leave an XCD's L2 between the waves that share them."
        % (fg, fl, wg, wl, total / 1e9, min(hit), max(hit)))
d = {"sdf_res": 256, "sdf_decode_bytes_per_launch": total, "note": note}
for path in ('gpurun_out/r03/pmc_traffic.json', 'profiles/pmc_traffic.json'):
    json.dump(d, open(path, 'w'), indent=1)
print(total)
PY
step bench
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cut -c1-300 $O/bench.json
step kernel-trace
rm -rf $O/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python3 bench.py --no-cpu-baseline --no-batch > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
DB=$(find $O/prof -name "*.db" | head -1)
python3 tools/prof_summary.py $DB > $O/bench_kernel_stats.txt && head -8 $O/bench_kernel_stats.txt | cut -c1-150
grep '^{' $O/prof.log | tail -1 > $O/bench_under_rocprof.json
rm -rf $O/prof
