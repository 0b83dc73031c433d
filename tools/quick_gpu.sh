#!/bin/bash
# quick GPU check: SDF parity tests + short bench (no CPU baseline)
timeout -k 10 300 python -u -m pytest tests/test_gpu_sdf.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
tail -n 3 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench1.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/bench1.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('ms_per_step %.2f decode_ms %.2f executed_TF %.0f issue_frac %.3f value %.3e' % (d['ms_per_step'], r['kernel_ms'], r['executed_mfma_tflops_bf16_rate'], r['mfma_issue_frac'], d['value']))
        break
else:
    print(open('gpurun_out/bench1.log').read()[-1500:])
PY
