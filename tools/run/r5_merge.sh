#!/bin/bash
# merged TAM launches: kernel tests, trunk tests, quick bench A/B
mkdir -p gpurun_out/merge
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "tam" > gpurun_out/merge/k.log 2>&1; echo "rc=$?" >> gpurun_out/merge/k.log
tail -5 gpurun_out/merge/k.log
timeout 900 python -m pytest tests/test_gpu_trunk.py tests/test_gpu_tta.py -x -q -k "not swin" > gpurun_out/merge/t.log 2>&1; echo "rc=$?" >> gpurun_out/merge/t.log
tail -5 gpurun_out/merge/t.log
for v in 1 0 1 0; do
  VITTA_TRUNK_TAM_MERGE=$v timeout 300 python bench.py --no-swin --no-cpu-baseline --no-sgd-all --no-streaming 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('merge=$v', round(d['value'],2), round(d['ms_per_step'],3), d['adapt_only_ms'], r['frac_of_fp32_matrix_peak'], r['kernel_ms_per_step'])" >> gpurun_out/merge/ab.log 2>&1
done
cat gpurun_out/merge/ab.log
