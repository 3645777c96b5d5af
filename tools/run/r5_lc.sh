#!/bin/bash
# launch-count work: affected tests, then the launch count of a replayed step
mkdir -p gpurun_out/lc
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "align or adam or optim or head or consis or prefetch" > gpurun_out/lc/k.log 2>&1; echo "rc=$?" >> gpurun_out/lc/k.log; tail -3 gpurun_out/lc/k.log
timeout 1500 python -m pytest tests/test_gpu_tta.py tests/test_gpu_trunk.py -x -q > gpurun_out/lc/t.log 2>&1; echo "rc=$?" >> gpurun_out/lc/t.log; tail -4 gpurun_out/lc/t.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config2" > gpurun_out/lc/f.log 2>&1; echo "rc=$?" >> gpurun_out/lc/f.log; tail -3 gpurun_out/lc/f.log
bash tools/run/r5_tl.sh r5n > /dev/null 2>&1
python - <<'PY'
import csv, json
rows = list(csv.DictReader(open('gpurun_out/r5n_graph_replay_kernel_stats.csv')))
calls = sum(int(r['calls']) for r in rows)
d = json.loads(open('gpurun_out/r5n_timed_only.json').read().strip().splitlines()[-1])
print('kernel calls', calls, 'ms/step', d['ms_per_step'], 'value', d['value'])
for r in rows[:45]:
    print(r['kernel'][:70], r['calls'], r['avg_us'])
PY
