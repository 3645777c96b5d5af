#!/bin/bash
mkdir -p gpurun_out/lc
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "align or adam or optim or head or consis" > gpurun_out/lc/k.log 2>&1; echo "rc=$?" >> gpurun_out/lc/k.log; tail -3 gpurun_out/lc/k.log
timeout 1500 python -m pytest tests/test_gpu_tta.py tests/test_gpu_trunk.py tests/test_gpu_entrypoints.py -x -q > gpurun_out/lc/t.log 2>&1; echo "rc=$?" >> gpurun_out/lc/t.log; tail -4 gpurun_out/lc/t.log
timeout 300 python bench.py --no-swin --no-cpu-baseline --no-streaming 2> gpurun_out/lc/bench.err | tail -1 > gpurun_out/lc/bench.json
python -c "
import json; d=json.load(open('gpurun_out/lc/bench.json')); print(d['value'], d['ms_per_step'], d['adapt_only_ms'], d['sgd_all']['value'] if d.get('sgd_all') else None, d['host_fed'])"
bash tools/run/r5_tl.sh r5n > /dev/null 2>&1
python - <<'PY'
import csv, json
rows = list(csv.DictReader(open('gpurun_out/r5n_graph_replay_kernel_stats.csv')))
calls = sum(int(r['calls']) for r in rows)
n = [int(r['calls']) for r in rows if r['kernel'].startswith('adam_step')][0]
print('kernel calls', calls, 'steps', n, 'per video', calls / n)
for r in rows:
    if 'stat_align' in r['kernel'] or 'copyBuffer' in r['kernel'] or 'Fill' in r['kernel'] or 'elementwise' in r['kernel']: print(r['kernel'][:60], r['calls'], r['avg_us'])
PY
