#!/bin/bash
# full GPU suite + smoke + two default bench lines, logs under gpurun_out/full
bash tools/run/r5_full.sh
for i in 1 2; do timeout 900 python bench.py > gpurun_out/full/bench_$i.json 2> gpurun_out/full/bench_$i.err; done
python - <<'PY'
import json
for i in (1, 2):
    try:
        d = json.loads(open(f"gpurun_out/full/bench_{i}.json").read().strip().splitlines()[-1])
        print(i, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e:
        print(i, "failed", e)
PY
