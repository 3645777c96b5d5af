#!/bin/bash
mkdir -p gpurun_out/hf
timeout 600 python -m pytest tests/test_gpu_tta.py -x -q -k "prefetcher or overlapped or online" > gpurun_out/hf/t.log 2>&1; echo "rc=$?" >> gpurun_out/hf/t.log
tail -4 gpurun_out/hf/t.log
timeout 300 python bench.py --no-swin --no-sgd-all --no-cpu-baseline --no-streaming 2> gpurun_out/hf/bench.err | tail -1 > gpurun_out/hf/bench.json
python -c "
import json; d=json.load(open('gpurun_out/hf/bench.json')); print(d['value'], d['ms_per_step'], d['host_fed'])"
