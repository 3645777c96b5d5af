#!/bin/bash
mkdir -p gpurun_out/merge
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "tam_merged" > gpurun_out/merge/k.log 2>&1; echo "rc=$?" >> gpurun_out/merge/k.log
tail -3 gpurun_out/merge/k.log
timeout 200 python tools/bench_tam_merge.py --out gpurun_out/merge/tam_merge_${1:-a}.json 2>&1 | tail -5
