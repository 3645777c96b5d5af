#!/bin/bash
mkdir -p gpurun_out/dtab
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "wmsa" > gpurun_out/dtab/k.log 2>&1; echo "rc=$?" >> gpurun_out/dtab/k.log; tail -4 gpurun_out/dtab/k.log
timeout 200 python tools/bench_wmsa.py --shift > gpurun_out/dtab/bench_frozen.txt 2>&1; cat gpurun_out/dtab/bench_frozen.txt | tail -3
timeout 200 python tools/bench_wmsa.py --shift --table-grad > gpurun_out/dtab/bench_tabgrad.txt 2>&1; cat gpurun_out/dtab/bench_tabgrad.txt | tail -3
