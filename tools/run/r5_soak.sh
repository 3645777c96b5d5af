#!/bin/bash
# the GPU suite twice (no -x): looks for order- / atomics-dependent flakes before the driver's own run
mkdir -p gpurun_out/soak
for i in 1 2; do
  timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/soak/pytest_$i.log 2>&1
  echo "rc=$?" >> gpurun_out/soak/pytest_$i.log
  tail -3 gpurun_out/soak/pytest_$i.log
done
