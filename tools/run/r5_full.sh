#!/bin/bash
# full GPU suite + smoke, logs under gpurun_out/full
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/full/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/full/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/full/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/full/smoke.log
tail -5 gpurun_out/full/pytest.log; tail -3 gpurun_out/full/smoke.log
