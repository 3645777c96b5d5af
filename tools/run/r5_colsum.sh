#!/bin/bash
# batched column sums: kernel + Swin suites, then A/B of VITTA_DEFER_COLSUMS on configs 3 and 5 (same box)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/colsum; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tta.py tests/test_gpu_fullsize.py -m gpu -x -q -k "column_sums or layernorm or swin or Swin" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
C5="--views 4 --frames 32 --window-depth 16 --wmsa-bf16 --dense-bf16"
C3="--views 2 --frames 16 --window-depth 8"
for rep in 1 2; do
for d in 1 0; do
  VITTA_DEFER_COLSUMS=$d timeout 600 python tools/bench_swin.py $C5 --steps 8 > $O/c5_defer${d}_$rep.json 2> $O/c5_defer${d}_$rep.err
  VITTA_DEFER_COLSUMS=$d timeout 600 python tools/bench_swin.py $C3 --steps 10 > $O/c3_defer${d}_$rep.json 2> $O/c3_defer${d}_$rep.err
done; done
for f in $O/c*_defer*.json; do echo $f; tail -1 $f | cut -c1-120; done
