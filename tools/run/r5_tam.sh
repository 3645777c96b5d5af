export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 200 python -m pytest tests/test_gpu_kernels.py -x -q -k "tam" 2>&1 | tail -5
timeout 100 python tools/bench_tam.py --out $O/r5c_tam.json 2>&1 | tail -4
for v in 1 0; do
  if [ $v == 1 ]; then export VITTA_TAM_FAST_OFF=1; else unset VITTA_TAM_FAST_OFF; fi
  timeout 200 python bench.py --no-swin --no-sgd-all --no-cpu-baseline --no-streaming 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('fast_off=$v', round(d['value'],2), round(d['ms_per_step'],3), d['adapt_only_ms'])"
done
