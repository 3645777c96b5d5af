export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_trunk.py -x -q -k "fused_head" 2>&1 | tail -8
for v in 0 1; do
  VITTA_FUSED_HEAD=$v timeout 200 python bench.py --no-swin --no-sgd-all --no-cpu-baseline --no-streaming 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('fused_head=$v', round(d['value'],2), round(d['ms_per_step'],3), d['adapt_only_ms'])"
done
