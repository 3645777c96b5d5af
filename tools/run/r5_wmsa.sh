export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "wmsa_bf16" 2>&1 | tail -6
FL="--views 4 --frames 32 --window-depth 16 --wmsa-bf16 --dense-bf16"
for f in two one; do
  VITTA_WMSA_BF16_BWD=$f timeout 300 python tools/bench_swin.py $FL --steps 8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bwd=$f', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('value','ms_per_step','ms_per_video','videos_per_s')})"
done
