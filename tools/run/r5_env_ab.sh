# bash tools/run/r5_env_ab.sh VAR v1 v2 ... : bench.py under VAR=value, same box
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for V in "$@"; do
  env $VAR=$V timeout 200 python bench.py --no-swin --no-cpu-baseline --no-sgd-all --no-streaming 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$VAR=$V', round(d['value'],2), round(d['ms_per_step'],3), d['adapt_only_ms'], r['frac_of_fp32_matrix_peak'], r['kernel_ms_per_step'], r['launches_per_step'])"
done
