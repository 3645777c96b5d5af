# bash tools/debug/sgd_ab.sh A B ...: swaps variants/<name>.so in as the library and times the SGD-all step (twice each, interleaved)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp vitta_amd/csrc/libvitta_hip.so /tmp/keep.so
for R in 1 2; do
for V in "$@"; do
  cp variants/$V.so vitta_amd/csrc/libvitta_hip.so
  python bench.py --optimizer sgd_all --steps 30 --warmup 8 --no-cpu-baseline --no-swin --no-sgd-all --no-streaming 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', round(d['value'],2), round(d['ms_per_step'],3))"
done
done
cp /tmp/keep.so vitta_amd/csrc/libvitta_hip.so
