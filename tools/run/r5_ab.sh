# bash tools/run/r5_ab.sh A B ... : bench.py with variants/<name>.so swapped in as the library, same box
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp vitta_amd/csrc/libvitta_hip.so /tmp/keep.so
for V in "$@"; do
  cp variants/$V.so vitta_amd/csrc/libvitta_hip.so
  timeout 200 python bench.py --no-swin --no-cpu-baseline --no-sgd-all --no-streaming 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$V', round(d['value'],2), round(d['ms_per_step'],3), d['adapt_only_ms'], r['frac_of_fp32_matrix_peak'], r['kernel_ms_per_step'])"
done
cp /tmp/keep.so vitta_amd/csrc/libvitta_hip.so
