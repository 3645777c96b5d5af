# bash tools/debug/variant_ab.sh A B ...  : swaps variants/<name>.so in as the library and runs the convolution bench + the step bench
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp vitta_amd/csrc/libvitta_hip.so /tmp/keep.so
for V in "$@"; do
  cp variants/$V.so vitta_amd/csrc/libvitta_hip.so
  echo "== $V"
  python tools/bench_conv.py --frames 16 --arith b3 --no-vendor --reps 40 --out gpurun_out/v_$V.json 2>/dev/null | tail -1 | cut -c60-200
  python bench.py --no-swin --no-cpu-baseline --no-sgd-all --no-streaming 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], d['adapt_only_ms'], r['frac_of_fp32_matrix_peak'], r['kernel_ms_per_step'])"
done
cp /tmp/keep.so vitta_amd/csrc/libvitta_hip.so
