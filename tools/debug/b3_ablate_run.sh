export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp vitta_amd/csrc/libvitta_hip.so /tmp/keep.so
for V in BASE NOMFMA NOSPLIT NODMA NOMFMADB3_ABL_NOSPLIT NOSPLITDB3_ABL_NODMA; do
  cp variants/$V.so vitta_amd/csrc/libvitta_hip.so
  echo "== $V"
  timeout 200 python tools/bench_conv.py --frames 16 --arith b3 --no-vendor --out gpurun_out/x.json 2>&1 | grep -v amdgpu | python -c "
import sys,ast
for l in sys.stdin:
    try: d=ast.literal_eval(l)
    except Exception: continue
    if 'fwd_us' in d and 'C' in d and d['name'] in ('layer1.0.conv2','layer1.0.conv3','layer2.1.conv1','layer2.1.conv2','layer3.1.conv1','layer3.1.conv2','layer4.1.conv2'): print(d['name'][5:], round(d['fwd_us'],1), end=' | ')
    if 'fwd_ms' in d: print('TOTAL fwd', round(d['fwd_ms'],3), 'dgrad', round(d['dgrad_ms'],3))
"
done
cp /tmp/keep.so vitta_amd/csrc/libvitta_hip.so
