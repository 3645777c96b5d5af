export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for W in 192 384 768; do for S in 2 4 8; do
echo "== MIN_WGS=$W MIN_STEPS=$S"
VITTA_CONV_B3_MIN_WGS=$W VITTA_CONV_B3_MIN_STEPS=$S timeout 200 python tools/bench_conv.py --frames 16 --arith b3 --no-vendor --out gpurun_out/x.json 2>&1 | grep -v amdgpu | python -c "
import sys,ast
tot=0
for l in sys.stdin:
    try: d=ast.literal_eval(l)
    except Exception: continue
    if 'fwd_us' in d and 'C' in d: print(d['name'][5:], int(d['fwd_us']*10)/10, int(d['dgrad_us']*10)/10, end=' | ')
    if 'fwd_ms' in d: print(); print('TOTAL fwd', round(d['fwd_ms'],3), 'dgrad', round(d['dgrad_ms'],3))
"
done; done
