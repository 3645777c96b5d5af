# Per-layer split-K sweep of conv_b3 (forward and data gradient of the 22 distinct trunk layers at 16 and 8 frames):
#   bash tools/debug/b3_ks_sweep.sh        -> gpurun_out/ks_<frames>_<ks>.json, then tools/debug/b3_ks_pick.py prints the table rows
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for F in 16 8; do for KS in 0 1 2 3 4 6 8 12 16; do
  VITTA_CONV_B3_FORCE_KS=$KS timeout 200 python tools/bench_conv.py --frames $F --arith b3 --no-vendor --reps 40 --out gpurun_out/ks_${F}_${KS}.json > /dev/null 2>&1
done; done
python tools/debug/b3_ks_pick.py
