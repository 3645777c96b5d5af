# bash tools/debug/sgd_ab_prof.sh A B ...: per-variant rocprofv3 kernel stats of the SGD-all step's graph-replay tail (top kernels)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp vitta_amd/csrc/libvitta_hip.so /tmp/keep.so
for V in "$@"; do
  cp variants/$V.so vitta_amd/csrc/libvitta_hip.so
  rm -rf /tmp/prof_$V
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$V -o p -- python bench.py --optimizer sgd_all --steps 128 --warmup 8 --timed-only > /dev/null 2>&1
  DB=$(ls /tmp/prof_$V/*.db /tmp/prof_$V/*/*.db 2>/dev/null | head -1)
  echo "== $V"
  python tools/prof_summary.py "$DB" gpurun_out/sgdab_$V.csv 250 > /dev/null
  head -16 gpurun_out/sgdab_$V.csv | cut -c1-50,90-200
done
cp /tmp/keep.so vitta_amd/csrc/libvitta_hip.so
