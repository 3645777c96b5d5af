export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp vitta_amd/csrc/libvitta_hip.so /tmp/keep.so
for V in "$@"; do
  cp variants/$V.so vitta_amd/csrc/libvitta_hip.so
  echo "== $V"; timeout 200 python tools/bench_wmsa.py --shift 2>&1 | tail -3
done
cp /tmp/keep.so vitta_amd/csrc/libvitta_hip.so
