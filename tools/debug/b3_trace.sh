# bash tools/debug/b3_trace.sh   (run `VITTA_EXTRA_CFLAGS=-DB3_TRACE python -m vitta_amd.build; cp vitta_amd/csrc/libvitta_hip.so variants/TRACE.so;
# python -m vitta_amd.build` first, where hipcc is)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp vitta_amd/csrc/libvitta_hip.so /tmp/keep.so
cp variants/TRACE.so vitta_amd/csrc/libvitta_hip.so
python tools/debug/b3_trace.py 2>&1 | grep -v amdgpu.ids
cp /tmp/keep.so vitta_amd/csrc/libvitta_hip.so
