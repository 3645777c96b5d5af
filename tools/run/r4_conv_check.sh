export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu > $O/r4b_tests_conv.txt 2>&1; tail -3 $O/r4b_tests_conv.txt
bash tools/debug/b3_trace.sh 2>&1 | grep -E "==|entry|  ->|main loop|of it|epilogue until|lifetime of" > $O/r4b_trace.txt; cat $O/r4b_trace.txt
timeout 300 python tools/bench_conv.py --frames 16 --arith b3 --no-vendor --out $O/r4b_conv16.json 2>&1 | grep -v amdgpu | tail -8
python tools/debug/conv_launch_probe.py 2>&1 | grep -v amdgpu
