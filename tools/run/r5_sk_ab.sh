export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/sk; mkdir -p $O
C3="--views 2 --frames 16 --window-depth 8"
for rep in 1 2 3; do for g in 512 0; do
  VITTA_GEMM_SK=$g timeout 600 python tools/bench_swin.py $C3 --steps 10 2>>gpurun_out/sk/ab.err | tail -1 | cut -c1-90 | sed "s/^/sk=$g /"
done; done
VITTA_GEMM_SK=512 timeout 600 python tools/bench_swin.py $C3 --steps 10 --sequential 2>>gpurun_out/sk/ab.err | tail -1 | cut -c1-90 | sed "s/^/seq sk=512 /"
VITTA_GEMM_SK=0 timeout 600 python tools/bench_swin.py $C3 --steps 10 --sequential 2>>gpurun_out/sk/ab.err | tail -1 | cut -c1-90 | sed "s/^/seq sk=0 /"
