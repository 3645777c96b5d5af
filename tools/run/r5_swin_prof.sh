# Video Swin-B config-5 shape with the bf16 recipe: timing + rocprofv3 kernel trace of the graph-replay tail.  bash tools/run/r4_swin_prof.sh [tag]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=${1:-r5_swin_c5}
FL="--views 4 --frames 32 --window-depth 16 --wmsa-bf16 --dense-bf16"
timeout 600 python tools/bench_swin.py $FL --steps 8 > $O/${T}.json 2> $O/${T}.err; tail -1 $O/${T}.json
rm -rf $O/prof_$T
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p -- python tools/bench_swin.py $FL --steps 12 > $O/${T}_profiled_run.json 2>> $O/${T}.err
DB=$(ls $O/prof_$T/*.db $O/prof_$T/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/${T}_graph_replay_kernel_stats.csv 300 > /dev/null
rm -rf $O/prof_$T
head -25 $O/${T}_graph_replay_kernel_stats.csv
