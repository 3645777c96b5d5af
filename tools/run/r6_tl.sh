export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=${1:-r6a}
rm -rf $O/prof_tl
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_tl -o g -- python bench.py --timed-only --steps 64 --no-cpu-baseline > $O/${T}_timed_only.json 2>> $O/${T}_prof.err
DB=$(ls $O/prof_tl/*.db $O/prof_tl/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/timeline.py "$DB" $O/${T}_timeline.csv 40 14 > /dev/null
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/${T}_graph_replay_kernel_stats.csv 250 > /dev/null
rm -rf $O/prof_tl
head -40 $O/${T}_graph_replay_kernel_stats.csv | cut -c1-160
