export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/prof_s
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_s -o g -- python bench.py --timed-only --steps 64 --no-cpu-baseline --optimizer sgd_all > $O/r5_timed_only_sgd.json 2>> $O/r5_prof.err
DB=$(ls $O/prof_s/*.db $O/prof_s/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/r5_sgd_all_graph_replay_kernel_stats.csv 250 > /dev/null
test -n "$DB" && timeout 120 python tools/timeline.py "$DB" $O/r5_sgd_timeline.csv 40 20
rm -rf $O/prof_s
head -24 $O/r5_sgd_all_graph_replay_kernel_stats.csv | cut -c1-120
