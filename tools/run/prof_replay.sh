# kernel-trace summary of the shipped hipGraph step: bash tools/run/prof_replay.sh <tag> [extra bench flags]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
TAG=${1:-r2}
shift
rm -rf $O/prof_$TAG
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o g -- python bench.py --timed-only --steps 64 --no-cpu-baseline "$@" > $O/${TAG}_timed_only.json 2> $O/${TAG}_prof.err
DB=$(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/${TAG}_graph_replay_kernel_stats.csv 250 > /dev/null
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/${TAG}_graph_replay_kernel_stats_by_grid.csv 250 --split-all > /dev/null
rm -rf $O/prof_$TAG
tail -1 $O/${TAG}_timed_only.json | cut -c1-200
