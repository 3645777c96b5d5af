# graph-replay kernel summary of the shipped step: bash tools/run/prof_replay.sh <tag> [bench args]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=$1
shift
rm -rf $O/prof_${T}g
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_${T}g -o g -- python bench.py --timed-only --steps 128 --no-cpu-baseline "$@" > $O/${T}_timed_only.json 2> $O/${T}_prof.err
DB=$(ls $O/prof_${T}g/*.db $O/prof_${T}g/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/${T}_graph_replay_kernel_stats.csv 250 > /dev/null
rm -rf $O/prof_${T}g
head -45 $O/${T}_graph_replay_kernel_stats.csv | cut -c1-150
tail -c 300 $O/${T}_timed_only.json
