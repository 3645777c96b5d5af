export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python bench.py > $O/r1p_bench.json 2> $O/r1p_bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_r1p -o r1p -- python bench.py > $O/r1p_bench_profiled_run.json 2> $O/r1p_prof.err
DB=$(ls $O/prof_r1p/*.db $O/prof_r1p/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/r1p_bench_kernel_stats.csv > /dev/null
rm -rf $O/prof_r1p
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_r1pg -o g -- python bench.py --timed-only --steps 128 > $O/r1p_timed_only.json 2>> $O/r1p_prof.err
DB=$(ls $O/prof_r1pg/*.db $O/prof_r1pg/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/r1p_graph_replay_kernel_stats.csv 250 > /dev/null
rm -rf $O/prof_r1pg
timeout 300 python bench.py --sequential > $O/r1p_bench_sequential.json 2> /dev/null
timeout 300 python bench.py --optimizer sgd_all > $O/r1p_bench_sgd_all.json 2> /dev/null
timeout 400 python bench.py --arch swin > $O/r1p_bench_swin.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_r1pf -o f -- python tools/bench_frames.py --iters 5 > $O/r1p_frames.json 2>> $O/r1p_prof.err
DB=$(ls $O/prof_r1pf/*.db $O/prof_r1pf/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/r1p_frames_kernel_stats.csv --split-all > /dev/null
rm -rf $O/prof_r1pf
timeout 300 python tools/bench_frames.py --clips 16 --chain 10 --iters 10 > $O/r1p_frames_16clips.json 2>/dev/null
for f in r1p_bench r1p_bench_profiled_run r1p_bench_sequential r1p_bench_sgd_all r1p_bench_swin; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print("$f", round(d["value"],2), round(d["ms_per_step"],3), d.get("adapt_only_ms"), round(d["roofline"]["frac"],3), round(d["roofline"]["avg_ms"]*1e3,1))
except Exception as e: print("$f", "ERR", e)
PY
done
grep -i "moments_nchw" $O/r1p_bench_kernel_stats.csv | head -3
grep -i "frames" $O/r1p_frames_kernel_stats.csv | head -3
