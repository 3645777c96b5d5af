# Every judged artefact of the round in one gpurun call: bash tools/run/round_artifacts.sh [tag]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=${1:-r2k}
timeout 400 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p -- python bench.py --no-sgd-all > $O/${T}_bench_profiled_run.json 2> $O/${T}_prof.err
DB=$(ls $O/prof_$T/*.db $O/prof_$T/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/${T}_bench_kernel_stats.csv > /dev/null
rm -rf $O/prof_$T
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_${T}g -o g -- python bench.py --timed-only --steps 128 --no-cpu-baseline > $O/${T}_timed_only.json 2>> $O/${T}_prof.err
DB=$(ls $O/prof_${T}g/*.db $O/prof_${T}g/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/${T}_graph_replay_kernel_stats.csv 250 > /dev/null
rm -rf $O/prof_${T}g
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_${T}s -o g -- python bench.py --timed-only --steps 128 --no-cpu-baseline --sequential > $O/${T}_timed_only_seq.json 2>> $O/${T}_prof.err
DB=$(ls $O/prof_${T}s/*.db $O/prof_${T}s/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/${T}_sequential_graph_replay_kernel_stats.csv 250 > /dev/null
rm -rf $O/prof_${T}s
timeout 300 python bench.py --sequential --no-cpu-baseline --no-sgd-all > $O/${T}_bench_sequential.json 2> /dev/null
timeout 300 python bench.py --optimizer sgd_all --no-cpu-baseline > $O/${T}_bench_sgd_all.json 2> /dev/null
timeout 300 python bench.py --force-exchanges --no-cpu-baseline --no-sgd-all > $O/${T}_bench_rccl_one_rank.json 2> /dev/null
timeout 400 python bench.py --arch swin --no-cpu-baseline > $O/${T}_bench_swin.json 2> /dev/null
timeout 300 python tools/bench_swin.py --library-dense > $O/${T}_swin_library_dense.json 2> /dev/null
timeout 300 python tools/bench_swin.py > $O/${T}_swin.json 2> /dev/null
timeout 300 python tools/bench_swin.py --wmsa-bf16 > $O/${T}_swin_bf16_wmsa.json 2> /dev/null
timeout 300 python tools/bench_gemm.py --out $O/${T}_gemm_bench.json > /dev/null 2>&1
timeout 200 python tools/bench_conv.py --frames 16 --out $O/${T}_conv_bench_16frames.json > /dev/null 2>&1
timeout 200 python tools/bench_conv.py --frames 8 --no-vendor --out $O/${T}_conv_bench_8frames.json > /dev/null 2>&1
timeout 200 python tools/debug/wgrad_probe.py > $O/${T}_wgrad_bench_16frames.txt 2>&1
for f in bench bench_profiled_run bench_sequential bench_sgd_all bench_rccl_one_rank bench_swin; do python - <<PY
import json
try:
    d=json.loads(open("$O/${T}_$f.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$f", round(d["value"],2), round(d["ms_per_step"],3), d.get("adapt_only_ms"), r["kernel"][:24], round(r["frac"],3), (d.get("sgd_all") or {}).get("value"))
except Exception as e: print("$f", "ERR", e)
PY
done
grep -i "conv_sk\|stem_conv" $O/${T}_bench_kernel_stats.csv | head -5
