export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
FL="--views 4 --frames 32 --window-depth 16 --wmsa-bf16 --dense-bf16"
rm -rf $O/prof_w1
VITTA_WMSA_BF16_BWD=${1:-one} timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_w1 -o p -- python tools/bench_swin.py $FL --steps 4 > /dev/null 2> $O/r5_wmsa_prof.err
DB=$(ls $O/prof_w1/*.db $O/prof_w1/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/r5_wmsa_${1:-one}_stats.csv 200 --split-all > /dev/null
rm -rf $O/prof_w1
grep -i wmsa $O/r5_wmsa_${1:-one}_stats.csv | head -20
