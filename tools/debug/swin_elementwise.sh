export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
FL="--views 4 --frames 32 --window-depth 16 --wmsa-bf16 --dense-bf16"
rm -rf /tmp/prof_ew
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_ew -o p -- python tools/bench_swin.py $FL --steps 12 > /dev/null 2>&1
DB=$(ls /tmp/prof_ew/*.db /tmp/prof_ew/*/*.db 2>/dev/null | head -1)
python tools/debug/elementwise_breakdown.py $DB 300 | cut -c1-330
