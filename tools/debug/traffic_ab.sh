export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
for M in 0 -1; do
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$C
  VITTA_CONV_B3_NFAST=$M timeout 400 rocprofv3 --pmc $C -d $O/pmc_$C -o t -- python bench.py --timed-only --no-graph --sequential --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>> $O/nf_prof.err
done
F=$(ls $O/pmc_FETCH_SIZE/*.db $O/pmc_FETCH_SIZE/*/*.db 2>/dev/null | head -1)
W=$(ls $O/pmc_WRITE_SIZE/*.db $O/pmc_WRITE_SIZE/*/*.db 2>/dev/null | head -1)
python tools/pmc_conv_traffic.py "$F" "$W" $O/nf_traffic_$M.json | grep -E "per_launch|over_algorithmic"
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
done
