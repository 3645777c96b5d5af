# HBM traffic of the step's convolution launches: bash tools/run/pmc_conv_traffic.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --timed-only --no-graph --sequential --steps 6 --warmup 2 --no-cpu-baseline --no-sgd-all"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o f -- $CMD > $R/gpurun_out/pmc_conv_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o w -- $CMD > $R/gpurun_out/pmc_conv_w.log 2>&1
F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1)
python $R/tools/pmc_conv_traffic.py $F $W $R/gpurun_out/r2h_conv_traffic_pmc.json
