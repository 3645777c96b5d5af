# PMC passes over the shipped convolution kernels: bash tools/run/pmc_b3.sh <tag> <shape> [<shape> ...]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=$1
shift
run() {  # name, counters...
  N=$1; shift
  rm -rf $O/pmc_$N
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$N -o c -- python tools/pmc_conv.py "${SHAPES[@]}" > /dev/null 2> $O/pmc_$N.err
  CC=$(ls $O/pmc_$N/*counter_collection.csv $O/pmc_$N/*/*counter_collection.csv 2>/dev/null | head -1)
  KT=$(ls $O/pmc_$N/*kernel_trace.csv $O/pmc_$N/*/*kernel_trace.csv 2>/dev/null | head -1)
  test -n "$CC" && python tools/pmc_conv.py --summarise "$CC" $O/${T}_conv_pmc_$N.json "$KT" > /dev/null
  rm -rf $O/pmc_$N
}
SHAPES=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
run sq2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run sq3 SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/${T}_conv_pmc_sq*.json")):
    d=json.load(open(f))
    for k,v in d.items():
        print(f.split('_')[-1], k)
        print("   ", {a: (b if not isinstance(b, (int, float)) else round(b,3) if b<100 else int(b)) for a,b in v.items()})
PY
