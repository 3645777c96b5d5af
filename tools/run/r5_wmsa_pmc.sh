#!/bin/bash
# SQ counters of the bf16 attention kernels at config 5's stage-0 shape (tools/bench_wmsa.py): gpurun_out/<tag>_wmsa_pmc_sq*.json
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=${1:-r5}
export PMC_MATCH='(wmsa\w*_kernel<[^>]*>|wmsa\w*_kernel)'
run() {
  N=$1; shift
  rm -rf $O/pmc_$N
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$N -o c -- python tools/bench_wmsa.py --shift --reps 2 > /dev/null 2> $O/pmc_$N.err
  CC=$(ls $O/pmc_$N/*counter_collection.csv $O/pmc_$N/*/*counter_collection.csv 2>/dev/null | head -1)
  KT=$(ls $O/pmc_$N/*kernel_trace.csv $O/pmc_$N/*/*kernel_trace.csv 2>/dev/null | head -1)
  test -n "$CC" && python tools/pmc_conv.py --summarise "$CC" $O/${T}_wmsa_pmc_$N.json "$KT" > /dev/null
  rm -rf $O/pmc_$N
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
run sq2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run sq3 SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/${T}_wmsa_pmc_sq*.json")):
    d=json.load(open(f))
    for k,v in d.items():
        print(f.split('_')[-1], k[:60], {a: round(b,3) for a,b in v.items() if isinstance(b,(int,float))})
PY
