# SQ counters of the Video Swin-B kernels (dense gemm_nt_kernel, window attention) over two sequential eager steps:
#   bash tools/run/pmc_swin.sh <tag> [bench_swin options]      (three --pmc passes; summaries in gpurun_out/<tag>_swin_pmc_sq*.json)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=$1
shift
export PMC_MATCH='(gemm_nt\w*_kernel<[^>]*>|wmsa\w*_kernel<[^>]*>|wmsa\w*_kernel|ln_\w+_kernel<[^>]*>)'
run() {  # name, counters...
  N=$1; shift
  rm -rf $O/pmc_$N
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$N -o c -- python tools/bench_swin.py --no-graph --sequential --steps 2 --warmup 1 "${OPTS[@]}" > /dev/null 2> $O/pmc_$N.err
  CC=$(ls $O/pmc_$N/*counter_collection.csv $O/pmc_$N/*/*counter_collection.csv 2>/dev/null | head -1)
  KT=$(ls $O/pmc_$N/*kernel_trace.csv $O/pmc_$N/*/*kernel_trace.csv 2>/dev/null | head -1)
  test -n "$CC" && python tools/pmc_conv.py --summarise "$CC" $O/${T}_swin_pmc_$N.json "$KT" > /dev/null
  rm -rf $O/pmc_$N
}
OPTS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
run sq2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/${T}_swin_pmc_sq*.json")):
    d=json.load(open(f))
    for k,v in d.items():
        print(f.split('_')[-1], k, {a: round(b,3) for a,b in v.items() if isinstance(b,(int,float)) and ('/' in a or 'share' in a or 'fraction' in a or a in ('launches','duration_us'))})
PY
