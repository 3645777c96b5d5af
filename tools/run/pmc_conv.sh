# PMC passes over chosen convolution shapes: bash tools/run/pmc_conv.sh "<shape> <shape> ..." <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SH=${1:-"256,128,56,1,1,64,64 64,64,56,3,1,64,64 256,256,14,3,1,64,64 1024,256,14,1,1,64,64"}
TAG=${2:-conv}
mkdir -p $R/gpurun_out/pmc
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p1 -o conv -- python $R/tools/pmc_conv.py $SH > $R/gpurun_out/pmc/${TAG}_p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_F32 SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p2 -o conv -- python $R/tools/pmc_conv.py $SH > $R/gpurun_out/pmc/${TAG}_p2.log 2>&1
python $R/tools/pmc_conv.py --summarise $(find /tmp/p1 -name "*counter_collection.csv" | head -1) $R/gpurun_out/pmc/${TAG}_p1.json $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) > /dev/null
python $R/tools/pmc_conv.py --summarise $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $R/gpurun_out/pmc/${TAG}_p2.json $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) > /dev/null
