#!/bin/bash
# usage: pmc3.sh <tag> : SVGF-oriented PMC passes (separate passes; --pmc only with --kernel-trace)
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$TAG
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_$TAG/p$i.log 2>&1
done
python3 $R/scratch/pmc_summary.py $R/gpurun_out/pmc_$TAG > $R/gpurun_out/pmc_$TAG/summary.txt 2>&1
grep -A40 "atrous_packed_kernel<2" $R/gpurun_out/pmc_$TAG/summary.txt | head -45
