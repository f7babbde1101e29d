#!/bin/bash
# usage (GPU box, via gpurun): scripts/pmc_run.sh <tag> [workload.py] [kernel-filter]
# Separate rocprofv3 --pmc passes (counters only: no trace domains) over a short workload; compact summary in gpurun_out/pmc_<tag>.txt
TAG=${1:-x}; WL=${2:-scripts/pmc_dual.py}; FLT=${3:-mlp_update}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
         "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
         "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA" \
         "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python3 $R/$WL > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 scripts/pmc_summary.py $OUT $FLT > gpurun_out/pmc_$TAG.txt 2>&1
cat gpurun_out/pmc_$TAG.txt | cut -c1-150
