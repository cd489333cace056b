#!/bin/bash
# rocprofv3 PMC passes (two counter sets, counters only - no trace domains) over one python tool, summed per kernel whose
# name contains SUBSTR.    usage (on the GPU box): tools/pmc_run.sh <tag> <SUBSTR> tools/ffn_stamps.py f16
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; sub=$2; shift 2
out=$ROOT/gpurun_out/${SFM_ROUND:-r03}/pmc_$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/p$i -o p$i --output-format csv -- python3 $ROOT/"$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 $ROOT/tools/pmc_any.py "$sub" $out/summary.json $(find $out -name "*counter_collection.csv")
