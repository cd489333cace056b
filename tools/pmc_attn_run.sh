#!/bin/bash
# rocprofv3 PMC passes over the attention micro-benchmark (separate passes per counter set), summaries to gpurun_out/
# usage: tools/pmc_attn_run.sh <tag> <attn_bench args...>
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
out=$ROOT/gpurun_out/${SFM_ROUND:-r03}/pmc_$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/p$i -o p$i --output-format csv -- python3 $ROOT/tools/attn_bench.py "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 $ROOT/tools/pmc_attention.py $out/summary.json $(find $out -name "*counter_collection.csv") > /dev/null
cat $out/summary.json
