#!/bin/bash
# rocprofv3 PMC passes over tools/conv_bench.py (conv16p on the PerceptionAgent layer shapes), summary per kernel instantiation
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
out=$ROOT/gpurun_out/${SFM_ROUND:-r04}/pmc_conv16p
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/p$i -o p$i --output-format csv -- python3 $ROOT/tools/conv_bench.py > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 $ROOT/tools/pmc_kernels.py $out/summary.json "conv16p_kernel<F16, 7, 2, 1, true, false>;conv16p_kernel<F16, 3, 1, 1, false, false>;conv16p_kernel<F16, 7, 2, 1, true, true>;conv16p_kernel<F16, 7, 2, 2;conv16p_kernel<F16, 3, 1, 2;conv16p_kernel<F16, 5, 2, 2;conv16p_kernel<F16, 1, 2, 2" $(find $out -name "*counter_collection.csv") > /dev/null
python3 - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in d.items():
    cyc=v["GRBM_GUI_ACTIVE"]/8
    print(k)
    print("   kernel cycles %.0f  waves*cycles(quad) %.3g  WAIT_ANY %.1f%%  WAIT_INST_ANY %.1f%%  ACTIVE_INST_ANY %.1f%%  VALU active %.1f%%  mfma util %.1f%%" % (cyc, v["SQ_WAVE_CYCLES"], 100*v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"], 100*v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"], 100*v["SQ_ACTIVE_INST_ANY"]/v["SQ_WAVE_CYCLES"], 100*v["SQ_ACTIVE_INST_VALU"]/v["SQ_WAVE_CYCLES"], v["mfma_util_pct"]))
    print("   insts: VALU %.3g MFMA %.3g LDS %.3g SALU %.3g ; LDS conflict %.1f%% of LDS active; LDS active %.1f%% WAIT_INST_LDS %.1f%% of wave cycles; VALU per MFMA %.1f" % (v["SQ_INSTS_VALU"], v["SQ_INSTS_MFMA"], v["SQ_INSTS_LDS"], v["SQ_INSTS_SALU"], v.get("lds_conflict_pct",0), 100*v["SQ_ACTIVE_INST_LDS"]/v["SQ_WAVE_CYCLES"], 100*v["SQ_WAIT_INST_LDS"]/v["SQ_WAVE_CYCLES"], v["SQ_INSTS_VALU"]/v["SQ_INSTS_MFMA"]))
PY
