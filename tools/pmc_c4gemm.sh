#!/bin/bash
# GPU box: issue / matrix-core counters of the C4 eval GEMM (qconv_gemm_wide_kernel).  bash tools/pmc_c4gemm.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c4gemm_pmc_${1:-a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/profile_wide.py c4gemm"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -- $CMD > $OUT/s$i.log 2>&1; echo "set $i rc=$?"
done
python3 $ROOT/tools/pmc_reduce.py $OUT > $OUT/summary.json
python3 - $OUT/summary.json <<'PY'
import json,sys
for r in json.load(open(sys.argv[1]))["per_launch_mean"]:
    if "gemm" in r["kernel"]:
        print(r["kernel"][:70]); print("   ", {k:round(v) for k,v in r.items() if k!="kernel"})
PY
