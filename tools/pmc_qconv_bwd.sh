#!/bin/bash
# GPU box: counters of the thin-product backward kernel (one layer shape).  bash tools/pmc_qconv_bwd.sh <tag> [layer args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}; shift || true
OUT=$ROOT/gpurun_out/qconv_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/profile_qconv_bwd.py $*"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -- $CMD > $OUT/s$i.log 2>&1; echo "set $i rc=$?"
done
python3 $ROOT/tools/pmc_reduce.py $OUT > $OUT/summary.json
python3 - $OUT/summary.json <<'PY'
import json,sys
for r in json.load(open(sys.argv[1]))["per_launch_mean"]:
    if "backward" in r["kernel"] or "fold" in r["kernel"]:
        print(r["kernel"][:70]); print("   ", {k:round(v) for k,v in r.items() if k!="kernel"})
PY
