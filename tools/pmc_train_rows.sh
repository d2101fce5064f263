#!/bin/bash
# GPU box: SQ issue counters of the training-step kernels at 1020 and 2560 rows.  bash tools/pmc_train_rows.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/train_pmc_${1:-a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in 102 256; do
  QIDDM_TRAIN_BATCH=$b timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/b$b -- python3 $ROOT/tools/profile_train.py adjoint > $OUT/b$b.log 2>&1 || exit 1
  python3 $ROOT/tools/pmc_reduce.py $OUT/b$b > $OUT/b$b.json
done
QIDDM_TRAIN_BATCH=102 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/c102 -- python3 $ROOT/tools/profile_train.py adjoint > $OUT/c102.log 2>&1
python3 $ROOT/tools/pmc_reduce.py $OUT/c102 > $OUT/c102.json
