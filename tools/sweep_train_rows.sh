#!/bin/bash
# GPU box: per-kernel times of the recorded adjoint training step against the number of rows (batch x tau 10):
# the staircase shows how the one-wavefront-per-row kernel fills the 1024 SIMDs.   bash tools/sweep_train_rows.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/train_rows_${1:-sweep}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in 26 102 154 205 256 307 410 820; do
  QIDDM_TRAIN_BATCH=$b timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b$b -- python3 $ROOT/tools/profile_train.py adjoint > $OUT/b$b.log 2>&1 || exit 1
  f=$(find $OUT/b$b -name "*kernel_stats.csv" | head -1)
  echo "batch $b rows $((b*10)): $(grep -h 'us/step' $OUT/b$b.log)" | tee -a $OUT/summary.txt
  grep -E "train_|adam" $f | awk -F'","' '{n=$1; sub(/\(.*/,"",n); gsub(/"/,"",n); printf "    %-60s avg %8.1f us\n", n, $4/1000}' | tee -a $OUT/summary.txt
done
