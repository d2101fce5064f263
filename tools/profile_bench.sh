#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:
#   bash tools/profile_bench.sh <tag>
# Produces, under gpurun_out/prof_<tag>/ :
#   stats/   rocprofv3 --kernel-trace --stats of `python3 bench.py` (per-kernel time)
#   pmc_rd/  rocprofv3 --pmc FETCH_SIZE      (separate pass, as MI355X_MICROARCH.md prescribes)
#   pmc_wr/  rocprofv3 --pmc WRITE_SIZE      (separate pass)
# and copies the small CSV summaries to gpurun_out/prof_<tag>/summary/ for committing under profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT/summary
cd /tmp && export TMPDIR=/tmp
# steps / warmup are multiples of 75 (15 steps per launch x 5 launches per graph): every profiled launch of the
# dominant kernel is a 15-step launch, so the per-kernel average is comparable with bench.py's own timing
BENCH="python3 $ROOT/bench.py --steps 750 --warmup 75 --no-cpu-baseline --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
echo "stats rc=$?" >> $OUT/stats.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_rd -- $BENCH > $OUT/pmc_rd.log 2>&1
echo "pmc_rd rc=$?" >> $OUT/pmc_rd.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_wr -- $BENCH > $OUT/pmc_wr.log 2>&1
echo "pmc_wr rc=$?" >> $OUT/pmc_wr.log
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/summary/kernel_stats.csv \;
python3 $ROOT/tools/pmc_traffic.py $OUT > $OUT/summary/pmc_traffic.json 2> $OUT/summary/pmc_traffic.err
tail -3 $OUT/stats.log; head -12 $OUT/summary/kernel_stats.csv; cat $OUT/summary/pmc_traffic.json
