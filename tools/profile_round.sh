#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:   bash tools/profile_round.sh <tag> [bench args]
# Produces gpurun_out/prof_<tag>/summary/ with
#   bench_line.json                      the JSON line of the profiled bench command (no profiler attached)
#   bench_kernel_stats.csv               rocprofv3 --kernel-trace --stats of the SAME bench command
#   bench_pmc_traffic.json               2*FETCH_SIZE + WRITE_SIZE per launch (separate --pmc passes) of that command
#   wide_kernel_stats.csv                --stats of tools/profile_wide.py (tiled n=16, wide adjoint, C4 GEMM, adjoint, engine)
#   wide_pmc_traffic.json                FETCH/WRITE per launch of the same
#   wide_pmc_l2.csv / *_sq.csv           TCC hit/miss and SQ issue counters (own passes)
set -u
TAG=${1:-r02}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
S=$OUT/summary
mkdir -p $S
cd /tmp && export TMPDIR=/tmp
BARGS=${*:---steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-train}
BENCH="python3 $ROOT/bench.py $BARGS"
WIDE="python3 $ROOT/tools/profile_wide.py"
run() { # name, then the rocprofv3 args..., then -- cmd
  local name=$1; shift
  timeout -k 10 400 rocprofv3 "$@" > $OUT/$name.log 2>&1
  echo "$name rc=$?" | tee -a $OUT/rc.log
}
$BENCH > $S/bench_line.json 2> $OUT/bench_plain.err; echo "bench rc=$?" | tee -a $OUT/rc.log
run bench_stats --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- $BENCH
run bench_rd --pmc FETCH_SIZE --output-format csv -d $OUT/bench/pmc_rd -- $BENCH
run bench_wr --pmc WRITE_SIZE --output-format csv -d $OUT/bench/pmc_wr -- $BENCH
run wide_stats --kernel-trace --stats --output-format csv -d $OUT/wide_stats -- $WIDE
run wide_rd --pmc FETCH_SIZE --output-format csv -d $OUT/wide/pmc_rd -- $WIDE
run wide_wr --pmc WRITE_SIZE --output-format csv -d $OUT/wide/pmc_wr -- $WIDE
run wide_l2 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/wide_l2 -- $WIDE
run unet_stats --kernel-trace --stats --output-format csv -d $OUT/unet_stats -- python3 $ROOT/tools/profile_unet_train.py 256 --no-table
run train_stats --kernel-trace --stats --output-format csv -d $OUT/train_stats -- python3 $ROOT/tools/profile_train.py adjoint
run train_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/train_sq -- python3 $ROOT/tools/profile_train.py adjoint
run qconv_bwd_stats --kernel-trace --stats --output-format csv -d $OUT/qconv_bwd_stats -- python3 $ROOT/tools/profile_qconv_bwd.py
python3 $ROOT/tools/stamp_qconv_train.py > $S/qconv_backward_phase_stamps.txt 2>> $OUT/rc.log
QIDDM_QCONV_VALU=1 python3 $ROOT/tools/stamp_qconv_train.py 2>> $OUT/rc.log | sed "s/per tile.*ticks;/VALU kernel:/" >> $S/qconv_backward_phase_stamps.txt
run wide_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/wide_sq -- $WIDE
find $OUT/bench_stats -name "*kernel_stats.csv" -exec cp {} $S/bench_kernel_stats.csv \;
find $OUT/wide_stats -name "*kernel_stats.csv" -exec cp {} $S/wide_kernel_stats.csv \;
find $OUT/unet_stats -name "*kernel_stats.csv" -exec cp {} $S/unet_simple_training_b256_kernel_stats.csv \;
find $OUT/train_stats -name "*kernel_stats.csv" -exec cp {} $S/train_step_adjoint_kernel_stats.csv \;
find $OUT/qconv_bwd_stats -name "*kernel_stats.csv" -exec cp {} $S/qconv_backward_c16_8_k3_kernel_stats.csv \;
grep -h "us/step" $OUT/train_stats.log > $S/train_step_adjoint.txt
python3 $ROOT/tools/pmc_reduce.py $OUT/train_sq > $S/train_step_pmc_sq.json 2>> $OUT/rc.log
grep -h "training step\|recorded in" $OUT/unet_stats.log > $S/unet_simple_training_b256.txt
python3 $ROOT/tools/pmc_traffic.py $OUT/bench > $S/bench_pmc_traffic.json 2>> $OUT/rc.log
python3 $ROOT/tools/pmc_traffic.py $OUT/wide > $S/wide_pmc_traffic.json 2>> $OUT/rc.log
python3 $ROOT/tools/pmc_reduce.py $OUT/wide_l2 > $S/wide_pmc_l2.json 2>> $OUT/rc.log
python3 $ROOT/tools/pmc_reduce.py $OUT/wide_sq > $S/wide_pmc_sq.json 2>> $OUT/rc.log
cat $OUT/rc.log; head -c 1500 $S/bench_line.json; echo; head -8 $S/bench_kernel_stats.csv; head -14 $S/wide_kernel_stats.csv
