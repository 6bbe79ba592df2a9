#!/bin/bash
# kernel trace + PMC passes (separate runs) of the sorted overlap job beside the id-order one (tools/ab_sorted.py); run on the GPU box
# usage: tools/profile_sorted.sh TAG [n_reads] [batch]
TAG=${1:-r3_sorted}; N=${2:-50000000}; B=${3:-20000000}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { sub=$1; shift; timeout 600 rocprofv3 "$@" --output-format csv -d $OUT/$sub -o t -- python tools/ab_sorted.py $N 0.0 $B 1 > $OUT/$sub.txt 2>&1; find $OUT/$sub -name "*kernel_trace.csv" -delete; }
run trace --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run pmc_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
python tools/summarize_sorted.py $OUT > $OUT/SUMMARY.md 2>&1
cat $OUT/SUMMARY.md
