#!/bin/bash
# Run on the GPU box: the sorted overlap job over all strands of 5*10^7 error-free reads with the radix sort of the parked strands looking at key bits b .. 31 only
# (FMD_PARK_SORT_FROM=b), HIP-event ms per pass.  Usage: tools/ab_sortbits.sh <tag> [n_reads] [err]
TAG=${1:-sortbits}; N=${2:-50000000}; E=${3:-0.0}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for b in 0 8 12 16 20 24 0; do
  FMD_PARK_SORT_FROM=$b timeout 600 python tools/ab_grp4.py $N $E 3 1 FMD_GRP4 2>&1 | grep "ms per pass" | sed "s/^/FMD_PARK_SORT_FROM=$b  /" | tee -a $OUT/ab_sortbits_${N}_${E}.txt
done
