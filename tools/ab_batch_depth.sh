#!/bin/bash
# one box: the sorted job at several batch sizes and prefix-table depths, a fresh process each (tables are built when the index is opened)
out=${1:-gpurun_out/r6_batch}; mkdir -p $out
export FMD_PAIR=1 AB_DIGEST=1
for b in 10000000 20000000 34000000 50000000 100000000; do
  echo "== batch $b" ; FMD_BENCH_OVLP_BATCH=$b timeout 600 python tools/ab_overlap.py 50000000 0.0 4 -- - 2>&1 | tail -1
done
for d in 14 15; do
  echo "== depth $d" ; FMD_PTAB_DEPTH=$d timeout 600 python tools/ab_overlap.py 50000000 0.0 4 -- - - 2>&1 | tail -2
done
