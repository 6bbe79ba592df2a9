#!/bin/bash
# Run on the GPU box: overlap-discovery leg of bench.py under a list of env settings ("VAR=val" each), one line per setting.
export FMD_BENCH_SMEM=0 FMD_BENCH_KMER=0 FMD_BENCH_PROBE=0 FMD_BENCH_CPU_SAMPLE=20000 FMD_BENCH_CPU_SAMPLE_OVLP=20000
for s in "$@"; do
  echo "== $s"
  env $s timeout 300 python bench.py --steps 3 --warmup 1 2>/dev/null | python tools/bench_pick.py
done
