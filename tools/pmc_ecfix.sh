#!/bin/bash
# SQ counters of k_ecfix (one rocprofv3 --pmc pass per library given): where do its waves spend their cycles?   tools/pmc_ecfix.sh libA.so [libB.so ...]
export FMD_BENCH_CPU_SAMPLE=20000 FMD_BENCH_CPU_SAMPLE_KMER=256 FMD_BENCH_PROBE=0 FMD_BENCH_PMC=0 FMD_BENCH_LEGS=kmer,ecfix TMPDIR=/tmp
OUT=gpurun_out/r6_ecfix/pmc; mkdir -p $OUT
for lib in "$@"; do
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    rm -rf $OUT/$lib.$tag
    FMD_HIP_LIB=$PWD/fermi_amd/lib/$lib timeout 900 rocprofv3 --pmc $pass --output-format csv -d $OUT/$lib.$tag -o t -- python bench.py --steps 1 --warmup 0 > /dev/null 2> $OUT/$lib.$tag.err
    f=$(find $OUT/$lib.$tag -name "*counter_collection.csv" | head -1)
    python - "$lib" "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc, n = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(sys.argv[2])):
    if r["Kernel_Name"].startswith("k_ecfix"):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print("%-24s %-22s %.4g per launch (%d launches)" % (sys.argv[1], k, acc[k] / max(1, n[k]), n[k]))
PY
    find $OUT/$lib.$tag -name "*.csv" -size +1M -delete
  done
done
