#!/bin/bash
# Run on the GPU box: kernel trace of the overlap-discovery leg of bench.py alone; prints the kernel table.
OUT=gpurun_out/trace_ovl
mkdir -p $OUT
export TMPDIR=/tmp
export FMD_BENCH_SMEM=0 FMD_BENCH_KMER=0 FMD_BENCH_PROBE=0 FMD_BENCH_CPU_SAMPLE=20000 FMD_BENCH_CPU_SAMPLE_OVLP=20000
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 2 --warmup 1 > $OUT/bench_traced.json 2> $OUT/bench_traced.err
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_ovl/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    if "ovl" in r["Name"]:
        print("%-50s calls %3s avg %.3f ms" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
