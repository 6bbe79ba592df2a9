#!/bin/bash
# Run on the GPU box: kernel trace of the sorted job with the lane-per-strand fast path and with the group form (FMD_NEI_LANE=1 / 0), error-free reads.
# Usage: tools/prof_lane.sh <tag> [n_reads=50000000] [err=0.0]
TAG=${1:-lane}; N=${2:-50000000}; E=${3:-0.0}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in 1 0; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_$v -o t -- python tools/ab_grp4.py $N $E 2 $v FMD_NEI_LANE > $OUT/run_lane$v.txt 2>&1
  f=$(find /tmp/pl_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" > $OUT/kernels_lane$v.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_ovl" in n and float(r["TotalDurationNs"]) > 1e5:
        print("%-60s calls %4s  total %9.2f ms  avg %8.3f ms" % (n.split("(")[0].replace("void ", "")[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
  grep "ms per pass" $OUT/run_lane$v.txt; cat $OUT/kernels_lane$v.txt
done
