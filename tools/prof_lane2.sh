#!/bin/bash
# Run on the GPU box: the sorted job under the kernel trace with the lane-per-strand fast path at several batch thresholds (FMD_LANE_BATCH) and with the group form.
TAG=${1:-lane}; N=${2:-50000000}; E=${3:-0.0}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "1 8" "1 16" "1 24" "1 32" "1 48" "0 0"; do
  set -- $cfg
  FMD_LANE_BATCH=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_$1_$2 -o t -- python tools/ab_grp4.py $N $E 2 $1 FMD_NEI_LANE > $OUT/run_$1_$2.txt 2>&1
  f=$(find /tmp/pl_$1_$2 -name "*kernel_stats.csv" | head -1)
  python - "$f" "$1 $2" <<'PY' | tee -a $OUT/summary.txt
import csv, sys
tot = 0.0; n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if "k_ovl_nei_lane" in r["Name"] or "k_ovl_nei_fast" in r["Name"]:
        tot += float(r["TotalDurationNs"]) / 1e6
print("FMD_NEI_LANE, FMD_LANE_BATCH = %s: fast-path kernels %.1f ms over 3 passes" % (sys.argv[2], tot))
PY
  grep "ms per pass" $OUT/run_$1_$2.txt | tee -a $OUT/summary.txt
done
