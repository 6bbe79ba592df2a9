#!/bin/bash
# per-kernel times of the sorted overlap job with and without the two-base head (separate rocprofv3 --kernel-trace --stats runs)
# usage: tools/prof_pair.sh [n_reads] [err] [extra env assignments for the pair run ...]
N=${1:-50000000}; E=${2:-0.0}; shift 2
OUT=gpurun_out/r6_pair/prof
mkdir -p $OUT
export TMPDIR=/tmp
for tag in nopair pair; do
  if [ $tag = nopair ]; then export FMD_PAIR_USE=0; else unset FMD_PAIR_USE; for kv in "$@"; do export "$kv"; done; fi
  rm -rf $OUT/$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o t -- python tools/ab_overlap.py $N $E 3 -- - > $OUT/$tag.txt 2>&1
  find $OUT/$tag -name "*kernel_trace.csv" -delete
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag: $(tail -1 $OUT/$tag.txt)"
  python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_ovl" in r["Name"] or "onesweep" in r["Name"] or "radix" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("  %-70s calls %5s  total %9.2f ms  avg %8.3f ms" % (r["Name"].replace("void ", "").split("(")[0][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
done
