#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats, separate runs) of the sorted overlap job under builds of the library:  tools/prof_libs.sh OUTDIR libA.so libB.so ...
OUT=$1; shift
mkdir -p $OUT; export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf $OUT/$lib
  FMD_HIP_LIB=$PWD/fermi_amd/lib/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$lib -o t -- python tools/ab_overlap.py ${N:-50000000} ${E:-0.0} 3 -- - > $OUT/$lib.txt 2>&1
  find $OUT/$lib -name "*kernel_trace.csv" -delete
  f=$(find $OUT/$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib: $(grep 'ms per pass' $OUT/$lib.txt | tail -1)"
  python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_ovl" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:10]:
    print("  %-60s calls %5s  total %9.2f ms  avg %8.3f ms" % (r["Name"].replace("void ", "").split("(")[0][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
done
