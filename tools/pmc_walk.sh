#!/bin/bash
# counters of the walk kernels of the sorted overlap job under environment settings (one rocprofv3 --pmc run per setting and counter group):
#   tools/pmc_walk.sh OUTDIR "ENV=VAL ..." ["ENV=VAL ..." ...]       ("-" = defaults)
OUT=$1; shift
mkdir -p $OUT; export TMPDIR=/tmp
i=0
for setting in "$@"; do
  i=$((i+1))
  for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
    tag=s$i.$(echo $pass | cut -d' ' -f1)
    rm -rf $OUT/$tag
    ( if [ "$setting" != "-" ]; then export $setting; fi; timeout 600 rocprofv3 --pmc $pass --output-format csv -d $OUT/$tag -o t -- python tools/ab_overlap.py ${N:-50000000} ${E:-0.0} 1 -- - > $OUT/$tag.txt 2>&1 )
    f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
    python - "$setting" "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc, n = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(sys.argv[2])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_ovl_walk") or k.startswith("k_ovl_nei_lane"):
        k = k.split(",")[0] if "lane" in k else k
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print("%-24s %-28s %-20s %.4g per launch (%d launches)" % (sys.argv[1], k[0][:28], k[1], acc[k] / n[k], n[k]))
PY
    find $OUT/$tag -name "*.csv" -size +1M -delete
  done
done
