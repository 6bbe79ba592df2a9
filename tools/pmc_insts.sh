#!/bin/bash
# instruction counts + times of the walk kernels of the sorted job (one PMC pass + one kernel trace); run on the GPU box
# usage: tools/pmc_insts.sh TAG [n_reads] [batch] [ENV=VAL ...]
TAG=${1:-x}; N=${2:-50000000}; B=${3:-20000000}; shift 3
OUT=gpurun_out/insts_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc -o t -- python tools/ab_sorted.py $N ${ERR:-0.0} $B 1 "$@" > $OUT/pmc.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/ab_sorted.py $N ${ERR:-0.0} $B 1 "$@" > $OUT/trace.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
python - $OUT <<'PY' | tee $OUT/TABLE.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
def short(n): return n.replace("void ", "").split("(")[0][:44]
f = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
t = {}
if f:
    for r in csv.DictReader(open(f[0])):
        if "k_ovl" in r["Name"]: t[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
acc = defaultdict(lambda: defaultdict(list))
f = glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if "k_ovl" in r["Kernel_Name"]: acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-46s %5s %9s %11s %11s %11s %11s" % ("kernel", "calls", "avg ms", "VALU", "SALU", "LDS", "VMEM_RD"))
for k in sorted(acc, key=lambda k: -t.get(k, (0, 0))[0] * t.get(k, (0, 0))[1]):
    a = acc[k]
    m = lambda c: sum(a[c]) / len(a[c]) if a.get(c) else 0
    if t.get(k, (0, 0))[1] > 0.3:
        print("%-46s %5d %9.3f %11.4g %11.4g %11.4g %11.4g" % (k, t[k][0], t[k][1], m("SQ_INSTS_VALU"), m("SQ_INSTS_SALU"), m("SQ_INSTS_LDS"), m("SQ_INSTS_VMEM_RD")))
PY
grep -E "index:|sorted job|id order|SAME|DIFF" $OUT/trace.txt | tee -a $OUT/TABLE.txt
