#!/bin/bash
# the get_nei kernels of the sorted job on reads with errors, with groups of 4 (FMD_GRP4=1, the default) and without (=0):
# per kernel the time (kernel trace) and VALU instructions / active cycles / thread-cycles (PMC passes of their own).  Run on the GPU box.
# usage: tools/pmc_grp4.sh TAG [n_reads=50000000]
TAG=${1:-x}; N=${2:-50000000}
OUT=gpurun_out/grp4_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for g in 0 1; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$g -o t -- python tools/ab_grp4.py $N 0.01 1 $g > $OUT/trace$g.txt 2>&1
  find $OUT/trace$g -name "*kernel_trace.csv" -delete
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc$g -o t -- python tools/ab_grp4.py $N 0.01 1 $g > $OUT/pmc$g.txt 2>&1
  timeout 600 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU --output-format csv -d $OUT/pmcb$g -o t -- python tools/ab_grp4.py $N 0.01 1 $g > $OUT/pmcb$g.txt 2>&1
done
python - $OUT <<'PY' | tee $OUT/TABLE.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
def short(n): return n.replace("void ", "").split("(")[0][:40]
for g in "01":
    t = {}
    for f in glob.glob(os.path.join(out, "trace" + g, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_ovl" in r["Name"]: t[short(r["Name"])] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    acc = defaultdict(lambda: defaultdict(float))
    for d in ("pmc", "pmcb"):
        for f in glob.glob(os.path.join(out, d + g, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_ovl" in r["Kernel_Name"]: acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    print("FMD_GRP4=%s  (sums over the 2 passes of the run: 10 batches of 2*10^7 strands)" % g)
    print("%-42s %5s %9s %11s %11s %11s %11s %9s" % ("kernel", "calls", "total ms", "INSTS_VALU", "ACTIVE_VALU", "THREAD_CYC", "INST_CYC", "thr/inst"))
    tot = 0.0
    for k in sorted(t, key=lambda k: -t[k][1]):
        if "nei" not in k and "classify" not in k: continue
        a = acc.get(k, {})
        tc, ic = a.get("SQ_THREAD_CYCLES_VALU", 0), a.get("SQ_INST_CYCLES_VALU", 0) or a.get("SQ_ACTIVE_INST_VALU", 0)
        tot += t[k][1]
        if t[k][1] > 0.5:
            print("%-42s %5d %9.2f %11.4g %11.4g %11.4g %11.4g %9s" % (k, t[k][0], t[k][1], a.get("SQ_INSTS_VALU", 0), a.get("SQ_ACTIVE_INST_VALU", 0), tc, a.get("SQ_INST_CYCLES_VALU", 0), "%.1f" % (tc / ic) if ic and tc else "-"))
    print("get_nei kernels together: %.1f ms" % tot)
    for l in open(os.path.join(out, "trace%s.txt" % g)):
        if "ms per pass" in l: print(l.strip())
PY
