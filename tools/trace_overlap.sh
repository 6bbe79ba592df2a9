#!/bin/bash
# Kernel timeline of overlap discovery under one or more settings (tools/ab_overlap.py syntax), rocprofv3 --kernel-trace.
# usage: tools/trace_overlap.sh OUTDIR N_READS ERR SETTING [SETTING...]
out=$(realpath -m "$1"); n=$2; err=$3; shift 3
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
    i=$((i + 1))
    d=/tmp/prof_ovl_$$_$i
    rm -rf $d
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $root/tools/ab_overlap.py $n $err 3 -- "$cfg" > $out/run_$i.txt 2>&1
    echo "== $cfg" >> $out/summary.txt
    grep "ms per pass" $out/run_$i.txt >> $out/summary.txt
    st=$(find $d -name "*kernel_stats.csv" | head -1)
    tr=$(find $d -name "*kernel_trace.csv" | head -1)
    [ -n "$st" ] && cp "$st" $out/kernel_stats_$i.csv
    [ -n "$tr" ] && python $root/tools/trace_tail.py "$tr" > $out/timeline_$i.txt
    rm -rf $d
done
cat $out/summary.txt
