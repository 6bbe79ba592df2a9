#!/usr/bin/env python3
"""Sums a counter per kernel over the two passes of tools/valu_table.sh and prints them side by side.  Usage: python tools/valu_table.py <dir with now/ and before/>"""
import csv, glob, os, sys
from collections import defaultdict


def read(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0, 0.0])
    if not f:
        return acc
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_ovl" in k and r.get("Counter_Name", "SQ_INSTS_VALU") == "SQ_INSTS_VALU":
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


root = sys.argv[1]
now, before = read(os.path.join(root, "now")), read(os.path.join(root, "before"))
nei = lambda k: k.startswith(("k_ovl_nei", "k_ovl_fix"))
print("SQ_INSTS_VALU (rocprofv3 --pmc, summed over the launches of ONE step of overlap discovery on 5*10^7 raw reads, 10^8 strands)")
print("before = FMD_NEI_LANE=0 FMD_GRP_DOWN=0 FMD_GRP_QUIET=0 (the get_nei kernels as round 4 ran them); now = the tree")
print("%-44s %8s %14s   %8s %14s" % ("kernel", "launches", "before", "launches", "now"))
for k in sorted(set(now) | set(before)):
    print("%-44s %8d %14.4g   %8d %14.4g" % (k[:44], before[k][0], before[k][1], now[k][0], now[k][1]))
tb, tn = sum(v[1] for k, v in before.items() if nei(k)), sum(v[1] for k, v in now.items() if nei(k))
print("%-44s %8s %14.4g   %8s %14.4g   %+.1f %%" % ("all get_nei kernels (k_ovl_nei*, k_ovl_fix)", "", tb, "", tn, 100.0 * (tn - tb) / tb if tb else 0.0))
wb, wn = sum(v[1] for k, v in before.items() if not nei(k)), sum(v[1] for k, v in now.items() if not nei(k))
print("%-44s %8s %14.4g   %8s %14.4g" % ("the other kernels of the step", "", wb, "", wn))
