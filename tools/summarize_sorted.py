"""Summary of a tools/profile_sorted.sh directory: per-kernel time (kernel trace) and PMC sums per kernel name."""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None
def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:60]
f = find("trace", "*kernel_stats.csv")
print("# %s\n\n## kernel trace (one warm-up pass + one timed pass of each order)\n" % out)
if f:
    print("| kernel | calls | total ms | avg ms |\n|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        if any(t in r["Name"] for t in ("k_ovl", "radix", "onesweep")) and float(r["TotalDurationNs"]) > 2e5:
            print("| %s | %s | %.2f | %.3f |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
print("\n## PMC (separate passes), summed over all launches of a kernel / number of launches\n")
for sub in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        print("- %s: no counter file" % sub); continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "k_ovl" in r["Kernel_Name"]:
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        for c, v in sorted(acc[k].items()):
            if sum(v) > 0:
                print("- %s %s: %.4g per launch (n = %d)" % (k, c, sum(v) / len(v), len(v)))
