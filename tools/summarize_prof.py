"""Summarise a tools/profile_*.sh output directory into markdown (kernel stats + PMC per launch)."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None


def kernel_stats():
    f = find("trace", "*kernel_stats.csv")
    if not f:
        return "no kernel_stats.csv\n"
    rows = list(csv.DictReader(open(f)))
    s = "| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|\n"
    for r in rows[:12]:
        s += "| %s | %s | %.3f | %.3f | %s |\n" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"])
    return s


def pmc(sub, prefix="bench"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        return {}
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


print("# rocprofv3 summary (%s)\n" % out)
print("## kernel trace (--kernel-trace --stats), bench.py --steps 5 --warmup 1\n")
print(kernel_stats())
bj = os.path.join(out, "bench_traced.json")
if os.path.exists(bj) and os.path.getsize(bj):
    print("bench line under tracing:\n```\n%s```\n" % open(bj).read())
print("## PMC (separate passes), per launch of each kernel\n")
cal = None
pr = pmc("pmc_probe")
for k, v in pr.items():
    if "k_probe" in k and "FETCH_SIZE" in v:
        kb = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
        # probe_gather(iters=1) launches twice (warm-up + timed), each reading 2^27 lines of PROBE_LINE bytes
        known = (1 << 27) * int(os.environ.get("PROBE_LINE", "128"))
        cal = known / (kb * 1024.0)
        print("calibration: k_probe FETCH_SIZE = %.0f KB per launch for %d known bytes -> true/reported = %.3f\n" % (kb, known, cal))
for sub in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq", "pmc_inst"):
    a = pmc(sub)
    for k, v in a.items():
        if any(t in k for t in ("k_bsearch", "k_ovl", "k_retrieve", "k_smem", "k_kmer")):
            for c, vals in v.items():
                m = sum(vals) / len(vals)
                extra = ""
                if c == "FETCH_SIZE" and cal:
                    extra = " -> %.2f GB HBM read per launch after calibration (x%.3f)" % (m * 1024 * cal / 1e9, cal)
                print("- %s %s = %.4g per launch (n=%d)%s" % (k, c, m, len(vals), extra))
