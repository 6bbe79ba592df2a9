#!/usr/bin/env python3
"""DESIGN.md section 0 from a bench line: python tools/fill_design.py profiles/r5_final/bench.json "192 passed in 14 min" "187 passed" (replaces the {{...}} fields; run once)."""
import json, sys
b = json.load(open(sys.argv[1]))
gpu_tests, cpu_tests = sys.argv[2], sys.argv[3]
def sci(x):
    e = 0
    while x >= 10: x /= 10; e += 1
    return "%.2f·10%s" % (x, "".join("⁰¹²³⁴⁵⁶⁷⁸⁹"[int(c)] for c in str(e)))
r = b["roofline"]; raw = b["overlap_discovery_on_raw_reads"]
def pm(d): return "%.0f GB per step = %.2f of 8 TB/s" % (d["roofline"]["traffic"] / 1e9, d["roofline"]["traffic_frac_of_peak"])
f = {
    "HEAD_MS": "%.1f" % b["ms_per_step"], "HEAD_RATE": sci(b["value"]), "HEAD_TRAFFIC": "%.1f" % (r["traffic"] / 1e9), "HEAD_FRAC": "%.2f" % r["traffic_frac_of_peak"],
    "HEAD_PROBE": "%.2f" % r["frac_of_random_gather_probe"]["traffic"], "HEAD_CPU": sci(b["cpu_baseline"]["value"]), "HEAD_RATIO": "%.0f" % b["speedup_vs_cpu_all_cores"],
    "RAW_MS": "%.1f" % raw["ms_with_the_fast_get_nei_path"], "RAW_RATE": sci(raw["reads_per_s"]), "RAW_FRAC": pm(raw),
    "CL_MS": "%.1f" % b["check_left"]["ms_per_step"],
    "BS_MS": "%.1f" % b["backward_search"]["ms_per_step"], "BS_RATE": sci(b["backward_search"]["value"]), "BS_FRAC": pm(b["backward_search"]), "BS_CPU": sci(b["backward_search"]["cpu_baseline"]["value"]),
    "SMEM_MS": "%.1f" % b["smem"]["ms_per_step"], "SMEM_RATE": sci(b["smem"]["value"]), "SMEM_FRAC": pm(b["smem"]), "SMEM_CPU": sci(b["smem"]["cpu_baseline"]["value"]),
    "KMER_MS": "%.1f" % b["kmer_harvest"]["ms_per_step"], "KMER_RATE": sci(b["kmer_harvest"]["value"]), "KMER_FRAC": pm(b["kmer_harvest"]), "KMER_CPU": sci(b["kmer_harvest"]["cpu_baseline"]["value"]),
    "EC_MS": "%.1f" % b["ec_fix"]["ms_per_step"], "EC_RATE": sci(b["ec_fix"]["value"]), "EC_FRAC": pm(b["ec_fix"]), "EC_CPU": sci(b["ec_fix"]["cpu_baseline"]["value"]),
    "GPU_TESTS": gpu_tests, "CPU_TESTS": cpu_tests,
}
s = open("DESIGN.md").read()
for k, v in f.items():
    s = s.replace("{{%s}}" % k, v)
assert "{{" not in s, [x for x in s.split("{{")[1:]][:3]
open("DESIGN.md", "w").write(s)
print({k: f[k] for k in ("HEAD_MS", "RAW_MS", "HEAD_TRAFFIC")})
