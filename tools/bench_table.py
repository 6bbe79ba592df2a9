#!/usr/bin/env python3
"""Markdown table of a bench.py JSON line (the rows DESIGN.md section 4 quotes).  Usage: python tools/bench_table.py bench.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def row(name, x, unit):
    r = x["roofline"]
    tr = "%.2f TB/s = **%.2f**" % (r["traffic_GBps"] / 1e3, r["traffic_frac_of_peak"]) if r.get("traffic") else "—"
    cb = x.get("cpu_baseline")
    cpu = "%.3g %s on %d threads (1 thread: %.3g)" % (cb["value"], cb["unit"], cb["cores"], cb["one_thread"]) if cb else "—"
    sp = "%.0f×" % x["speedup_vs_cpu_all_cores"] if "speedup_vs_cpu_all_cores" in x else "—"
    fr = "%.2f TB/s = %.2f" % (r["achieved_requested"] / 1e3, r["frac_requested"]) if r.get("frac_requested") is not None else "—"
    return "| %s | %.4g %s (%.1f ms) | %s | %s | %.1f TB/s | %s | %s |" % (name, x["value"], unit, r["kernel_ms"], fr, tr, r["algorithmic_equivalent_GBps"] / 1e3, cpu, sp)


print("| leg | GPU (HIP-event ms per step) | requested device bytes ÷ time (`roofline.frac_requested`) | HBM bytes from PMC ÷ time (`roofline.frac`) | SURVEY 8(d) algorithmic equivalent | reference CPU, same box | ratio |")
print("|---|---|---|---|---|---|---|")
n = d["config"]["reads"]
top = dict(d); top["value"] = d["value"]
print(row("overlap discovery, %d M reads, all %d M strands (headline)" % (n // 10**6, 2 * n // 10**6), d, "reads/s"))
cl = d.get("check_left")
if cl:
    print("| check_left over the same table (`fmd_ovlp_link_dev`: verdicts from `lfork`, row map, links) | %.3g strands/s (%.1f ms = %.3f of the discovery; exact kernel on every edge: %.0f ms) | streaming | — | — | — | — |"
          % (cl["value"], cl["roofline"]["kernel_ms"], cl["fraction_of_discovery_time"], cl["exact_kernel_on_every_edge_ms"]))
if d.get("backward_search"):
    print(row("backward search, 10 M reads (configs[1])", d["backward_search"], "reads/s"))
if d.get("smem"):
    print(row("SMEM (`fm6_smem`), %d M reads with 1 %% errors (configs[2])" % (n // 10**6), d["smem"], "reads/s"))
if d.get("kmer_harvest"):
    print(row("k-mer harvest of `correct`, same index (k = %d)" % d["kmer_harvest"]["k"], d["kmer_harvest"], "k-mers/s"))
