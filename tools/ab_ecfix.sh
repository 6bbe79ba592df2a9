#!/bin/bash
# same-box A/B of builds of the library over the ec_fix leg (k-mer harvest for its table, then k_ecfix over 5*10^7 raw reads; PMC traffic measured in the run):
#   tools/ab_ecfix.sh libA.so libB.so ...
export FMD_BENCH_CPU_SAMPLE=20000 FMD_BENCH_CPU_SAMPLE_KMER=256 FMD_BENCH_PROBE=1
mkdir -p gpurun_out/ab_ecfix
for lib in "$@"; do
  FMD_HIP_LIB=$PWD/fermi_amd/lib/$lib FMD_BENCH_LEGS=kmer,ecfix timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/ab_ecfix/$lib.json 2> gpurun_out/ab_ecfix/$lib.err
  python - $lib gpurun_out/ab_ecfix/$lib.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    e = d["ec_fix"]; r = e["roofline"]
    print("%-26s ec_fix %7.1f ms  traffic %s GB (%s)  requested %.1f GB  parity %s  overflowed %s" % (sys.argv[1], e["ms_per_step"], ("%.1f" % (r["traffic"] / 1e9)) if r.get("traffic") else "-",
          (r.get("traffic_source") or "")[:24], (r["achieved_requested"] or 0) * r["kernel_ms"] * 1e-3, e["parity_vs_cpu_on_sample"][:24], e["reads_whose_trace_overflowed"]))
except Exception as ex:
    print(sys.argv[1], "failed:", ex)
PY
done
