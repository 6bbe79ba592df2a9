#!/bin/bash
# same-box A/B of two builds of the library over the bench legs (reduced CPU samples): tools/ab_libs.sh libA.so libB.so [reps=2]
A=$1; B=$2; R=${3:-2}
export FMD_BENCH_ID_ORDER_AB=0 FMD_BENCH_HOST_API=0 FMD_BENCH_CPU_SAMPLE=50000 FMD_BENCH_CPU_SAMPLE_OVLP=20000 FMD_BENCH_CPU_SAMPLE_SMEM=20000 FMD_BENCH_CPU_SAMPLE_KMER=512 FMD_BENCH_CPU_SAMPLE_OVLP_RAW=20000 FMD_BENCH_PROBE=0
mkdir -p gpurun_out/ab_libs
for r in $(seq 1 $R); do for lib in $A $B; do
  FMD_HIP_LIB=$PWD/fermi_amd/lib/$lib timeout 700 python bench.py --steps 3 --warmup 1 > gpurun_out/ab_libs/$lib.$r.json 2>/dev/null
  python - $lib gpurun_out/ab_libs/$lib.$r.json <<PY
import json,sys
d=json.load(open(sys.argv[2]))
print("%-22s overlap %6.1f  raw %6.1f  bsearch %6.2f  smem %6.1f  kmer %6.1f  check_left %5.1f   %s %s %s %s" % (sys.argv[1], d["ms_per_step"], d["overlap_discovery_on_raw_reads"]["ms_with_the_fast_get_nei_path"], d["backward_search"]["ms_per_step"], d["smem"]["ms_per_step"], d["kmer_harvest"]["ms_per_step"], d["check_left"]["ms_per_step"], d["parity_vs_cpu_on_sample"], d["smem"]["parity_vs_cpu_on_sample"], d["backward_search"]["parity_vs_cpu_on_sample"], d["kmer_harvest"]["parity_vs_cpu_on_sample"]))
PY
done; done
