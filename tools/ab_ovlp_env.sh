#!/bin/bash
# A/B the overlap leg: lane-per-strand only (FMD_OVLP_SLOW_ONLY=1) vs group kernels (default)
export FMD_BENCH_CPU_SAMPLE=20000 FMD_BENCH_CPU_SAMPLE_OVLP=${FMD_BENCH_CPU_SAMPLE_OVLP:-100000}
for mode in grp slow; do
  if [ $mode = slow ]; then export FMD_OVLP_SLOW_ONLY=1; else unset FMD_OVLP_SLOW_ONLY; fi
  python bench.py --steps 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); o=d['overlap_discovery']
print('$mode', 'overlap %.2f M reads/s' % (o['value']/1e6), '%.1f ms/step' % o['ms_per_step'], 'frac %.3f' % o['roofline']['frac'], o['parity_vs_cpu_on_sample'], 'ovf', o['overflow_records'])"
done
