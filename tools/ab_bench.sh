#!/bin/bash
# A/B the overlap leg of bench.py between HIP library variants (run on the GPU box).
# usage: tools/ab_bench.sh name1 name2 ...   (name "base" = fermi_amd/lib/libfmdhip.so)
export FMD_BENCH_CPU_SAMPLE=20000 FMD_BENCH_CPU_SAMPLE_OVLP=20000
for v in "$@"; do
  if [ "$v" = base ]; then unset FMD_HIP_LIB; else export FMD_HIP_LIB=$PWD/fermi_amd/lib/libfmdhip_$v.so; fi
  python bench.py --steps 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); o=d['overlap_discovery']
print('$v', 'bsearch %.1f M reads/s' % (d['value']/1e6), 'overlap %.2f M reads/s' % (o['value']/1e6), '%.1f ms/step' % o['ms_per_step'], 'frac %.3f' % o['roofline']['frac'], o['parity_vs_cpu_on_sample'], d['parity_vs_cpu_on_sample'])"
done
