#!/bin/bash
# Chunk sizes of the dealt-out get_nei work lists (FMD_DEAL_CHUNK, fmd_ovlp_grp.hip) and the group-occupancy counters with them.  Build the variants first:
#   make variant NAME=dc32 EXTRA=-DFMD_DEAL_CHUNK=32; make variant NAME=dc128 EXTRA=-DFMD_DEAL_CHUNK=128; make variant NAME=stats EXTRA=-DGRP_STATS=1
# then run this on the GPU box.  Separate processes: differences below ~2 % are box noise (profiles/r4_dyn/chunk_sizes.txt).
mkdir -p gpurun_out/r4_dyn
for e in 0.01 0; do
  for v in "" _dc32 _dc128; do
    FMD_HIP_LIB=$PWD/fermi_amd/lib/libfmdhip$v.so timeout 200 python tools/ab_grp4.py 50000000 $e 3 1 FMD_NEI_DYN 2>&1 | grep "ms per pass" | sed "s/^/chunk variant '$v' e=$e: /" >> gpurun_out/r4_dyn/chunk_sizes.txt
  done
done
FMD_HIP_LIB=$PWD/fermi_amd/lib/libfmdhip_stats.so timeout 200 python tools/ab_grp4.py 10000000 0.01 1 1 FMD_NEI_DYN > gpurun_out/r4_dyn/grp_stats_raw_10M_dyn.txt 2>&1
FMD_HIP_LIB=$PWD/fermi_amd/lib/libfmdhip_stats.so timeout 200 python tools/ab_grp4.py 10000000 0 1 1 FMD_NEI_DYN > gpurun_out/r4_dyn/grp_stats_clean_10M_dyn.txt 2>&1
cat gpurun_out/r4_dyn/chunk_sizes.txt
