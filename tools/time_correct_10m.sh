#!/bin/bash
# Run on the GPU box after tools/time_unitig_10m.py (uses its FASTQ): fermi-amd build on reads with 1 % errors is
# not needed for timing -- correct the error-free reads against their own index and print the stage times.
D=/tmp/fmd_time_unitig
A=fermi_amd/bin/fermi-amd
for t in 64 16; do echo "correct -t$t"; FMD_TIMING=1 $A correct -t$t $D/a.fmd $D/r.fq 2>&1 >/dev/null | grep "M::"; done
