#!/bin/bash
# Run on the GPU box: `fermi-amd unitig -l50` on 10^7 reads with the process kept on one NUMA node (default) and free to roam (FMD_NUMA=off), three times each.
python tools/time_unitig_10m.py 10000000 > /dev/null 2>&1     # leaves /tmp/fmd_time_unitig/a.fmd
lscpu | grep -i "numa node"
for rep in 1 2 3; do for v in on off; do
  if [ $v = off ]; then export FMD_NUMA=off; else unset FMD_NUMA; fi
  FMD_TIMING=1 fermi_amd/bin/fermi-amd unitig -l50 /tmp/fmd_time_unitig/a.fmd 2> /tmp/numa_err.txt > /tmp/numa.mag
  echo "FMD_NUMA=$v: $(grep -h 'walks run at the commit' /tmp/numa_err.txt | sed 's/.*appended in/commit/;s/, turning.*//') | $(grep -h 'table of' /tmp/numa_err.txt | sed 's/.*GPU(s): /table /;s/, .*//') | $(grep -h 'skip list over' /tmp/numa_err.txt | sed 's/.*links: /skip list /;s/;.*//') | $(grep -h 'from the start' /tmp/numa_err.txt | sed 's/.*unitig: //') | $(md5sum < /tmp/numa.mag | cut -c1-8)"
done; done
