#!/bin/bash
# prefix-table depth 14 against 15, alternating, a fresh process each (process-to-process spread is 6 ms: one pair of runs says nothing)
export FMD_PAIR=1
for i in 1 2 3; do for d in 14 15; do echo -n "depth $d run $i: "; FMD_PTAB_DEPTH=$d timeout 600 python tools/ab_overlap.py 50000000 0.0 4 -- - 2>&1 | tail -1 | cut -c60-130; done; done
