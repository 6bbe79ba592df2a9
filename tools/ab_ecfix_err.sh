#!/bin/bash
# k_ecfix at other error rates than the bench's 1 %: is what round 6 did to it (hop loop, gates) a gain everywhere?   tools/ab_ecfix_err.sh libOld.so libNew.so
for e in 0.0 0.03 0.002; do for lib in "$@"; do echo -n "e=$e "; FMD_BENCH_SMEM_ERR=$e bash tools/ab_ecfix.sh $lib | tail -1; done; done
