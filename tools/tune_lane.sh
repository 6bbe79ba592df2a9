#!/bin/bash
# Run on the GPU box: the sorted job (5*10^7 error-free reads, HIP-event ms per pass) over a few settings of the lane kernel's knobs
OUT=gpurun_out/${1:-tune}; mkdir -p $OUT
run() { env "$@" timeout 300 python tools/ab_grp4.py 50000000 0.0 3 1 FMD_NEI_LANE 2>&1 | grep "ms per pass" | sed "s/^/$* /" | tee -a $OUT/tune_lane.txt; }
run FMD_LANE_BATCH=32 FMD_LANE_TICKETS=256
run FMD_LANE_BATCH=32 FMD_LANE_TICKETS=64
run FMD_LANE_BATCH=32 FMD_LANE_TICKETS=1024
run FMD_LANE_BATCH=28 FMD_LANE_TICKETS=256
run FMD_LANE_BATCH=40 FMD_LANE_TICKETS=256
run FMD_LANE_BATCH=32 FMD_LANE_TICKETS=256 FMD_LANE_WAVES=6
run FMD_LANE_BATCH=32 FMD_LANE_TICKETS=256
