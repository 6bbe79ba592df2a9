#!/bin/bash
# The short form of tools/final_measure.sh for a change that touches the get_nei kernels only (fmd_ovlp_grp.hip): the GPU tests that
# run those kernels, the PMC passes of the raw-read leg (the headline's are taken by bench.py in its own run), the bench line plain and
# under rocprofv3 --kernel-trace --stats.  Usage: tools/final_measure_short.sh <tag>
TAG=${1:-final_short}
OUT=gpurun_out/$TAG; P=gpurun_out/pmc_$TAG
mkdir -p $OUT $P
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_sorted.py tests/test_gpu_dist.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu \
   -k "sorted or dist or rccl or head_and_tail or overlap or unitig or lfork or check_left or unforked or fuzz or packed or 10m or repeat_rich" 2>&1) | grep -v "amdgpu.ids" | tail -8 > $OUT/pytest_get_nei_subset.log; cat $OUT/pytest_get_nei_subset.log
PMC_LEGS=overlap_raw timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/raw_fetch -o legs -- python tools/pmc_legs.py 2 > $P/raw_fetch.log 2>&1
PMC_LEGS=overlap_raw timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/raw_write -o legs -- python tools/pmc_legs.py 2 > $P/raw_write.log 2>&1
PROBE_LINE=64 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_probe -o probe -- python tools/probe_once.py > $P/probe_once.txt 2>&1
python tools/pmc_to_json.py $P 2 50000000 10000000 "profiles/r4_final/pmc (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/pmc_legs.py, 2 steps per leg)" > $P/pmc_traffic_summary.txt 2>&1
cp profiles/pmc_traffic.json $P/pmc_traffic.json
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
FMD_BENCH_PMC=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$TAG -o bench -- python bench.py --steps 5 --warmup 2 > $OUT/bench_traced.json 2> $OUT/bench_traced.err
cp $(find /tmp/trace_$TAG -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python tools/trace_tail.py $(find /tmp/trace_$TAG -name "*kernel_trace.csv" | head -1) 400 > $OUT/overlap_timeline_tail.txt 2>/dev/null
python tools/bench_table.py $OUT/bench.json
