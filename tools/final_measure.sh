#!/bin/bash
# Run on the GPU box (gpurun), on the FINAL kernel sources of a round: GPU tests, smoke, the PMC passes (which stamp
# profiles/pmc_traffic.json with the source sha), then the bench line twice -- plain (what the driver will time) and under
# rocprofv3 --kernel-trace --stats -- and the two CLI timings.  Everything lands in gpurun_out/<tag>/ (+ gpurun_out/pmc_<tag>/).
# Usage: tools/final_measure.sh <tag> [steps=20] [warmup=5]
TAG=${1:-final}; K=${2:-20}; W=${3:-5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path\|amdgpu.ids" | tail -6 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
FMD_PAIR=1 tools/pmc_collect.sh $TAG 2 > $OUT/pmc_collect.log 2>&1; tail -3 $OUT/pmc_collect.log   # (FMD_PAIR=1: the legs on indexes with two-base blocks, as bench.py runs them)
timeout 1800 python bench.py --steps $K --warmup $W > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
FMD_BENCH_PMC=0 timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$TAG -o bench -- python bench.py --steps 5 --warmup 2 > $OUT/bench_traced.json 2> $OUT/bench_traced.err
cp $(find /tmp/trace_$TAG -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python tools/trace_tail.py $(find /tmp/trace_$TAG -name "*kernel_trace.csv" | head -1) 400 > $OUT/overlap_timeline_tail.txt 2>/dev/null
timeout 900 python tools/time_unitig_10m.py 10000000 > $OUT/time_unitig_10M.txt 2>&1; tail -3 $OUT/time_unitig_10M.txt
timeout 900 python tools/time_correct.py 10000000 > $OUT/time_correct_10M.txt 2>&1; tail -3 $OUT/time_correct_10M.txt
python tools/bench_table.py $OUT/bench.json
