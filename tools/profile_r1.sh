#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of bench.py, then PMC passes (separate runs,
# --pmc never combined with tracing) for HBM traffic, calibrated on the gather probe whose byte
# count is known.  Results land in gpurun_out/prof_$TAG/.
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export FMD_BENCH_CPU_SAMPLE=${FMD_BENCH_CPU_SAMPLE:-200000}
# 1. kernel trace + stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 5 --warmup 1 > $OUT/bench_traced.json 2> $OUT/bench_traced.err
# 2. PMC: HBM read bytes of the search kernel
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_write.err
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o bench -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_l2.err
timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -o bench -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_sq.err
# 3. calibration: the probe reads exactly n_access*128 bytes of a 64 GiB working set (no reuse)
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_probe -o probe -- python tools/probe_once.py > $OUT/probe_once.txt 2> $OUT/pmc_probe.err
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.md 2>&1
cat $OUT/SUMMARY.md
