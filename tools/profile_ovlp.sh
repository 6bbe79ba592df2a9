#!/bin/bash
# kernel trace + PMC passes for the overlap-discovery leg of bench.py (run on the GPU box)
TAG=${1:-r1_ovlp}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export FMD_BENCH_CPU_SAMPLE=100000 FMD_BENCH_CPU_SAMPLE_OVLP=${FMD_BENCH_CPU_SAMPLE_OVLP:-100000}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 2 --warmup 1 > $OUT/bench_traced.json 2> $OUT/bench_traced.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python bench.py --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python bench.py --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_write.err
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o bench -- python bench.py --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_sq.err
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o bench -- python bench.py --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_l2.err
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.md 2>&1
cat $OUT/SUMMARY.md
