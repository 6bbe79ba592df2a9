#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes (separate runs, --pmc never combined with
# tracing) for the WHOLE default bench.py line (backward search, overlap discovery, SMEM, k-mer
# harvest), plus the FETCH_SIZE calibration on the gather probe with the line size of the block
# geometry in use.  Results land in gpurun_out/prof_$TAG/.
TAG=${1:-all}
export PROBE_LINE=${2:-64}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export FMD_BENCH_CPU_SAMPLE=50000 FMD_BENCH_CPU_SAMPLE_OVLP=20000 FMD_BENCH_CPU_SAMPLE_SMEM=20000 FMD_BENCH_CPU_SAMPLE_KMER=512
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 2 --warmup 1 > $OUT/bench_traced.json 2> $OUT/bench_traced.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python bench.py --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python bench.py --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_write.err
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o bench -- python bench.py --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_l2.err
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o bench -- python bench.py --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_sq.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_probe -o probe -- python tools/probe_once.py > $OUT/probe_once.txt 2> $OUT/pmc_probe.err
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.md 2>&1
cat $OUT/SUMMARY.md
