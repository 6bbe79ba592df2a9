#!/bin/bash
# kernel trace + PMC passes for k_smem (run on the GPU box)
TAG=${1:-r1_smem}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python tools/run_smem.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/run.txt 2> $OUT/run.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $CMD > /dev/null 2> $OUT/pmc_write.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o bench -- $CMD > /dev/null 2> $OUT/pmc_sq.err
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o bench -- $CMD > /dev/null 2> $OUT/pmc_l2.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_inst -o bench -- $CMD > /dev/null 2> $OUT/pmc_inst.err
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.md 2>&1
cat $OUT/SUMMARY.md; cat $OUT/run.txt
