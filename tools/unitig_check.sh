#!/bin/bash
# Run on the GPU box (gpurun): the GPU tests that go through `unitig` (fixtures, replicas, file pages, the in-memory API, the 1 M / 2 M / 10 M md5s), then
# `unitig` of N error-free and N raw reads with phase times and the peak resident set (FMD_TIMING).  Usage: tools/unitig_check.sh <tag> [n_reads=10000000]
TAG=${1:-uc}; N=${2:-10000000}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu -k "unitig or api or md5 or seqsort or sorted_table or file_pages" 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path\|amdgpu.ids" | tail -15 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 900 python tools/time_unitig_10m.py $N > $OUT/time_unitig.txt 2>&1; grep -v "^\[M::packed\|walk_parallel\]" $OUT/time_unitig.txt | tail -40
timeout 900 python tools/time_unitig_err.py $N > $OUT/time_unitig_err.txt 2>&1; tail -25 $OUT/time_unitig_err.txt
