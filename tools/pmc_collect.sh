#!/bin/bash
# Run on the GPU box (gpurun): the two HBM-traffic PMC passes (separate runs, --pmc never combined with tracing) over
# tools/pmc_legs.py and the calibration probe; writes profiles/pmc_traffic.json stamped with the kernel-source sha and
# copies the raw CSVs next to it.  Usage: [PMC_LEGS=smem,kmer] tools/pmc_collect.sh <tag> [steps=2]
# (PMC_LEGS = a partial re-collection: only those legs run, the other entries of profiles/pmc_traffic.json stay as long as their sources did)
TAG=${1:-r2}; K=${2:-2}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
NR=${FMD_BENCH_READS:-50000000}; NB=${FMD_BENCH_BSEARCH_READS:-10000000}
timeout 1200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o legs -- python tools/pmc_legs.py $K > $OUT/pmc_fetch.log 2>&1
timeout 1200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o legs -- python tools/pmc_legs.py $K > $OUT/pmc_write.log 2>&1
if [ -z "$PMC_LEGS" ] || [[ ",$PMC_LEGS," == *",overlap_raw,"* ]]; then
PMC_LEGS=overlap_raw timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/raw_fetch -o legs -- python tools/pmc_legs.py $K > $OUT/raw_fetch.log 2>&1
PMC_LEGS=overlap_raw timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/raw_write -o legs -- python tools/pmc_legs.py $K > $OUT/raw_write.log 2>&1
fi
if [ -z "$PMC_LEGS" ] || [[ ",$PMC_LEGS," == *",ecfix,"* ]]; then   # the correction pass: harvest + table in the child, K steps of k_ecfix
PMC_LEGS=ecfix timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ec_fetch -o legs -- python tools/pmc_legs.py $K > $OUT/ec_fetch.log 2>&1
PMC_LEGS=ecfix timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/ec_write -o legs -- python tools/pmc_legs.py $K > $OUT/ec_write.log 2>&1
fi
PROBE_LINE=64 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_probe -o probe -- python tools/probe_once.py > $OUT/probe_once.txt 2>&1
python tools/pmc_to_json.py $OUT $K $NR $NB "profiles/${TAG}_pmc (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/pmc_legs.py, $K steps per leg)" > $OUT/pmc_traffic_summary.txt 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
tail -40 $OUT/pmc_traffic_summary.txt
