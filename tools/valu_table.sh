#!/bin/bash
# Run on the GPU box (gpurun): vector instructions issued by the get_nei kernels over one step of overlap discovery on 5*10^7 RAW reads (1 % substitutions),
# as the tree runs it and with round 5's three changes to those kernels switched off (FMD_NEI_LANE=0: the unforked path in the group form;
# FMD_GRP_DOWN=0: no second pass through smaller groups; FMD_GRP_QUIET=0: no quiet round in the group kernels) -- rocprofv3 --pmc SQ_INSTS_VALU,
# counters only, one pass each (MI355X_MICROARCH.md).  Usage: tools/valu_table.sh <tag>  -> gpurun_out/<tag>/TABLE.txt
TAG=${1:-r5_grp}; OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PMC_LEGS=overlap_raw timeout 900 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $OUT/now -o legs -- python tools/pmc_legs.py 1 > $OUT/now.log 2>&1
PMC_LEGS=overlap_raw FMD_NEI_LANE=0 FMD_GRP_DOWN=0 FMD_GRP_QUIET=0 timeout 900 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $OUT/before -o legs -- python tools/pmc_legs.py 1 > $OUT/before.log 2>&1
python tools/valu_table.py $OUT > $OUT/TABLE.txt 2>&1
cat $OUT/TABLE.txt
