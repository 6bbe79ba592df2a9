T=/tmp/fmd_validate
python tools/validate_large.py 1000000 0.01 > /dev/null 2>&1 &
PID=$!
# wait for the inputs to exist, then stop the validator (we only want its files)
while [ ! -f $T/b.fmd ]; do sleep 2; done
kill $PID 2>/dev/null
A=fermi_amd/bin/fermi-amd
for i in 1 2; do FMD_TIMING=1 $A unitig -l50 $T/a.fmd 2>&1 >/dev/null | grep "M::"; done
TIMEFORMAT="%R s wall"
for t in 1 8 64; do echo "correct -t$t"; FMD_TIMING=1 $A correct -t$t $T/a.fmd $T/r.fq 2>&1 >/dev/null | grep "M::"; done
