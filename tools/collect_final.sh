#!/bin/bash
# After tools/final_measure.sh <tag> ran on the GPU box and gpurun merged its output: copy what is to be judged from gpurun_out/ (scratch)
# into profiles/<round>_final/ (tracked) -- the bench line, the traced run's kernel statistics, the test log, the PMC passes summed per
# kernel -- and profiles/pmc_traffic.json.  Usage: tools/collect_final.sh <tag> [dest=profiles/r3_final]
TAG=$1; DST=${2:-profiles/r6_final}
[ -d gpurun_out/$TAG ] || { echo "no gpurun_out/$TAG"; exit 1; }
rm -rf $DST; mkdir -p $DST/pmc
cp gpurun_out/$TAG/* $DST/
cp gpurun_out/pmc_$TAG/pmc_traffic_summary.txt gpurun_out/pmc_$TAG/probe_once.txt gpurun_out/pmc_$TAG/pmc_traffic.json $DST/pmc/
for d in pmc_fetch pmc_write raw_fetch raw_write ec_fetch ec_write pmc_probe; do
  f=$(find gpurun_out/pmc_$TAG/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $DST/pmc/${d}_per_kernel.csv <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith(("k_", "fmd")) or "probe" in k:
        acc[(k, r["Counter_Name"])][0] += 1; acc[(k, r["Counter_Name"])][1] += float(r["Counter_Value"])
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["kernel", "counter", "launches", "sum"])
for (k, c), (n, v) in sorted(acc.items()): w.writerow([k, c, n, v])
PY
done
cp gpurun_out/pmc_$TAG/pmc_traffic.json profiles/pmc_traffic.json
sed -i "s#profiles/${TAG}_pmc#$DST/pmc#g" profiles/pmc_traffic.json $DST/pmc/pmc_traffic.json $DST/pmc/pmc_traffic_summary.txt $DST/bench.json $DST/bench_traced.json
[ -f gpurun_out/pytest_scale_700M.txt ] && cp gpurun_out/pytest_scale_700M.txt $DST/pytest_config5_700M.txt
python - $DST <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
d = json.load(open("profiles/pmc_traffic.json"))
for k, v in d.items():
    if isinstance(v, dict): print("%-24s %s %s %.1f GB per step" % (k, v["csrc_sha"], "current" if v["csrc_sha"] == bench.csrc_sha(k.split("@")[0]) else "STALE", (v["fetch_kb"] * v["fetch_calibration"] + v["write_kb"]) * 1024 / 1e9))
b = json.load(open(sys.argv[1] + "/bench.json"))
r = b["overlap_discovery_on_raw_reads"]
print("ec_fix %.1f ms (%s)" % (b["ec_fix"]["ms_per_step"], b["ec_fix"]["parity_vs_cpu_on_sample"][:9]))
print("headline %.1f ms (id order %.1f), raw %.1f ms (general only %.1f; %s), bsearch %.2f, smem %.1f (%s), kmer %.1f, check_left %.1f; sha %s" % (
    b["ms_per_step"], b["overlap_discovery"]["id_order_one_pass_walk"]["ms_per_step"], r["ms_with_the_fast_get_nei_path"], r["ms_general_group_kernels_only"], r["parity_vs_cpu_on_sample"],
    b["backward_search"]["ms_per_step"], b["smem"]["ms_per_step"], b["smem"]["parity_vs_cpu_on_sample"], b["kmer_harvest"]["ms_per_step"], b["check_left"]["ms_per_step"], b["kernel_sources_sha"]))
PY
cat $DST/pytest.log
