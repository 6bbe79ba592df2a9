#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/pmc_legs.py -> profiles/pmc_traffic.json, keyed by leg@reads and stamped with the sha
of the kernel sources they were measured on (bench.py prints roofline.traffic only when that sha is the tree's).
FETCH_SIZE / WRITE_SIZE are KB (MI355X_MICROARCH.md, HBM section); FETCH_SIZE is calibrated on the gather probe (64-byte
lines, known byte count), as that section prescribes for anything but wide streaming reads.
Usage: python tools/pmc_to_json.py <dir with pmc_fetch/ pmc_write/ pmc_probe/> <steps> <n_reads> <n_bsearch_reads> <source label>"""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

d, steps, n_reads, n_bs, label = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
OVL = ("k_ovl_head_adm", "k_ovl_walk", "k_ovl_pair", "k_ovl_strag_adm", "k_ovl_park_keys", "k_ovl_seq_out", "k_ovl_seq_redo", "k_ovl_classify", "k_ovl_nei_fast", "k_ovl_nei_lane", "k_ovl_nei_grp", "k_ovl_nei", "k_ovl_fix")
# (the two 32-bit radix sorts of the sorted job -- ~10 GB of streaming per 10^8 strands -- run in rocprim kernels whose names they
# share with the sorts of the index build in the same profile: not in these sums)
LEGS = {"overlap@%d" % n_reads: OVL,
        "check_left@%d" % n_reads: ("k_link_rows", "k_link_edges", "k_link_row_of", "k_link_rows32", "k_link_edges32", "k_ovl_cls"), "k_bsearch@%d" % n_bs: ("k_bsearch", "k_bsearch_pair"), "smem@%d" % n_reads: ("k_smem",),
        "kmer@%d" % n_reads: ("k_kmer_level", "k_kmer_emit")}


def sums(sub, counter):
    f = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(float)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]] += float(r["Counter_Value"])
    return acc


fetch, write, probe = sums("pmc_fetch", "FETCH_SIZE"), sums("pmc_write", "WRITE_SIZE"), sums("pmc_probe", "FETCH_SIZE")
cal = 1.0
if probe.get("k_probe"):
    known = 2 * (1 << 27) * 64            # probe_once: warm-up + timed launch, 2^27 lines of 64 bytes each
    cal = known / (probe["k_probe"] * 1024.0)
out = {"_comment": "HBM bytes per bench step from rocprofv3 --pmc passes over tools/pmc_legs.py (tools/pmc_collect.sh): sum over the launches of a leg's "
                   "kernels / steps.  KB units; FETCH_SIZE x fetch_calibration (gather probe, 64-byte lines: %.4f).  Every entry is valid for the kernel sources its csrc_sha names (the leg's files, bench.csrc_sha(leg))." % cal}
for key, names in LEGS.items():
    fk = sum(v for k, v in fetch.items() if k in names) / steps
    wk = sum(v for k, v in write.items() if k in names) / steps
    if fk:
        out[key] = {"fetch_kb": fk, "write_kb": wk, "fetch_calibration": cal, "csrc_sha": bench.csrc_sha(key.split("@")[0]), "source": label,
                    "per_kernel_fetch_kb": {k: v / steps for k, v in fetch.items() if k in names}}
fr, wr = sums("raw_fetch", "FETCH_SIZE"), sums("raw_write", "WRITE_SIZE")   # overlap discovery on the raw-read index: its own profile (same kernel names)
fk = sum(v for k, v in fr.items() if k in OVL) / steps
if fk:
    out["overlap_raw@%d" % n_reads] = {"fetch_kb": fk, "write_kb": sum(v for k, v in wr.items() if k in OVL) / steps, "fetch_calibration": cal, "csrc_sha": bench.csrc_sha("overlap_raw"),
                                       "source": label, "per_kernel_fetch_kb": {k: v / steps for k, v in fr.items() if k in OVL}}
fe, we = sums("ec_fetch", "FETCH_SIZE"), sums("ec_write", "WRITE_SIZE")     # the correction pass (PMC_LEGS=ecfix: a profile of its own)
if fe.get("k_ecfix"):
    out["ecfix@%d" % n_reads] = {"fetch_kb": fe["k_ecfix"] / steps, "write_kb": we.get("k_ecfix", 0.0) / steps, "fetch_calibration": cal, "csrc_sha": bench.csrc_sha("ecfix"),
                                 "source": label, "per_kernel_fetch_kb": {"k_ecfix": fe["k_ecfix"] / steps}}
dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
try:   # a partial collection (PMC_LEGS=...) keeps the other legs' entries; each is valid for the sha it carries
    old = json.load(open(dst))
    for k, v in old.items():
        if k not in out and isinstance(v, dict) and v.get("csrc_sha") == bench.csrc_sha(k.split("@")[0]):
            out[k] = v
except Exception:
    pass
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
