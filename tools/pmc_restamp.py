#!/usr/bin/env python3
"""profiles/pmc_traffic.json entries measured when bench.csrc_sha() covered ALL of fermi_amd/csrc -> entries stamped per leg
(bench.csrc_sha(leg): the headers, the index layout and the files the leg's kernels live in), so that a change to one leg's
kernels invalidates that leg's counters only.  An entry is restamped only if its old stamp IS the whole-tree sha of the git
revision named (i.e. it provably was measured on that revision's files); the per-leg sha is computed from that revision's files.
Usage: python tools/pmc_restamp.py <git revision the entries were measured on> [json=profiles/pmc_traffic.json]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
rev = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "pmc_traffic.json")
at_rev = lambda fn: subprocess.run(["git", "-C", ROOT, "show", "%s:fermi_amd/csrc/%s" % (rev, fn)], check=True, capture_output=True).stdout
names = subprocess.run(["git", "-C", ROOT, "ls-tree", "--name-only", rev, "fermi_amd/csrc/"], check=True, capture_output=True, text=True).stdout.split()
assert sorted(os.path.basename(n) for n in names if n.endswith((".hip", ".h"))) == sorted(f for f in os.listdir(os.path.join(ROOT, "fermi_amd", "csrc")) if f.endswith((".hip", ".h"))), "file set changed"
whole = bench.csrc_sha(None, read=at_rev)
d = json.load(open(path))
for k, v in d.items():
    if isinstance(v, dict) and v.get("csrc_sha") == whole:
        v["csrc_sha"] = bench.csrc_sha(k.split("@")[0], read=at_rev)
        v["measured_on_revision"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", rev], check=True, capture_output=True, text=True).stdout.strip()
        print("%-24s -> %s (%s)" % (k, v["csrc_sha"], "current" if v["csrc_sha"] == bench.csrc_sha(k.split("@")[0]) else "STALE: the leg's sources changed since"))
    elif isinstance(v, dict):
        print("%-24s kept (stamp %s is not the whole-tree sha of %s)" % (k, v.get("csrc_sha"), rev))
if "_comment" in d:
    d["_comment"] = d["_comment"].replace("Valid for kernel sources with this sha only.", "Every entry is valid for the kernel sources its csrc_sha names (the leg's files, bench.csrc_sha(leg)).")
json.dump(d, open(path, "w"), indent=1)
