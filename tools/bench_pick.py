import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{"metric"'):
        d = json.loads(l)
        print("bsearch ms", round(d["ms_per_step"], 2))
        for k in ("overlap_discovery", "smem", "kmer_harvest"):
            if k in d:
                o = d[k]
                print(k, "ms", round(o["ms_per_step"], 2), "frac", round(o["roofline"]["frac"], 3), "parity", o.get("parity_vs_cpu_on_sample"),
                      ("| pipelined vs serial: " + o["pipelined_vs_serial_order"]) if "pipelined_vs_serial_order" in o else "")
