"""Print the start/end (ms, relative) of the overlap kernels of the last bench step from a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_ovl" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = rows after the last long gap
last_walk0 = max(i for i, r in enumerate(rows) if "k_ovl_walk" in r["Kernel_Name"] and (i == 0 or "k_ovl_walk" not in rows[i - 1]["Kernel_Name"]) and
                 (i < 2 or int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 200000))
t0 = int(rows[last_walk0]["Start_Timestamp"])
for r in rows[last_walk0:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print("%-22s q%-3s %8.2f -> %8.2f  (%.2f ms)" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6,
                                                 (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
