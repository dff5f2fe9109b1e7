"""Prints the kernel timeline of one steady-state step from a rocprofv3 --kernel-trace CSV:
start and end (microseconds from the step's first kernel), stream/queue and name.
usage: python benchmarks/timeline.py <kernel_trace.csv> [step_index_from_end=2]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step begins with the kernel that normalises the adjacency (the model half's first)
starts = [i for i, r in enumerate(rows)
          if "gcn_norm_adj" in r["Kernel_Name"] or "model_fwd_" in r["Kernel_Name"]]
i0, i1 = starts[-back - 1], starts[-back]
t0 = int(rows[i0]["Start_Timestamp"])
print(f"step of {i1 - i0} kernels, {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us to the next step's first kernel")
for r in rows[i0:i1 + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0][-48:]
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
