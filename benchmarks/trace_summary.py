"""Print per-launch durations of one kernel from a rocprofv3 kernel-trace CSV.
usage: python benchmarks/trace_summary.py <dir> <kernel substring> [launches per group]"""
import csv
import glob
import sys

d, sub = sys.argv[1], sys.argv[2]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 50
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(dur), "launches; grid", rows[0]["Grid_Size_X"], "wg", rows[0]["Workgroup_Size_X"], "vgpr", rows[0]["VGPR_Count"],
      "lds", rows[0]["LDS_Block_Size"])
last = dur[-per:]
print("last group (us):", " ".join(f"{x:.0f}" for x in last))
print("sum of last group: %.1f us" % sum(last))
