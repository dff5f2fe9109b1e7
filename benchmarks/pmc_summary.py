"""Sum the counters of a rocprofv3 --pmc run per kernel (names cut to 120 characters).
usage: python benchmarks/pmc_summary.py <rocprofv3 output dir> <destination csv> [kernel substring]"""
import csv
import glob
import sys
from collections import defaultdict

src = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
sub = sys.argv[3] if len(sys.argv) > 3 else ""
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(dict)
for r in csv.DictReader(open(src)):
    name = r["Kernel_Name"][:120]
    if sub not in name:
        continue
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[name][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
counters = sorted({c for v in acc.values() for c in v})
with open(sys.argv[2], "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "dispatches", "total_us"] + counters)
    for name in sorted(acc, key=lambda n: -sum(acc[n].values())):
        w.writerow([name, len(disp[name]), f"{sum(disp[name].values()):.1f}"] + [f"{acc[name].get(c, 0.0):.6g}" for c in counters])
print("wrote", sys.argv[2], "from", src)
