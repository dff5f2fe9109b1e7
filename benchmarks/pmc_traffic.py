"""HBM traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are
collected in separate runs) -> an entry of profiles/pmc_traffic.json and rows of a csv.

usage: python benchmarks/pmc_traffic.py <key> <fetch dir> <write dir> <kernel substring>
                                        <nodes R> <executed graph-iterations> <command text>
                                        [algorithmic bytes of the launch, for ragged batches]

gfx950 correction: on wide coalesced streams FETCH_SIZE counts 64 B per 128-B request, so it
is doubled; WRITE_SIZE is taken as read; both are in KiB."""
import csv
import glob
import json
import sys
from pathlib import Path

key, d_fetch, d_write, sub, R, gi, cmd = sys.argv[1:8]
R, gi = int(R), int(gi)
ROOT = Path(__file__).resolve().parents[1]


def rows(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    return [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]


def mean_for(d, counter):
    v = [float(r["Counter_Value"]) for r in rows(d, counter) if sub in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)


fetch_kb, n_f = mean_for(d_fetch, "FETCH_SIZE")
write_kb, n_w = mean_for(d_write, "WRITE_SIZE")
hbm = 2.0 * fetch_kb * 1024 + write_kb * 1024
alg = float(sys.argv[8]) if len(sys.argv) > 8 else gi * (4 * R ** 4 + 12 * R ** 2)
entry = {
    "kernel": sub, "command": cmd, "launches_averaged": [n_f, n_w],
    "fetch_size_kb_mean": fetch_kb, "write_size_kb_mean": write_kb, "graph_iterations": gi,
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> doubled; "
                  "WRITE_SIZE as read; KB = 1024 B",
    "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
    "hbm_bytes_per_graph_iteration": hbm / gi, "ratio_to_algorithmic": hbm / alg,
}
p = ROOT / "profiles" / "pmc_traffic.json"
allv = json.load(open(p)) if p.exists() else {}
allv[key] = entry
json.dump(allv, open(p, "w"), indent=1)
import os
out = ROOT / "profiles" / f"{os.environ.get('GVAE_PMC_ROUND', 'r2')}_pmc_matching_{key}.csv"
with open(out, "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["pass", "kernel", "grid", "workgroup", "counter", "value_KB", "duration_us"])
    for name, d, counter in (("pmc_fetch", d_fetch, "FETCH_SIZE"), ("pmc_write", d_write, "WRITE_SIZE")):
        for r in rows(d, counter):
            if "gvae::" in r["Kernel_Name"]:
                w.writerow([name, r["Kernel_Name"].split("(")[0], r["Grid_Size"], r["Workgroup_Size"], counter,
                            r["Counter_Value"], "%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)])
print(json.dumps(entry, indent=1))
