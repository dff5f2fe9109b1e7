"""Copy a rocprofv3 --stats kernel summary with kernel names cut to 160 characters
(torch's templated names run to kilobytes).
usage: python benchmarks/trim_stats.py <rocprofv3 output dir> <destination csv>"""
import csv
import glob
import sys

src = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
with open(src) as fi, open(sys.argv[2], "w", newline="") as fo:
    w = csv.writer(fo)
    for row in csv.reader(fi):
        row[0] = row[0][:160]
        w.writerow(row)
print("wrote", sys.argv[2], "from", src)
