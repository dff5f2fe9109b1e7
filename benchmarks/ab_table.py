"""Table of `mpm_stream wall` times from a benchmarks/ab_libs.sh log: one row per case, one column per library."""
import re
import sys
rows, lib, case = {}, None, None
for l in open(sys.argv[1]):
    m = re.match(r'== tmp_ab/(\S+) :: (.*)', l)
    if m:
        lib, case = m.group(1), m.group(2)
        continue
    m = re.search(r'wall\s+([\d.]+) ms', l)
    if m:
        rows.setdefault(case, {})[lib] = float(m.group(1))
libs = sorted({k for v in rows.values() for k in v})
print("case".ljust(18), " ".join(x.ljust(14) for x in libs))
for c, v in rows.items():
    print(c.ljust(18), " ".join(str(v.get(x, '')).ljust(14) for x in libs))
