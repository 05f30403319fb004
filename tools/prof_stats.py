#!/usr/bin/env python
"""Print the top kernels of a rocprofv3 *_kernel_stats.csv (short names). usage: prof_stats.py <csv> [n] [filter]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
flt = sys.argv[3] if len(sys.argv) > 3 else ""
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"\(.*", "", name)[:64]
    if flt and not re.search(flt, name):
        continue
    print("%-66s calls %5s avg %8.1f us  %5.2f%%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, 100.0 * int(r["TotalDurationNs"]) / tot))
    n -= 1
    if n <= 0:
        break
print("total kernel time %.2f ms" % (tot / 1e6))
