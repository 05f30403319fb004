#!/usr/bin/env python
"""rocprofv3 kernel trace -> average duration per (kernel, grid size).  `--stats` aggregates by kernel NAME only, and one
template instance now serves several layers (e.g. wino_conv16g_kernel<8, 2, 4, 1, true> is the backward-data launch of the
32x32 layer AND of the 16x16 layer of the bench model), so the per-launch figure bench.py reports for the dominant launch has
to be read against the rows of ITS grid.  usage: trace_by_grid.py <*_kernel_trace.csv> > kernel_stats_by_grid.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    name = name.split("(")[0] if "<" not in name else name[:name.index(">") + 1]
    grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "?"
    agg.setdefault((name, grid, wg), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
total = sum(sum(v) for v in agg.values())
w = csv.writer(sys.stdout)
w.writerow(["Name", "GridSize", "WorkgroupSize", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
for (name, grid, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, grid, wg, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v), round(100.0 * sum(v) / total, 2)])
