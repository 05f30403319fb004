#!/usr/bin/env python
"""rocprofv3 kernel trace -> where the time of a pass goes between and inside launches: per pass (the launches between two
first-layer forward launches) the sum of kernel durations, the sum of the idle gaps between consecutive launches (end of one to the
start of the next) and the span; then the gap in front of each kernel of the pass, averaged over the passes.
usage: trace_gaps.py <*_kernel_trace.csv> [first_kernel_substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "c3w64_relu_pool"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if first in e[2]]
passes = [ev[a:b] for a, b in zip(starts, starts[1:])]
passes = passes[2:]                      # (warm-up passes)
if not passes:
    sys.exit("no passes found")
n = len(passes[0])
passes = [p for p in passes if len(p) == n]
tot_k = sum(sum(e[1] - e[0] for e in p) for p in passes) / len(passes) / 1e3
tot_g = sum(sum(max(0, p[i + 1][0] - p[i][1]) for i in range(len(p) - 1)) for p in passes) / len(passes) / 1e3
span = sum(p[-1][1] - p[0][0] for p in passes) / len(passes) / 1e3
print("passes %d, launches per pass %d: kernel time %.1f us, gaps between launches %.1f us (%.2f us per boundary), span %.1f us"
      % (len(passes), n, tot_k, tot_g, tot_g / (n - 1), span))
print("%-72s %9s %9s" % ("kernel (in launch order)", "dur us", "gap before us"))
for i in range(n):
    name = passes[0][i][2].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:72]
    dur = sum(p[i][1] - p[i][0] for p in passes) / len(passes) / 1e3
    gap = sum(max(0, p[i][0] - p[i - 1][1]) for p in passes) / len(passes) / 1e3 if i else 0.0
    print("%-72s %9.1f %9.2f" % (name, dur, gap))
