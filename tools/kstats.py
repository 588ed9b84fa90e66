#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv (found under the given directory) with short kernel names."""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True))[-1]
for r in csv.DictReader(open(f)):
    m = re.search(r"mmvae::(k_\w+)(<[^>]*>)?", r["Name"])
    if m:
        print("%-28s calls %4s avg %8.1f us  tot %5.1f%%  min %7.1f max %7.1f" % (
            m.group(1) + (m.group(2) or ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"]),
            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
