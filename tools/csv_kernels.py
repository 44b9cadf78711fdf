#!/usr/bin/env python3
"""Print name prefix, calls and average microseconds from a rocprofv3 *_kernel_stats.csv. Dev tool."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or sys.argv[2] in r["Name"]:
        print("%-60s %5s calls  %9.1f us avg" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
