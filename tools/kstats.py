#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv with short kernel names (development aid).
    python tools/kstats.py gpurun_out/prof_x/stats [min_calls]"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
f = sorted(glob.glob(d + "/*/*kernel_stats.csv"), key=os.path.getmtime)[-1]
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) < mn:
        continue
    name = re.sub(r"^void ", "", r["Name"])
    name = re.sub(r"\(.*$", "", name)[:70]
    print("%-70s calls %5s  avg %9.1f us  min %9.1f  max %9.1f" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                   float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
