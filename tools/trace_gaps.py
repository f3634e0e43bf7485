#!/usr/bin/env python3
"""Largest idle gaps of the training queue in a rocprofv3 kernel trace, with what the other queues ran meanwhile (development aid).
    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [min_gap_ms]"""
import csv
import glob
import os
import re
import sys


def short(n):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", n))[:48]


d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
kt = sorted(glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(kt)) if r["Kind"] == "KERNEL_DISPATCH"]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
main_q = [r for r in rows if "k_embbag_fwd" in r["Kernel_Name"]][0]["Queue_Id"]
mq = [r for r in rows if r["Queue_Id"] == main_q]
print("training queue %s, %d launches; queues: %s" % (main_q, len(mq), sorted({r["Queue_Id"] for r in rows})))
for a, b in zip(mq, mq[1:]):
    gap = (b["s"] - a["e"]) / 1e6
    if gap >= min_gap:
        print("gap %.1f ms after %s (ended %.3f ms), next %s" % (gap, short(a["Kernel_Name"]), a["e"] / 1e6 % 100000, short(b["Kernel_Name"])))
        inside = [r for r in rows if r["e"] > a["e"] and r["s"] < b["s"] and r["Queue_Id"] != main_q]
        agg = {}
        for r in inside:
            k = (r["Queue_Id"], short(r["Kernel_Name"]))
            v = agg.setdefault(k, [0, 0.0])
            v[0] += 1
            v[1] += (min(r["e"], b["s"]) - max(r["s"], a["e"])) / 1e6
        for (q, n), (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            print("      q%-3s %-48s x%-4d %.2f ms inside the gap" % (q, n, cnt, ms))

# steps (gather launch to gather launch) longer than 3 ms: where their time went
anch = [r for r in rows if "k_embbag_fwd" in r["Kernel_Name"]]
print("\nslow steps:")
shown = 0
for a, b in zip(anch, anch[1:]):
    dur = (b["s"] - a["s"]) / 1e6
    if dur < 3.0 or shown >= 3:
        continue
    shown += 1
    print("step of %.1f ms:" % dur)
    last = {}
    for r in rows:
        if r["s"] < a["s"] or r["s"] >= b["s"]:
            continue
        q = r["Queue_Id"]
        gap = (r["s"] - last[q]) / 1e6 if q in last else 0.0
        last[q] = r["e"]
        k = (r["e"] - r["s"]) / 1e6
        if k > 0.2 or gap > 0.2:
            print("   +%8.3f ms  q%-2s gap %7.3f  ran %7.3f ms  %s" % ((r["s"] - a["s"]) / 1e6, q, gap, k, short(r["Kernel_Name"])))
