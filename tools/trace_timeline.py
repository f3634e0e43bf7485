#!/usr/bin/env python3
"""Print the per-stream kernel timeline of one steady-state training step from a rocprofv3 --kernel-trace CSV
(development aid: where the step's critical path and its idle gaps are).

    python tools/trace_timeline.py gpurun_out/prof/.../NNN_kernel_trace.csv [--step -20] [--anchor k_embbag_fwd]
"""
import argparse
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--anchor", default="k_embbag_fwd", help="kernel whose launches delimit the steps")
    ap.add_argument("--step", type=int, default=-20, help="which step to print (index into the anchor launches)")
    ap.add_argument("--nsteps", type=int, default=1)
    a = ap.parse_args()
    rows = [r for r in csv.DictReader(open(a.csv)) if r["Kind"] == "KERNEL_DISPATCH"]
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    anchors = [i for i, r in enumerate(rows) if a.anchor in r["Kernel_Name"]]
    if a.anchor == "k_embbag_fwd":
        # the step's gather fused into the interaction forward (cdlrm_gather_interact_fwd): that kernel anchors the steps (the
        # stand-alone gather launches of such a trace are bench.py's operator timing after the timed region)
        fused = [i for i, r in enumerate(rows) if "k_interact_fwd_s<" in r["Kernel_Name"] and "true>" in r["Kernel_Name"]]
        anchors = fused or anchors
    if not anchors:
        sys.exit("no launch of " + a.anchor)
    i0 = anchors[a.step]
    i1 = anchors[a.step + a.nsteps] if a.step + a.nsteps < 0 or a.step >= 0 else len(rows)
    t0 = rows[i0]["s"]
    qs = sorted({r["Queue_Id"] for r in rows[i0:i1]})
    print("step of %.1f us, %d launches, queues %s" % ((rows[i1]["s"] - t0) / 1e3, i1 - i0, qs))
    last_end = {}
    for r in rows[i0:i1]:
        q = r["Queue_Id"]
        gap = (r["s"] - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = r["e"]
        print("%8.1f %8.1f  q%-2s %6.1f us  gap %6.1f  grid %7d  %s" % (
            (r["s"] - t0) / 1e3, (r["e"] - t0) / 1e3, q, (r["e"] - r["s"]) / 1e3, gap,
            int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]),
            short(r["Kernel_Name"])))
    # steady-state step time over the last anchors
    d = [(rows[anchors[k + 1]]["s"] - rows[anchors[k]]["s"]) / 1e3 for k in range(max(0, len(anchors) - 60), len(anchors) - 1)]
    d.sort()
    print("median step (anchor to anchor) %.1f us" % d[len(d) // 2])


if __name__ == "__main__":
    main()
