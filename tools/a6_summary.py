#!/usr/bin/env python3
"""SURVEY row a-6 (per-iteration tag probe + cached EmbeddingBag forward, model_no_ddp.py:149-212) AS A WHOLE, from a rocprofv3
kernel trace of bench.py: the roofline kernel is only the gather half of the row.  Per training step the row also pays

  * the take of the batch's slot ids and miss rows (k_take), and
  * its share of the window-resident probe: one look-ahead chunk of CH batches is resolved per CH steps
    (k_probe + k_victim_pos + k_resolve_seg: cdlrm_window_resolve), i.e. 1 / CH of a chunk per step,

both on side queues (beside the step's GEMMs), so their in-step durations are contended ones.

    python tools/a6_summary.py gpurun_out/prof_c3/stats profiles/r04_a6_whole_c3.json [lookups_per_step] [D]
"""
import csv
import glob
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import roofkernel  # noqa: E402

TAKE = "k_take"
RESOLVE = ("k_probe", "k_victim_pos", "k_resolve_seg")


def main():
    d, out = sys.argv[1], sys.argv[2]
    lookups = int(sys.argv[3]) if len(sys.argv) > 3 else 8192 * 26
    D = int(sys.argv[4]) if len(sys.argv) > 4 else 128
    kt = sorted(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
    dur = {}
    rows = list(csv.DictReader(open(kt)))
    which = roofkernel.pick({r["Kernel_Name"] for r in rows})
    for r in rows:
        name = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
        if roofkernel.kind(r["Kernel_Name"]) == which:
            name = "<the gather>"       # the stand-alone gather, or the interaction forward that does the gather (fused)
        dur.setdefault(name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    gname = "<the gather>"
    steps = len(dur.get(gname, []))
    g = dur[gname][10:] if steps > 30 else dur[gname]
    take = dur.get(TAKE, [])
    take = take[10:] if len(take) > 30 else take
    res_total = sum(sum(dur.get(k, [])) for k in RESOLVE)
    res_launches = {k: len(dur.get(k, [])) for k in RESOLVE}
    gather_us, take_us = statistics.mean(g), (statistics.mean(take) if take else 0.0)
    resolve_us = res_total / max(1, steps)
    whole = gather_us + take_us + resolve_us
    alg = lookups * (8 * D + 16)
    if which == "fused":
        # the row's work is now: read the rows (+ slot ids) -- the pooled output is never written; the fused kernel's time also
        # holds the interaction forward itself (a-9), which no longer has a launch of its own
        # priced on what the fused operator moves (rows + slot ids, the dense feature, the interaction rows): tools/roofkernel.py
        T = 26
        alg = roofkernel.bytes_per_launch("fused", lookups // T, T, D)[1]
    doc = {"source": kt.split("gpurun_out/")[-1], "steps_in_trace": steps, "gather_kernel": which,
           "gather_us": gather_us, "take_us_in_step": take_us, "window_resolve_us_per_step_amortised": resolve_us,
           "resolve_launches": res_launches, "a6_us_per_step": whole,
           "algorithmic_bytes_per_step": alg, "roofline_us_at_8TBps": alg / 8e6,
           "frac_of_8TBps_whole_row": alg / whole / 1e3 / 8000.0,
           "basis": ("fused: the gather IS the interaction forward's operand load -- gather_us is that one kernel (rows a-6 + a-9 forward), "
                     "priced on its own bytes" if which == "fused" else "SURVEY 8(d): 8D + 16 per lookup"),
           "note": "gather: the roofline kernel alone on the training queue; take and resolve run on side queues beside the "
                   "step's GEMMs (contended durations: k_take is ~10-19 us stand-alone); the resolve's total GPU time in the "
                   "trace is spread over the steps it covers"}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
