#!/usr/bin/env python3
"""Per-launch durations of the roofline kernel (the cached EmbeddingBag gather) from a rocprofv3 kernel trace, so that
bench.py's `roofline.frac` can be recomputed from the profiler's side:

    python tools/gather_launches.py gpurun_out/prof_c3/stats profiles/r02_c3_gather_launches.json [bytes_per_launch]

Writes p10 / p50 / p90 / mean / min / max over the launches, the same again without the launches that overlapped another
stream's long-running kernel of the look-ahead plan (k_bm_*, k_uniq_probe, k_assign, k_host_rows ...: a plan running
beside the training step), and the implied GB/s for the algorithmic bytes per launch."""
import csv
import glob
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import roofkernel  # noqa: E402
PLAN = ("k_bm_", "k_uniq_probe", "k_kept_flags", "k_assign", "k_winner", "k_host_rows", "k_victim", "k_cf_", "k_scan_tops",
        "k_prot_clear", "k_commit", "k_writeback")


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(round(q / 100.0 * (len(v) - 1))))]


def main():
    d, out = sys.argv[1], sys.argv[2]
    kt = sorted(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(kt)))
    which = roofkernel.pick({r["Kernel_Name"] for r in rows})
    # bytes per launch: argv[3] = lookups per table and launch (default 8192; T = 26, D = 128) -- priced per kernel kind
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    survey_bytes, nbytes = roofkernel.bytes_per_launch(which, B, 26, 128)
    if which == "gather":
        nbytes = survey_bytes       # (the stand-alone gather is priced on the SURVEY's 8D + 16, as in every earlier round)
    plan = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if any(p in r["Kernel_Name"] for p in PLAN))
    g = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if roofkernel.kind(r["Kernel_Name"]) == which]
    g = g[10:] if len(g) > 30 else g                       # warm-up launches
    dur, quiet = [], []
    import bisect
    starts = [p[0] for p in plan]
    for s, e in g:
        us = (e - s) / 1e3
        dur.append(us)
        i = bisect.bisect_right(starts, e)
        overl = any(pe > s for ps, pe in plan[max(0, i - 64):i])
        if not overl:
            quiet.append(us)

    def summ(v):
        if not v:
            return None
        return {"launches": len(v), "mean_us": statistics.mean(v), "p10_us": pct(v, 10), "p50_us": pct(v, 50), "p90_us": pct(v, 90),
                "min_us": min(v), "max_us": max(v), "GBps_at_mean": nbytes / statistics.mean(v) / 1e3,
                "GBps_at_p50": nbytes / pct(v, 50) / 1e3, "frac_of_8TBps_at_mean": nbytes / statistics.mean(v) / 1e3 / 8000.0}
    gname = sorted({r["Kernel_Name"].replace("void ", "").split("(")[0] for r in rows if roofkernel.kind(r["Kernel_Name"]) == which})
    doc = {"source": kt.split("gpurun_out/")[-1], "kernel": gname[0] if gname else None, "kernel_kind": which,
           "algorithmic_bytes_per_launch": nbytes,
           "basis": ("fused gather + interaction: what the kernel moves -- per lookup the row and its slot id (4D + 4), per sample the "
                     "dense feature and the interaction row; the pooled rows are never written (the SURVEY's 8D + 16 per lookup "
                     "= %d bytes would price a write that does not exist)" % survey_bytes) if which == "fused"
                    else "SURVEY 8(d): 8D + 16 per lookup",
           "all_launches": summ(dur), "launches_not_beside_a_window_plan": summ(quiet),
           "note": "durations are rocprofv3 kernel-trace End - Start; the profiled run is slower per step than the "
                   "un-profiled one (host-bound under the tracer) but a kernel's own duration is not"}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["all_launches"]))
    print(json.dumps(doc["launches_not_beside_a_window_plan"]))


if __name__ == "__main__":
    main()
