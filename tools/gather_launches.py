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

GATHER = "k_embbag_fwd_arange"
PLAN = ("k_bm_", "k_uniq_probe", "k_kept_flags", "k_assign", "k_winner", "k_host_rows", "k_victim", "k_cf_", "k_scan_tops",
        "k_prot_clear", "k_commit", "k_writeback")


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(round(q / 100.0 * (len(v) - 1))))]


def main():
    d, out = sys.argv[1], sys.argv[2]
    nbytes = float(sys.argv[3]) if len(sys.argv) > 3 else 8192 * 26 * (8 * 128 + 16)
    kt = sorted(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(kt)))
    plan = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if any(p in r["Kernel_Name"] for p in PLAN))
    g = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if GATHER in r["Kernel_Name"]]
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
    doc = {"source": kt.split("gpurun_out/")[-1], "kernel": GATHER, "algorithmic_bytes_per_launch": nbytes,
           "all_launches": summ(dur), "launches_not_beside_a_window_plan": summ(quiet),
           "note": "durations are rocprofv3 kernel-trace End - Start; the profiled run is slower per step than the "
                   "un-profiled one (host-bound under the tracer) but a kernel's own duration is not"}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["all_launches"]))
    print(json.dumps(doc["launches_not_beside_a_window_plan"]))


if __name__ == "__main__":
    main()
