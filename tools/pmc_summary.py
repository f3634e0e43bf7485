#!/usr/bin/env python3
"""Condense two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each with --kernel-trace --output-format csv) of
`bench.py` into the JSON `bench.py` reads the gather's HBM traffic from:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_gather_pmc.json

Corrections as MI355X_MICROARCH.md prescribes: counters are KB; on gfx950 FETCH_SIZE reports half of the bytes of a
16-B-per-lane coalesced read (doubled here); WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import roofkernel  # noqa: E402


def load(d, ctr, skip):
    f = sorted(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr:
            continue
        per.setdefault(r["Kernel_Name"].replace("void ", "").split("(")[0], []).append(float(r["Counter_Value"]))
    kt = sorted(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur.setdefault(r["Kernel_Name"].replace("void ", "").split("(")[0], []).append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k, v in per.items():
        vv = v[skip:] if len(v) > 2 * skip else v
        dd = dur.get(k, [0.0])
        dd = dd[skip:] if len(dd) > 2 * skip else dd
        out[k] = dict(launches=len(v), median_counter_KB=statistics.median(vv), median_us=statistics.median(dd))
    return out


def main():
    fetch_dir, write_dir, dst = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "c3"
    alpha = float(sys.argv[5]) if len(sys.argv) > 5 else 1.05
    fe, wr = load(fetch_dir, "FETCH_SIZE", 10), load(write_dir, "WRITE_SIZE", 10)
    which = roofkernel.pick(fe)        # the fused gather + interaction kernel when the step ran it, else the stand-alone gather
    gk = [k for k in fe if roofkernel.kind(k) == which][0]
    rd = fe[gk]["median_counter_KB"] * 1024 * 2
    wb = wr[gk]["median_counter_KB"] * 1024
    doc = {
        "command": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python3 "
                   "bench.py --no-cpu-baseline --steps 50 --warmup 10 [--alpha A] (two separate passes, MI355X, 1 GPU)",
        "workload": workload,
        "alpha": alpha,
        "n_gpus": 1,
        "kernel": gk,
        "FETCH_SIZE_KB_median": fe[gk]["median_counter_KB"],
        "WRITE_SIZE_KB_median": wr[gk]["median_counter_KB"],
        "kernel_us_median_during_pmc_pass": fe[gk]["median_us"],
        "correction": "counters are KB; gfx950 FETCH_SIZE = 1/2 of the bytes of a 16-B-per-lane coalesced read -> "
                      "doubled; WRITE_SIZE exact",
        "hbm_read_bytes_per_launch": rd,
        "hbm_write_bytes_per_launch": wb,
        "hbm_bytes_per_launch": rd + wb,
        "kernel_kind": which,
        "note": ("fused gather + interaction: reads = cache rows + slot ids + the dense feature (repeated ids of a Zipf batch are "
                 "served by L2 / Infinity Cache), writes = the interaction rows; the pooled rows are never written"
                 if which == "fused" else
                 "reads are below the algorithmic 109 MB because repeated ids of a Zipf batch are served by L2 / "
                 "Infinity Cache; writes equal the pooled output exactly"),
        "all_kernels": {k: {"FETCH_SIZE": fe[k], "WRITE_SIZE": wr.get(k)} for k in sorted(fe)},
    }
    json.dump(doc, open(dst, "w"), indent=1)
    print(gk, "read %.1f MB + write %.1f MB per launch" % (rd / 1e6, wb / 1e6))


if __name__ == "__main__":
    main()
