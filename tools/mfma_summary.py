#!/usr/bin/env python3
"""MFMA utilisation per kernel from ONE rocprofv3 PMC pass of bench.py:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma \
              -- python3 bench.py --no-cpu-baseline --steps 300 --warmup 50
    python tools/mfma_summary.py gpurun_out/pmc_mfma profiles/r01_mfma_pmc.json

SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the chip's 1024 SIMDs (a v_mfma_f32_32x32x2_f32 holds its SIMD's
matrix pipe for 64 cycles = 4096 flop); GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs (MI355X_MICROARCH.md), so a
dispatch's shader cycles are GUI_ACTIVE / 8 and its MFMA utilisation = BUSY / (GUI_ACTIVE / 8 * 1024).  Counter passes
serialise the dispatches, so these are per-kernel figures without the step's cross-stream contention."""
import csv
import glob
import json
import os
import statistics
import sys

SIMDS = 1024
PEAK_CLK_HZ = 2.4e9        # 1024 SIMDs x 64 flop/clk x 2.4 GHz = 157.3 TFLOP/s FP32 matrix peak


def short(n):
    return n.replace("void ", "").split("(")[0]


def main():
    d, out = sys.argv[1], sys.argv[2]
    f = sorted(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
    per = {}
    for r in csv.DictReader(open(f)):
        k = (short(r["Kernel_Name"]), r["Dispatch_Id"])
        e = per.setdefault(k, {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                               "grid": int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    agg, shapes = {}, {}
    for (name, _), e in per.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in e or "GRBM_GUI_ACTIVE" not in e or e["GRBM_GUI_ACTIVE"] <= 0:
            continue
        a = agg.setdefault(name, {"busy": [], "gui": [], "us": []})
        a["busy"].append(e["SQ_VALU_MFMA_BUSY_CYCLES"]); a["gui"].append(e["GRBM_GUI_ACTIVE"]); a["us"].append(e["us"])
        # one template instance serves layers of different sizes (the 512-wide layers and the half-size ones): a launch's
        # busy cycles ARE its problem size (64 cycles per 32x32x2 MFMA), so they split the instance by layer shape
        if name.startswith("k_gemm2"):
            mflop = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 64.0 * 4096.0 / 1e6
            sh = shapes.setdefault((name, e["grid"], round(mflop, -1)), {"busy": [], "us": []})
            sh["busy"].append(e["SQ_VALU_MFMA_BUSY_CYCLES"]); sh["us"].append(e["us"])
    rows = {}
    for name, a in agg.items():
        busy, gui = sum(a["busy"]), sum(a["gui"])
        if busy <= 0:
            continue
        util = busy / (gui / 8.0 * SIMDS)
        flops = busy / 64.0 * 4096.0                       # FP32 32x32x2 MFMA: 64 busy cycles, 4096 flop
        secs = sum(a["us"]) * 1e-6
        rows[name] = {"launches": len(a["us"]), "median_us": statistics.median(a["us"]),
                      # primary: matrix-pipe busy cycles against wall time at the 2.4 GHz peak clock (= TF/s / 157.3)
                      "mfma_util": busy / (secs * PEAK_CLK_HZ * SIMDS),
                      "tflops_during_pass": flops / secs / 1e12,
                      # GRBM_GUI_ACTIVE over-counts on dispatches shorter than ~0.3 ms (the quotient below reads 2.7-4.3
                      # GHz here), so the counter-only ratio under-states the utilisation: kept for reference
                      "mfma_util_vs_gui_active": util,
                      "gui_active_per_second_ghz": gui / 8.0 / secs / 1e9}
    res = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 "
                      "bench.py --no-cpu-baseline --steps 50 --warmup 10 (tools/profile_round.sh; MI355X, 1 GPU, config c3)",
           "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time * 2.4 GHz * 1024 SIMDs); tflops = busy / 64 * 4096 / time; "
                      "mfma_util_vs_gui_active = busy / (GRBM_GUI_ACTIVE / 8 * 1024)",
           "kernels": dict(sorted(rows.items(), key=lambda kv: -kv[1]["mfma_util"])),
           # the LDS-DMA GEMM by layer shape: MFLOP per launch (8192 x 512 x 512 x 2 = 4295), workgroups, launches,
           # median duration IN THE COUNTER PASS (dispatches serialised, ~5 % longer than in the training step), utilisation
           "k_gemm2_by_shape": [
               {"kernel": n, "workgroups": g, "mflop": mf, "launches": len(v["us"]), "median_us": statistics.median(v["us"]),
                "mfma_util": sum(v["busy"]) / (sum(v["us"]) * 1e-6 * PEAK_CLK_HZ * SIMDS),
                "tflops": sum(v["busy"]) / 64.0 * 4096.0 / (sum(v["us"]) * 1e-6) / 1e12}
               for (n, g, mf), v in sorted(shapes.items(), key=lambda kv: (-kv[0][2], kv[0][0]))]}
    json.dump(res, open(out, "w"), indent=1)
    for sh in res["k_gemm2_by_shape"]:
        print("  %-34s %5d MFLOP  %4d workgroups  util %5.1f %%  %6.1f TF/s  median %6.1f us  x%d" % (
            sh["kernel"], sh["mflop"], sh["workgroups"], 100 * sh["mfma_util"], sh["tflops"], sh["median_us"], sh["launches"]))
    for k, v in res["kernels"].items():
        print("%-60s util %5.1f %%  %6.1f TF/s  median %7.1f us  x%d" % (
            k[:60], 100 * v["mfma_util"], v["tflops_during_pass"], v["median_us"], v["launches"]))


if __name__ == "__main__":
    main()
