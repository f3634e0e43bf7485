#!/usr/bin/env python3
"""Wall time of consecutive blocks of training steps right after set-up (development aid: what the first steps of a short
bench run pay that the steady state does not).

    python tools/step_ramp.py [--config c3] [--block 5] [--blocks 12]
"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cdlrm_amd.engine import WindowResolver  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--block", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--resolver-chunk", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = a.block * a.blocks
    L = n + 8
    wl = bench.build_workload(a.config, lookahead=L, dev=dev)
    eng, pipe, syn, B = wl["eng"], wl["pipe"], wl["syn"], wl["B"]
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    win = syn.window(0, L)
    pipe.plan_window(win)
    if pipe._worker is not None:
        pipe._worker.join()
    torch.cuda.synchronize()
    pipe.commit()
    rs = WindowResolver(eng, win, B, chunk=a.resolver_chunk)
    out = []
    for blk in range(a.blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(blk * a.block, (blk + 1) * a.block):
            idx = win[:, j * B:(j + 1) * B]
            nxt = win[:, (j + 1) * B:(j + 2) * B]
            X, T = syn.dense(j)
            eng.step(X, idx, T, j=j, next_idx=nxt, res=rs.batch(j), next_res=rs.batch(j + 1), loss_sync=False)
            rs.ensure(j + rs.CH + 2)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / a.block * 1e3)
    eng.finish()
    print("ms/step per block of %d: %s" % (a.block, " ".join("%.4f" % x for x in out)))


if __name__ == "__main__":
    main()
