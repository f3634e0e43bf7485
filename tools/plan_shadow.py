#!/usr/bin/env python3
"""What does a window plan in the background cost the steps beside it?  Trains config c3 in blocks of 10 steps, launches the
next window's plan after block 12 and prints every block's ms/step next to what the plan thread was doing.

A block is timed on the TRAINING streams only (an event behind `eng.finish()` on the training stream): a device-wide
synchronisation would also wait for the plan's own copies in flight and charge them to the block -- that instrument error sent
round 4 looking for a 75 ms "stall" that was the plan's 4 GB row copy finishing under the timer.

    python tools/plan_shadow.py [--threads 13] [--blocks 60] [--alpha 1.05]
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
    ap.add_argument("--threads", type=int, default=-1)
    ap.add_argument("--blocks", type=int, default=60)
    ap.add_argument("--alpha", type=float, default=1.05)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = 3000
    wl = bench.build_workload("c3", dev=dev, alpha=a.alpha)
    eng, pipe, syn, B = wl["eng"], wl["pipe"], wl["syn"], wl["B"]
    if a.threads > 0:
        pipe.gather_threads = a.threads
    main_stream = torch.cuda.Stream(device=dev, priority=-1)
    torch.cuda.set_stream(main_stream)
    win, nxt_win = syn.window(0, L), syn.window(1, L)
    pipe.plan_window(win)
    pipe._worker.join()
    pipe.commit()
    pipe.plan_window(nxt_win)            # a first background plan pins the staging: not the one we look at
    pipe._worker.join()
    torch.cuda.synchronize()
    pipe._worker, pipe.planned, pipe._exchange = None, None, []
    rs = WindowResolver(eng, win, B)
    j = 0
    print("gather threads %d" % pipe.gather_threads)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for blk in range(a.blocks):
        if blk == 12:
            pipe.plan_window(nxt_win)
        eng.finish()
        e0.record(main_stream)
        t0 = time.perf_counter()
        for _ in range(10):
            idx = win[:, j * B:(j + 1) * B]
            nx = win[:, (j + 1) * B:(j + 2) * B]
            X, T = syn.dense(j)
            eng.step(X, idx, T, j=j + 1, next_idx=nx, res=rs.batch(j), next_res=rs.batch(j + 1), loss_sync=False)
            rs.ensure(j + rs.CH + 2)
            j += 1
        t_issue = time.perf_counter() - t0
        eng.finish()
        e1.record(main_stream)
        e1.synchronize()                 # the training streams only: the plan's copies in flight are not waited for
        bd = pipe._bd or {}
        alive = pipe._worker is not None and pipe._worker.is_alive()
        state = "" if blk < 12 else ("plan thread alive; " if alive else "plan thread done; ") + " ".join(
            "%s=%.0f" % (k, v) for k, v in bd.items() if isinstance(v, float) and v > 0)
        print("block %2d  %.4f ms/step  (issue %.4f)  %s" % (blk, e0.elapsed_time(e1) / 10, t_issue / 10 * 1e3, state))
    eng.finish()
    pipe.close()


if __name__ == "__main__":
    main()
