#!/usr/bin/env python3
"""Does a process that has replayed multi-lane launch tapes stay quiet afterwards?  (the tape's helper threads
re-spun after every timed-out wait and burned ~25 % of a core each for the life of the process.)  Trains a few hundred small
steps (three lanes), then measures the process's own CPU time over two idle seconds.

    python tools/idle_cpu_check.py
"""
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
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.build_workload("c2", lookahead=64, batch=1024, dev=dev, max_ind_range=200000)
    eng, pipe, syn, B = wl["eng"], wl["pipe"], wl["syn"], wl["B"]
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    win = syn.window(0, 64)
    pipe.plan_window(win)
    if pipe._worker is not None:
        pipe._worker.join()
    pipe.commit()
    rs = WindowResolver(eng, win, B)
    for j in range(60):
        idx = win[:, j * B:(j + 1) * B]
        nxt = win[:, (j + 1) * B:(j + 2) * B]
        X, T = syn.dense(j)
        eng.step(X, idx, T, j=j + 1, next_idx=nxt, res=rs.batch(j), next_res=rs.batch(j + 1), loss_sync=False)
        rs.ensure(j + rs.CH + 2)
    eng.finish()
    torch.cuda.synchronize()
    lanes = max((t["native"].lanes if t["native"] is not None else 1) for t in eng._tapes.values())
    time.sleep(0.5)
    t0, c0 = time.perf_counter(), time.process_time()
    time.sleep(2.0)
    busy = (time.process_time() - c0) / (time.perf_counter() - t0)
    print("tape lanes %d; idle process CPU use over 2 s: %.1f %% of one core" % (lanes, 100.0 * busy))
    sys.exit(0 if busy < 0.10 else 1)


if __name__ == "__main__":
    main()
