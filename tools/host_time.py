#!/usr/bin/env python3
"""Where the HOST's time per training step goes (per-rank batch 1024: the thread that issues a step, not the GPU, is within 10 %
of setting the step time).  Runs the bench loop's body on one GPU and accumulates wall-clock around its parts.

    python tools/host_time.py [--batch 1024] [--steps 2000]
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
from cdlrm_amd import _lib  # noqa: E402
from cdlrm_amd.engine import TrainEngine, WindowResolver  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--ops", type=int, default=1, help="1: clock every recorded call of the step's tape (costs ~2 us per step)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = a.steps + 40
    wl = bench.build_workload(a.config, lookahead=L, batch=a.batch, dev=dev)
    eng, pipe, syn, B = wl["eng"], wl["pipe"], wl["syn"], wl["B"]
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    win = syn.window(0, L)
    pipe.plan_window(win)
    if pipe._worker is not None:
        pipe._worker.join()
    pipe.commit()
    rs = WindowResolver(eng, win, B)
    acc = {"replay": 0.0, "taped": 0.0, "step": 0.0, "ensure": 0.0, "args": 0.0}
    pc = time.perf_counter

    orig_replay = _lib.NativeTape.replay

    def replay(self):
        t = pc()
        r = orig_replay(self)
        acc["replay"] += pc() - t
        return r
    _lib.NativeTape.replay = replay
    orig_taped = TrainEngine._step_taped

    def taped(self, *x):
        t = pc()
        r = orig_taped(self, *x)
        acc["taped"] += pc() - t
        return r
    TrainEngine._step_taped = taped

    def run(j0, n):
        for j in range(j0, j0 + n):
            t0 = pc()
            idx = win[:, j * B:(j + 1) * B]
            nxt = win[:, (j + 1) * B:(j + 2) * B]
            X, T = syn.dense(j)
            r0, r1 = rs.batch(j), rs.batch(j + 1)
            t1 = pc()
            eng.step(X, idx, T, j=j + 1, next_idx=nxt, res=r0, next_res=r1, loss_sync=False)
            t2 = pc()
            rs.ensure(j + rs.CH + 2)
            t3 = pc()
            acc["args"] += t1 - t0
            acc["step"] += t2 - t1
            acc["ensure"] += t3 - t2

    run(0, 30)
    torch.cuda.synchronize()
    assert _lib.raw().cdlrm_debug_set(3, 1 if a.ops else 0) == 0
    for k in acc:
        acc[k] = 0.0
    t0 = pc()
    run(30, a.steps)
    t_issue = pc() - t0
    torch.cuda.synchronize()
    dt = pc() - t0
    eng.finish()
    n = a.steps
    print("B = %d: %.1f us/step wall, %.1f us/step to issue" % (B, dt / n * 1e6, t_issue / n * 1e6))
    print("  bench-loop arguments (views, resolver lookups)   %6.1f us" % (acc["args"] / n * 1e6))
    print("  eng.step                                         %6.1f us" % (acc["step"] / n * 1e6))
    print("    of it _step_taped                              %6.1f us" % (acc["taped"] / n * 1e6))
    print("      of it the native replay (one library call)   %6.1f us" % (acc["replay"] / n * 1e6))
    print("  resolver.ensure                                  %6.1f us" % (acc["ensure"] / n * 1e6))
    if a.ops:
        _lib.raw().cdlrm_debug_set(3, 0)
        best = None
        for tp in eng._tapes.values():
            if tp["native"] is not None:
                ot = tp["native"].op_times()
                if best is None or sum(o[2] for o in ot) > sum(o[2] for o in best):
                    best = ot
        if best:
            print("the most replayed tape, per replay: op, lane, us in the call, us waiting for another lane's op")
            lane_sum = {}
            for k, (name, lane, calls, call_us, wait_us, max_us) in enumerate(best):
                print("  %3d  lane %d  %-28s %6.2f  %6.2f   (longest %.0f)" % (k, lane, name.replace("cdlrm_", ""), call_us, wait_us, max_us))
                c = lane_sum.setdefault(lane, [0.0, 0.0, 0])
                c[0] += call_us; c[1] += wait_us; c[2] += 1
            for lane, (c, w, k) in sorted(lane_sum.items()):
                print("  lane %d: %d calls, %.1f us in calls, %.1f us waiting" % (lane, k, c, w))


if __name__ == "__main__":
    main()
