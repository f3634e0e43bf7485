#!/usr/bin/env python3
"""Print the launch sequence one recorded training step issues (function, stream) -- the program order the queues see.

    python tools/tape_dump.py [--config c3] [--batch N]
"""
import argparse
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cdlrm_amd.engine import WindowResolver  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--batch", type=int, default=-1)
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = 64
    wl = bench.build_workload(a.config, lookahead=L, batch=a.batch, dev=dev, max_ind_range=-1)
    eng, pipe, syn, B = wl["eng"], wl["pipe"], wl["syn"], wl["B"]
    main_s = torch.cuda.Stream(device=dev, priority=-1)
    torch.cuda.set_stream(main_s)
    win = syn.window(0, L)
    pipe.plan_window(win)
    if pipe._worker is not None:
        pipe._worker.join()
    pipe.commit()
    rs = WindowResolver(eng, win, B)
    names = {main_s.cuda_stream: "main", eng.side.cuda_stream: "side", eng.pref.cuda_stream: "pref"}
    for j in range(a.steps):
        idx = win[:, j * B:(j + 1) * B]
        nxt = win[:, (j + 1) * B:(j + 2) * B]
        X, T = syn.dense(j)
        eng.step(X, idx, T, j=j + 1, next_idx=nxt, res=rs.batch(j), next_res=rs.batch(j + 1), loss_sync=False)
        rs.ensure(j + rs.CH + 2)
    torch.cuda.synchronize()
    key, tape = list(eng._tapes.items())[-1]
    print("tapes:", len(eng._tapes), " last key:", key)
    for fn, args, is_lib in tape["prog"]:
        nm = getattr(fn, "__name__", None) or getattr(fn, "name", repr(fn))
        st = ""
        for v in args:
            val = getattr(v, "value", v)
            if isinstance(val, int) and val in names:
                st = names[val]
        print("  %-34s %s" % (nm, st))
    eng.finish()


if __name__ == "__main__":
    main()
