#!/usr/bin/env python3
"""A/B timing of engine schedule knobs on ONE box in ONE process (box-to-box spread is +-1.5 %, more than most knobs are
worth): builds a bench workload once, then alternates rounds of N steps with the knob set to each value.

    python tools/ab_step.py --attr rows_chain_min --values 4096,1000000000 [--config c3] [--batch 1024] [--steps 300]
"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cdlrm_amd.engine import WindowResolver  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--attr", required=True, help="TrainEngine attribute to switch; debug:<key> = a development toggle "
                                                  "inside the library (cdlrm_debug_set, when the build has one)")
    ap.add_argument("--values", required=True, help="values separated by ';' (python literals)")
    ap.add_argument("--config", default="c3")
    ap.add_argument("--batch", type=int, default=-1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--max-ind-range", type=int, default=-1)
    ap.add_argument("--set", default="", help="other TrainEngine attributes held fixed for the whole run: 'name=value,name=value'")
    a = ap.parse_args()
    vals = [eval(v) for v in a.values.split(";")]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = a.steps * 2 + 8
    wl = bench.build_workload(a.config, lookahead=L, batch=a.batch, dev=dev, max_ind_range=a.max_ind_range)
    eng, pipe, syn, B = wl["eng"], wl["pipe"], wl["syn"], wl["B"]
    for kv in filter(None, a.set.split(",")):
        k_, v_ = kv.split("=")
        assert hasattr(eng, k_), k_
        setattr(eng, k_, eval(v_))
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    win = syn.window(0, L)
    pipe.plan_window(win)
    if pipe._worker is not None:
        pipe._worker.join()
    pipe.commit()
    rs = WindowResolver(eng, win, B)
    pos = [0]

    def run(n):
        for _ in range(n):
            j = pos[0] % (L - 1)
            idx = win[:, j * B:(j + 1) * B]
            nxt = win[:, (j + 1) * B:(j + 2) * B] if j + 2 < L else None
            X, T = syn.dense(j)
            eng.step(X, idx, T, j=j + 1, next_idx=nxt, res=rs.batch(j), next_res=rs.batch(j + 1) if nxt is not None else None,
                     loss_sync=False)
            rs.ensure(j + rs.CH + 2)
            pos[0] += 1
            if pos[0] % (L - 1) == 0:       # wrapped: the resolver's chunks are behind us -- start over
                raise SystemExit("window exhausted: raise --steps margin")

    if a.attr.startswith("debug:"):
        from cdlrm_amd import _lib
        key = int(a.attr.split(":")[1])
        fn = _lib.raw().cdlrm_debug_set

        def set_knob(v):
            torch.cuda.synchronize()
            assert fn(key, int(v)) == 0
    elif a.attr in ("pref_priority", "side_priority"):
        # the priority of the engine's prefetch / weight-gradient stream or of its side stream (numerically lower = more
        # urgent; the training queue runs at -1): a stream of that priority replaces the engine's, recorded tapes are dropped
        from cdlrm_amd import _lib
        made = {}

        def set_knob(v):
            torch.cuda.synchronize()
            if v not in made:
                h = _lib.raw().cdlrm_stream_create(int(v))
                made[v] = torch.cuda.ExternalStream(int(h), device=dev)
            if a.attr == "pref_priority":
                eng.pref = eng.wst = made[v]
            else:
                eng.side = made[v]
            eng._tapes.clear()
            eng._pref = None
    else:
        def set_knob(v):
            setattr(eng, a.attr, v)
    res = {repr(v): [] for v in vals}
    nper = max(20, a.steps // (a.rounds * len(vals)) - 10)
    for v in vals:                           # warm every control path
        set_knob(v)
        run(10)
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for v in vals:
            set_knob(v)
            run(3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(nper)
            torch.cuda.synchronize()
            res[repr(v)].append((time.perf_counter() - t0) / nper * 1e3)
    eng.finish()
    wl["cg"].ctx.check()
    for v in vals:
        xs = res[repr(v)]
        print("%s = %-12r  ms/step: %s   median %.4f  min %.4f" % (a.attr, v, " ".join("%.4f" % x for x in xs),
                                                                 float(np.median(xs)), min(xs)))


if __name__ == "__main__":
    main()
