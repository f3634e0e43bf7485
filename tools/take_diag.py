#!/usr/bin/env python3
"""Development aid: the per-step take (cdlrm_embbag_take) of a real c3 window stand-alone -- how many lookups of a batch
go to aux rows, and what the kernel costs without anything beside it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cdlrm_amd import ops  # noqa: E402
from cdlrm_amd.engine import WindowResolver  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L = int(os.environ.get("L", "3000"))
wl = bench.build_workload("c3", lookahead=L, batch=-1, seed=1, dev=dev, rank=0, world=1, barrier=lambda: None,
                          alpha=float(os.environ.get("ALPHA", "1.05")), max_ind_range=-1)
eng, pipe, syn, B = wl["eng"], wl["pipe"], wl["syn"], wl["B"]
win = syn.window(0, L)
pipe.plan_window(win)
if pipe._worker is not None:
    pipe._worker.join()
torch.cuda.synchronize()
pipe.commit()
torch.cuda.synchronize()
res = WindowResolver(eng, win, B)
res.ensure(40)
torch.cuda.synchronize()
ctx = eng.ctx
T = win.shape[0]
for j in (0, 7, 33):
    ws, wsrc, ev = res.batch(j)
    idx = win[:, j * B:(j + 1) * B]
    slots = torch.empty(T, B, dtype=torch.int32, device=dev)
    ops.embbag_take(ctx, idx, ws, wsrc, slots, aux_phase=0)
    torch.cuda.synchronize()
    first_aux = torch.tensor([eng.cg.cache_sizes[k] * eng.cg.num_ways for k in range(T)], device=dev).view(T, 1)
    naux = int((slots >= first_aux).sum())
    nvict = int(((slots >= first_aux) & (wsrc >= 0)).sum())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        ops.embbag_take(ctx, idx, ws, wsrc, slots, aux_phase=0)
    e0.record()
    for _ in range(20):
        ops.embbag_take(ctx, idx, ws, wsrc, slots, aux_phase=0)
    e1.record()
    torch.cuda.synchronize()
    print("batch %d: %d of %d lookups in aux rows (%d from victim rows), take %.1f us" % (
        j, naux, T * B, nvict, e0.elapsed_time(e1) / 20 * 1e3))
