#!/usr/bin/env python3
"""Development aid: one launch of the forward layer chain with the sync words kept, printed afterwards."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdlrm_amd import ops  # noqa

DEV = torch.device("cuda:0")
M = int(os.environ.get("M", "64"))
dims = [480, 512, 512, 256]
g = torch.Generator().manual_seed(1)
X = (torch.randn(M, dims[0], generator=g) * 0.5).to(DEV)
Ws = [(torch.randn(dims[i + 1], dims[i], generator=g) / np.sqrt(dims[i])).to(DEV) for i in range(3)]
bs = [torch.randn(dims[i + 1], generator=g).to(DEV) for i in range(3)]
Yc = [torch.full((M, dims[i + 1]), float("nan"), device=DEV) for i in range(3)]
Yl = [torch.empty(M, dims[i + 1], device=DEV) for i in range(3)]
plan = ops.ChainPlan("fwd", X, [(Ws[i], bs[i], Yc[i], 1) for i in range(3)], M, DEV)
torch.cuda.synchronize()
print("plan built", flush=True)
ops.mlp_chain(plan)
print("launched", flush=True)
import time
time.sleep(3.0)
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    snap = torch.empty(plan.sync.numel(), dtype=torch.int32).pin_memory()
    snap.copy_(plan.sync, non_blocking=True)
s2.synchronize()
sn = snap.numpy()
print("after 3 s: queue heads", sn[0:256:32], "exited", sn[256], "err", sn[257], flush=True)
print("done op0..2:", sn[288:292], sn[352:356], sn[416:420], flush=True)
d0 = 288 + 256
for b in range(int(os.environ.get("CDLRM_CHAIN_GRID", "8"))):
    print("  wg %d: xcd %d  item %d  stage %d  iter %d" % (b, sn[d0 + 4 * b] - 100, sn[d0 + 4 * b + 1], sn[d0 + 4 * b + 2], sn[d0 + 4 * b + 3]), flush=True)
torch.cuda.synchronize()
print("synchronised", flush=True)
s = plan.sync.cpu().numpy()
print("queue heads:", s[0:256:32], "exited", s[256], "err", s[257])
rbs = (M + 31) // 32
for op in range(3):
    print("done op%d:" % op, s[288 + op * 64: 288 + op * 64 + rbs])
cur = X
for i in range(3):
    ops.linear_fwd(cur, Ws[i], bs[i], Yl[i], 1)
    cur = Yl[i]
torch.cuda.synchronize()
for i in range(3):
    bad = ~(Yc[i] == Yl[i])
    print("layer %d: equal %s, mismatching rows %s" % (i, bool(torch.equal(Yc[i], Yl[i])), sorted(set(bad.nonzero()[:, 0].tolist()))[:12]))
