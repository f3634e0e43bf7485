"""Is the fused gather + interaction forward bound by where its rows come from?  c3, one GPU: the kernel alone on the last
batch's slot ids, (a) back to back (every row of the previous launch still in L2 / the 256 MiB Infinity Cache: 113 MB of rows),
(b) with 1 GB of scratch copied in between (cold), (c) behind a loads-only pass over the same rows.  Launch-attached HIP events.

    python tools/gather_warm_cold.py
"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cdlrm_amd import ops  # noqa: E402
from cdlrm_amd.engine import WindowResolver  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = 64
    wl = bench.build_workload("c3", lookahead=L, dev=dev)
    eng, pipe, syn, B, cg = wl["eng"], wl["pipe"], wl["syn"], wl["B"], wl["cg"]
    win = syn.window(0, L)
    pipe.plan_window(win)
    if pipe._worker is not None:
        pipe._worker.join()
    pipe.commit()
    rs = WindowResolver(eng, win, B)
    for j in range(12):
        idx = win[:, j * B:(j + 1) * B]
        nxt = win[:, (j + 1) * B:(j + 2) * B]
        X, T = syn.dense(j)
        eng.step(X, idx, T, j=j + 1, next_idx=nxt, res=rs.batch(j), next_res=rs.batch(j + 1), loss_sync=False)
        rs.ensure(j + rs.CH + 2)
    eng.finish()
    torch.cuda.synchronize()
    idx = win[:, 12 * B:13 * B]
    slots, _, _ = ops.embbag_probe(cg.ctx, idx, aux_phase=0)
    buf = eng._buffers(B)
    feat, R = buf["feat"], buf["R"]
    flush = torch.empty(2, 1 << 28, device=dev)

    def run(mode, reps=30):
        pairs = [(ops.TimingEvent(), ops.TimingEvent()) for _ in range(reps)]
        for e0, e1 in pairs:
            if mode == "cold":
                flush[1].copy_(flush[0])
            ops.time_next_gather(cg.ctx, e0, e1)
            ops.gather_interact_fwd(cg.ctx, slots, feat[:, 0, :], eng.itself, R)
        torch.cuda.synchronize()
        us = [a.elapsed_us(b) for a, b in pairs[5:]]
        return float(np.mean(us)), float(np.percentile(us, 10)), float(np.percentile(us, 90))

    from cdlrm_amd import _lib
    ref = None
    for dbg, name in ((2, "single slice"), (0, "double-buffered"), (2, "single slice"), (0, "double-buffered")):
        assert _lib.raw().cdlrm_debug_set(7, dbg) == 0
        for mode in ("warm", "cold"):
            m, p10, p90 = run(mode)
            print("%-16s %-5s  mean %.2f us  p10 %.2f  p90 %.2f" % (name, mode, m, p10, p90), flush=True)
        torch.cuda.synchronize()
        if ref is None:
            ref = R.clone()
        else:
            assert torch.equal(ref, R), "the two forms differ"
    assert _lib.raw().cdlrm_debug_set(7, 0) == 0
    print("bit-identical outputs")
    cg.ctx.check()


if __name__ == "__main__":
    main()
