#!/usr/bin/env python3
"""Race check "the strong way": the same N training steps from the same initial state under different SCHEDULES (launch-tape
lanes, attached events, ...) must end on the same bits -- losses, dense parameters, cache rows, tags.  Same kernels, same
inputs; only who issues what, and when, differs.

    python tools/race_check.py [--config c3] [--batch 1024] [--steps 1500] [--max-ind-range 2000000]
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

VARIANTS = {
    "default (three lanes, attached events, folded wait, the gather fused into the interaction kernels)": {},
    "two lanes": {"tape_lanes": 2},
    "one lane": {"tape_lanes": 1},
    "recorded events": {"attach_events": False},
    "wait on the training queue": {"fold_top_wait": False},
    "unchained take (round-1 schedule)": {"chain_take": False},
    "chained take at every batch (one aux region)": {"gather_alone_min": 1},
    "two aux regions at every batch": {"gather_alone_min": 1 << 30},
    "slot sort behind the interaction forward": {"sort_after_fwd": True},
    "slot sort behind the interaction forward, chained take": {"sort_after_fwd": True, "gather_alone_min": 1},
    "top weight gradients behind the interaction backward": {"top_wgrad_after": "interacted"},
    "top weight gradients behind the bottom input gradients": {"top_wgrad_after": "bot_dz"},
    "top weight gradients behind the bottom weight gradients": {"top_wgrad_after": "bot_wg"},
    "gather + interaction as two launches (the block written and read back)": {"fuse_gather": False},
    "slot sort per step (no chunk slices, no folded once-only update)": {"sort_chunks": False},
    "chunk slices, once-only slots in the sorted path": {"fuse_once": False},
    "slices of one batch": {"sort_slice": 1},
    "slices of five batches, sorted on the side stream": {"sort_slice": 5, "sort_on": "side"},
    "slices sorted on the side stream, chained take": {"sort_on": "side", "gather_alone_min": 1},
    "slices behind the interaction backward": {"sort_after": "interacted"},
    "slices on a least-priority stream of their own": {"sort_on": "own"},
    "slice sorts 0.2 ms late on their own stream (two aux regions)": {"sort_on": "own", "sort_delay": 400000, "gather_alone_min": 1 << 30},
    "slice sorts 0.2 ms late on their own stream (chained take)": {"sort_on": "own", "sort_delay": 400000, "gather_alone_min": 1},
    "slice sorts 0.2 ms late, slices of one batch, no tape": {"sort_on": "own", "sort_delay": 400000, "sort_slice": 1, "use_tape": False},
    "slice sorts 0.2 ms late on the prefetch stream": {"sort_delay": 400000},
    "python tape": {"native_tape": False},
    "no tape": {"use_tape": False},
}


def run(a, knobs, host):
    dev = torch.device("cuda", 0)
    L = a.steps // 2
    wl = bench.build_workload(a.config, lookahead=L, batch=a.batch, dev=dev, max_ind_range=a.max_ind_range, host=host,
                              cache_init="zeros", write_back=False)
    eng, pipe, syn, B, cg = wl["eng"], wl["pipe"], wl["syn"], wl["B"], wl["cg"]
    eng.tape_lanes_below = 1 << 30          # (multi-lane replay at every batch size here)
    for k, v in knobs.items():
        setattr(eng, k, v)
    own = torch.cuda.Stream(device=dev, priority=-1)
    own.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(own):
        for w in range(2):                                  # two windows: one boundary inside the run
            win = syn.window(w, L)
            pipe.plan_window(win)
            pipe.commit()
            rs = WindowResolver(eng, win, B)
            for j in range(L):
                idx = win[:, j * B:(j + 1) * B]
                nxt = win[:, (j + 1) * B:(j + 2) * B] if j + 1 < L else None
                X, T = syn.dense(w * L + j)
                eng.step(X, idx, T, j=j, next_idx=nxt, res=rs.batch(j), next_res=rs.batch(j + 1) if nxt is not None else None,
                         loss_sync=False)
                rs.ensure(j + rs.CH + 2)
        eng.finish()
    torch.cuda.synchronize()
    cg.ctx.check()
    # (cache rows only: which aux region a batch's miss rows pass through differs between the take schedules)
    rows = sum(cg.emb_l[k].weight.data[: cg.num_ways * cg.cache_sizes[k]].sum(dtype=torch.float64).item()
               for k in range(len(cg.cache_sizes)))
    out = (float(eng._buffers(B)["loss"][0]), eng.param_flat.clone(), rows, cg.tags.clone(), eng.stat_acc.clone())
    del wl, eng, pipe, syn, cg, rs
    import gc
    gc.collect()            # (tapes, resolver rings and plans reference each other: cycles would keep ~10 GB per variant alive)
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--negative", action="store_true", help="negative control of the slice-sort ordering (exit 0 = noticed)")
    ap.add_argument("--config", default="c3")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--max-ind-range", type=int, default=2000000)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    host = bench.build_host_tables(a.config, seed=123, dev=torch.device("cuda", 0), max_ind_range=a.max_ind_range)
    ref, bad = None, 0
    if a.negative:
        # negative control: slice sorts arrive late and NOTHING waits for them -- the run must leave the reference's bits
        ref = run(a, {}, host)
        r = run(a, {"sort_on": "own", "sort_delay": 2000000, "slice_wait": False, "gather_alone_min": 1}, host)
        same = (r[0] == ref[0] and torch.equal(r[1], ref[1]) and r[2] == ref[2] and torch.equal(r[3], ref[3]))
        print("late slices, nobody waits:", "bit-identical (the check is blind)" if same else "DIFFERS (as it must)")
        sys.exit(1 if same else 0)
    for name, knobs in VARIANTS.items():
        r = run(a, knobs, host)
        if ref is None:
            ref = r
            print("%-52s loss %.10f" % (name, r[0]))
            continue
        same = (r[0] == ref[0] and torch.equal(r[1], ref[1]) and r[2] == ref[2] and torch.equal(r[3], ref[3])
                and torch.equal(r[4], ref[4]))
        bad += 0 if same else 1
        print("%-52s loss %.10f  %s" % (name, r[0], "bit-identical" if same else "DIFFERS"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
