#!/usr/bin/env python3
"""Baseline fidelity: is the oracle -- the CPU restatement bench.py times as
`cpu_baseline` (kind "port") -- a stand-in for the reference in COST as well as in results?

Runs only in the build container (imports /root/reference exactly as tools/make_golden.py does; the GPU box never runs it).
For each shape, the imported reference (Prefetcher.process_batch_slice -> CacheEmbeddings -> eviction_manager ->
Embedding_Table_Cache_Group.forward -> DLRM_Net -> BCELoss -> backward -> the two SGD steps, world size 1, everything on the
CPU) and the oracle's trainer (oracle/cdlrm_oracle.py: OracleTrainer.refill / .step) train on the SAME batches from the same
seeds; the loss trajectories are compared (1e-5) and the wall time is taken per phase:

    refill  = window scan + host row gather + insert / evict + write-back, per window
    step    = tag probe + aux fill + EmbeddingBag + MLPs + interaction + loss + backward + both SGD steps, per iteration

    PYTHONDONTWRITEBYTECODE=1 python tools/time_reference_vs_oracle.py [--threads 8] [--json profiles/rNN_reference_vs_oracle.json]

Shapes: c1 (BASELINE configs[0]: 8 tables x 10 k rows, D = 16, B = 128, L = 32, cache 2 k x 4-way) and a c2-like shape
(26 Kaggle cardinalities capped at 2 M rows, D = 32, B = 2048, L = 20, cache 50 k x 8-way: BASELINE.md section 2's shape with
the host tables cut to 1.1 GB so that both sides fit the container beside each other).
"""
import argparse
import json
import os
import queue
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

KAGGLE = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
          5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]

SHAPES = {
    "c1": dict(ln_emb=[10000] * 8, m_spa=16, ln_bot=[13, 64, 16], top=[64, 32, 1], cache_size=2000, ways=4, B=128, L=32,
               nbatch=128, seed=123, lr=0.1, lr_emb=0.3, alpha=1.2),
    "c2-like": dict(ln_emb=[min(n, 2000000) for n in KAGGLE], m_spa=32, ln_bot=[13, 512, 256, 32], top=[512, 256, 1],
                    cache_size=50000, ways=8, B=2048, L=20, nbatch=40, seed=7, lr=0.1, lr_emb=0.3, alpha=1.1),
}


def make_batches(c, zipf_indices):
    rng = np.random.RandomState(c["seed"] + 1)
    T, B = len(c["ln_emb"]), c["B"]
    out = []
    for _ in range(c["nbatch"]):
        X = torch.from_numpy(rng.rand(B, c["ln_bot"][0]).astype(np.float32))
        lS_i = torch.stack([torch.from_numpy(zipf_indices(rng, c["ln_emb"][k], B, c["alpha"])) for k in range(T)])
        lS_o = torch.arange(B, dtype=torch.int64).repeat(T, 1)
        Tt = torch.from_numpy(np.round(rng.rand(B, 1)).astype(np.float32))
        out.append((X, lS_o, lS_i, Tt))
    return out


def time_reference(G, mods, c, batches):
    M, MD, CM, QR = mods
    rank = G.RankCpu(0)
    T, L = len(c["ln_emb"]), c["L"]
    np.random.seed(c["seed"])
    torch.manual_seed(c["seed"])
    eg = MD.Embedding_Table_Group(c["m_spa"], np.array(c["ln_emb"]))
    nf = T + 1
    ln_top = np.array([c["m_spa"] + nf * (nf - 1) // 2] + list(c["top"]))
    np.random.seed(c["seed"])
    torch.manual_seed(c["seed"])
    cg = MD.Embedding_Table_Cache_Group(c["m_spa"], np.array(c["ln_emb"]), c["cache_size"], c["B"], c["ways"])
    dl = MD.DLRM_Net(np.array(c["ln_bot"]), ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0)
    loss_fn = torch.nn.BCELoss(reduction="mean")
    opt_m = torch.optim.SGD(dl.parameters(), lr=c["lr"])
    opt_e = torch.optim.SGD(cg.parameters(), lr=c["lr_emb"])
    losses, t_refill, t_step = [], [], []
    for j, (X, lS_o, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            t0 = time.perf_counter()
            win = torch.cat([b[2] for b in batches[j:j + L]], dim=1)
            rows, uniqs, maps = CM.Prefetcher.process_batch_slice(win, eg)
            fifo = queue.Queue()
            torch.manual_seed(5000 + j)
            M.CacheEmbeddings(rows, uniqs, maps, cg, fifo, rank)
            evq = queue.Queue()
            evq.put(fifo.get())
            aff = os.sched_getaffinity(0)
            # (the eviction manager applies the write-back and then sits in `eviction_fifo.get(timeout)` until the timeout
            #  ends it, cache_manager.py:57-64: that wait is not work -- a short timeout, taken off the clock)
            EV_TIMEOUT = 0.05
            CM.Prefetcher.eviction_manager(eg, evq, False, min(aff), EV_TIMEOUT)
            os.sched_setaffinity(0, aff)
            t_refill.append(time.perf_counter() - t0 - EV_TIMEOUT)
        t0 = time.perf_counter()
        lookups, _ = cg(lS_o, lS_i, eg, rank)
        Z = dl(X, lookups)
        E = loss_fn(Z, Tt)
        opt_m.zero_grad()
        opt_e.zero_grad()
        E.backward()
        opt_e.step()
        opt_m.step()
        t_step.append(time.perf_counter() - t0)
        losses.append(float(E.detach()))
    return np.array(losses), t_refill, t_step, ln_top


def time_oracle(c, batches, ln_top):
    from oracle import cdlrm_oracle as O
    L = c["L"]
    np.random.seed(c["seed"])
    torch.manual_seed(c["seed"])
    tr = O.OracleTrainer(c["ln_emb"], c["m_spa"], np.array(c["ln_bot"]), ln_top, cache_size=c["cache_size"],
                         num_ways=c["ways"], mini_batch_size=c["B"], lr=c["lr"], lr_embeds=c["lr_emb"], lookahead=L,
                         table_agg_freq=10 ** 9, seed=c["seed"])
    t_refill, t_step = [], []
    for j, (X, lS_o, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            t0 = time.perf_counter()
            torch.manual_seed(5000 + j)
            tr.refill(torch.cat([b[2] for b in batches[j:j + L]], dim=1))
            t_refill.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        tr.step(j, X, lS_o, lS_i, Tt)
        t_step.append(time.perf_counter() - t0)
    return np.array([l[0] for l in tr.losses]), t_refill, t_step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--shapes", default="c1,c2-like")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    if not os.path.isdir("/root/reference"):
        sys.exit("tools/time_reference_vs_oracle.py runs in the build container only (/root/reference is not here)")
    import make_golden as G
    mods = G.import_reference()
    out = {"threads": None, "host_cores": os.cpu_count(), "torch": torch.__version__, "shapes": {}}
    for name in a.shapes.split(","):
        c = SHAPES[name]
        threads = 1 if name == "c1" else a.threads         # (c1's tensors are tiny: more threads are slower)
        torch.set_num_threads(threads)
        batches = make_batches(c, G.zipf_indices)
        res = {}
        ln_top = None
        for side in ("reference", "oracle", "reference", "oracle"):          # interleaved, two rounds each
            if side == "reference":
                losses, tr_, ts_, ln_top = time_reference(G, mods, c, batches)
            else:
                losses, tr_, ts_ = time_oracle(c, batches, ln_top)
            r = res.setdefault(side, dict(refill_ms=[], step_ms=[], losses=None))
            r["refill_ms"].append(float(np.median(tr_)) * 1e3)
            r["step_ms"].append(float(np.median(ts_[len(ts_) // 4:])) * 1e3)
            r["losses"] = losses
        rel = float(np.max(np.abs(res["oracle"]["losses"] - res["reference"]["losses"]) / np.abs(res["reference"]["losses"])))
        row = {"threads": threads, "iterations": c["nbatch"], "lookahead": c["L"], "batch": c["B"],
               "loss_max_rel_diff": rel}
        for side in ("reference", "oracle"):
            r = res[side]
            st, rf = min(r["step_ms"]), min(r["refill_ms"])
            row[side] = {"step_ms": st, "refill_ms_per_window": rf, "ms_per_it_incl_refill": st + rf / c["L"],
                         "samples_per_s": c["B"] / ((st + rf / c["L"]) * 1e-3)}
        row["oracle_over_reference_wall"] = row["oracle"]["ms_per_it_incl_refill"] / row["reference"]["ms_per_it_incl_refill"]
        out["shapes"][name] = row
        print("%-8s threads %d  loss max rel diff %.2e" % (name, threads, rel))
        for side in ("reference", "oracle"):
            r = row[side]
            print("   %-9s step %8.3f ms/it   refill %9.2f ms/window   -> %8.3f ms/it incl. refill = %9.0f samples/s"
                  % (side, r["step_ms"], r["refill_ms_per_window"], r["ms_per_it_incl_refill"], r["samples_per_s"]))
        print("   oracle / reference wall time: %.3f" % row["oracle_over_reference_wall"])
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
