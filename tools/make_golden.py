#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (lkp411/cDLRM).

Runs only in the build container (needs /root/reference, read-only).  Nothing from the reference's
source travels: the outputs are data (inputs + the reference's outputs for them), committed as small
.npz fixtures.  The GPU box never runs this script.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--only NAME]

Oracle hygiene: torch.set_num_threads(1) (duplicate-write determinism),
torch.manual_seed(s) immediately before every CacheEmbeddings call, prefetch distance 0, eviction
write-back applied synchronously.
"""
import argparse
import math
import os
import queue
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.modules["setproctitle"] = types.SimpleNamespace(setproctitle=lambda s: None)
    import cache_manager as CM  # noqa
    import main_no_ddp as M  # noqa
    import model_no_ddp as MD  # noqa
    from tricks import qr_embedding_bag as QR  # noqa
    return M, MD, CM, QR


class RankCpu(str):
    """A rank object that is the CPU device for `.to(rank)` / `device=rank` and compares equal to its
    integer rank id for `if rank == 0` / `if i == rank` (main_no_ddp.py:208, 255)."""

    def __new__(cls, rid=0):
        o = str.__new__(cls, "cpu")
        o.rid = rid
        return o

    def __eq__(self, other):
        if isinstance(other, int):
            return other == self.rid
        return str.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = str.__hash__


class SampleRecorder:
    """Records, for every Categorical.sample() the reference makes, the Exp(1) draw `q` it consumed
    and the ways it returned."""

    def __init__(self):
        self.records = []
        self._orig = torch.distributions.Categorical.sample

    def __enter__(self):
        rec = self

        def patched(dist_self, sample_shape=torch.Size()):
            st = torch.get_rng_state()
            w = rec._orig(dist_self, sample_shape)
            st_after = torch.get_rng_state()
            torch.set_rng_state(st)
            q = torch.empty(dist_self.probs.shape, dtype=torch.float32)
            if q.numel() > 0:
                q.exponential_(1)
            assert torch.equal(torch.get_rng_state(), st_after), "q draw does not mirror Categorical.sample"
            if q.numel() > 0:
                assert torch.equal(torch.argmax(dist_self.probs / q, -1), w)
            rec.records.append((q.clone(), w.clone()))
            return w

        torch.distributions.Categorical.sample = patched
        return self

    def __exit__(self, *a):
        torch.distributions.Categorical.sample = self._orig


def zipf_indices(rng, n, size, alpha=1.1):
    """Skewed indices in [0, n): a permuted Zipf so hot rows are not the low ids."""
    r = rng.zipf(alpha, size=size).astype(np.int64)
    return (r * 2654435761 % n).astype(np.int64)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(path, **conv)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------------------------------


def g_isprime(M, MD, CM, QR):
    tab = np.array([bool(MD.isPrime(n)) for n in range(1, 5000)], dtype=np.uint8)
    cg = MD.Embedding_Table_Cache_Group.__new__(MD.Embedding_Table_Cache_Group)
    ins = [4, 5, 8, 100, 2000, 2048, 10240, 50000, 150000, 500000]
    outs = [MD.Embedding_Table_Cache_Group.find_next_prime(cg, c) for c in ins]
    save("isprime", isprime_1_4999=tab, next_prime_in=np.array(ins), next_prime_out=np.array(outs))


def build_ref_cache_group(MD, m_spa, ln_emb, cache_size, aux, ways, zero=False):
    cg = MD.Embedding_Table_Cache_Group(m_spa, np.array(ln_emb), cache_size, aux, ways)
    if zero:
        for e in cg.emb_l:
            e.weight.data.zero_()
    return cg


def build_ref_host(MD, m_spa, ln_emb, rows_fn=None):
    eg = MD.Embedding_Table_Group(m_spa, np.array(ln_emb))
    if rows_fn is not None:
        for k, e in enumerate(eg.emb_l):
            e.weight.data = rows_fn(k, e.weight.data.shape)
    return eg


def g_appendix_a(M, MD, CM, QR):
    """one table n=50, D=2, P=5, 2 ways, aux 4; two windows + one probe."""
    rank = RankCpu(0)
    eg = build_ref_host(MD, 2, [50], lambda k, s: torch.stack(
        [torch.arange(50, dtype=torch.float32), 0.5 * torch.arange(50, dtype=torch.float32)], 1))
    cg = build_ref_cache_group(MD, 2, [50], 5, 4, 2, zero=True)
    out = {"host": eg.emb_l[0].weight.data.clone(), "P": np.array(cg.cache_sizes)}
    wins = [[1, 3, 6, 8, 11, 13, 6, 1], [1, 16, 21, 3, 26, 4]]
    for w, raw in enumerate(wins):
        sl = torch.tensor([raw], dtype=torch.int64)
        rows, uniqs, maps = CM.Prefetcher.process_batch_slice(sl, eg)
        fifo = queue.Queue()
        torch.manual_seed(w)
        with SampleRecorder() as rec:
            M.CacheEmbeddings(rows, uniqs, maps, cg, fifo, rank)
        ev = fifo.get()
        out.update({f"w{w}_raw": np.array(raw), f"w{w}_uniq": uniqs[0], f"w{w}_q": rec.records[0][0],
                    f"w{w}_way": rec.records[0][1], f"w{w}_occ": cg.occupancy_tables[0].clone(),
                    f"w{w}_weight": cg.emb_l[0].weight.data.clone(), f"w{w}_ev_idx": ev[0][0],
                    f"w{w}_ev_rows": ev[0][1], f"w{w}_map_shape": np.array(maps[0].shape)})
    lS_i = torch.tensor([[1, 16, 7, 26]], dtype=torch.int64)
    lS_o = torch.tensor([[0, 1, 2, 3]], dtype=torch.int64)
    ly, cgi = cg(lS_o, lS_i, eg, rank)
    out.update(fwd_lS_i=lS_i, fwd_lS_o=lS_o, fwd_ly=ly[0].detach(), fwd_idx=cgi[0],
               fwd_weight=cg.emb_l[0].weight.data.clone())
    save("appendix_a", **out)


def g_cache_windows(M, MD, CM, QR, name, ln_emb, m_spa, cache_size, ways, B, L, nwin, seed, alpha):
    """Consecutive windows of CacheEmbeddings with hits / full sets / evictions / contested slots,
    eviction write-back applied between windows, plus a forward probe after each window."""
    rank = RankCpu(0)
    rng = np.random.RandomState(seed)
    np.random.seed(seed)
    eg = build_ref_host(MD, m_spa, ln_emb)
    torch.manual_seed(seed)
    cg = build_ref_cache_group(MD, m_spa, ln_emb, cache_size, B, ways)
    T = len(ln_emb)
    out = dict(ln_emb=np.array(ln_emb), m_spa=m_spa, cache_size=cache_size, ways=ways, B=B, L=L,
               nwin=nwin, seed=seed, cache_sizes=np.array(cg.cache_sizes),
               P=np.array(cg.max_cache_size))
    for k in range(T):
        out[f"host0_{k}"] = eg.emb_l[k].weight.data.clone()
        out[f"weight0_{k}"] = cg.emb_l[k].weight.data.clone()
    for w in range(nwin):
        win = torch.stack([torch.from_numpy(
            zipf_indices(rng, ln_emb[k], L * B, alpha) if alpha > 0 else rng.randint(0, ln_emb[k], L * B).astype(np.int64))
            for k in range(T)])
        rows, uniqs, maps = CM.Prefetcher.process_batch_slice(win, eg)
        fifo = queue.Queue()
        torch.manual_seed(1000 + w)
        with SampleRecorder() as rec:
            M.CacheEmbeddings(rows, uniqs, maps, cg, fifo, rank)
        ev = fifo.get()
        # synchronous write-back with the reference's own eviction_manager body
        evq = queue.Queue()
        evq.put(ev)
        aff = os.sched_getaffinity(0)
        CM.Prefetcher.eviction_manager(eg, evq, False, min(aff), 1)
        os.sched_setaffinity(0, aff)
        out[f"w{w}_win"] = win
        out[f"w{w}_qseed"] = 1000 + w
        for k in range(T):
            out[f"w{w}_uniq_{k}"] = uniqs[k]
            out[f"w{w}_rows_{k}"] = rows[k]
            out[f"w{w}_q_{k}"] = rec.records[k][0]
            out[f"w{w}_way_{k}"] = rec.records[k][1]
            out[f"w{w}_occ_{k}"] = cg.occupancy_tables[k].clone()
            out[f"w{w}_weight_{k}"] = cg.emb_l[k].weight.data.clone()
            out[f"w{w}_ev_idx_{k}"] = ev[k][0]
            out[f"w{w}_ev_rows_{k}"] = ev[k][1]
            out[f"w{w}_host_{k}"] = eg.emb_l[k].weight.data.clone()
        # forward probe: a batch drawn from this window (mostly hits) plus some never-seen ids
        lS_i = win[:, rng.randint(0, L * B, B)].clone()
        for k in range(T):
            lS_i[k, :max(1, B // 16)] = torch.from_numpy(rng.randint(0, ln_emb[k], max(1, B // 16)))
        lS_o = torch.arange(B, dtype=torch.int64).repeat(T, 1)
        ly, cgi = cg(lS_o, lS_i, eg, rank)
        out[f"w{w}_fwd_lS_i"] = lS_i
        for k in range(T):
            out[f"w{w}_fwd_ly_{k}"] = ly[k].detach()
            out[f"w{w}_fwd_idx_{k}"] = cgi[k]
            out[f"w{w}_fwd_weight_{k}"] = cg.emb_l[k].weight.data.clone()
    save(name, **out)


def g_writeback(M, MD, CM, QR):
    for avg in (False, True):
        np.random.seed(7)
        eg = build_ref_host(MD, 4, [40, 9])
        before = [e.weight.data.clone() for e in eg.emb_l]
        rng = np.random.RandomState(3)
        ev = []
        for k, n in enumerate([40, 9]):
            idx = torch.from_numpy(rng.choice(n, 5, replace=False).astype(np.int64))
            emb = torch.from_numpy(rng.randn(5, 4).astype(np.float32))
            # repeated entries carry identical rows (App. A window 2)
            idx = torch.cat([idx, idx[:2]])
            emb = torch.cat([emb, emb[:2]])
            ev.append((idx, emb))
        evq = queue.Queue()
        evq.put(ev)
        aff = os.sched_getaffinity(0)
        CM.Prefetcher.eviction_manager(eg, evq, avg, min(aff), 1)
        os.sched_setaffinity(0, aff)
        out = {}
        for k in range(2):
            out[f"before_{k}"] = before[k]
            out[f"idx_{k}"] = ev[k][0]
            out[f"emb_{k}"] = ev[k][1]
            out[f"after_{k}"] = eg.emb_l[k].weight.data.clone()
        save("writeback_avg%d" % int(avg), **out)


def g_init(M, MD, CM, QR):
    seed = 123
    ln_emb, m_spa = [1460, 583, 305, 24], 16
    ln_bot, ln_top = np.array([13, 64, 16]), np.array([16 + 10, 32, 1])
    np.random.seed(seed)
    torch.manual_seed(seed)
    eg = MD.Embedding_Table_Group(m_spa, np.array(ln_emb))
    out = {"seed": seed, "ln_emb": np.array(ln_emb), "m_spa": m_spa, "ln_bot": ln_bot, "ln_top": ln_top}
    for k, e in enumerate(eg.emb_l):
        out[f"host_head_{k}"] = e.weight.data[:4].clone()
        out[f"host_sum_{k}"] = e.weight.data.double().sum()
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = MD.Embedding_Table_Cache_Group(m_spa, np.array(ln_emb), 100, 32, 4)
    dl = MD.DLRM_Net(ln_bot, ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0)
    for k, e in enumerate(cg.emb_l):
        out[f"cache_head_{k}"] = e.weight.data[:4].clone()
        out[f"cache_shape_{k}"] = np.array(e.weight.shape)
    out["cache_sizes"] = np.array(cg.cache_sizes)
    i = 0
    for l in dl.bot_l:
        if isinstance(l, torch.nn.Linear):
            out[f"bot_w{i}"], out[f"bot_b{i}"] = l.weight.data.clone(), l.bias.data.clone()
            i += 1
    i = 0
    for l in dl.top_l:
        if isinstance(l, torch.nn.Linear):
            out[f"top_w{i}"], out[f"top_b{i}"] = l.weight.data.clone(), l.bias.data.clone()
            i += 1
    save("init", **out)


def g_dense(M, MD, CM, QR):
    """DLRM_Net forward/backward (a-8, a-9, a-10): weights, X, ly -> R, Z, loss, all grads."""
    for itself in (False, True):
        np.random.seed(11)
        torch.manual_seed(11)
        D, Tn, B = 16, 5, 24
        nf = Tn + 1
        nint = (nf * (nf + 1)) // 2 if itself else (nf * (nf - 1)) // 2
        ln_bot, ln_top = np.array([13, 32, D]), np.array([D + nint, 24, 8, 1])
        dl = MD.DLRM_Net(ln_bot, ln_top, "dot", itself, True, -1, ln_top.size - 2, 0.0)
        X = torch.rand(B, 13)
        ly = [torch.randn(B, D, requires_grad=True) for _ in range(Tn)]
        Tt = torch.round(torch.rand(B, 1))
        x = dl.bot_l(X)
        R = dl.interact_features(x, ly)
        Z = dl(X, ly)
        E = torch.nn.BCELoss(reduction="mean")(Z, Tt)
        E.backward()
        out = dict(itself=int(itself), X=X, T=Tt, R=R.detach(), Z=Z.detach(), loss=E.detach(),
                   ln_bot=ln_bot, ln_top=ln_top)
        for k in range(Tn):
            out[f"ly_{k}"], out[f"ly_grad_{k}"] = ly[k].detach(), ly[k].grad
        for nm, seq in (("bot", dl.bot_l), ("top", dl.top_l)):
            i = 0
            for l in seq:
                if isinstance(l, torch.nn.Linear):
                    out[f"{nm}_w{i}"], out[f"{nm}_b{i}"] = l.weight.data.clone(), l.bias.data.clone()
                    out[f"{nm}_gw{i}"], out[f"{nm}_gb{i}"] = l.weight.grad.clone(), l.bias.grad.clone()
                    i += 1
        save("dense_itself%d" % int(itself), **out)


def g_dense_variants(M, MD, CM, QR):
    """The non-default arms of the dense path, through the reference's own DLRM_Net + loss_fn_wrap
    (main_no_ddp.py:212-221, 364-372; model_no_ddp.py:272-316): "cat" interaction, MSE, weighted BCE
    (`--loss-weights`), and the `--loss-threshold` clamp of the prediction."""
    variants = dict(cat_bce=("cat", "bce", "1.0-1.0", 0.0), dot_mse=("dot", "mse", "1.0-1.0", 0.0),
                    dot_wbce=("dot", "wbce", "0.4-2.5", 0.0), dot_bce_thr=("dot", "bce", "1.0-1.0", 0.1),
                    cat_wbce_thr=("cat", "wbce", "1.5-0.7", 0.25))
    for name, (op, loss, lw, thr) in variants.items():
        np.random.seed(13)
        torch.manual_seed(13)
        D, Tn, B = 16, 5, 24
        nf = Tn + 1
        n_in = D + (nf * (nf - 1)) // 2 if op == "dot" else nf * D
        ln_bot, ln_top = np.array([13, 32, D]), np.array([n_in, 24, 8, 1])
        dl = MD.DLRM_Net(ln_bot, ln_top, op, False, True, -1, ln_top.size - 2, thr)
        X = torch.rand(B, 13)
        ly = [torch.randn(B, D, requires_grad=True) for _ in range(Tn)]
        Tt = torch.round(torch.rand(B, 1))
        args = types.SimpleNamespace(loss_function=loss, loss_weights=lw)
        if loss == "mse":
            loss_fn, loss_ws = torch.nn.MSELoss(reduction="mean"), None
        elif loss == "bce":
            loss_fn, loss_ws = torch.nn.BCELoss(reduction="mean"), None
        else:       # main_no_ddp.py:370-372 (np.fromstring there; same parse)
            loss_ws = torch.tensor(np.array([float(s) for s in lw.split("-")], dtype=float))
            loss_fn = torch.nn.BCELoss(reduction="none")
        Z = dl(X, ly)
        E = M.loss_fn_wrap(Z, Tt, loss_fn, args, loss_ws)
        E.backward()
        out = dict(op=op, loss_kind=loss, loss_weights=np.array([float(s) for s in lw.split("-")]), loss_threshold=thr,
                   X=X, T=Tt, Z=Z.detach(), loss=E.detach().to(torch.float64), ln_bot=ln_bot, ln_top=ln_top)
        for k in range(Tn):
            out[f"ly_{k}"], out[f"ly_grad_{k}"] = ly[k].detach(), ly[k].grad
        for nm, seq in (("bot", dl.bot_l), ("top", dl.top_l)):
            i = 0
            for l in seq:
                if isinstance(l, torch.nn.Linear):
                    out[f"{nm}_w{i}"], out[f"{nm}_b{i}"] = l.weight.data.clone(), l.bias.data.clone()
                    out[f"{nm}_gw{i}"], out[f"{nm}_gb{i}"] = l.weight.grad.clone(), l.bias.grad.clone()
                    i += 1
        save("dense_" + name, **out)


def g_random_data(M, MD, CM, QR):
    """The reference's random front end (dlrm_data_pytorch.py:551-684, 752-805): three consecutive batches of a
    RandomDataset (seed reset on the access to item 0), non-fixed and fixed bag sizes."""
    import dlrm_data_pytorch as DP
    out = {}
    for tag, fixed in (("var", False), ("fix", True)):
        ln_emb = np.array([60, 7, 1500, 3])
        ds = DP.RandomDataset(5, ln_emb, 0, 3, 12, 6, fixed, 1, True, "random", "", False, reset_seed_on_access=True,
                              rand_seed=31)
        out[tag + "_ln_emb"] = ln_emb
        for j in range(3):
            X, lS_o, lS_i, T = ds[j]
            out[f"{tag}_X{j}"], out[f"{tag}_T{j}"] = X, T
            for k in range(len(ln_emb)):
                out[f"{tag}_o{j}_{k}"], out[f"{tag}_i{j}_{k}"] = lS_o[k], lS_i[k]
    save("random_data", **out)


def g_synthetic_data(M, MD, CM, QR):
    """The reference's `synthetic` front end (dlrm_data_pytorch.py:808-1129): two recorded traces profiled into their
    stack-distance distributions (trace_profile + the module's main, :1086-1117), written with write_dist_to_file, then
    batches of RandomDataset(data_generation="synthetic") over them -- non-fixed and fixed bag sizes, with and without
    padding -- and a stand-alone trace_generate_lru / trace_generate_rand run."""
    import collections
    import operator
    import tempfile
    import dlrm_data_pytorch as DP
    rng = np.random.RandomState(77)
    out = {}
    ln_emb = np.array([40, 9])
    traces = [(rng.zipf(1.4, 300) % 40).astype(np.uint64).tolist(), (rng.randint(0, 9, 120)).astype(np.uint64).tolist()]
    td = "/tmp/cdlrm_synth_golden"        # no letter "j" anywhere in the path: the reference replaces every one of them
    os.makedirs(td, exist_ok=True)
    if True:
        for pad in (False, True):
            for i, trace in enumerate(traces):
                _, sds, uniq = DP.trace_profile(trace, pad)
                sds.reverse()
                uniq.reverse()
                l = len(sds)
                dc = sorted(collections.Counter(sds).items(), key=operator.itemgetter(0))
                list_sd = [x for x, _ in dc]
                cumm_sd = []
                for q, (_, k) in enumerate(dc):
                    cumm_sd.append(k / float(l) if q == 0 else cumm_sd[q - 1] + (k / float(l)))
                # file name template: every "j" becomes the table number
                # (ints: under numpy 2 str([np.uint64(3)]) is "np.uint64(3)", which the reference's reader cannot parse;
                #  the numpy 1.x it was written for prints "3")
                DP.write_dist_to_file(os.path.join(td, "dist%d_%d.log" % (int(pad), i)), [int(x) for x in uniq], list_sd,
                                      cumm_sd)
                tag = "p%d_t%d" % (int(pad), i)
                out[tag + "_trace"] = np.array(trace, dtype=np.uint64)
                out[tag + "_uniq"] = np.array(uniq, dtype=np.uint64)
                out[tag + "_list_sd"] = np.array(list_sd)
                out[tag + "_cumm_sd"] = np.array(cumm_sd, dtype=np.float64)
                out[tag + "_file"] = np.frombuffer(open(os.path.join(td, "dist%d_%d.log" % (int(pad), i)), "rb").read(),
                                                   dtype=np.uint8)
            for fixed in (False, True):
                ds = DP.RandomDataset(3, ln_emb, 0, 2, 6, 5, fixed, 1, True, "synthetic",
                                      os.path.join(td, "dist%d_j.log" % int(pad)), pad, reset_seed_on_access=True,
                                      rand_seed=19)
                for b in range(2):
                    X, lS_o, lS_i, T = ds[b]
                    tag = "p%d_f%d_b%d" % (int(pad), int(fixed), b)
                    out[tag + "_X"], out[tag + "_T"] = X, T
                    for k in range(len(ln_emb)):
                        out[tag + "_o%d" % k], out[tag + "_i%d" % k] = lS_o[k], lS_i[k]
        uniq, list_sd, cumm_sd = DP.read_dist_from_file(os.path.join(td, "dist0_0.log"))
        np.random.seed(5)
        out["lru_trace"] = np.array(DP.trace_generate_lru(list(uniq), list_sd, cumm_sd, 200, False), dtype=np.uint64)
        np.random.seed(5)
        out["rand_trace"] = np.array(DP.trace_generate_rand(list(uniq), list_sd, cumm_sd, 200, False), dtype=np.uint64)
    out["ln_emb"] = ln_emb
    save("synthetic_data", **out)


def g_embbag_sgd(M, MD, CM, QR):
    """nn.EmbeddingBag(sum, sparse) backward + optim.SGD step on cache rows (a-7), with repeated
    slots and a multi-hot case."""
    for name, multihot in (("embsgd_onehot", False), ("embsgd_multihot", True)):
        torch.manual_seed(5)
        rng = np.random.RandomState(5)
        rows, D, nb = 200, 8, 48
        E = torch.nn.EmbeddingBag(rows, D, mode="sum", sparse=True)
        w0 = E.weight.data.clone()
        if multihot:
            lens = rng.randint(1, 5, nb)
            offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
            n = int(lens.sum())
        else:
            offs = np.arange(nb, dtype=np.int64)
            n = nb
        slots = rng.randint(0, 30, n).astype(np.int64)  # many repeats
        opt = torch.optim.SGD(E.parameters(), lr=0.3)
        V = E(torch.from_numpy(slots), torch.from_numpy(offs))
        G = torch.randn(nb, D)
        opt.zero_grad()
        V.backward(G)
        opt.step()
        save(name, w0=w0, slots=slots, offsets=offs, V=V.detach(), grad=G, lr=0.3, w1=E.weight.data.clone())


def ref_train(M, MD, CM, QR, *, ln_emb, m_spa, ln_bot, top, cache_size, ways, B, L, nbatch, seed,
              lr, lr_emb, alpha, reseed=True):
    """World-size-1 replay of Run's loop body (main_no_ddp.py:387-415) around the reference's own
    objects: Prefetcher.process_batch_slice -> CacheEmbeddings -> cache_group -> DLRM_Net -> BCELoss
    -> backward -> optimizer_embeds.step -> optimizer_mlps.step.  Schedule: prefetch distance 0,
    synchronous write-back."""
    rank = RankCpu(0)
    T = len(ln_emb)
    np.random.seed(seed)
    torch.manual_seed(seed)
    eg = MD.Embedding_Table_Group(m_spa, np.array(ln_emb))
    nf = T + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2] + list(top))
    # trainer process re-seeds (main_no_ddp.py:335-337)
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = MD.Embedding_Table_Cache_Group(m_spa, np.array(ln_emb), cache_size, B, ways)
    dl = MD.DLRM_Net(np.array(ln_bot), ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0)
    loss_fn = torch.nn.BCELoss(reduction="mean")
    opt_m = torch.optim.SGD(dl.parameters(), lr=lr)
    opt_e = torch.optim.SGD(cg.parameters(), lr=lr_emb)
    rng = np.random.RandomState(seed + 1)
    batches = []
    for j in range(nbatch):
        X = torch.from_numpy(rng.rand(B, ln_bot[0]).astype(np.float32))
        lS_i = torch.stack([torch.from_numpy(zipf_indices(rng, ln_emb[k], B, alpha)) for k in range(T)])
        lS_o = torch.arange(B, dtype=torch.int64).repeat(T, 1)
        Tt = torch.from_numpy(np.round(rng.rand(B, 1)).astype(np.float32))
        batches.append((X, lS_o, lS_i, Tt))
    losses, evs = [], 0
    for j, (X, lS_o, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            win = torch.cat([b[2] for b in batches[j:j + L]], dim=1)
            rows, uniqs, maps = CM.Prefetcher.process_batch_slice(win, eg)
            fifo = queue.Queue()
            if reseed:      # False: ONE generator stream from the trainer's seeding on, as Run itself runs
                torch.manual_seed(5000 + j)
            M.CacheEmbeddings(rows, uniqs, maps, cg, fifo, rank)
            evq = queue.Queue()
            evq.put(fifo.get())
            aff = os.sched_getaffinity(0)
            CM.Prefetcher.eviction_manager(eg, evq, False, min(aff), 1)
            os.sched_setaffinity(0, aff)
        lookups, cgi = cg(lS_o, lS_i, eg, rank)
        Z = dl(X, lookups)
        E = loss_fn(Z, Tt)
        opt_m.zero_grad()
        opt_e.zero_grad()
        E.backward()
        opt_e.step()
        opt_m.step()
        losses.append(float(E.detach()))
    return dict(eg=eg, cg=cg, dl=dl, batches=batches, losses=np.array(losses, dtype=np.float64), ln_top=ln_top)


def g_train_w1(M, MD, CM, QR):
    cfgs = {
        # BASELINE config 1 shape: 8 tables x 10k rows, D=16, B=128, L=32, cache 2k x 4-way
        "train_c1": dict(ln_emb=[10000] * 8, m_spa=16, ln_bot=[13, 64, 16], top=[64, 32, 1], cache_size=2000,
                         ways=4, B=128, L=32, nbatch=96, seed=123, lr=0.1, lr_emb=0.3, alpha=1.2),
        # small, eviction-heavy
        "train_small": dict(ln_emb=[3000, 50, 7, 1200], m_spa=8, ln_bot=[4, 16, 8], top=[16, 1], cache_size=40,
                            ways=4, B=32, L=4, nbatch=40, seed=9, lr=0.1, lr_emb=0.3, alpha=1.3),
        # no re-seeding before the refills: the way choices depend on every draw the trainer made since its
        # seeding (cache-table init, nn.Linear default init) -- pins the generator-stream contract of Run
        "train_stream": dict(ln_emb=[3000, 50, 7, 1200, 40000], m_spa=16, ln_bot=[13, 32, 16], top=[32, 1],
                             cache_size=40, ways=4, B=64, L=4, nbatch=14, seed=11, lr=0.1, lr_emb=0.3, alpha=1.2,
                             reseed=False),
    }
    for name, c in cfgs.items():
        r = ref_train(M, MD, CM, QR, **c)
        out = {k: np.array(v) for k, v in c.items()}
        out["losses"] = r["losses"]
        out["ln_top"] = r["ln_top"]
        for k in range(len(c["ln_emb"])):
            out[f"occ_{k}"] = r["cg"].occupancy_tables[k]
            w = r["cg"].emb_l[k].weight.data
            out[f"weight_sum_{k}"] = w[: c["ways"] * r["cg"].cache_sizes[k]].double().sum()
            out[f"host_sum_{k}"] = r["eg"].emb_l[k].weight.data.double().sum()
        i = 0
        for l in r["dl"].top_l:
            if isinstance(l, torch.nn.Linear):
                out[f"top_w{i}"] = l.weight.data.clone()
                i += 1
        save(name, **out)



def g_train_shapes(M, MD, CM, QR):
    """BASELINE.json configs c2 / c3 / c5 at the SHAPE the bench runs them -- 26 tables with the public Criteo
    cardinalities (capped so the host tables fit this container), the config's embedding width, way count, MLP widths --
    through the reference's own objects (ref_train).  These reach the kernel instantiations the headline bench uses
    (16-lane / 16-way probes, the 26-table gather, D = 128 / 32 rows, the multi-pass slot sort at > 8192 lookups per
    table) that the small fixtures above do not."""
    KAGGLE = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
              5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
    TERABYTE = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208,
                11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]
    cap = lambda c, m: [min(n, m) for n in c]
    cfgs = {
        # c3 (README.md:7): D=128, 16-way, bot 13-512-256-128, top 512-512-256-1; B / L / cache scaled down
        "train_c3shape": dict(ln_emb=cap(TERABYTE, 50000), m_spa=128, ln_bot=[13, 512, 256, 128], top=[512, 512, 256, 1],
                              cache_size=1500, ways=16, B=1024, L=3, nbatch=9, seed=31, lr=0.8, lr_emb=0.8, alpha=1.05),
        # c2: Kaggle cardinalities, D=32, 8-way, B=2048
        "train_c2shape": dict(ln_emb=cap(KAGGLE, 50000), m_spa=32, ln_bot=[13, 512, 256, 32], top=[512, 256, 1],
                              cache_size=1000, ways=8, B=2048, L=4, nbatch=12, seed=32, lr=0.1, lr_emb=0.3, alpha=1.1),
        # c5: more than 8192 lookups per table and step (the slot sort takes its merge passes), 16-way
        "train_c5shape": dict(ln_emb=cap(TERABYTE, 40000), m_spa=128, ln_bot=[13, 512, 256, 128], top=[512, 512, 256, 1],
                              cache_size=4000, ways=16, B=12288, L=2, nbatch=6, seed=33, lr=0.8, lr_emb=0.8, alpha=1.05),
        # c4 (BASELINE configs[3]): embed-dim 256 -- bot 13-512-256-256, a 256 + 351 = 607-wide top input (608-pitch), 26 tables
        # x 16-way, 1024-byte cache rows; the QR operator itself is stand-alone in the reference (g_qr / g_qr_c4)
        "train_c4shape": dict(ln_emb=cap(TERABYTE, 40000), m_spa=256, ln_bot=[13, 512, 256, 256], top=[512, 512, 256, 1],
                              cache_size=1500, ways=16, B=1024, L=3, nbatch=9, seed=34, lr=0.8, lr_emb=0.8, alpha=1.05),
    }
    torch.set_num_threads(1)
    for name, c in cfgs.items():
        r = ref_train(M, MD, CM, QR, **c)
        out = {k: np.array(v) for k, v in c.items()}
        out["losses"] = r["losses"]
        out["ln_top"] = r["ln_top"]
        for k in range(len(c["ln_emb"])):
            occ = r["cg"].occupancy_tables[k]
            assert int(occ.max()) < 2 ** 31
            out[f"occ_{k}"] = occ.to(torch.int32)           # ids < 2^31: int32 keeps the fixture small
            w = r["cg"].emb_l[k].weight.data
            out[f"weight_sum_{k}"] = w[: c["ways"] * r["cg"].cache_sizes[k]].double().sum()
            out[f"host_sum_{k}"] = r["eg"].emb_l[k].weight.data.double().sum()
        i = 0
        for l in r["dl"].top_l:
            if isinstance(l, torch.nn.Linear):
                if l.weight.numel() <= 4096:                  # the narrow last layers only (fixture size)
                    out[f"top_w{i}"] = l.weight.data.clone()
                i += 1
        save(name, **out)


# ---- world-size-2 harness over gloo with the reference's own collective call sites -----------------


def _w2_worker(rid, cfg, occ_shared, host_shared, ret_q):
    M, MD, CM, QR = import_reference()
    import torch.distributed as dist
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(cfg["port"])
    dist.init_process_group("gloo", rank=rid, world_size=2)
    # the legacy *_multigpu collectives no longer exist in torch 2.10: same semantics on 1 tensor
    dist.all_reduce_multigpu = lambda ts, op=dist.ReduceOp.SUM, async_op=False: dist.all_reduce(ts[0], op=op, async_op=async_op)
    dist.broadcast_multigpu = lambda ts, src, async_op=False: dist.broadcast(ts[0], src=src, async_op=async_op)
    rank = RankCpu(rid)
    seed, B, L, T = cfg["seed"], cfg["B"], cfg["L"], len(cfg["ln_emb"])
    lbs = math.ceil(B / 2)
    np.random.seed(seed)
    torch.manual_seed(seed)
    eg = MD.Embedding_Table_Group(cfg["m_spa"], np.array(cfg["ln_emb"]))
    for k in range(T):
        eg.emb_l[k].weight.data = host_shared[k]
    nf = T + 1
    ln_top = np.array([cfg["m_spa"] + nf * (nf - 1) // 2] + list(cfg["top"]))
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = MD.Embedding_Table_Cache_Group(cfg["m_spa"], np.array(cfg["ln_emb"]), cfg["cache_size"], B, cfg["ways"])
    dl = MD.DLRM_Net(np.array(cfg["ln_bot"]), ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0)
    cg.occupancy_tables = occ_shared            # share_occupancy_tables (main_no_ddp.py:295-306)
    loss_fn = torch.nn.BCELoss(reduction="mean")
    opt_m = torch.optim.SGD(dl.parameters(), lr=cfg["lr"])
    opt_e = torch.optim.SGD(cg.parameters(), lr=cfg["lr_emb"])
    rng = np.random.RandomState(seed + 1)
    batches = []
    for j in range(cfg["nbatch"]):
        X = torch.from_numpy(rng.rand(B, cfg["ln_bot"][0]).astype(np.float32))
        lS_i = torch.stack([torch.from_numpy(zipf_indices(rng, cfg["ln_emb"][k], B, cfg["alpha"])) for k in range(T)])
        lS_o = torch.arange(B, dtype=torch.int64).repeat(T, 1)
        Tt = torch.from_numpy(np.round(rng.rand(B, 1)).astype(np.float32))
        batches.append((X, lS_o, lS_i, Tt))
    losses, window = [], []
    for j, (X, lS_o, lS_i, Tt) in enumerate(batches):
        X = X[rid * lbs:(rid + 1) * lbs]
        lS_i_r = lS_i[:, rid * lbs:(rid + 1) * lbs]
        lS_o_r = lS_o[:, :lbs]
        Tt = Tt[rid * lbs:(rid + 1) * lbs]
        if j % L == 0:
            fifo_b, fifo_e = queue.Queue(), queue.Queue()
            if rid == 0:
                win = torch.cat([b[2] for b in batches[j:j + L]], dim=1)
                fifo_b.put(CM.Prefetcher.process_batch_slice(win, eg))
                torch.manual_seed(5000 + j)
            with torch.no_grad():
                reqs = M.load_caches_and_broadcast(cg, fifo_b, fifo_e, rank)
            M.wait_wrap(reqs)
            if rid == 0:
                evq = queue.Queue()
                evq.put(fifo_e.get())
                aff = os.sched_getaffinity(0)
                CM.Prefetcher.eviction_manager(eg, evq, False, min(aff), 1)
                os.sched_setaffinity(0, aff)
            dist.barrier()
        lookups, cgi = cg(lS_o_r, lS_i_r, eg, rank)
        Z = dl(X, lookups)
        E = loss_fn(Z, Tt)
        opt_m.zero_grad()
        opt_e.zero_grad()
        E.backward()
        reqs = M.aggregate_gradients(dl)
        opt_e.step()
        M.wait_wrap(reqs)
        opt_m.step()
        if j > 0 and j % cfg["agg_freq"] == 0:
            idxs = torch.cat(window + [torch.stack(cgi)], dim=1)
            M.broadcast_and_aggregate(cg, idxs, rank, cfg["agg_op"])
            window = []
        else:
            window.append(torch.stack(cgi))
        losses.append(float(E.detach()))
        dist.barrier()
    res = dict(losses=np.array(losses))
    for k in range(T):
        res[f"weight_sum_{k}"] = float(cg.emb_l[k].weight.data[: cfg["ways"] * cg.cache_sizes[k]].double().sum())
    i = 0
    for l in dl.top_l:
        if isinstance(l, torch.nn.Linear):
            res[f"top_w{i}"] = l.weight.data.clone().numpy()
            res[f"top_b{i}"] = l.bias.data.clone().numpy()
            i += 1
    ret_q.put((rid, res))
    dist.barrier()
    dist.destroy_process_group()


def g_train_w2(M, MD, CM, QR):
    import torch.multiprocessing as mp
    only = os.environ.get("GOLDEN_W2_ONLY", "")
    for name, agg_op, agg_freq in (("train_w2_mean", "mean", 3), ("train_w2_freq1", "mean", 1), ("train_w2_max", "max", 2),
                                   ("train_w2_sum", "sum", 4)):
        if only and name != only:
            continue
        cfg = dict(ln_emb=[3000, 50, 7, 1200], m_spa=8, ln_bot=[4, 16, 8], top=[16, 1], cache_size=40, ways=4,
                   B=32, L=4, nbatch=24, seed=9, lr=0.1, lr_emb=0.3, alpha=1.3, agg_op=agg_op, agg_freq=agg_freq,
                   port=29731)
        np.random.seed(cfg["seed"])
        torch.manual_seed(cfg["seed"])
        eg = MD.Embedding_Table_Group(cfg["m_spa"], np.array(cfg["ln_emb"]))
        host = [e.weight.data.clone().share_memory_() for e in eg.emb_l]
        P = MD.Embedding_Table_Cache_Group(cfg["m_spa"], np.array(cfg["ln_emb"]), cfg["cache_size"], 4, cfg["ways"])
        occ = [t.clone().share_memory_() for t in P.occupancy_tables]
        ctx = mp.get_context("spawn")
        rq = ctx.Queue()
        procs = [ctx.Process(target=_w2_worker, args=(r, cfg, occ, host, rq)) for r in range(2)]
        for p in procs:
            p.start()
        got = dict(rq.get(timeout=300) for _ in range(2))
        for p in procs:
            p.join()
        out = {k: np.array(v) for k, v in cfg.items() if k != "port"}
        for r in range(2):
            for k, v in got[r].items():
                out[f"r{r}_{k}"] = v
        for k in range(len(host)):
            out[f"occ_{k}"] = occ[k]
            out[f"host_sum_{k}"] = host[k].double().sum()
        save(name, **out)


def g_qr(M, MD, CM, QR):
    torch.manual_seed(3)
    n, D, c = 40_000_000, 8, 4
    # keep tables small: weights given explicitly via _weight for a *small* category count, plus the
    # index arithmetic alone at n = 4e7 (float32 division quirk above 2**24)
    big = torch.tensor([0, 1, 5, 16777217, 39999999, 33554433, 25000003], dtype=torch.int64)
    out = dict(big_idx=big, big_q=(big / c).long(), big_r=torch.remainder(big, c).long(), c=c)
    ncat = 103
    for op in ("mult", "add", "concat"):
        wq = torch.randn(math.ceil(ncat / c), D)
        wr = torch.randn(c, D)
        E = QR.QREmbeddingBag(ncat, D, c, operation=op, mode="sum", sparse=True, _weight=[wq.clone(), wr.clone()])
        idx = torch.randint(0, ncat, (37,))
        offs = torch.tensor([0, 1, 2, 5, 9, 20, 30])
        V = E(idx, offs)
        G = torch.randn_like(V)
        V.backward(G)
        out.update({f"{op}_wq": wq, f"{op}_wr": wr, f"{op}_idx": idx, f"{op}_offs": offs, f"{op}_V": V.detach(),
                    f"{op}_G": G, f"{op}_gq": E.weight_q.grad.to_dense(), f"{op}_gr": E.weight_r.grad.to_dense()})
    save("qr", **out)


def qr_c4_weights(rows, D, salt):
    """Deterministic table contents for fixtures whose tables are too large to store: w[i, d] = ((37 i + 11 d + salt) mod
    1024) / 1024 - 0.5 -- exact in float32 on any device (tests/test_hip_kernels.py restates the formula)."""
    out = torch.empty(rows, D, dtype=torch.float32)
    d = torch.arange(D, dtype=torch.int64).view(1, -1)
    for r0 in range(0, rows, 1 << 20):
        i = torch.arange(r0, min(rows, r0 + (1 << 20)), dtype=torch.int64).view(-1, 1)
        out[r0:r0 + i.shape[0]] = ((i * 37 + d * 11 + salt) & 1023).to(torch.float32) / 1024.0 - 0.5
    return out


def g_qr_c4(M, MD, CM, QR):
    """The QR operator at BASELINE configs[3]'s table size: the largest Terabyte table (39 884 406 categories, 4 collisions
    -> a 9 971 102-row quotient table), embed-dim 256, lookups mostly above 2**24 where `(input / c).long()` is a float32
    division (tricks/qr_embedding_bag.py:157).  Table contents come from qr_c4_weights (10 GB: not stored); the fixture
    holds indices, outputs and the SPARSE weight gradients."""
    nthreads = torch.get_num_threads()
    torch.set_num_threads(8)
    n, D, c = 39884406, 256, 4
    rows_q = math.ceil(n / c)
    wq = qr_c4_weights(rows_q, D, 5)
    wr = qr_c4_weights(c, D, 901)
    torch.set_num_threads(1)
    rng = np.random.RandomState(44)
    idx = rng.randint(0, n, size=1024).astype(np.int64)
    idx[:8] = [0, 16777217, 33554433, 39884403, 39884405, 25000003, 16777216, 3]
    idx[8:64] = idx[:56]                                   # repeats: gradient rows accumulate
    idx = torch.from_numpy(idx)
    offs = torch.arange(0, 1024, 4, dtype=torch.int64)
    out = dict(n=n, c=c, D=D, idx=idx, offs=offs, q=(idx / c).long(), q_exact=idx // c)
    G = torch.from_numpy(rng.randn(offs.numel(), D).astype(np.float32))
    out["G"] = G
    for op in ("mult", "add"):
        E = QR.QREmbeddingBag(n, D, c, operation=op, mode="sum", sparse=True, _weight=[wq, wr])
        V = E(idx, offs)
        V.backward(G)
        gq = E.weight_q.grad.coalesce()
        out.update({f"{op}_V": V.detach(), f"{op}_gq_rows": gq.indices()[0], f"{op}_gr": E.weight_r.grad.to_dense()})
        if op == "mult":
            out[f"{op}_gq_vals"] = gq.values()
        else:       # (for "add" a quotient row's gradient is a plain sum of G rows: a per-row checksum pins it)
            out[f"{op}_gq_rowsum"] = gq.values().double().sum(dim=1)
        E.weight_q.grad = E.weight_r.grad = None
    assert int((out["q"] != out["q_exact"]).sum()) > 30     # the float32-division quirk is exercised
    save("qr_c4", **out)
    torch.set_num_threads(nthreads)


def g_md(M, MD, CM, QR):
    """Mixed-dimension trick: md_solver on a few size lists, PrEmbeddingBag forward/backward with and without the
    projection, including widths 1 and 2."""
    from tricks import md_embedding_bag as MDT  # noqa
    torch.manual_seed(5)
    out = {}
    cases = [("criteo", [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145], 0.3, 16, None, True),
             ("b_budget", [100, 5000, 70, 900000], 0.25, None, 40000.0, True),
             ("noround", [12, 400, 33000, 7], 0.4, 32, None, False),
             ("alpha0", [50, 60, 70], 0.0, 8, None, True)]
    for name, n, alpha, d0, B, rd in cases:
        d = MDT.md_solver(torch.tensor(n), alpha, d0=d0, B=B, round_dim=rd)
        out[f"solver_{name}_n"] = np.array(n)
        out[f"solver_{name}_alpha"] = alpha
        out[f"solver_{name}_d0"] = -1 if d0 is None else d0
        out[f"solver_{name}_B"] = -1.0 if B is None else B
        out[f"solver_{name}_round"] = int(rd)
        out[f"solver_{name}_d"] = d
    k = torch.tensor([1.0, 2.0, 1.0, 4.0])
    out["solver_k_d"] = MDT.md_solver(torch.tensor([100, 5000, 70, 900000]), 0.25, d0=16, k=k)
    out["solver_k_k"] = k
    for name, ncat, ed, bd in (("proj", 211, 4, 16), ("ident", 97, 8, 8), ("w1", 53, 1, 8), ("w2", 64, 2, 32)):
        E = MDT.PrEmbeddingBag(ncat, ed, bd)
        idx = torch.randint(0, ncat, (41,))
        offs = torch.tensor([0, 1, 1, 4, 9, 20, 33, 41])       # an empty bag inside and at the end
        V = E(idx, offs)
        G = torch.randn_like(V)
        V.backward(G)
        out.update({f"{name}_W": E.embs.weight.detach().clone(), f"{name}_idx": idx, f"{name}_offs": offs,
                    f"{name}_V": V.detach(), f"{name}_G": G, f"{name}_gW": E.embs.weight.grad.to_dense(),
                    f"{name}_base": bd})
        if ed < bd:
            out[f"{name}_P"] = E.proj.weight.detach().clone()
            out[f"{name}_gP"] = E.proj.weight.grad
    save("md", **out)


def g_window_groups(M, MD, CM, QR):
    """Drive the reference's Prefetcher.run() for real on a fake loader and record which batches end
    up in which FIFO entry (a-1)."""
    import threading
    import torch.multiprocessing as mp
    res = {}
    cases = [(14, 3, 2), (13, 3, 2), (7, 2, 1), (12, 4, 1), (5, 8, 2), (9, 2, 2)]
    for ci, (nb, L, cw) in enumerate(cases):
        B, T = 4, 2
        # batch j carries the ids j*B .. j*B+B-1 in table 0, so uniq lists identify the batches
        ld = [(None, None, torch.stack([torch.arange(j * B, (j + 1) * B), torch.arange(j * B, (j + 1) * B) % 7]), None)
              for j in range(nb)]
        np.random.seed(1)
        eg = MD.Embedding_Table_Group(2, np.array([nb * B + 1, 7]))
        eg.share_memory()
        args = types.SimpleNamespace(lookahead=L, mini_batch_size=B, cache_workers=cw, nepochs=1, main_start_core=0,
                                     average_on_writeback=False, eviction_fifo_timeout=1)
        bf, ef, ev = queue.Queue(), mp.Manager().Queue(), mp.Event()
        ev.set()
        pf = CM.Prefetcher(args, eg, bf, ef, ev, ld)
        aff = os.sched_getaffinity(0)
        pf.run()
        os.sched_setaffinity(0, aff)
        groups = []
        while not bf.empty():
            rows, uniqs, maps = bf.get()
            groups.append(sorted(set((uniqs[0] // B).tolist())))
        flat = np.full((len(groups), max(len(g) for g in groups)), -1, dtype=np.int64)
        for i, g in enumerate(groups):
            flat[i, :len(g)] = g
        res[f"case{ci}_cfg"] = np.array([nb, L, cw])
        res[f"case{ci}_groups"] = flat
    save("window_groups", **res)


def g_criteo_loader(M, MD, CM, QR):
    """data_loader_terabyte.DataLoader over three tiny day files (day sizes chosen to hit the file-boundary
    carry-over, a tail of exactly one batch, and the short last batch), train / val / test splits."""
    import tempfile
    import data_loader_terabyte as DL
    rng = np.random.RandomState(31)
    sizes = [23, 14, 17]
    B = 7
    out = dict(sizes=np.array(sizes), B=B, max_ind_range=50)
    with tempfile.TemporaryDirectory() as d:
        for day, n in enumerate(sizes):
            xi = rng.randint(0, 1000, size=(n, 13)).astype(np.int32)
            xc = rng.randint(0, 100000, size=(n, 26)).astype(np.int32)
            y = rng.randint(0, 2, size=n).astype(np.int32)
            np.savez(os.path.join(d, "day_%d_reordered.npz" % day), X_int=xi, X_cat=xc, y=y)
            out["xi_%d" % day], out["xc_%d" % day], out["y_%d" % day] = xi, xc, y
        np.savez(os.path.join(d, "day_day_count.npz"), total_per_file=np.array(sizes))
        cases = dict(train=([0, 1, 2], "train", False), train_drop=([0, 1, 2], "train", True), val=([2], "val", False),
                     test=([1, 2], "test", False))
        for name, (days, split, drop) in cases.items():
            ld = DL.DataLoader("day", d, days, B, max_ind_range=50, split=split, drop_last_batch=drop)
            batches = list(ld)
            out[name + "_len"] = len(ld)
            out[name + "_nb"] = len(batches)
            out[name + "_sizes"] = np.array([b[3].shape[0] for b in batches])
            out[name + "_X"] = torch.cat([b[0] for b in batches])
            out[name + "_lS_i"] = torch.cat([b[2] for b in batches], dim=1)
            out[name + "_T"] = torch.cat([b[3] for b in batches])
            out[name + "_lS_o_last"] = batches[-1][1]
    save("criteo_loader", **out)


GENS = dict(criteo_loader=g_criteo_loader, isprime=g_isprime, appendix_a=g_appendix_a, writeback=g_writeback, init=g_init, dense=g_dense,
            dense_variants=g_dense_variants, random_data=g_random_data, synthetic_data=g_synthetic_data,
            embbag_sgd=g_embbag_sgd, train_w1=g_train_w1, train_shapes=g_train_shapes, train_w2=g_train_w2, qr=g_qr, qr_c4=g_qr_c4, md=g_md,
            window_groups=g_window_groups)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.set_num_threads(1)
    M, MD, CM, QR = import_reference()
    gens = dict(GENS)
    gens["cache_windows_small"] = lambda *m: g_cache_windows(
        *m, name="cache_windows_small", ln_emb=[3000, 50, 7, 1200], m_spa=8, cache_size=40, ways=4, B=16, L=4,
        nwin=4, seed=21, alpha=1.2)
    gens["cache_windows_uniform"] = lambda *m: g_cache_windows(
        *m, name="cache_windows_uniform", ln_emb=[5000, 300], m_spa=4, cache_size=100, ways=8, B=32, L=8,
        nwin=3, seed=22, alpha=0.0)
    for name, fn in gens.items():
        if a.only and a.only != name:
            continue
        print("==", name)
        fn(M, MD, CM, QR)


if __name__ == "__main__":
    main()
