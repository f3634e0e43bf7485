"""TEST DOUBLE of cdlrm_amd.ops for the CPU multi-rank (gloo) orchestration tests.

It has the same call surface as cdlrm_amd/ops.py, but every "kernel" is the CPU oracle (oracle/cdlrm_oracle.py)
or a plain torch-CPU op.  It exists so that the data-parallel control flow of TrainEngine / WindowPipeline (which
collective runs when, on which rows, the bias-not-reduced quirk, sync-to-rank-0 at a refill, touched-row
bookkeeping) can be exercised with world_size 2 on a machine without a GPU.  It is never importable from the
product package and is installed only by tests through monkeypatching.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import cdlrm_oracle as O

ACT = {"none": 0, "relu": 1, "sigmoid": 2}
HOST_REGISTRY = {}      # fake "device pointer" -> host table tensor


def register_host(tables):
    ptrs = []
    for t in tables:
        key = 1000 + len(HOST_REGISTRY)
        HOST_REGISTRY[key] = t
        ptrs.append(key)
    return ptrs


class CacheCtx:
    def __init__(self, table_rows, cache_sets, dim, num_ways, aux_rows, device, aux_phases=1):
        self.T = len(table_rows)
        self.D, self.ways, self.aux = int(dim), int(num_ways), int(aux_rows)
        self.aux_phases = max(1, int(aux_phases))
        self.table_rows = [int(x) for x in table_rows]
        self.cache_sets = [int(x) for x in cache_sets]
        self.device = torch.device(device)
        self.rows = [self.ways * p + self.aux * self.aux_phases for p in self.cache_sets]
        self.row_base, self.tag_base = [0], [0]
        for k in range(self.T):
            self.row_base.append(self.row_base[-1] + self.rows[k])
            self.tag_base.append(self.tag_base[-1] + self.cache_sets[k] * self.ways)
        self.total_rows, self.total_tags = self.row_base[-1], self.tag_base[-1]
        self.host = None

    def bind_cache(self, tags, weight):
        self.tags, self.weight = tags, weight

    def bind_host_tables(self, ptrs):
        self.host = [HOST_REGISTRY[int(p)] for p in ptrs]

    def occ(self, k):
        return self.tags[self.tag_base[k]:self.tag_base[k + 1]].view(self.cache_sets[k], self.ways)

    def w(self, k):
        return self.weight[self.row_base[k]:self.row_base[k + 1]]

    def bind_victims(self, victims):
        pass                       # the test double always reads the host tables: same values

    def check(self, stream=None):
        pass


class Victims:
    def __init__(self, ctx, cap):
        self.cap = cap


def embbag_probe(ctx, idx, stream=None, aux_phase=0, out=None):
    T, n = idx.shape
    assert 0 <= aux_phase < ctx.aux_phases
    if out is not None:
        slots, miss_pos, miss_count = out
        miss_pos.zero_()
        miss_count.zero_()
    else:
        slots = torch.empty(T, n, dtype=torch.int32)
        miss_pos = torch.zeros(T, n, dtype=torch.int32)
        miss_count = torch.zeros(T, dtype=torch.int32)
    for k in range(T):
        s, mp, mi = O.probe_table(ctx.occ(k), idx[k], ctx.cache_sets[k])
        if mp.numel():
            s[mp] += aux_phase * ctx.aux
            ctx.w(k)[s[mp]] = ctx.host[k][mi]
        slots[k] = s.to(torch.int32)
        miss_pos[k, :mp.numel()] = mp.to(torch.int32)
        miss_count[k] = mp.numel()
    return slots, miss_pos, miss_count


def embbag_fwd(ctx, slots, offsets, out, ld_bag, ld_table, n_bags=None, stream=None):
    for k in range(ctx.T):
        off = torch.arange(slots.shape[1]) if offsets is None else offsets[k]
        out[:, k, :] = torch.nn.functional.embedding_bag(slots[k].long(), ctx.w(k), off, mode="sum")


def embbag_bwd_work(ctx, n, device):
    return {"slots": None}


def embbag_bwd_prepare(ctx, slots, work, stream=None):
    work["slots"] = slots.clone()


def embbag_bwd_apply(ctx, n, offsets, grad, ld_bag, ld_table, lr, work, touched=None, stream=None):
    slots = work["slots"]
    for k in range(ctx.T):
        off = torch.arange(n) if offsets is None else offsets[k]
        O.embbag_bwd_sgd(ctx.w(k), slots[k].long(), off, grad[:, k, :], lr)
        if touched is not None:     # like the kernel: aux rows are never flagged
            sl = slots[k].long()
            touched[ctx.row_base[k] + sl[sl < ctx.ways * ctx.cache_sets[k]]] = 1


class WindowPlan:
    def __init__(self, ctx, max_window, cap_uniq=None, cap_win=None):
        self.ctx = ctx
        self.uniqs = None
        self.cap_uniq = cap_uniq if cap_uniq is not None else sum(min(int(max_window), n) for n in ctx.table_rows)

    def unique(self, idx, stream=None):
        self.uniqs = [torch.from_numpy(np.unique(idx[k].numpy())) for k in range(self.ctx.T)]

    def set_unique(self, uniqs):
        self.uniqs = [u.clone() for u in uniqs]

    def victims(self, victims, stream=None):
        pass

    def probe(self, stream=None):
        self.kept_counts = []
        for k, u in enumerate(self.uniqs):
            occ, P = self.ctx.occ(k), self.ctx.cache_sets[k]
            s = u % P
            eq = occ[s] == u.view(-1, 1)
            hit = eq.any(1)
            avail = torch.ones(occ.shape, dtype=torch.bool)
            avail[s[hit], eq.nonzero(as_tuple=True)[1]] = False
            full = ~avail.any(1)
            self.kept_counts.append(int((~hit & ~full[s]).sum()))

    def offsets(self, stream=None):
        uo = [0]
        ko = [0]
        for k in range(self.ctx.T):
            uo.append(uo[-1] + int(self.uniqs[k].numel()))
            ko.append(ko[-1] + self.kept_counts[k])
        return uo, ko, None

    def assign(self, q=None, seed=0, stream=None):
        self.q = q

    def fetch(self, src_ptrs, by_position, stream=None):
        assert not by_position
        self.src = [HOST_REGISTRY[int(p)] for p in src_ptrs]

    def commit(self, stream=None):
        ctx = self.ctx
        pos = [0]

        def qsrc(M, ways):
            out = self.q[pos[0]:pos[0] + M].reshape(M, ways).clone()
            pos[0] += M
            return out

        self.ev = []
        for k in range(ctx.T):
            rows = self.src[k][self.uniqs[k]]
            d = O.cache_embeddings_table(self.uniqs[k], rows, ctx.occ(k), ctx.w(k), ctx.cache_sets[k], qsrc)
            self.ev.append((d["evict_idx"], d["evict_rows"]))

    def writeback(self, dst_ptrs, average, stream=None):
        O.eviction_writeback([HOST_REGISTRY[int(p)] for p in dst_ptrs], self.ev, average)


def agg_compact(ctx, touched, rows_out, count_out, stream=None):
    nz = touched.nonzero().flatten()
    rows_out[:nz.numel()] = nz
    count_out[0] = nz.numel()
    touched.zero_()


def agg_gather(ctx, rows, count, scale, buf, cap, stream=None, first=0):
    U = max(0, min(int(count[0]) - first, cap))
    buf[:U] = ctx.weight[rows[:U]] / scale if scale != 1.0 else ctx.weight[rows[:U]]


def agg_scatter(ctx, rows, count, buf, cap, stream=None, first=0):
    U = max(0, min(int(count[0]) - first, cap))
    ctx.weight[rows[:U]] = buf[:U]


def interact_fwd(feat, itself, R, stream=None):
    out = O.interact_features(feat[:, 0, :], [feat[:, k, :] for k in range(1, feat.shape[1])], "dot", itself)
    R[:, :out.shape[1]].copy_(out)          # R may carry zero pad columns (row pitch rounded up to 4)


def _act_bwd(d, y, act):
    return d * (y > 0) if act == 1 else d * ((1 - y) * y) if act == 2 else d


def interact_bwd(feat, dR, itself, dfeat, stream=None, x_act=0):
    f = feat.detach().clone().requires_grad_(True)
    out = O.interact_features(f[:, 0, :], [f[:, k, :] for k in range(1, f.shape[1])], "dot", itself)
    out.backward(dR[:, :out.shape[1]])
    g = f.grad.clone()
    g[:, 0, :] = _act_bwd(g[:, 0, :], feat[:, 0, :], x_act)
    dfeat.copy_(g)


def linear_fwd(X, W, b, Y, act, stream=None, alone=False):
    y = torch.nn.functional.linear(X, W, b)
    Y.copy_(torch.relu(y) if act == 1 else torch.sigmoid(y) if act == 2 else y)


def linear_bwd_work(M, N, K, device):
    return torch.empty(1)


def linear_bwd(X, W, Y, dY, dX, dW, db, act, work, stream=None, x_act=0, alone=False):
    if act == 1:
        dY.copy_(dY * (Y > 0))
    elif act == 2:
        dY.copy_(dY * ((1 - Y) * Y))
    if dX is not None:
        dX.copy_(_act_bwd(dY @ W, X, x_act))
    if dW is not None:
        dW.copy_(dY.t() @ X)
    if db is not None:
        db.copy_(dY.sum(0))


def mlp_wgrad_work(M, Ns, Ks, device):
    return torch.empty(1)


class WgradPlan:
    def __init__(self, Xs, dZs, dWs, dbs, work):
        self.Xs, self.dZs, self.dWs, self.dbs = list(Xs), list(dZs), list(dWs), list(dbs)

    def set_x(self, i, x):
        self.Xs[i] = x

    def set_params(self, Ws, bs):
        self.Ws, self.bs = list(Ws), list(bs)


def mlp_wgrad(plan, stream=None, lr=None):
    for i, (x, dz, dw, db) in enumerate(zip(plan.Xs, plan.dZs, plan.dWs, plan.dbs)):
        dw[:, :x.shape[1]].copy_(dz.t() @ x)
        if dw.shape[1] > x.shape[1]:
            dw[:, x.shape[1]:].zero_()
        if db is not None:
            db.copy_(dz.sum(0))
        if lr is not None:
            plan.Ws[i].add_(dw, alpha=-lr)
            if db is not None and plan.bs[i] is not None:
                plan.bs[i].add_(db, alpha=-lr)


def bce_fwd_bwd(Z, target, loss_buf, dZ, stream=None, sigmoid_bwd=False):
    z = Z.detach().clone().requires_grad_(True)
    l = torch.nn.functional.binary_cross_entropy(z, target, reduction="mean")
    l.backward()
    loss_buf[0] = l.detach()
    if dZ is not None:
        dZ.copy_(_act_bwd(z.grad, Z, 2) if sigmoid_bwd else z.grad)


def sgd_step(param, grad, lr, stream=None):
    param.add_(grad, alpha=-lr)


def scale_div(x, divisor, stream=None):
    x.div_(divisor)


LOSS = {"bce": 0, "mse": 1, "wbce": 2}


def _loss(z, target, kind, weights, threshold):
    zc = torch.clamp(z, min=threshold, max=1.0 - threshold) if 0.0 < threshold < 1.0 else z
    name = {0: "bce", 1: "mse", 2: "wbce"}[int(kind)]
    ws = torch.tensor([float(weights[0]), float(weights[1])], dtype=torch.float64)
    return O.loss_fn(zc, target, name, ws), zc


def loss_fwd_bwd(Z, target, loss_buf, dZ, *, kind=0, weights=(1.0, 1.0), threshold=0.0, Zc=None, sigmoid_bwd=False,
                 stream=None):
    z = Z.detach().clone().requires_grad_(True)
    l, zc = _loss(z, target, kind, weights, threshold)
    l.backward()
    loss_buf[0] = l.detach().float()
    if Zc is not None:
        Zc.copy_(zc.detach())
    if dZ is not None:
        dZ.copy_(_act_bwd(z.grad, Z, 2) if sigmoid_bwd else z.grad)


def head_scratch(device):
    return torch.zeros(1)


def head_fwd_bwd(Y, w, bias, target, Z, dZ, dY, loss_buf, scratch, *, x_act=0, kind=0, weights=(1.0, 1.0),
                 threshold=0.0, Zc=None, finish=True, stream=None):
    K = Y.shape[1]
    wrow = w.reshape(-1)[:K]
    pre = Y @ wrow.view(K, 1) + (bias if bias is not None else 0.0)
    Z.copy_(torch.sigmoid(pre).view_as(Z))
    loss_fwd_bwd(Z, target.view_as(Z), loss_buf, dZ, kind=kind, weights=weights, threshold=threshold, Zc=Zc,
                 sigmoid_bwd=True)
    if dY is not None:
        dY[:, :K].copy_(_act_bwd(dZ.view(-1, 1) * wrow.view(1, K), Y, x_act))


def head_finish(scratch, B, loss_buf, stream=None, acc=None):
    if acc is not None:                    # the stand-in's head_fwd_bwd always completes the loss; only the running sums are left
        acc[0] += float(loss_buf[1])
        acc[1] += float(loss_buf[2])


def sgd_step2(param, grad, off0, n0, off1, n1, lr, stream=None):
    param[off0:off0 + n0].add_(grad[off0:off0 + n0], alpha=-lr)
    param[off1:off1 + n1].add_(grad[off1:off1 + n1], alpha=-lr)


def act_bwd(dX, X, act, stream=None):
    dX.copy_(_act_bwd(dX, X, act))
