"""Mixed-dimension trick (reference: tricks/md_embedding_bag.py:20-78) on the HIP kernels.

    md_solver(n, alpha, d0=None, B=None, round_dim=True, k=None)   per-table embedding widths by the alpha-power rule
    PrEmbeddingBag(num_embeddings, embedding_dim, base_dim)         EmbeddingBag(sum) of width embedding_dim + a bias-free
                                                                    projection to base_dim (identity when equal)

Same names, argument order and results as the reference's module.  The pooled lookup is `cdlrm_bag_fwd/bwd`
(csrc/qr.hip), the projection the FP32-MFMA Linear kernels.  Stand-alone operator: the reference parses `--md-flag` but
builds its host tables without it (main_no_ddp.py:612-621), and cache rows have ONE width, so mixed widths have no cached
semantics to match -- neither here.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib, ops


def pow_2_round(dims):
    """Nearest power of two (in log2), md_embedding_bag.py:56-57."""
    return 2 ** torch.round(torch.log2(dims.type(torch.float)))


def alpha_power_rule(n, alpha, d0=None, B=None):
    """Widths d_i = lambda * n_i^(-alpha), rounded, at least 1 (md_embedding_bag.py:39-53).  lambda from the baseline
    width d0 of the smallest table, or from the parameter budget B."""
    nf = n.type(torch.float)
    if d0 is not None:
        lamb = d0 * (nf[0] ** alpha)
    elif B is not None:
        lamb = B / torch.sum(nf ** (1 - alpha))
    else:
        raise ValueError("Must specify either d0 or B")
    d = torch.ones(len(n)) * lamb * (nf ** (-alpha))
    for i in range(len(d)):
        if i == 0 and d0 is not None:
            d[i] = d0
        else:
            d[i] = 1 if d[i] < 1 else d[i]
    return torch.round(d).type(torch.long)


def md_solver(n, alpha, d0=None, B=None, round_dim=True, k=None):
    """md_embedding_bag.py:20-36: tables sorted by size (ascending), sizes divided by the query counts k, alpha-power
    rule, optional rounding to powers of two.  NOTE (reference behaviour, kept): the result is in SORTED-table order."""
    n, indices = torch.sort(n)
    k = k[indices] if k is not None else torch.ones(len(n))
    d = alpha_power_rule(n.type(torch.float) / k, alpha, d0=d0, B=B)
    if round_dim:
        d = pow_2_round(d)
    return d


class _BagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, offsets, weight):
        dev = weight.device
        if dev.type != "cuda":
            raise RuntimeError("cdlrm_amd: PrEmbeddingBag needs the MI355X (no CPU path)")
        idx = input.to(dev, torch.int64).contiguous()
        if offsets is None:
            assert input.dim() == 2, "offsets may be omitted only for 2-D input"
            offsets = torch.arange(0, idx.numel(), input.shape[1], device=dev)
        off = offsets.to(dev, torch.int64).contiguous()
        nb, D = off.numel(), weight.shape[1]
        out = torch.empty(nb, D, dtype=torch.float32, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().cdlrm_bag_fwd(idx.data_ptr(), off.data_ptr(), idx.numel(), nb, weight.data_ptr(),
                                            weight.shape[0], D, out.data_ptr(), err.data_ptr(), s))
        if int(err.item()) != 0:
            raise IndexError("PrEmbeddingBag: index outside the table")
        ctx.save_for_backward(idx, off)
        ctx.shape = tuple(weight.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, off = ctx.saved_tensors
        gW = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().cdlrm_bag_bwd(idx.data_ptr(), off.data_ptr(), idx.numel(), off.numel(), g.contiguous().data_ptr(),
                                            ctx.shape[0], ctx.shape[1], gW.data_ptr(), s))
        return None, None, gW


class _ProjFn(torch.autograd.Function):
    """y = x W^T (no bias) on the Linear kernels."""

    @staticmethod
    def forward(ctx, x, W):
        x = x.contiguous()
        Wc = W.contiguous()
        y = torch.empty(x.shape[0], W.shape[0], dtype=torch.float32, device=x.device)
        ops.linear_fwd(x, Wc, None, y, ops.ACT["none"])
        ctx.save_for_backward(x, Wc)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g = g.contiguous().clone()
        dX = torch.empty_like(x)
        dW = torch.empty_like(W)
        work = ops.linear_bwd_work(x.shape[0], W.shape[0], W.shape[1], x.device)
        ops.linear_bwd(x, W, None, g, dX, dW, None, ops.ACT["none"], work)
        return dX, dW


class _Embs(nn.Module):
    """`embs` of the reference's PrEmbeddingBag: holds `.weight` like nn.EmbeddingBag does."""

    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))

    def forward(self, input, offsets=None, per_sample_weights=None):
        if per_sample_weights is not None:
            raise NotImplementedError("per-sample weights are not implemented by the HIP operator")
        return _BagFn.apply(input, offsets, self.weight)


class _Proj(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))

    def forward(self, x):
        return _ProjFn.apply(x, self.weight)


class PrEmbeddingBag(nn.Module):
    def __init__(self, num_embeddings, embedding_dim, base_dim):
        super().__init__()
        self.embs = _Embs(num_embeddings, embedding_dim)
        torch.nn.init.xavier_uniform_(self.embs.weight)
        if embedding_dim < base_dim:
            self.proj = _Proj(embedding_dim, base_dim)
            torch.nn.init.xavier_uniform_(self.proj.weight)
        elif embedding_dim == base_dim:
            self.proj = nn.Identity()
        else:
            raise ValueError("Embedding dim " + str(embedding_dim) + " > base dim " + str(base_dim))

    def forward(self, input, offsets=None, per_sample_weights=None):
        return self.proj(self.embs(input, offsets=offsets, per_sample_weights=per_sample_weights))
