"""QREmbeddingBag (reference: tricks/qr_embedding_bag.py:25-185) on the HIP kernels of csrc/qr.hip.

Same constructor and forward signature as the reference's module (mode="sum" only, which is what the
reference instantiates, model_no_ddp.py:54-55).  Tables: weight_q [ceil(n/c), D], weight_r [c, D];
init uniform_(w, sqrt(1/n)) = U(sqrt(1/n), 1) as in the reference (:152-154, a quirk of nn.init.uniform_'s
positional arguments).  q = (input / c).long() is a float32 true division (:157): kept, it is wrong above 2**24.
The reference never wires this module into the cached training path; neither does this build.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from .. import _lib

_OPS = {"mult": 0, "add": 1, "concat": 2}


class _QRFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, offsets, weight_q, weight_r, num_collisions, op):
        lib = _lib.lib()
        dev = weight_q.device
        if dev.type != "cuda":
            raise RuntimeError("cdlrm_amd: QREmbeddingBag needs the MI355X (no CPU path)")
        idx = input.to(dev, torch.int64).contiguous()
        n = idx.numel()
        if offsets is None:
            assert input.dim() == 2, "offsets may be omitted only for 2-D input"
            offsets = torch.arange(0, n, input.shape[1], device=dev)
        off = offsets.to(dev, torch.int64).contiguous()
        nb, D = off.numel(), weight_q.shape[1]
        out = torch.empty(nb, D * (2 if op == 2 else 1), dtype=torch.float32, device=dev)
        eq = torch.empty(nb, D, dtype=torch.float32, device=dev)
        er = torch.empty(nb, D, dtype=torch.float32, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.cdlrm_qr_embbag_fwd(idx.data_ptr(), off.data_ptr(), n, nb, weight_q.data_ptr(), weight_r.data_ptr(),
                                           weight_q.shape[0], int(num_collisions), D, op, out.data_ptr(), eq.data_ptr(),
                                           er.data_ptr(), err.data_ptr(), s))
        if int(err.item()) != 0:
            raise IndexError("QREmbeddingBag: quotient index outside weight_q (float32 division of a large id?)")
        ctx.save_for_backward(idx, off, eq, er)
        ctx.meta = (weight_q.shape, weight_r.shape, int(num_collisions), op)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, off, eq, er = ctx.saved_tensors
        sq, sr, c, op = ctx.meta
        gq = torch.zeros(sq, dtype=torch.float32, device=g.device)
        gr = torch.zeros(sr, dtype=torch.float32, device=g.device)
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().cdlrm_qr_embbag_bwd(idx.data_ptr(), off.data_ptr(), idx.numel(), off.numel(), eq.data_ptr(),
                                                  er.data_ptr(), g.contiguous().data_ptr(), sq[0], c, sq[1], op,
                                                  gq.data_ptr(), gr.data_ptr(), s))
        return None, None, gq, gr, None, None


class QREmbeddingBag(nn.Module):
    def __init__(self, num_categories, embedding_dim, num_collisions, operation='mult', max_norm=None, norm_type=2.,
                 scale_grad_by_freq=False, mode='mean', sparse=False, _weight=None):
        super().__init__()
        assert operation in ['concat', 'mult', 'add'], 'Not valid operation!'
        self.num_categories = num_categories
        if isinstance(embedding_dim, int) or len(embedding_dim) == 1:
            self.embedding_dim = [embedding_dim, embedding_dim]
        else:
            self.embedding_dim = embedding_dim
        self.num_collisions = num_collisions
        self.operation = operation
        self.max_norm, self.norm_type, self.scale_grad_by_freq = max_norm, norm_type, scale_grad_by_freq
        if self.operation in ('add', 'mult'):
            assert self.embedding_dim[0] == self.embedding_dim[1], 'Embedding dimensions do not match!'
        self.num_embeddings = [int(np.ceil(num_categories / num_collisions)), num_collisions]
        if _weight is None:
            self.weight_q = Parameter(torch.Tensor(self.num_embeddings[0], self.embedding_dim[0]))
            self.weight_r = Parameter(torch.Tensor(self.num_embeddings[1], self.embedding_dim[1]))
            self.reset_parameters()
        else:
            assert list(_weight[0].shape) == [self.num_embeddings[0], self.embedding_dim[0]]
            assert list(_weight[1].shape) == [self.num_embeddings[1], self.embedding_dim[1]]
            self.weight_q = Parameter(_weight[0])
            self.weight_r = Parameter(_weight[1])
        self.mode = mode
        self.sparse = sparse

    def reset_parameters(self):
        nn.init.uniform_(self.weight_q, np.sqrt(1 / self.num_categories))
        nn.init.uniform_(self.weight_r, np.sqrt(1 / self.num_categories))

    def forward(self, input, offsets=None, per_sample_weights=None):
        if self.mode != "sum" or per_sample_weights is not None or self.max_norm is not None:
            raise NotImplementedError("the HIP operator implements mode='sum' without per-sample weights / max_norm "
                                      "(what the reference instantiates, model_no_ddp.py:54-55)")
        if self.embedding_dim[0] != self.embedding_dim[1]:
            raise NotImplementedError("different quotient / remainder dimensions")
        return _QRFn.apply(input, offsets, self.weight_q, self.weight_r, self.num_collisions, _OPS[self.operation])
