"""Criteo-shaped synthetic input for the cached training path.

The reference has no Criteo-shaped synthetic generator (RandomDataset is uniform multi-hot,
dlrm_data_pytorch.py:763-805, and random mode is broken in main_no_ddp.py:546/630), so this module only
restates the batch LAYOUT the hot path consumes, exactly as data_loader_terabyte.py:68-87 emits it:

    X    fp32  [B, 13]      dense features
    lS_o int64 [T, B]       = arange(B) per table (one index per bag)   -> passed as None here
    lS_i int64 [T, B]       sparse indices
    T    fp32  [B, 1]       click targets in {0, 1}

Indices are counter-based: window w is a pure function of (seed, w), generated on the device, so the
trainer-side batches and the look-ahead side see identical data without a second loader
(dlrm_data_pytorch.py:465-483 relies on two loaders over the same data for this).
"""
from __future__ import annotations

from typing import List, Sequence

import torch

# public Criteo cardinalities: the reference reads them from *_fea_count.npz
KAGGLE_COUNTS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
                 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
TERABYTE_COUNTS = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208,
                   11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]


class CriteoSynth:
    def __init__(self, ln_emb: Sequence[int], m_den: int, batch_size: int, *, seed: int = 123, alpha: float = 1.05,
                 device="cuda", pool: int = 8):
        self.ln_emb = [int(n) for n in ln_emb]
        self.m_den, self.B, self.seed, self.alpha = int(m_den), int(batch_size), int(seed), float(alpha)
        self.device = torch.device(device)
        g = torch.Generator(device=self.device)
        g.manual_seed(self.seed * 7919 + 17)
        self.X_pool = torch.rand(pool, self.B, self.m_den, generator=g, device=self.device)
        self.T_pool = torch.round(torch.rand(pool, self.B, 1, generator=g, device=self.device))
        self.pool = pool

    def _table_indices(self, g, n: int, count: int) -> torch.Tensor:
        u = torch.rand(count, generator=g, device=self.device, dtype=torch.float64)
        if self.alpha <= 0.0:                      # uniform: worst-case hit rate
            r = torch.floor(u * n).to(torch.int64)
        else:
            a = self.alpha
            if abs(a - 1.0) < 1e-9:
                x = torch.exp(u * torch.log(torch.tensor(float(n + 1), dtype=torch.float64, device=self.device)))
            else:
                top = float(n + 1) ** (1.0 - a) - 1.0
                x = (top * u + 1.0) ** (1.0 / (1.0 - a))
            r = torch.floor(x).to(torch.int64) - 1   # Zipf-like rank in [0, n)
            r.clamp_(0, n - 1)
            r = (r * 2654435761 + 40503) % n         # scatter the hot ranks over the id space
        return r

    def window(self, w: int, num_batches: int) -> torch.Tensor:
        """Indices of `num_batches` consecutive batches starting at batch w*num_batches: int64 [T, num_batches*B];
        batch j of the window is columns [j*B, (j+1)*B)."""
        g = torch.Generator(device=self.device)
        g.manual_seed((self.seed * 1000003 + int(w)) & 0x7FFFFFFFFFFF)
        count = num_batches * self.B
        out = torch.empty(len(self.ln_emb), count, dtype=torch.int64, device=self.device)
        for k, n in enumerate(self.ln_emb):
            out[k] = self._table_indices(g, n, count)
        return out

    def dense(self, j: int):
        return self.X_pool[j % self.pool], self.T_pool[j % self.pool]
