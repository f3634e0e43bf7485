"""Criteo-shaped synthetic input for the cached training path.

The reference has no Criteo-shaped synthetic generator (RandomDataset is uniform multi-hot,
dlrm_data_pytorch.py:763-805, and random mode is broken in main_no_ddp.py:546/630), so this module only
restates the batch LAYOUT the hot path consumes, exactly as data_loader_terabyte.py:68-87 emits it:

    X    fp32  [B, 13]      dense features
    lS_o int64 [T, B]       = arange(B) per table (one index per bag)   -> passed as None here
    lS_i int64 [T, B]       sparse indices
    T    fp32  [B, 1]       click targets in {0, 1}

Indices are counter-based: lookup p of table k is a pure function of (seed, k, p) (csrc/synth.hip, one launch per
table), so the trainer-side batches and the look-ahead side regenerate identical data independently, in any chunking,
without a second loader (dlrm_data_pytorch.py:465-483 relies on two loaders over the same data for this).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
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

    @staticmethod
    def _mix64(z: np.ndarray) -> np.ndarray:
        """splitmix64 finaliser on uint64 arrays (csrc/synth.hip: synth_mix64)."""
        with np.errstate(over="ignore"):
            z = z + np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def _key(self, k: int) -> int:
        """The key of table k's lookup stream."""
        z = (self.seed * 0x9E3779B97F4A7C15 + (k + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        return int(self._mix64(np.array([z], dtype=np.uint64))[0])

    def _table_indices_host(self, k: int, n: int, first: int, count: int) -> torch.Tensor:
        """The stream of csrc/synth.hip restated in numpy (CPU device: the oracle's baseline run, CPU tests).  Same integers
        wherever float64 pow rounds alike on host and device (a last-place difference can move a rank across an integer
        boundary for a few lookups in a million)."""
        with np.errstate(over="ignore"):
            z = self._mix64(np.uint64(self._key(k)) + np.arange(first, first + count, dtype=np.uint64))
        u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        a = self.alpha
        if a <= 0.0:
            r = np.minimum(np.floor(u * n).astype(np.int64), n - 1)
        else:
            if abs(a - 1.0) < 1e-9:
                x = np.exp(u * np.log(float(n) + 1.0))
            else:
                x = (((float(n) + 1.0) ** (1.0 - a) - 1.0) * u + 1.0) ** (1.0 / (1.0 - a))
            r = np.clip(np.floor(x).astype(np.int64) - 1, 0, n - 1)
            r = ((r.astype(np.uint64) * np.uint64(2654435761) + np.uint64(40503)) % np.uint64(n)).astype(np.int64)
        return torch.from_numpy(r)

    def window(self, w: int, num_batches: int) -> torch.Tensor:
        """Indices of `num_batches` consecutive batches starting at batch w*num_batches: int64 [T, num_batches*B];
        batch j of the window is columns [j*B, (j+1)*B).  Table k's row is lookups [w * num_batches * B, ...) of its
        counter-based stream: the same batches whatever the window / chunk size they are asked for in."""
        count = num_batches * self.B
        first = int(w) * count
        out = torch.empty(len(self.ln_emb), count, dtype=torch.int64, device=self.device)
        if self.device.type != "cuda":
            for k, n in enumerate(self.ln_emb):
                out[k] = self._table_indices_host(k, n, first, count)
            return out
        from . import _lib
        L = _lib.lib()
        st = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            for k, n in enumerate(self.ln_emb):
                _lib.check(L.cdlrm_synth_indices(out[k].data_ptr(), count, first, n, float(self.alpha), self._key(k), st))
        return out

    def dense(self, j: int):
        return self.X_pool[j % self.pool], self.T_pool[j % self.pool]
