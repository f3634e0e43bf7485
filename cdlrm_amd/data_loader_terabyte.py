"""Reader of the pre-processed Criteo day files, mirroring the reference's `data_loader_terabyte.DataLoader`
(data_loader_terabyte.py:19-172): `<dir>/<name>_<day>_reordered.npz` with X_int [n, 13], X_cat [n, 26], y [n] and
`<dir>/<name>_day_count.npz` with total_per_file.  A batch is the tuple the trainer loop consumes,

    (X fp32 [B, 13] = log(X_int + 1),  lS_o int64 [26, B] = arange(B) per row,  lS_i int64 [26, B] = X_cat^T,  T fp32 [B, 1])

(`_transform_features`, :68-87).  Batching quirks kept: batches run across day-file boundaries (the tail of a file is
carried into the first batch of the next), a file's rows are consumed while `start < rows - batch_size` (strict, :115
-- a tail of exactly batch_size rows is carried over too), "test" reads the first half of each file and "val" the
second half (:107-112), and the last short batch is emitted unless drop_last_batch.  Host-side only: no GPU work."""
from __future__ import annotations

import math
import os
from typing import Iterator, Sequence, Tuple

import numpy as np
import torch

Batch = Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]


def transform_features(x_int: np.ndarray, x_cat: np.ndarray, y: np.ndarray, max_ind_range: int) -> Batch:
    """data_loader_terabyte.py:68-87."""
    if max_ind_range > 0:
        x_cat = x_cat % max_ind_range
    X = torch.log(torch.as_tensor(np.asarray(x_int), dtype=torch.float) + 1)
    cat = torch.as_tensor(np.asarray(x_cat), dtype=torch.long)
    T = torch.as_tensor(np.asarray(y), dtype=torch.float32).view(-1, 1)
    B, F = cat.shape[0], cat.shape[1]
    lS_o = torch.arange(B).reshape(1, -1).repeat(F, 1)
    return X, lS_o, cat.t(), T


class DataLoader:
    def __init__(self, data_filename: str, data_directory: str, days: Sequence[int], batch_size: int,
                 max_ind_range: int = -1, split: str = "train", drop_last_batch: bool = False):
        self.data_filename, self.data_directory = data_filename, data_directory
        self.days, self.batch_size, self.max_ind_range = list(days), int(batch_size), int(max_ind_range)
        with np.load(os.path.join(data_directory, data_filename + "_day_count.npz")) as data:
            total = int(sum(data["total_per_file"][np.array(self.days)]))
        self.length = int(np.ceil(total / 2.)) if split in ("test", "val") else total
        self.split, self.drop_last_batch = split, drop_last_batch

    def __len__(self) -> int:
        return self.length // self.batch_size if self.drop_last_batch else math.ceil(self.length / self.batch_size)

    def __iter__(self) -> Iterator[Batch]:
        B = self.batch_size
        carry = None            # rows left over from the previous file(s): (x_int, x_cat, y)
        for day in self.days:
            path = os.path.join(self.data_directory, "%s_%d_reordered.npz" % (self.data_filename, day))
            with np.load(path) as data:
                x_int, x_cat, y = data["X_int"], data["X_cat"], data["y"]
            end, start = y.shape[0], 0
            if self.split in ("test", "val"):
                half = int(np.ceil(end / 2.))
                if self.split == "test":
                    end = half
                else:
                    start = end - half
            while start < end - B:
                take = B - (carry[2].shape[0] if carry is not None else 0)
                sl = slice(start, start + take)
                xi, xc, yy = x_int[sl], x_cat[sl], y[sl]
                if carry is not None:
                    xi, xc, yy = (np.concatenate([carry[0], xi]), np.concatenate([carry[1], xc]),
                                  np.concatenate([carry[2], yy]))
                    carry = None
                if xi.shape[0] != B:
                    raise ValueError("should not happen")
                yield transform_features(xi, xc, yy, self.max_ind_range)
                start += take
            if start != end:
                sl = slice(start, end)
                carry = (x_int[sl], x_cat[sl], y[sl]) if carry is None else (
                    np.concatenate([carry[0], x_int[sl]]), np.concatenate([carry[1], x_cat[sl]]),
                    np.concatenate([carry[2], y[sl]]))
        if not self.drop_last_batch and carry is not None:
            yield transform_features(carry[0], carry[1], carry[2], self.max_ind_range)
