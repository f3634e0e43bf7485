"""The reference's random data front end (`--data-generation=random`, its CLI default): uniform multi-hot bags.

Mirrors dlrm_data_pytorch.py:551-684 (RandomDataset, collate_wrapper_random, make_random_data_and_loader) and
:752-805 (generate_random_output_batch, generate_uniform_input_batch): same numpy draws in the same order, so a seeded
run sees the reference's batches (pinned by tests/golden/random_data.npz).  A batch is
(X fp32 [n, m_den], lS_o int64 [T, n], lS_i list of T int64 tensors of DIFFERENT lengths, T fp32 [n, 1]): every bag holds
1 .. num_indices_per_lookup distinct, sorted indices (np.unique), so tables are ragged against each other.
"""
import numpy as np
import torch
from numpy import random as ra


def generate_random_output_batch(n, num_targets, round_targets=False):
    """dlrm_data_pytorch.py:752-760."""
    P = ra.rand(n, num_targets).astype(np.float32)
    if round_targets:
        P = np.round(P).astype(np.float32)
    return torch.tensor(P)


def generate_uniform_input_batch(m_den, ln_emb, n, num_indices_per_lookup, num_indices_per_lookup_fixed):
    """dlrm_data_pytorch.py:763-805.  Draw order: the dense block, then per table, per bag: (the bag size unless fixed),
    the bag's indices."""
    Xt = torch.tensor(ra.rand(n, m_den).astype(np.float32))
    offsets_per_table, indices_per_table = [], []
    for size in ln_emb:
        offs = np.empty(n, dtype=np.int64)
        chunks = []
        offset = 0
        for b in range(n):
            if num_indices_per_lookup_fixed:
                group = np.int64(num_indices_per_lookup)
            else:
                r = ra.random(1)
                group = np.int64(np.round(max([1.0], r * min(size, num_indices_per_lookup))))
            r = ra.random(group)
            bag = np.unique(np.round(r * (size - 1)).astype(np.int64))      # duplicates removed, sorted
            offs[b] = offset
            chunks.append(bag)
            offset += bag.size
        offsets_per_table.append(torch.from_numpy(offs))
        indices_per_table.append(torch.from_numpy(np.concatenate(chunks) if chunks else np.empty(0, dtype=np.int64)))
    return Xt, offsets_per_table, indices_per_table


class RandomDataset(torch.utils.data.Dataset):
    """dlrm_data_pytorch.py:551-646: one item = one whole batch, generated on access; the numpy seed is reset on the
    access to item 0 (`reset_seed_on_access`), which is how the trainer-side and Prefetcher-side loaders see the same
    stream."""

    def __init__(self, m_den, ln_emb, data_size, num_batches, mini_batch_size, num_indices_per_lookup,
                 num_indices_per_lookup_fixed, num_targets=1, round_targets=False, data_generation="random",
                 trace_file="", enable_padding=False, reset_seed_on_access=False, rand_seed=0):
        if data_generation != "random":
            raise SystemExit("ERROR: --data-generation=" + data_generation + " is not supported")
        nbatches = int(np.ceil((data_size * 1.0) / mini_batch_size))
        if num_batches != 0:
            nbatches = num_batches
            data_size = nbatches * mini_batch_size
        self.m_den, self.ln_emb = m_den, ln_emb
        self.data_size, self.num_batches, self.mini_batch_size = data_size, nbatches, mini_batch_size
        self.num_indices_per_lookup = num_indices_per_lookup
        self.num_indices_per_lookup_fixed = num_indices_per_lookup_fixed
        self.num_targets, self.round_targets = num_targets, round_targets
        self.reset_seed_on_access, self.rand_seed = reset_seed_on_access, rand_seed

    def __getitem__(self, index):
        if isinstance(index, slice):
            return [self[i] for i in range(index.start or 0, index.stop or len(self), index.step or 1)]
        if self.reset_seed_on_access and index == 0:
            np.random.seed(self.rand_seed)
        n = min(self.mini_batch_size, self.data_size - (index * self.mini_batch_size))
        X, lS_o, lS_i = generate_uniform_input_batch(self.m_den, self.ln_emb, n, self.num_indices_per_lookup,
                                                     self.num_indices_per_lookup_fixed)
        T = generate_random_output_batch(n, self.num_targets, self.round_targets)
        return X, lS_o, lS_i, T

    def __len__(self):
        return self.num_batches


def collate_wrapper_random(list_of_tuples):
    X, lS_o, lS_i, T = list_of_tuples[0]
    return X, torch.stack(lS_o), lS_i, T


def make_random_data_and_loader(args, ln_emb, m_den):
    """dlrm_data_pytorch.py:658-684."""
    train_data = RandomDataset(m_den, ln_emb, args.data_size, args.num_batches, args.mini_batch_size,
                               args.num_indices_per_lookup, args.num_indices_per_lookup_fixed, 1, args.round_targets,
                               args.data_generation, getattr(args, "data_trace_file", ""),
                               getattr(args, "data_trace_enable_padding", False), reset_seed_on_access=True,
                               rand_seed=args.numpy_rand_seed)
    train_loader = torch.utils.data.DataLoader(train_data, batch_size=1, shuffle=False,
                                               num_workers=getattr(args, "num_workers", 0),
                                               collate_fn=collate_wrapper_random, pin_memory=False, drop_last=False)
    return train_data, train_loader
