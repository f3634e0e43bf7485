"""cdlrm_amd.data_loader_terabyte.DataLoader against the reference's loader (tests/golden/criteo_loader.npz, made by
tools/make_golden.py from data_loader_terabyte.DataLoader over three tiny day files): same batches, same order, same
length, for the train / val / test splits, with and without drop_last_batch."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture()
def day_files(golden, tmp_path):
    g = golden("criteo_loader")
    for day in range(len(g["sizes"])):
        np.savez(os.path.join(tmp_path, "day_%d_reordered.npz" % day), X_int=g["xi_%d" % day], X_cat=g["xc_%d" % day],
                 y=g["y_%d" % day])
    np.savez(os.path.join(tmp_path, "day_day_count.npz"), total_per_file=g["sizes"])
    return g, str(tmp_path)


@pytest.mark.parametrize("name,days,split,drop", [("train", [0, 1, 2], "train", False),
                                                  ("train_drop", [0, 1, 2], "train", True), ("val", [2], "val", False),
                                                  ("test", [1, 2], "test", False)])
def test_criteo_day_loader_matches_reference(day_files, name, days, split, drop):
    from cdlrm_amd.data_loader_terabyte import DataLoader
    g, d = day_files
    ld = DataLoader("day", d, days, int(g["B"]), max_ind_range=int(g["max_ind_range"]), split=split, drop_last_batch=drop)
    batches = list(ld)
    assert len(ld) == int(g[name + "_len"])
    assert len(batches) == int(g[name + "_nb"])
    assert [b[3].shape[0] for b in batches] == g[name + "_sizes"].tolist()
    assert torch.equal(torch.cat([b[0] for b in batches]), torch.from_numpy(g[name + "_X"]))
    assert torch.equal(torch.cat([b[2] for b in batches], dim=1), torch.from_numpy(g[name + "_lS_i"]))
    assert torch.equal(torch.cat([b[3] for b in batches]), torch.from_numpy(g[name + "_T"]))
    assert torch.equal(batches[-1][1], torch.from_numpy(g[name + "_lS_o_last"]))
    X, lS_o, lS_i, T = batches[0]
    assert X.dtype == torch.float32 and lS_i.dtype == torch.int64 and lS_o.dtype == torch.int64 and T.shape[1] == 1
