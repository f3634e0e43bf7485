"""`main_no_ddp.Run` (the reference's trainer entry, main_no_ddp.py:324-502) end to end on the MI355X: the loop that a
user of the reference's CLI gets, against the oracle's trainer on the same loader, host tables and seeds -- the loss
printed every iteration (print-freq 1) within 1e-5 relative, the final tag state bit-exact; and the module's
`main()` with the reference's flag spelling on synthetic Criteo-shaped data."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import cdlrm_oracle as O

pytestmark = pytest.mark.gpu

FLAGS = ["--arch-sparse-feature-size=16", "--arch-mlp-bot=13-32-16", "--arch-mlp-top=32-1",
         "--arch-embedding-size=3000-50-7-1200-40000", "--mini-batch-size=64", "--lookahead=4", "--cache-size=40",
         "--num-ways=4", "--loss-function=bce", "--round-targets=True", "--learning-rate=0.1", "--lr-embeds=0.3",
         "--print-freq=1", "--world-size=1", "--numpy-rand-seed=11", "--table-agg-freq=5"]


def _loader(ln_emb, B, nb, seed):
    rng = np.random.RandomState(seed)
    lS_o = torch.arange(B).repeat(len(ln_emb), 1)
    out = []
    for _ in range(nb):
        X = torch.from_numpy(rng.rand(B, 13).astype(np.float32))
        idx = torch.stack([torch.from_numpy((rng.zipf(1.2, size=B).astype(np.int64) * 2654435761 % n)) for n in ln_emb])
        T = torch.from_numpy(np.round(rng.rand(B, 1)).astype(np.float32))
        out.append((X, lS_o, idx, T))
    return out


class _Loader(list):
    pass


def test_run_matches_oracle_trainer(capsys):
    from cdlrm_amd.main_no_ddp import ProcessArgs, Run
    from cdlrm_amd.model_no_ddp import Embedding_Table_Group
    args = ProcessArgs(FLAGS + ["--test-freq=5"])
    ln_emb = np.array([3000, 50, 7, 1200, 40000])
    m_spa, B, L, nb, seed = 16, 64, 4, 14, 11            # 14 batches: three full windows and a short last one
    ln_bot = np.array([13, 32, 16])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2, 32, 1])
    batches = _Loader(_loader(ln_emb, B, nb, 5))
    test_batches = _loader(ln_emb, B, 3, 77)              # the rank-0 test loop (main_no_ddp.py:478-494)
    want_acc, want_auc = [], []
    # oracle
    torch.set_num_threads(1)
    np.random.seed(seed)
    torch.manual_seed(seed)
    host_o = O.init_host_tables([int(n) for n in ln_emb], m_spa)
    otr = O.OracleTrainer([int(n) for n in ln_emb], m_spa, ln_bot, ln_top, cache_size=40, num_ways=4, mini_batch_size=B,
                          lr=0.1, lr_embeds=0.3, lookahead=L, table_agg_freq=5, seed=seed,
                          host_tables=[h.clone() for h in host_o])
    for j, (X, lS_o, idx, T) in enumerate(batches):
        if j % L == 0:
            otr.refill(torch.cat([b[2] for b in batches[j:j + L]], dim=1))
        otr.step(j, X, lS_o, idx, T)
        if (j > 0 and j % 5 == 0) or j == nb - 1:
            ok = tot = 0
            Zs = []
            for Xt, lS_ot, idxt, Tt in test_batches:
                Z = otr.evaluate(Xt, lS_ot, idxt)
                Zs.append(Z)
                ok += int((torch.round(Z) == Tt).sum())
                tot += Tt.shape[0]
            want_acc.append(100 * ok / tot)
            from sklearn.metrics import roc_auc_score
            want_auc.append(roc_auc_score(torch.cat([b[3] for b in test_batches]).numpy().ravel(), torch.cat(Zs).numpy().ravel()))
            # a sample whose score sits within ~1e-5 of 0.5 could round either way on the GPU
            assert min(float((Z - 0.5).abs().min()) for Z in Zs) > 2e-5
    # Run
    eg = Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
    for k in range(len(ln_emb)):
        eg.emb_l[k].weight.data = host_o[k].clone()
    eg.pin()
    capsys.readouterr()
    eng = Run(0, m_spa, ln_emb, ln_bot, ln_top, batches, test_batches, None, None, None, eg, args)
    printed = capsys.readouterr().out
    got = [float(x) for x in re.findall(r"Loss = ([0-9.eE+-]+),", printed)]
    want = np.array([l[0] for l in otr.losses])
    # the reference prints for j > 0 and resets its running sums at every print (main_no_ddp.py:455-477): the first
    # line averages iterations 0 and 1, every later line is one iteration
    assert len(got) == nb - 1
    expect = np.concatenate([[(want[0] + want[1]) / 2], want[2:]])
    np.testing.assert_allclose(np.array(got), expect, rtol=1e-5)
    got_acc = [float(x) for x in re.findall(r"Test accuracy = ([0-9.eE+-]+)%", printed)]
    assert printed.count("Testing at") == 3 and len(want_acc) == 3          # j = 5, 10 and the last iteration
    np.testing.assert_allclose(got_acc, want_acc, rtol=0, atol=1e-9)
    # ... and the AUC of the same scores (ops.roc_auc: device-side rank sum) against sklearn on the oracle's scores
    got_auc = [float(x) for x in re.findall(r"Test AUC = ([0-9.eE+-]+)", printed)]
    np.testing.assert_allclose(got_auc, want_auc, rtol=0, atol=2e-4)        # (192 test samples: one swapped pair moves it 1e-4)
    eng.cg.ctx.check()
    for k in range(len(ln_emb)):
        assert torch.equal(eng.cg.occupancy_tables[k].cpu(), otr.occ[k]), k
        np.testing.assert_allclose(eg.emb_l[k].weight.data.double().sum().item(), otr.host[k].double().sum().item(),
                                   rtol=1e-6, atol=1e-6)


def test_save_model_flushes_cache_and_load_model_resumes(tmp_path, capsys):
    """--save-model: every valid cache row is written to its host row (the reference only writes EVICTED rows back,
    so without the flush the trained values of cached rows would be lost), MLPs + host tables saved; --load-model
    starts a new run from them: its first loss equals a forward pass of the oracle over the flushed state."""
    from cdlrm_amd.main_no_ddp import ProcessArgs, Run
    from cdlrm_amd.model_no_ddp import Embedding_Table_Group
    ln_emb = np.array([3000, 50, 7, 1200, 40000])
    m_spa, B, L, nb, seed = 16, 64, 4, 8, 11
    ln_bot = np.array([13, 32, 16])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2, 32, 1])
    batches = _Loader(_loader(ln_emb, B, nb, 5))
    path = str(tmp_path / "model.pt")
    torch.set_num_threads(1)
    np.random.seed(seed)
    torch.manual_seed(seed)
    host_o = O.init_host_tables([int(n) for n in ln_emb], m_spa)
    otr = O.OracleTrainer([int(n) for n in ln_emb], m_spa, ln_bot, ln_top, cache_size=40, num_ways=4, mini_batch_size=B,
                          lr=0.1, lr_embeds=0.3, lookahead=L, table_agg_freq=5, seed=seed,
                          host_tables=[h.clone() for h in host_o])
    for j, (X, lS_o, idx, T) in enumerate(batches):
        if j % L == 0:
            otr.refill(torch.cat([b[2] for b in batches[j:j + L]], dim=1))
        otr.step(j, X, lS_o, idx, T)
    # oracle flush: W_host[k][tag] = cache row of every valid tag (slot = P*way + set)
    flushed = [h.clone() for h in otr.host]
    for k in range(len(ln_emb)):
        P = otr.cache_sizes[k]
        occ = otr.occ[k]
        for s_, w_ in (occ != -1).nonzero().tolist():
            flushed[k][int(occ[s_, w_])] = otr.weights[0][k][P * w_ + s_]
    eg = Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
    for k in range(len(ln_emb)):
        eg.emb_l[k].weight.data = host_o[k].clone()
    eg.pin()
    Run(0, m_spa, ln_emb, ln_bot, ln_top, batches, None, None, None, None, eg, ProcessArgs(FLAGS + ["--save-model=" + path]))
    saved = torch.load(path)
    for k in range(len(ln_emb)):
        np.testing.assert_allclose(saved["emb"][k].numpy(), flushed[k].numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(eg.emb_l[k].weight.data.numpy(), flushed[k].numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(saved["dlrm"]["top_l.0.weight"].numpy(), otr.top[0][0][0].numpy(), rtol=1e-4, atol=1e-6)
    # resume: a fresh run from the file; its first window is planned on an empty cache over the flushed tables
    eg2 = Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
    for k in range(len(ln_emb)):
        eg2.emb_l[k].weight.data = torch.zeros_like(host_o[k])
    eg2.pin()
    capsys.readouterr()
    Run(0, m_spa, ln_emb, ln_bot, ln_top, _Loader(batches[:3]), None, None, None, None, eg2,
        ProcessArgs(FLAGS + ["--load-model=" + path]))
    got = [float(x) for x in re.findall(r"Loss = ([0-9.eE+-]+),", capsys.readouterr().out)]
    # oracle: same MLPs, flushed tables, empty cache
    otr2 = O.OracleTrainer([int(n) for n in ln_emb], m_spa, ln_bot, ln_top, cache_size=40, num_ways=4, mini_batch_size=B,
                           lr=0.1, lr_embeds=0.3, lookahead=L, table_agg_freq=5, seed=seed,
                           host_tables=[h.clone() for h in flushed])
    otr2.bot[0], otr2.top[0] = otr.bot[0], otr.top[0]
    for j, (X, lS_o, idx, T) in enumerate(batches[:3]):
        if j % L == 0:
            otr2.refill(torch.cat([b[2] for b in batches[:3][j:j + L]], dim=1))
        otr2.step(j, X, lS_o, idx, T)
    want = np.array([l[0] for l in otr2.losses])
    np.testing.assert_allclose(np.array(got), np.concatenate([[(want[0] + want[1]) / 2], want[2:]]), rtol=1e-5)


def test_main_cli_criteo_day_files(tmp_path, capsys):
    """--data-generation=dataset over pre-processed day files in the reference's format (<raw>_<day>_reordered.npz,
    <raw>_day_count.npz, <raw>_fea_count.npz): trains on all days but the last, tests on the last."""
    import os
    from cdlrm_amd import main_no_ddp
    rng = np.random.RandomState(3)
    counts = np.array([900, 40, 7, 300, 1500])
    sizes = [64 * 5 + 9, 64 * 4 + 30, 64 * 2]
    for day, n in enumerate(sizes):
        np.savez(os.path.join(tmp_path, "day_%d_reordered.npz" % day), X_int=rng.randint(0, 500, size=(n, 13)).astype(np.int32),
                 X_cat=np.stack([rng.randint(0, c, size=n) for c in counts], axis=1).astype(np.int32),
                 y=rng.randint(0, 2, size=n).astype(np.int32))
    np.savez(os.path.join(tmp_path, "day_day_count.npz"), total_per_file=np.array(sizes))
    np.savez(os.path.join(tmp_path, "day_fea_count.npz"), counts=counts)
    flags = [f for f in FLAGS if not f.startswith("--arch-embedding-size")]
    main_no_ddp.main(flags + ["--data-generation=dataset", "--raw-data-file=" + os.path.join(tmp_path, "day"),
                              "--device-rng"])
    out = capsys.readouterr().out
    n_train = (sizes[0] + sizes[1]) // 64                      # drop_last_batch=True, batches run across the file boundary
    assert out.count("Epoch 0: Finished") == n_train - 1
    assert out.count("Testing at") == 1 and "Test accuracy = " in out
    losses = [float(x) for x in re.findall(r"Loss = ([0-9.eE+-]+),", out)]
    assert all(np.isfinite(losses))


def test_main_cli_synthetic(capsys):
    """python -m cdlrm_amd.main_no_ddp <reference flags> on Criteo-shaped synthetic data: runs, prints the
    reference's progress line, the loss stays finite and the cache state is consistent."""
    from cdlrm_amd import main_no_ddp
    main_no_ddp.main(FLAGS + ["--data-generation=criteo-synthetic", "--num-batches=21", "--device-rng"])
    out = capsys.readouterr().out
    lines = [l for l in out.splitlines() if l.startswith("Epoch 0: Finished")]
    assert len(lines) == 20
    losses = [float(x) for x in re.findall(r"Loss = ([0-9.eE+-]+),", out)]
    assert all(np.isfinite(losses)) and 0.3 < losses[-1] < 1.0
    assert "Caching overhead" in lines[0] and "Train Acc" in lines[0]


def test_run_random_multihot_matches_oracle_trainer(capsys):
    """--data-generation=random (the reference CLI's default front end: uniform multi-hot bags, ragged tables) with the
    CLI's default loss (mse) through `Run`: printed loss per iteration and final tags against the oracle's trainer fed
    the same RandomDataset batches."""
    from cdlrm_amd import dlrm_data_pytorch as DP
    from cdlrm_amd.main_no_ddp import ProcessArgs, Run
    from cdlrm_amd.model_no_ddp import Embedding_Table_Group
    flags = [f for f in FLAGS if not f.startswith(("--loss-function", "--arch-embedding-size", "--mini-batch-size",
                                                    "--cache-size", "--arch-mlp-bot"))]
    args = ProcessArgs(flags + ["--arch-embedding-size=900-40-6-2500", "--mini-batch-size=32", "--cache-size=300",
                                "--arch-mlp-bot=5-32-16", "--data-generation=random", "--num-batches=10",
                                "--num-indices-per-lookup=6"])
    assert args.loss_function == "mse"                   # the reference's default (main_no_ddp.py:48)
    ln_emb = np.array([900, 40, 6, 2500])
    m_spa, B, L, nb, seed = 16, 32, 4, 10, 11
    ln_bot = np.array([5, 32, 16])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2, 32, 1])
    _, loader = DP.make_random_data_and_loader(args, ln_emb, 5)
    loader.multi_hot = True
    batches = [(X, [o for o in lS_o], lS_i, T) for X, lS_o, lS_i, T in loader]
    aux = (B * 6 + 255) // 256 * 256
    torch.set_num_threads(1)
    np.random.seed(seed)
    torch.manual_seed(seed)
    host_o = O.init_host_tables([int(n) for n in ln_emb], m_spa)
    otr = O.OracleTrainer([int(n) for n in ln_emb], m_spa, ln_bot, ln_top, cache_size=300, num_ways=4,
                          mini_batch_size=aux, lr=0.1, lr_embeds=0.3, lookahead=L, table_agg_freq=5, seed=seed,
                          host_tables=[h.clone() for h in host_o], loss="mse")
    # `for ... in train_ld` on a torch DataLoader draws the iterator's base seed from the global torch generator
    # (one int64 random_()), in the reference's loop as in Run's: the way choices of the refills come after that draw
    torch.empty((), dtype=torch.int64).random_()
    for j, (X, lS_o, lS_i, T) in enumerate(batches):
        if j % L == 0:
            otr.refill([torch.cat([b[2][k] for b in batches[j:j + L]]) for k in range(len(ln_emb))])
        otr.step(j, X, lS_o, lS_i, T)
    eg = Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
    for k in range(len(ln_emb)):
        eg.emb_l[k].weight.data = host_o[k].clone()
    eg.pin()
    capsys.readouterr()
    eng = Run(0, m_spa, ln_emb, ln_bot, ln_top, loader, None, None, None, None, eg, args)
    printed = capsys.readouterr().out
    got = [float(x) for x in re.findall(r"Loss = ([0-9.eE+-]+),", printed)]
    want = np.array([l[0] for l in otr.losses])
    assert len(got) == nb - 1
    # multi-hot bags: the pooled sums (up to 6 rows) and their gradients accumulate in another order than ATen's, and
    # the differences compound over the iterations -- 3e-5 here; the 1e-5 bar is the Criteo layout's (one row per bag)
    np.testing.assert_allclose(np.array(got), np.concatenate([[(want[0] + want[1]) / 2], want[2:]]), rtol=3e-5)
    eng.cg.ctx.check()
    for k in range(len(ln_emb)):
        assert torch.equal(eng.cg.occupancy_tables[k].cpu(), otr.occ[k]), k


def test_main_cli_random_default_front_end(capsys):
    """python -m cdlrm_amd.main_no_ddp with the reference's DEFAULT data mode and loss (random multi-hot, mse)."""
    from cdlrm_amd import main_no_ddp
    flags = [f for f in FLAGS if not f.startswith(("--loss-function", "--cache-size"))]
    main_no_ddp.main(flags + ["--num-batches=9", "--cache-size=4000", "--num-indices-per-lookup=4"])
    out = capsys.readouterr().out
    losses = [float(x) for x in re.findall(r"Loss = ([0-9.eE+-]+),", out)]
    assert len(losses) == 8 and all(np.isfinite(losses)) and 0.0 < losses[-1] < 1.0


def test_main_cli_synthetic_trace_front_end(capsys, tmp_path):
    """python -m cdlrm_amd.main_no_ddp --data-generation=synthetic: bags drawn from per-table stack-distance profiles
    (--data-trace-file with "j" = table number, dlrm_data_pytorch.py:808-883) train through the same multi-hot path as the
    random front end."""
    from cdlrm_amd import dlrm_data_pytorch as DP
    from cdlrm_amd import main_no_ddp
    d = str(tmp_path)
    assert "j" not in d
    emb = [int(x) for x in [f for f in FLAGS if f.startswith("--arch-embedding-size")][0].split("=")[1].split("-")]
    rng = np.random.RandomState(3)
    for i, n in enumerate(emb):
        trace = (rng.zipf(1.3, 400) % n).astype(np.uint64).tolist()
        DP.write_dist_to_file(os.path.join(d, "dist_emb_%d.log" % i), *DP.trace_distribution(trace))
    flags = [f for f in FLAGS if not f.startswith(("--loss-function", "--cache-size"))]
    main_no_ddp.main(flags + ["--num-batches=9", "--cache-size=4000", "--num-indices-per-lookup=4",
                              "--data-generation=synthetic", "--data-trace-file=" + os.path.join(d, "dist_emb_j.log")])
    out = capsys.readouterr().out
    losses = [float(x) for x in re.findall(r"Loss = ([0-9.eE+-]+),", out)]
    assert len(losses) == 8 and all(np.isfinite(losses)) and 0.0 < losses[-1] < 1.0


def test_run_device_rng_lookahead_plan_equals_boundary_plan(capsys, monkeypatch):
    """--device-rng: Run plans window w+1 in the background while window w trains (the reference's Prefetcher role).
    Same printed losses and final tags as planning every window at its boundary."""
    from cdlrm_amd.main_no_ddp import ProcessArgs, Run
    from cdlrm_amd.model_no_ddp import Embedding_Table_Group
    ln_emb = np.array([3000, 50, 7, 1200, 40000])
    m_spa, B, nb, seed = 16, 64, 14, 11
    ln_bot = np.array([13, 32, 16])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2, 32, 1])
    args = ProcessArgs(FLAGS + ["--device-rng"])
    np.random.seed(seed)
    host0 = O.init_host_tables([int(n) for n in ln_emb], m_spa)
    outs = []
    for at_boundary in (False, True):
        args.plan_at_boundary = at_boundary
        eg = Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
        for k in range(len(ln_emb)):
            eg.emb_l[k].weight.data = host0[k].clone()
        eg.pin()
        capsys.readouterr()
        eng = Run(0, m_spa, ln_emb, ln_bot, ln_top, _Loader(_loader(ln_emb, B, nb, 5)), None, None, None, None, eg, args)
        printed = capsys.readouterr().out
        eng.cg.ctx.check()
        outs.append(([float(x) for x in re.findall(r"Loss = ([0-9.eE+-]+),", printed)], eng.cg.tags.cpu().clone(),
                     [e.weight.data.clone() for e in eg.emb_l]))
    assert len(outs[0][0]) == nb - 1 and outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)


def test_roc_auc_rank_sum_against_sklearn():
    """ops.roc_auc on the device = sklearn.metrics.roc_auc_score, ties included (scores quantised to 64 levels), and NaN for a
    one-class target."""
    from sklearn.metrics import roc_auc_score
    from cdlrm_amd import ops
    rng = np.random.RandomState(3)
    for n, levels in ((10, 0), (4097, 0), (20000, 64), (20000, 2)):
        s = rng.rand(n).astype(np.float32)
        if levels:
            s = np.round(s * levels) / levels
        t = (rng.rand(n) < 0.3 + 0.4 * s).astype(np.float32)
        got = ops.roc_auc(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda())
        assert abs(got - roc_auc_score(t, s)) < 1e-9, (n, levels)
    assert np.isnan(ops.roc_auc(torch.rand(8).cuda(), torch.ones(8).cuda()))


def test_mlperf_auc_threshold_stops_training(capsys):
    """--mlperf-auc-threshold (parsed and unused by the reference, main_no_ddp.py:119-120): training stops at the first test
    whose AUC reaches it."""
    from cdlrm_amd.main_no_ddp import ProcessArgs, Run
    from cdlrm_amd.model_no_ddp import Embedding_Table_Group
    args = ProcessArgs(FLAGS + ["--test-freq=3", "--mlperf-auc-threshold=0.01"])
    ln_emb = np.array([3000, 50, 7, 1200, 40000])
    m_spa, B = 16, 64
    ln_bot = np.array([13, 32, 16])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2, 32, 1])
    batches = _Loader(_loader(ln_emb, B, 12, 5))
    test_batches = _loader(ln_emb, B, 2, 77)
    np.random.seed(11)
    torch.manual_seed(11)
    host = O.init_host_tables([int(n) for n in ln_emb], m_spa)
    eg = Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
    for k in range(len(ln_emb)):
        eg.emb_l[k].weight.data = host[k].clone()
    eg.pin()
    capsys.readouterr()
    Run(0, m_spa, ln_emb, ln_bot, ln_top, batches, test_batches, None, None, None, eg, args)
    printed = capsys.readouterr().out
    assert printed.count("Testing at") == 1 and "MLPerf threshold reached at 3/12" in printed
    assert len(re.findall(r"Loss = ", printed)) == 3        # iterations 1, 2, 3 printed, nothing after the stop


def test_evict_victim_cache_matches_oracle(capsys):
    """--evict-victim-cache (main_no_ddp.py:96 parses it, model_no_ddp.py:187 records victim_cache_entries, nothing uses
    either): here the trained aux rows of a batch's MISSES go back to their host rows (and to their copies among the window's
    victim rows) behind every step -- `emb_tables[k].weight[missing] = cache[k].weight[aux]`, last occurrence of an index
    wins.  Run against the oracle's arm of the same definition: loss per iteration 1e-5, tags bit-exact, host tables equal;
    and the flag changes the result (without it the host rows of missed indices keep their initial values)."""
    from cdlrm_amd.main_no_ddp import ProcessArgs, Run
    from cdlrm_amd.model_no_ddp import Embedding_Table_Group
    ln_emb = np.array([3000, 50, 7, 1200, 40000])
    m_spa, B, L, nb, seed = 16, 64, 4, 14, 11
    ln_bot = np.array([13, 32, 16])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2, 32, 1])
    batches = _Loader(_loader(ln_emb, B, nb, 5))
    torch.set_num_threads(1)
    np.random.seed(seed)
    torch.manual_seed(seed)
    host_o = O.init_host_tables([int(n) for n in ln_emb], m_spa)
    otr = O.OracleTrainer([int(n) for n in ln_emb], m_spa, ln_bot, ln_top, cache_size=40, num_ways=4, mini_batch_size=B,
                          lr=0.1, lr_embeds=0.3, lookahead=L, table_agg_freq=5, seed=seed,
                          host_tables=[h.clone() for h in host_o], evict_victim_cache=True)
    n_dup = 0
    for j, (X, lS_o, idx, T) in enumerate(batches):
        if j % L == 0:
            otr.refill(torch.cat([b[2] for b in batches[j:j + L]], dim=1))
        otr.step(j, X, lS_o, idx, T)
    changed = sum(int((otr.host[k] != host_o[k]).any(1).sum()) for k in range(len(ln_emb)))
    assert changed > 100            # the arm does something: rows of missed indices were written

    def run(flags):
        args = ProcessArgs(FLAGS + flags)
        eg = Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
        for k in range(len(ln_emb)):
            eg.emb_l[k].weight.data = host_o[k].clone()
        eg.pin()
        capsys.readouterr()
        eng = Run(0, m_spa, ln_emb, ln_bot, ln_top, batches, None, None, None, None, eg, args)
        printed = capsys.readouterr().out
        return eng, eg, [float(x) for x in re.findall(r"Loss = ([0-9.eE+-]+),", printed)]

    eng, eg, got = run(["--evict-victim-cache"])
    want = np.array([l[0] for l in otr.losses])
    expect = np.concatenate([[(want[0] + want[1]) / 2], want[2:]])
    np.testing.assert_allclose(np.array(got), expect, rtol=1e-5)
    eng.cg.ctx.check()
    for k in range(len(ln_emb)):
        assert torch.equal(eng.cg.occupancy_tables[k].cpu(), otr.occ[k]), k
        np.testing.assert_allclose(eg.emb_l[k].weight.data.numpy(), otr.host[k].numpy(), rtol=1e-5, atol=1e-6)
    _, eg0, got0 = run([])
    assert sum(int((eg0.emb_l[k].weight.data != eg.emb_l[k].weight.data).any(1).sum()) for k in range(len(ln_emb))) > 100
    assert not np.allclose(got0, got, rtol=1e-7, atol=0)


def test_evict_victim_cache_refuses_more_than_one_rank():
    from cdlrm_amd.main_no_ddp import ProcessArgs
    args = ProcessArgs(FLAGS + ["--evict-victim-cache"])
    assert args.evict_victim_cache is True
