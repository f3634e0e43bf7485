"""BASELINE.json's GPU configurations at the SHAPE the bench runs them, against the REFERENCE's own run of the same
workload (tests/golden/train_c{2,3,4,5}shape.npz, captured by tools/make_golden.py:g_train_shapes from the imported
reference: main_no_ddp.py:148-209 CacheEmbeddings, model_no_ddp.py:149-212 cache forward, DLRM_Net, BCELoss, both SGDs).

  c3-shape: 26 Terabyte-cardinality tables (capped at 50 k rows), D=128, 16-WAY, bot 13-512-256-128, top 512-512-256-1
  c2-shape: 26 Kaggle-cardinality tables, D=32, 8-way, B=2048, L=4
  c4-shape: embed-dim 256 (BASELINE configs[3]): bot 13-512-256-256, the 607-wide top input on its 608-float pitch, 26
            tables x 16-way with 1024-byte cache rows -- pipelined and window-resolved
  c5-shape: 12288 lookups per table and step (the backward's slot sort takes its merge passes), the look-ahead window
            STREAMED into the plan in 3 chunks (cdlrm_window_unique_add / _finish) in front of a real insert

BCE loss per iteration within 1e-5 relative, cache tag state bit-exact, cache-row / host-row checksums.  These are the
kernel instantiations the headline bench uses (k_probe<16>, k_uniq_probe<16>, the 16-way k_assign, the 26-table gather)
which the small fixtures never reach.

Plus full-size PROPERTY tests of c3 and c5 exactly as bench.py builds them (bench.build_workload): no oracle can run
at that size, so the checks are the invariants the reference's data structure guarantees.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("name,chunks,pipelined,resolved,fused", [
    ("train_c3shape", 0, True, False, True), ("train_c2shape", 0, True, True, True), ("train_c5shape", 3, True, False, True),
    ("train_c3shape", 2, False, False, True), ("train_c3shape", 0, True, True, True), ("train_c5shape", 0, True, True, True),
    ("train_c4shape", 0, True, False, True), ("train_c4shape", 0, True, True, True),
    # gather + interaction as two launches (the path of multi-hot bags, "cat" and shapes outside the fused kernels)
    ("train_c3shape", 0, True, True, False), ("train_c2shape", 0, True, False, False), ("train_c5shape", 3, True, False, False),
    ("train_c4shape", 0, False, False, False)])
def test_config_shaped_training_vs_reference(golden, name, chunks, pipelined, resolved, fused):
    from test_engine_parity import build, make_batches
    g = golden(name)
    host, cg, dl, eng, pipe = build(g)
    eng.fuse_gather = fused     # True (the default): the gather rides in the interaction kernels (cdlrm_gather_interact_fwd / _bwd)
    assert eng._fused_gather(None) == fused
    assert cg.num_ways == int(g["ways"]) and len(cg.cache_sizes) == 26
    L, ways = int(g["L"]), int(g["ways"])
    batches = make_batches(g)
    dev_idx = [b[1].to(DEV) for b in batches]
    losses = []
    for j, (X, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            win = torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(DEV)
            torch.manual_seed(5000 + j)          # the q stream the reference consumed for this refill
            if chunks:
                parts = [p.contiguous() for p in torch.chunk(win, chunks, dim=1)]
                assert len(parts) == chunks
                pipe.plan_window(lambda parts=parts: iter(parts))
            else:
                pipe.plan_window(win)
            pipe.commit()
            pipe.wait_writeback()
            rs = None
            if resolved:        # window-resident probe: tag match once per window, cdlrm_embbag_take per step
                from cdlrm_amd.engine import WindowResolver
                rs = WindowResolver(eng, win, int(g["B"]), chunk=2)
        nxt = dev_idx[j + 1] if pipelined and j + 1 < len(batches) and (j + 1) % L != 0 else None
        loss = eng.step(X.to(DEV), dev_idx[j], Tt.to(DEV), j=j, next_idx=nxt,
                        res=rs.batch(j % L) if rs is not None else None,
                        next_res=rs.batch(j % L + 1) if (rs is not None and nxt is not None) else None)
        if rs is not None:
            rs.ensure(j % L + rs.CH + 2)
        losses.append(loss[0:1].clone())
    losses = np.array([float(x) for x in losses])
    cg.ctx.check()
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    occ = cg.occupancy_tables
    n_resident = 0
    for k in range(26):
        want = t(g[f"occ_{k}"]).to(torch.int64)
        assert torch.equal(occ[k].cpu(), want), k                                  # bit-exact tag state
        n_resident += int((want != -1).sum())
        w = cg.emb_l[k].weight[: ways * cg.cache_sizes[k]].double().sum().item()
        np.testing.assert_allclose(w, float(g[f"weight_sum_{k}"]), rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(host.emb_l[k].weight.data.double().sum().item(), float(g[f"host_sum_{k}"]),
                                   rtol=1e-6, atol=1e-4)
    assert n_resident > 1000
    from cdlrm_amd.model_no_ddp import _linears
    for i, l in enumerate(_linears(dl.top_l)):
        if f"top_w{i}" in g.files:
            np.testing.assert_allclose(l.weight.data.cpu().numpy(), g[f"top_w{i}"], rtol=1e-4, atol=1e-6)


# --------------------------------------------------------------------------------------------------------------------
# full-size property tests: c3 and c5 as bench.py builds them
# --------------------------------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def terabyte_host():
    """The 96 GB Terabyte-shape host tables (D = 128), built once for both full-size cases."""
    import bench
    host = bench.build_host_tables("c3", seed=123, dev=torch.device(DEV))
    yield host
    del host


def _check_cache_invariants(cg):
    """What the reference's data structure guarantees after any number of CacheEmbeddings calls
    (main_no_ddp.py:155, 203-204): a resident tag sits in set `tag % P`, and no tag is resident twice."""
    ways = cg.num_ways
    for k, occ in enumerate(cg.occupancy_tables):
        P = cg.cache_sizes[k]
        valid = occ != -1
        sets = torch.arange(P, device=occ.device).view(-1, 1).expand(-1, ways)
        assert torch.equal(occ[valid] % P, sets[valid]), "table %d: a tag outside its set" % k
        v = occ[valid]
        assert v.numel() == torch.unique(v).numel(), "table %d: a tag resident twice" % k
        assert int(v.min()) >= 0 and int(v.max()) < int(cg.ln_emb[k]) if v.numel() else True


def _run_full_size(config, host, L, n_windows, steps_per_window, max_ind_range=-1, fuse_gather=True):
    import bench
    w = bench.build_workload(config, lookahead=L, host=host, seed=123, cache_init="zeros", write_back=False,
                             max_ind_range=max_ind_range)
    cg, eng, pipe, syn, B = w["cg"], w["eng"], w["pipe"], w["syn"], w["B"]
    T = len(cg.cache_sizes)
    # fuse_gather (the default): the rows are the interaction kernels' operand loads and the gathered block never exists;
    # False: gather + interaction as two launches, whose gather output is checked against the rows the probe resolved
    eng.fuse_gather = bool(fuse_gather)
    assert eng._fused_gather(None) == bool(fuse_gather)
    losses, feats = [], None
    for wi in range(n_windows):
        win = syn.window(wi, L)
        pipe.plan_window(win)
        pipe.commit()
        _check_cache_invariants(cg)
        from cdlrm_amd.engine import WindowResolver
        rs = WindowResolver(eng, win, B)              # as bench.py: the window's lookups resolved once
        for jj in range(steps_per_window):
            idx = win[:, jj * B:(jj + 1) * B]
            X, Tt = syn.dense(wi * L + jj)
            nxt = win[:, (jj + 1) * B:(jj + 2) * B] if jj + 1 < steps_per_window else None
            # the probe result of THIS batch, taken before the step updates the rows it points at
            check_rows = wi == n_windows - 1 and jj == 0
            if check_rows:
                eng.finish()
                torch.cuda.synchronize()
                from cdlrm_amd import ops
                slots, miss_pos, miss_count = ops.embbag_probe(cg.ctx, idx, aux_phase=eng._phase)
                torch.cuda.synchronize()
                rb = torch.tensor(cg.row_base[:T], device=DEV).view(T, 1)
                rows_before = slots.to(torch.int64) + rb
                snapshot = cg.weight.data[rows_before[:, :4096].reshape(-1)].view(T, -1, cg.m_spa).clone()
                eng._pref = None                     # the in-line probe above replaced any prefetched one
            lossbuf = eng.step(X, idx, Tt, j=jj, next_idx=nxt, res=rs.batch(jj),
                               next_res=rs.batch(jj + 1) if nxt is not None else None)
            rs.ensure(jj + rs.CH + 2)
            losses.append(lossbuf[0:1].clone())
            if check_rows and not fuse_gather:
                # gather == row copy: feat[:, k+1] is bit-exactly the cache row the probe resolved (one lookup per bag)
                eng.finish()
                torch.cuda.synchronize()
                feat = eng._buffers(B)["feat"]
                got = feat[:4096, 1:, :].permute(1, 0, 2)
                assert torch.equal(got, snapshot), "gather output differs from the cache rows it resolved"
                feats = got.double().sum().item()
    eng.finish()
    cg.ctx.check()
    torch.cuda.synchronize()
    # (in pieces: a float64 sum of the whole cache converts it at once -- 65 GiB at c5, beside everything else of the run)
    wsum = sum(c.sum(dtype=torch.float64).item() for c in cg.weight.data.split(1 << 21))
    out = dict(losses=torch.cat(losses).cpu(), tags=cg.tags.clone(), wsum=wsum, feats=feats, params=eng.param_flat.clone())
    del w, cg, eng, pipe, syn
    torch.cuda.empty_cache()
    return out


@pytest.mark.parametrize("config,L,steps", [("c3", 1000, 4), ("c5", 125, 2)])
def test_full_size_invariants_and_bitwise_repeat(terabyte_host, config, L, steps):
    """c3 (B=8192, 150 k x 16-way) and c5 (B=65536, 500 k x 16-way) at FULL size, two look-ahead windows of 8.2 M
    indices per table each (c3: 1000 batches; c5: 125, one chunk of the bench's streamed window) -- more unique indices
    than the big tables' caches have slots, so full sets, contested slots, evictions and window victims all occur:
    every resident tag in its set and unique, the gather bit-exact against the rows the probe resolved, finite losses
    near ln 2, and a second run from the same state bitwise identical (tags, every parameter, the loss trajectory) --
    no atomics-order or cross-stream race dependence anywhere in the step.  Three runs: the default schedule (the gather
    fused into the interaction kernels) TWICE -- the bitwise repeat of one schedule --, and once with gather + interaction as
    two launches: the fused kernels are held to the two operators bit for bit at full size, through every step.  (Write-back
    is off in all runs so that each sees the host tables of the first.)"""
    a = _run_full_size(config, terabyte_host, L, 2, steps)
    a2 = _run_full_size(config, terabyte_host, L, 2, steps)
    assert torch.equal(a["losses"], a2["losses"]) and torch.equal(a["tags"], a2["tags"]) and torch.equal(a["params"], a2["params"]) \
        and a["wsum"] == a2["wsum"], "the same schedule run twice from the same state differs"
    del a2
    b = _run_full_size(config, terabyte_host, L, 2, steps, fuse_gather=False)
    assert torch.isfinite(a["losses"]).all() and 0.3 < float(a["losses"][-1]) < 2.0
    assert torch.equal(a["losses"], b["losses"]), "loss trajectory differs between the fused and the two-launch schedule"
    assert torch.equal(a["tags"], b["tags"])
    assert torch.equal(a["params"], b["params"])
    assert a["wsum"] == b["wsum"] and a["feats"] is None and b["feats"] is not None
    assert int((a["tags"] != -1).sum()) > 1_000_000


def _run_window_boundary(host, L, gather_alone_min, steps_after=6):
    """One whole look-ahead window of c3 at full size AS bench.py's whole-window leg trains it: the next window's plan launched
    in the background at iteration plan_at (least-priority stream + CPU row gather beside the steps), the commit at the
    boundary with the eviction write-back ON, the look-ahead resolver, the given take schedule.  Returns the end state; the
    host rows the write-back changed are put back (the module's other cases read the same host tables)."""
    import bench
    from cdlrm_amd.engine import WindowResolver
    w = bench.build_workload("c3", lookahead=L, host=host, seed=123, cache_init="zeros", write_back=True)
    cg, eng, pipe, syn, B = w["cg"], w["eng"], w["pipe"], w["syn"], w["B"]
    T = len(cg.cache_sizes)
    eng.gather_alone_min = gather_alone_min
    plan_at = max(1, min(L // 2, 64))                   # bench.py's
    win = syn.window(0, L)
    pipe.plan_window(win)
    pipe.commit()
    rs = WindowResolver(eng, win, B)
    nxt_win = None
    losses = []
    for jj in range(L):
        if jj == plan_at:
            pipe.wait_writeback()
            nxt_win = syn.window(1, L)
            pipe.plan_window(nxt_win)                   # runs beside the steps that follow
        idx = win[:, jj * B:(jj + 1) * B]
        X, Tt = syn.dense(jj)
        nxt = win[:, (jj + 1) * B:(jj + 2) * B] if (jj + 1 < L and jj + 1 != plan_at) else None
        lossbuf = eng.step(X, idx, Tt, j=jj, next_idx=nxt, res=rs.batch(jj), next_res=rs.batch(jj + 1) if nxt is not None else None,
                           loss_sync=False)
        rs.ensure(jj + rs.CH + 2)
    # ---- the boundary.  The plan named its evictions at iteration plan_at; the rows they carry to the host tables are the rows
    #      as the LAST step of the window left them.  Sample up to 4096 evicted tags per table: their cache rows now, their host
    #      rows now (for the undo) -- then commit + write-back and compare.
    if pipe._worker is not None:
        pipe._worker.join()
    eng.finish()
    torch.cuda.synchronize()
    # (the plan holds, per winner of the window, the tag word it will overwrite and the cache row it will take: what the commit
    #  kernel evicts is whatever VALID tag sits in that word at commit time, main_no_ddp.py:190-199)
    _, _, wo = pipe.plan.offsets()
    undo, expect = [], []
    n_evicted = 0
    for k in range(T):
        tp = pipe.plan.win_tag[wo[k]:wo[k + 1]]
        rows = pipe.plan.win_row[wo[k]:wo[k + 1]]
        old = cg.tags[tp]
        valid = old != -1
        tags = old[valid]
        n_evicted += int(tags.numel())
        if tags.numel() == 0:
            undo.append(None); expect.append(None)
            continue
        assert tags.numel() == torch.unique(tags).numel(), "table %d: a tag evicted twice" % k
        tags_h = tags.cpu()
        undo.append((tags_h, host.emb_l[k].weight.data[tags_h].clone()))
        pick = torch.linspace(0, tags.numel() - 1, min(4096, tags.numel())).long().to(tags.device)
        P = int(cg.cache_sizes[k])
        sample = tags[pick]
        assert bool((cg.occupancy_tables[k][sample % P] == sample.view(-1, 1)).any(1).all()), \
            "table %d: a tag named for eviction is not resident before the commit" % k
        expect.append((sample.cpu(), cg.weight.data[rows[valid][pick]].clone().cpu()))
    assert n_evicted > 100_000, "the boundary evicts too little to test anything (%d)" % n_evicted
    pipe.commit()
    pipe.wait_writeback()
    torch.cuda.synchronize()
    _check_cache_invariants(cg)
    for k in range(T):
        if expect[k] is not None:
            tags_s, rows = expect[k]
            assert torch.equal(host.emb_l[k].weight.data[tags_s], rows), "table %d: host rows of evicted tags != the cache rows they left" % k
            # ... and an evicted tag is gone from its set
            P = int(cg.cache_sizes[k])
            assert not bool((cg.occupancy_tables[k][tags_s.to(DEV) % P] == tags_s.to(DEV).view(-1, 1)).any())
    # ---- a few steps into the next window (its resolver against the new tags)
    rs = WindowResolver(eng, nxt_win, B)
    for jj in range(steps_after):
        idx = nxt_win[:, jj * B:(jj + 1) * B]
        X, Tt = syn.dense(L + jj)
        nxt = nxt_win[:, (jj + 1) * B:(jj + 2) * B] if jj + 1 < steps_after else None
        lossbuf = eng.step(X, idx, Tt, j=jj, next_idx=nxt, res=rs.batch(jj), next_res=rs.batch(jj + 1) if nxt is not None else None)
        rs.ensure(jj + rs.CH + 2)
        losses.append(lossbuf[0:1].clone())
    eng.finish()
    cg.ctx.check()
    torch.cuda.synchronize()
    pipe.close()
    # (the cache rows proper: the aux rows behind them hold the last batches' transient miss rows, in one region or two
    #  depending on the take schedule)
    wsum = sum(cg.emb_l[k].weight.data[:cg.num_ways * int(cg.cache_sizes[k])].sum(dtype=torch.float64).item() for k in range(T))
    out = dict(losses=torch.cat(losses).cpu(), tags=cg.tags.clone(), wsum=wsum, params=eng.param_flat.clone(), n_evicted=n_evicted)
    for k in range(T):                                  # the host tables as the other cases expect them
        if undo[k] is not None:
            host.emb_l[k].weight.data[undo[k][0]] = undo[k][1]
    del w, cg, eng, pipe, syn
    torch.cuda.empty_cache()
    return out


def test_whole_window_with_background_plan_commit_and_writeback(terabyte_host):
    """What bench.py's whole-window leg does (config.whole_window: the figure that matches the metric's "wall incl. refills"),
    asserted: c3 at full size through ONE window boundary -- L = 200 steps with the next window's plan running in the
    background from iteration 64, the commit, the eviction write-back ON, then six steps of the next window -- under BOTH
    take schedules (two aux regions: gather_alone_min above the batch; chained take: at it).  Checked at the boundary: every
    tag named for eviction is resident before and gone after, the host rows of 4096 evicted tags per table equal the
    trained cache rows they left, the structure's invariants hold; and the two schedules end on the same bits (tags, every
    parameter, the loss trajectory): the background plan, the commit and the write-back order with the training step under
    either."""
    a = _run_window_boundary(terabyte_host, 200, 16384)
    b = _run_window_boundary(terabyte_host, 200, 8192)
    assert torch.isfinite(a["losses"]).all() and 0.3 < float(a["losses"][-1]) < 2.0
    assert a["n_evicted"] == b["n_evicted"]
    assert torch.equal(a["tags"], b["tags"]), "tags differ between the take schedules"
    assert torch.equal(a["losses"], b["losses"]) and torch.equal(a["params"], b["params"]) and a["wsum"] == b["wsum"]


def test_c4_capped_invariants_and_bitwise_repeat():
    """BASELINE configs[3] (embed-dim 256, 26 tables x 150 k x 16-way, B = 8192) exactly as `bench.py --config c4
    --max-ind-range 2000000` builds it: tables capped at 2 M rows (the uncapped host tables are 192 GB of pinned memory;
    capped: 20 GB), everything else at the configuration's size -- the 1024-byte-row gather / backward, the 13-512-256-256
    bottom MLP, the 607 -> 608-pitch top input at M = 8192.  Same invariants as the c3 / c5 cases, two windows of 8.2 M
    indices per table: the eight 2 M-row tables overflow their 2.4 M slots' sets, so evictions and victims occur."""
    import bench
    cap = 2_000_000
    host = bench.build_host_tables("c4", seed=123, dev=torch.device(DEV), max_ind_range=cap)
    try:
        a = _run_full_size("c4", host, 1000, 2, 4, max_ind_range=cap)
        b = _run_full_size("c4", host, 1000, 2, 4, max_ind_range=cap, fuse_gather=False)
    finally:
        del host
    assert torch.isfinite(a["losses"]).all() and 0.3 < float(a["losses"][-1]) < 2.0
    assert torch.equal(a["losses"], b["losses"]), "loss trajectory differs between two identical runs"
    assert torch.equal(a["tags"], b["tags"])
    assert torch.equal(a["params"], b["params"])
    assert a["wsum"] == b["wsum"] and a["feats"] is None and b["feats"] is not None
    assert int((a["tags"] != -1).sum()) > 1_000_000
