"""End-to-end parity of the fused HIP training engine against the loss trajectories / final state the
REFERENCE produced (tests/golden/train_*.npz, captured by tools/make_golden.py): BCE loss within 1e-5
relative per iteration, cache tag state bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.asarray(a))


def make_batches(g):
    """The batch stream tools/make_golden.py:ref_train draws (numpy RandomState(seed+1))."""
    ln_emb = [int(x) for x in g["ln_emb"]]
    B, seed, alpha = int(g["B"]), int(g["seed"]), float(g["alpha"])
    rng = np.random.RandomState(seed + 1)
    out = []
    for j in range(int(g["nbatch"])):
        X = torch.from_numpy(rng.rand(B, int(g["ln_bot"][0])).astype(np.float32))
        lS_i = torch.stack([torch.from_numpy((rng.zipf(alpha, size=B).astype(np.int64) * 2654435761 % n).astype(np.int64))
                            for n in ln_emb])
        Tt = torch.from_numpy(np.round(rng.rand(B, 1)).astype(np.float32))
        out.append((X, lS_i, Tt))
    return out


def build(g, world=1, rank=0, host=None, aux_phases=2):
    from cdlrm_amd.engine import TrainEngine, WindowPipeline
    from cdlrm_amd.model_no_ddp import DLRM_Net, Embedding_Table_Cache_Group, Embedding_Table_Group
    ln_emb = np.array([int(x) for x in g["ln_emb"]])
    m_spa, seed, B, L = int(g["m_spa"]), int(g["seed"]), int(g["B"]), int(g["L"])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2] + [int(x) for x in g["top"]])
    if host is None:
        np.random.seed(seed)
        torch.manual_seed(seed)
        host = Embedding_Table_Group(m_spa, ln_emb).pin()
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = Embedding_Table_Cache_Group(m_spa, ln_emb, int(g["cache_size"]), B, int(g["ways"]), aux_phases=aux_phases).to(DEV)
    dl = DLRM_Net(np.array(g["ln_bot"]), ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0).to(DEV)
    eng = TrainEngine(cg, dl, host, lr=float(g["lr"]), lr_embeds=float(g["lr_emb"]), world_size=world, rank=rank,
                      table_agg_freq=int(g["agg_freq"]) if "agg_freq" in g.files else 10 ** 9,
                      table_agg_op=str(g["agg_op"]) if "agg_op" in g.files else "mean")
    pipe = WindowPipeline(cg, host, L * B, parity_rng=True, rank=rank, world_size=world)
    return host, cg, dl, eng, pipe


@pytest.mark.parametrize("name,pipelined,aux_phases,resolved", [
    ("train_small", False, 2, False), ("train_c1", False, 2, False), ("train_small", True, 2, False),
    ("train_c1", True, 2, False), ("train_small", True, 1, False), ("train_stream", False, 2, False),
    ("train_stream", True, 2, False),
    # window-resident probe (engine.WindowResolver): the window's lookups resolved once per window, cdlrm_embbag_take per step
    ("train_small", True, 2, True), ("train_c1", True, 2, True), ("train_small", False, 2, True),
    ("train_small", True, 1, True), ("train_stream", True, 2, True),
    # loss completed off the training queue (step(loss_sync=False), what bench.py and Run use): read behind finish()
    ("train_small", True, 2, "async"), ("train_c1", True, 2, "async")])
def test_loss_trajectory_and_tag_state(golden, name, pipelined, aux_phases, resolved):
    """pipelined: the next batch's indices are handed to step() (as bench.py does inside a window), so its tag probe
    and aux-row fill run during the current step, into the other aux region (aux_phases = 2) or behind the current
    embedding update (aux_phases = 1).  Same trajectory either way."""
    g = golden(name)
    host, cg, dl, eng, pipe = build(g, aux_phases=aux_phases)
    async_loss = resolved == "async"
    L = int(g["L"])
    batches = make_batches(g)
    dev_idx = [b[1].to(DEV) for b in batches]
    losses = []
    for j, (X, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            win = torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(DEV)
            if "reseed" not in g.files or bool(g["reseed"]):
                torch.manual_seed(5000 + j)      # the q stream the reference consumed for this refill
            # (train_stream: no re-seeding -- the draws continue the stream the trainer's construction left behind)
            pipe.plan_window(win)
            pipe.commit()
            pipe.wait_writeback()
            rs = None
            if resolved:
                from cdlrm_amd.engine import WindowResolver
                rs = WindowResolver(eng, win, int(g["B"]), chunk=3)
        nxt = dev_idx[j + 1] if pipelined and j + 1 < len(batches) and (j + 1) % L != 0 else None
        loss = eng.step(X.to(DEV), dev_idx[j], Tt.to(DEV), j=j, next_idx=nxt,
                        res=rs.batch(j % L) if rs is not None else None,
                        next_res=rs.batch(j % L + 1) if (rs is not None and nxt is not None) else None,
                        loss_sync=not async_loss)
        if rs is not None:
            rs.ensure(j % L + rs.CH + 2)
        if async_loss:
            eng.finish()
        losses.append(loss[0:3].clone())
    stats = eng.stat_acc.tolist()
    np.testing.assert_allclose(stats, [sum(float(x[1]) for x in losses), sum(float(x[2]) for x in losses)], rtol=1e-12)
    losses = [float(x[0]) for x in losses]
    cg.ctx.check()
    np.testing.assert_allclose(np.array(losses), g["losses"], rtol=1e-5)
    occ = cg.occupancy_tables
    for k in range(len(g["ln_emb"])):
        assert torch.equal(occ[k].cpu(), t(g[f"occ_{k}"])), k                      # bit-exact tag state
        w = cg.emb_l[k].weight[: int(g["ways"]) * cg.cache_sizes[k]].double().sum().item()
        np.testing.assert_allclose(w, float(g[f"weight_sum_{k}"]), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(host.emb_l[k].weight.data.double().sum().item(), float(g[f"host_sum_{k}"]),
                                   rtol=1e-6, atol=1e-5)
    from cdlrm_amd.model_no_ddp import _linears
    for i, l in enumerate(_linears(dl.top_l)):
        np.testing.assert_allclose(l.weight.data.cpu().numpy(), g[f"top_w{i}"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name,defer,chunk,delay", [("train_small", True, 2, False), ("train_small", False, 2, False),
                                                    ("train_c1", True, 2, False), ("train_stream", True, 2, False),
                                                    # the host far ahead of a slow side stream, a ring slot recycled every 3 steps
                                                    ("train_c1", True, 1, True), ("train_c1", False, 2, True)])
def test_chained_take_long_batch_path(golden, name, defer, chunk, delay):
    """The long-batch schedule (gather alone on the main stream, B >= gather_alone_min) on the window-resident probe: the
    next batch's take follows the embedding update on the side stream and the next gather waits for ONE event recorded
    behind it (and behind the deferred top-MLP update).  Forced here at the goldens' small batches; same trajectory,
    tags and weights as the reference, taped and untaped steps alike.
    delay: every step's side-stream work (embedding update, the next batch's take) is held back by a ~1 ms spin kernel while
    the host issues the whole window without a sync, so resolves of later chunks are issued long before the takes that read
    the ring slots they recycle have run (engine.WindowResolver's ring + side-stream event must order them)."""
    from cdlrm_amd.engine import TrainEngine, WindowResolver
    g = golden(name)
    host, cg, dl, eng0, pipe = build(g, aux_phases=2)
    eng = TrainEngine(cg, dl, host, lr=eng0.lr, lr_embeds=eng0.lr_embeds, table_agg_freq=eng0.agg_freq,
                      table_agg_op=eng0.agg_op, defer_top_update=defer)
    eng.gather_alone_min = 1
    assert eng.chain_take
    L = int(g["L"])
    batches = make_batches(g)
    dev_idx = [b[1].to(DEV) for b in batches]
    losses, chained = [], 0
    for j, (X, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            win = torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(DEV)
            if "reseed" not in g.files or bool(g["reseed"]):
                torch.manual_seed(5000 + j)
            pipe.plan_window(win)
            pipe.commit()
            pipe.wait_writeback()
            rs = WindowResolver(eng, win, int(g["B"]), chunk=chunk)
        nxt = dev_idx[j + 1] if j + 1 < len(batches) and (j + 1) % L != 0 else None
        if delay:
            with torch.cuda.stream(eng.side):
                torch.cuda._sleep(2_500_000)
        loss = eng.step(X.to(DEV), dev_idx[j], Tt.to(DEV), j=j, next_idx=nxt, res=rs.batch(j % L),
                        next_res=rs.batch(j % L + 1) if nxt is not None else None)
        chained += int(eng._pref is not None and bool(eng._pref.get("chained_top")) == (defer and eng.world == 1))
        rs.ensure(j % L + rs.CH + 2)
        losses.append(loss[0:1].clone())
    eng.finish()
    losses = [float(x) for x in losses]
    cg.ctx.check()
    assert chained >= len(batches) - len(batches) // L - 1
    np.testing.assert_allclose(np.array(losses), g["losses"], rtol=1e-5)
    occ = cg.occupancy_tables
    for k in range(len(g["ln_emb"])):
        assert torch.equal(occ[k].cpu(), t(g[f"occ_{k}"])), k
        w = cg.emb_l[k].weight[: int(g["ways"]) * cg.cache_sizes[k]].double().sum().item()
        np.testing.assert_allclose(w, float(g[f"weight_sum_{k}"]), rtol=1e-5, atol=1e-4)
    from cdlrm_amd.model_no_ddp import _linears
    for i, l in enumerate(_linears(dl.top_l)):
        np.testing.assert_allclose(l.weight.data.cpu().numpy(), g[f"top_w{i}"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name,long_batch", [("train_small", False), ("train_c1", False), ("train_c1", True)])
def test_native_tape_replays_the_recorded_step(golden, name, long_batch):
    """The C-side launch tape (csrc/tape.hip: one library call re-issues the ~45 recorded calls of a step, event records
    and stream waits included) against the same tape replayed from Python: bit-identical losses, tags and weights; and
    the native tapes really are the ones that ran -- single-lane on the null stream, and in TWO LANES on a stream of the
    trainer's own (what bench.py and Run use): the training queue's calls issued by this thread, the side queues' by the
    library's helper thread, ordered on the host by the events they share.  long_batch: the chained-take schedule."""
    from cdlrm_amd import _lib
    from cdlrm_amd.engine import WindowResolver
    assert _lib.native_tape_ok()
    g = golden(name)
    L = int(g["L"])
    batches = make_batches(g)
    runs = []
    for native in (False, True, "lanes"):
        if native == "lanes":
            torch.cuda.synchronize()
            own = torch.cuda.Stream(priority=-1)
            own.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(own):
                runs.append(_native_tape_run(g, batches, L, True, long_batch, want_lanes=2))
            torch.cuda.current_stream().wait_stream(own)
        else:
            runs.append(_native_tape_run(g, batches, L, native, long_batch, want_lanes=1))
    for r in runs[1:]:
        assert torch.equal(runs[0][0], r[0])
        assert torch.equal(runs[0][1], r[1]) and torch.equal(runs[0][2], r[2])
        for a, b in zip(runs[0][3], r[3]):
            assert torch.equal(a, b)
    np.testing.assert_allclose(runs[1][0].numpy(), g["losses"], rtol=1e-5)


def _native_tape_run(g, batches, L, native, long_batch, want_lanes):
    from cdlrm_amd import _lib
    from cdlrm_amd.engine import WindowResolver
    if True:
        host, cg, dl, eng, pipe = build(g, aux_phases=2)
        eng.native_tape = bool(native)
        if long_batch:
            eng.gather_alone_min = 1
        dev_idx = [b[1].to(DEV) for b in batches]
        losses = []
        for j, (X, lS_i, Tt) in enumerate(batches):
            if j % L == 0:
                win = torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(DEV)
                if "reseed" not in g.files or bool(g["reseed"]):
                    torch.manual_seed(5000 + j)
                pipe.plan_window(win)
                pipe.commit()
                pipe.wait_writeback()
                rs = WindowResolver(eng, win, int(g["B"]), chunk=3)
            nxt = dev_idx[j + 1] if j + 1 < len(batches) and (j + 1) % L != 0 else None
            loss = eng.step(X.to(DEV), dev_idx[j], Tt.to(DEV), j=j, next_idx=nxt, res=rs.batch(j % L),
                            next_res=rs.batch(j % L + 1) if nxt is not None else None)
            rs.ensure(j % L + rs.CH + 2)
            losses.append(loss[0:1].clone())
        eng.finish()
        torch.cuda.synchronize()
        n_native = sum(1 for t in eng._tapes.values() if t["native"] is not None)
        assert (n_native > 0) == bool(native) and len(eng._tapes) > 0 and not eng.tape_fallbacks
        if native:
            assert all(int(_lib.raw().cdlrm_tape_length(t["native"]._h)) >= len(t["prog"]) for t in eng._tapes.values())
            assert all((t["native"].lanes == 1) if want_lanes == 1 else (2 <= t["native"].lanes <= eng.tape_lanes)
                       for t in eng._tapes.values())
            if want_lanes > 1:      # main + side (+ the prefetch / weight-gradient queue of the short-batch schedule; at long
                                    # batches the prefetch queue appears in the steps that place a look-ahead chunk's resolve)
                lanes = sorted({t["native"].lanes for t in eng._tapes.values()})
                assert lanes in ([2], [2, 3]) if long_batch else max(lanes) == 3, lanes
        return (torch.cat(losses).cpu(), cg.tags.cpu().clone(), cg.weight.data.cpu().clone(),
                [l.weight.data.cpu().clone() for l in dl.top_l if hasattr(l, "weight")])


def test_host_gather_plan_equals_device_fetch(golden):
    """WindowPipeline(host_gather=True) -- CPU threads gather the winners' / victims' rows, one DMA copy each, from a
    background thread -- leaves exactly the cache, tag and victim state of the default plan (GPU waves reading the
    host tables), window after window, evictions written back included."""
    from cdlrm_amd.engine import WindowPipeline
    g = golden("train_small")
    L = int(g["L"])
    batches = make_batches(g)
    states = []
    for hg in (False, True):
        host, cg, dl, eng, _ = build(g)
        pipe = WindowPipeline(cg, host, L * int(g["B"]), parity_rng=False, seed=5, host_gather=hg, gather_threads=3)
        assert pipe.host_gather == hg
        snaps = []
        for j in range(0, len(batches), L):
            win = torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(DEV)
            pipe.plan_window(win)
            pipe.commit()
            pipe.wait_writeback()
            torch.cuda.synchronize()
            vic = pipe.victims[pipe._vnext ^ 1]
            voff = vic.off.cpu().tolist()
            nv = voff[-2]               # off[T]: entries listed (off[T + 1]: what the window has, before the cap)
            snaps.append((cg.tags.cpu().clone(), [cg.emb_l[k].weight[: int(g["ways"]) * cg.cache_sizes[k]].cpu().clone()
                                                  for k in range(len(cg.cache_sizes))], voff,
                          vic.idx[:nv].cpu().clone(), vic.rows[:nv].cpu().clone(),
                          [h.weight.data.clone() for h in host.emb_l]))
        cg.ctx.check()
        states.append(snaps)
    assert sum(s[2][-1] for s in states[0]) > 0, "the fixture must produce victims"
    for a, b in zip(*states):
        assert torch.equal(a[0], b[0])
        for x, y in zip(a[1], b[1]):
            assert torch.equal(x, y)
        assert a[2] == b[2] and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
        for x, y in zip(a[5], b[5]):
            assert torch.equal(x, y)


def test_dropin_autograd_surface_matches_engine(golden):
    """The reference-shaped loop (cache_group(...) -> dlrm(...) -> loss -> backward -> optimizer steps) through the
    autograd wrappers gives the same trajectory as the fused engine."""
    from cdlrm_amd.model_no_ddp import CacheSGD, HipBCELoss
    from cdlrm_amd.engine import WindowPipeline
    g = golden("train_small")
    host, cg, dl, eng, pipe = build(g)
    del eng
    L = int(g["L"])
    loss_fn = HipBCELoss()
    opt_m = torch.optim.SGD(dl.parameters(), lr=float(g["lr"]))
    opt_e = CacheSGD(cg, lr=float(g["lr_emb"]))
    batches = make_batches(g)
    losses = []
    for j, (X, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            win = torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(DEV)
            torch.manual_seed(5000 + j)
            pipe.plan_window(win)
            pipe.commit()
            pipe.wait_writeback()
        lS_o = torch.arange(X.shape[0]).repeat(lS_i.shape[0], 1)
        lookups, cgi = cg(lS_o, lS_i, host, 0)
        Z = dl(X.to(DEV), lookups)
        E = loss_fn(Z, Tt.to(DEV))
        opt_m.zero_grad()
        opt_e.zero_grad()
        E.backward()
        opt_e.step()
        opt_m.step()
        losses.append(float(E))
        assert cgi[0].dtype == torch.int32
    np.testing.assert_allclose(np.array(losses), g["losses"], rtol=1e-5)
    for k in range(len(g["ln_emb"])):
        assert torch.equal(cg.occupancy_tables[k].cpu(), t(g[f"occ_{k}"]))


@pytest.mark.parametrize("name,pipelined", [("train_c1", True), ("train_small", False), ("train_stream", True)])
def test_deferred_top_update_same_trajectory(golden, name, pipelined):
    """defer_top_update: the top MLP's weight gradients + SGD run on the side stream beside the rest of the backward
    and the head of the next step.  A schedule change only: the reference's loss trajectory and final weights."""
    g = golden(name)
    from cdlrm_amd.engine import TrainEngine
    host, cg, dl, eng0, pipe = build(g)
    eng = TrainEngine(cg, dl, host, lr=eng0.lr, lr_embeds=eng0.lr_embeds, table_agg_freq=eng0.agg_freq,
                      table_agg_op=eng0.agg_op, defer_top_update=True)
    L = int(g["L"])
    batches = make_batches(g)
    dev_idx = [b[1].to(DEV) for b in batches]
    losses = []
    for j, (X, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            win = torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(DEV)
            if "reseed" not in g.files or bool(g["reseed"]):
                torch.manual_seed(5000 + j)
            pipe.plan_window(win)
            pipe.commit()
            pipe.wait_writeback()
        nxt = dev_idx[j + 1] if pipelined and j + 1 < len(batches) and (j + 1) % L != 0 else None
        loss = eng.step(X.to(DEV), dev_idx[j], Tt.to(DEV), j=j, next_idx=nxt)
        losses.append(loss[0:1].clone())
    eng.finish()
    np.testing.assert_allclose(np.array([float(x) for x in losses]), g["losses"], rtol=1e-5)
    for k in range(len(g["ln_emb"])):
        assert torch.equal(cg.occupancy_tables[k].cpu(), t(g[f"occ_{k}"])), k
    from cdlrm_amd.model_no_ddp import _linears
    for i, l in enumerate(_linears(dl.top_l)):
        np.testing.assert_allclose(l.weight.data.cpu().numpy(), g[f"top_w{i}"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("op,loss,ws,thr,defer", [("cat", "bce", None, 0.0, False), ("dot", "mse", None, 0.0, True),
                                                  ("dot", "wbce", (0.4, 2.5), 0.0, False),
                                                  ("cat", "wbce", (1.5, 0.7), 0.45, True),
                                                  ("dot", "bce", None, 0.48, False)])
def test_engine_loss_and_interaction_arms_vs_oracle(op, loss, ws, thr, defer):
    """--arch-interaction-op=cat, --loss-function=mse|wbce, --loss-threshold through the fused engine: loss per
    iteration, prediction and tag state against the oracle's trainer (itself pinned on the reference's DLRM_Net +
    loss_fn_wrap by tests/golden/dense_*.npz)."""
    from cdlrm_amd.engine import TrainEngine, WindowPipeline
    from cdlrm_amd.model_no_ddp import DLRM_Net, Embedding_Table_Cache_Group, Embedding_Table_Group
    from oracle import cdlrm_oracle as O
    ln_emb, m_spa, B, L, ways, cache_size, seed = [3000, 50, 7, 1200], 16, 48, 3, 4, 40, 17
    ln_bot = np.array([13, 32, m_spa])
    nf = len(ln_emb) + 1
    ln_top = np.array([(m_spa + nf * (nf - 1) // 2) if op == "dot" else nf * m_spa, 24, 1])
    rng = np.random.RandomState(3)
    batches = []
    for j in range(9):
        X = torch.from_numpy(rng.rand(B, 13).astype(np.float32))
        idx = torch.stack([torch.from_numpy((rng.zipf(1.2, size=B).astype(np.int64) * 2654435761 % n)) for n in ln_emb])
        Tt = torch.from_numpy(np.round(rng.rand(B, 1)).astype(np.float32))
        batches.append((X, idx, Tt))
    torch.set_num_threads(1)
    otr = O.OracleTrainer(ln_emb, m_spa, ln_bot, ln_top, cache_size=cache_size, num_ways=ways, mini_batch_size=B,
                          lr=0.1, lr_embeds=0.3, lookahead=L, table_agg_freq=10 ** 9, seed=seed, loss=loss, op=op,
                          loss_weights=ws, loss_threshold=thr)
    lS_o = torch.arange(B).repeat(len(ln_emb), 1)
    for j, (X, idx, Tt) in enumerate(batches):
        if j % L == 0:
            torch.manual_seed(700 + j)
            otr.refill(torch.cat([b[1] for b in batches[j:j + L]], dim=1))
        otr.step(j, X, lS_o, idx, Tt)
    Zo = otr.evaluate(batches[-1][0], lS_o, batches[-1][1])
    np.random.seed(seed)
    torch.manual_seed(seed)
    host = Embedding_Table_Group(m_spa, np.array(ln_emb)).pin()
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = Embedding_Table_Cache_Group(m_spa, np.array(ln_emb), cache_size, B, ways).to(DEV)
    dl = DLRM_Net(ln_bot, ln_top, op, False, True, -1, ln_top.size - 2, thr).to(DEV)
    eng = TrainEngine(cg, dl, host, lr=0.1, lr_embeds=0.3, loss=loss, loss_weights=ws or (1.0, 1.0),
                      defer_top_update=defer)
    pipe = WindowPipeline(cg, host, L * B, parity_rng=True)
    losses = []
    for j, (X, idx, Tt) in enumerate(batches):
        if j % L == 0:
            torch.manual_seed(700 + j)
            pipe.plan_window(torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(DEV))
            pipe.commit()
            pipe.wait_writeback()
        lossbuf = eng.step(X.to(DEV), idx.to(DEV), Tt.to(DEV), j=j)
        losses.append(float(lossbuf[0]))
        # the print statistics the loss kernel leaves for Run (main_no_ddp.py:431-433): #correct and loss * mbs
        Zp = eng.prediction(B)
        assert float(lossbuf[1]) == float((torch.round(Zp) == Tt.to(DEV)).sum())
        assert float(lossbuf[2]) == float(lossbuf[0] * B)
    Zg = eng.evaluate(batches[-1][0].to(DEV), batches[-1][1].to(DEV))
    cg.ctx.check()
    np.testing.assert_allclose(np.array(losses), np.array([l[0] for l in otr.losses]), rtol=1e-5)
    np.testing.assert_allclose(Zg.cpu().numpy(), Zo.numpy(), rtol=1e-5, atol=1e-6)
    for k in range(len(ln_emb)):
        assert torch.equal(cg.occupancy_tables[k].cpu(), otr.occ[k]), k


@pytest.mark.parametrize("fixed,op", [(False, "dot"), (True, "cat")])
def test_ragged_multihot_bags_vs_oracle(fixed, op):
    """The reference's default data mode (--data-generation=random: uniform multi-hot bags, tables ragged against each
    other, dlrm_data_pytorch.py:763-805) through the fused engine: engine.square_bags() squares the tables off with one
    scratch bag; loss per iteration, prediction and tag state against the oracle's trainer on the raw ragged lists."""
    from types import SimpleNamespace
    from cdlrm_amd import dlrm_data_pytorch as DP
    from cdlrm_amd.engine import TrainEngine, WindowPipeline, pad_window, square_bags
    from cdlrm_amd.model_no_ddp import DLRM_Net, Embedding_Table_Cache_Group, Embedding_Table_Group
    from oracle import cdlrm_oracle as O
    ln_emb, m_spa, B, L, ways, cache_size, seed = [900, 40, 6, 2500], 16, 32, 3, 4, 300, 23
    aux = 512                       # >= lookups per table and batch (the squared-off width): misses get their own aux row
    ln_bot = np.array([5, 32, m_spa])
    nf = len(ln_emb) + 1
    ln_top = np.array([(m_spa + nf * (nf - 1) // 2) if op == "dot" else nf * m_spa, 24, 1])
    args = SimpleNamespace(data_size=0, num_batches=9, mini_batch_size=B, num_indices_per_lookup=7,
                           num_indices_per_lookup_fixed=fixed, round_targets=True, data_generation="random",
                           numpy_rand_seed=seed)
    _, loader = DP.make_random_data_and_loader(args, np.array(ln_emb), 5)
    batches = [(X, [o for o in lS_o], lS_i, Tt) for X, lS_o, lS_i, Tt in loader]
    assert len({int(x.numel()) for x in batches[0][2]}) > 1, "ragged tables expected"
    torch.set_num_threads(1)
    otr = O.OracleTrainer(ln_emb, m_spa, ln_bot, ln_top, cache_size=cache_size, num_ways=ways, mini_batch_size=aux,
                          lr=0.1, lr_embeds=0.3, lookahead=L, table_agg_freq=10 ** 9, seed=seed, op=op)
    for j, (X, lS_o, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            torch.manual_seed(800 + j)
            otr.refill([torch.cat([b[2][k] for b in batches[j:j + L]]) for k in range(len(ln_emb))])
        otr.step(j, X, lS_o, lS_i, Tt)
    np.random.seed(seed)
    torch.manual_seed(seed)
    host = Embedding_Table_Group(m_spa, np.array(ln_emb)).pin()
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = Embedding_Table_Cache_Group(m_spa, np.array(ln_emb), cache_size, aux, ways).to(DEV)
    dl = DLRM_Net(ln_bot, ln_top, op, False, True, -1, ln_top.size - 2, 0.0).to(DEV)
    eng = TrainEngine(cg, dl, host, lr=0.1, lr_embeds=0.3)
    pipe = WindowPipeline(cg, host, 4096, parity_rng=True)
    losses = []
    for j, (X, lS_o, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            torch.manual_seed(800 + j)
            pipe.plan_window(pad_window([torch.cat([b[2][k] for b in batches[j:j + L]]) for k in range(len(ln_emb))], DEV))
            pipe.commit()
            pipe.wait_writeback()
        off, idx = square_bags(lS_o, lS_i, DEV)
        assert off.shape == (len(ln_emb), B + 1) and idx.shape[1] % 256 == 0
        losses.append(float(eng.step(X.to(DEV), idx, Tt.to(DEV), lS_o=off, j=j)[0]))
    cg.ctx.check()
    np.testing.assert_allclose(np.array(losses), np.array([l[0] for l in otr.losses]), rtol=1e-5)
    for k in range(len(ln_emb)):
        assert torch.equal(cg.occupancy_tables[k].cpu(), otr.occ[k]), k
        nrow = ways * cg.cache_sizes[k]
        np.testing.assert_allclose(cg.emb_l[k].weight[:nrow].cpu().numpy(), otr.weights[0][k][:nrow].numpy(),
                                   rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("m_spa", [256, 64])
def test_embed_dim_256_engine_vs_oracle(m_spa):
    """BASELINE config c4's embedding width (256; 64 for the middle template) through every kernel of the step: probe,
    gather, interaction (MFMA), MLPs, embedding backward + sparse SGD, window insert / evict -- loss per iteration,
    tags, cache rows against the oracle's trainer."""
    from cdlrm_amd.engine import TrainEngine, WindowPipeline
    from cdlrm_amd.model_no_ddp import DLRM_Net, Embedding_Table_Cache_Group, Embedding_Table_Group
    from oracle import cdlrm_oracle as O
    ln_emb, B, L, ways, cache_size, seed = [2000, 30, 5, 900, 20000, 64], 96, 3, 4, 48, 29
    ln_bot = np.array([13, 64, m_spa])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2, 64, 1])
    rng = np.random.RandomState(8)
    batches = []
    for j in range(9):
        X = torch.from_numpy(rng.rand(B, 13).astype(np.float32))
        idx = torch.stack([torch.from_numpy((rng.zipf(1.2, size=B).astype(np.int64) * 2654435761 % n)) for n in ln_emb])
        Tt = torch.from_numpy(np.round(rng.rand(B, 1)).astype(np.float32))
        batches.append((X, idx, Tt))
    torch.set_num_threads(1)
    otr = O.OracleTrainer(ln_emb, m_spa, ln_bot, ln_top, cache_size=cache_size, num_ways=ways, mini_batch_size=B,
                          lr=0.05, lr_embeds=0.2, lookahead=L, table_agg_freq=10 ** 9, seed=seed)
    lS_o = torch.arange(B).repeat(len(ln_emb), 1)
    for j, (X, idx, Tt) in enumerate(batches):
        if j % L == 0:
            torch.manual_seed(600 + j)
            otr.refill(torch.cat([b[1] for b in batches[j:j + L]], dim=1))
        otr.step(j, X, lS_o, idx, Tt)
    np.random.seed(seed)
    torch.manual_seed(seed)
    host = Embedding_Table_Group(m_spa, np.array(ln_emb)).pin()
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = Embedding_Table_Cache_Group(m_spa, np.array(ln_emb), cache_size, B, ways).to(DEV)
    dl = DLRM_Net(ln_bot, ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0).to(DEV)
    eng = TrainEngine(cg, dl, host, lr=0.05, lr_embeds=0.2)
    pipe = WindowPipeline(cg, host, L * B, parity_rng=True)
    losses = []
    dev_idx = [b[1].to(DEV) for b in batches]
    for j, (X, idx, Tt) in enumerate(batches):
        if j % L == 0:
            torch.manual_seed(600 + j)
            pipe.plan_window(torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(DEV))
            pipe.commit()
            pipe.wait_writeback()
        nxt = dev_idx[j + 1] if j + 1 < len(batches) and (j + 1) % L != 0 else None
        losses.append(eng.step(X.to(DEV), dev_idx[j], Tt.to(DEV), j=j, next_idx=nxt)[0:1].clone())
    cg.ctx.check()
    np.testing.assert_allclose(np.array([float(x) for x in losses]), np.array([l[0] for l in otr.losses]), rtol=1e-5)
    for k in range(len(ln_emb)):
        assert torch.equal(cg.occupancy_tables[k].cpu(), otr.occ[k]), k
        nrow = ways * cg.cache_sizes[k]
        np.testing.assert_allclose(cg.emb_l[k].weight[:nrow].cpu().numpy(), otr.weights[0][k][:nrow].numpy(),
                                   rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(host.emb_l[k].weight.data.numpy(), otr.host[k].numpy(), rtol=2e-5, atol=1e-7)
