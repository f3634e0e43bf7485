"""GPU parity tests of the C-ABI kernels against the CPU oracle and the reference's golden vectors.
Run on the MI355X box:  python -m pytest tests -m gpu -x -q"""
import numpy as np
import pytest
import torch

from oracle import cdlrm_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def ops():
    from cdlrm_amd import ops as _ops
    from cdlrm_amd import _lib
    _lib.lib()
    return _ops


class DevState:
    """Flat device cache state + pinned host tables built from per-table CPU tensors."""

    def __init__(self, ops, ln_emb, cache_sizes, D, ways, aux, occ, weights, host):
        self.ops = ops
        self.ctx = ops.CacheCtx(ln_emb, cache_sizes, D, ways, aux, torch.device(DEV))
        self.tags = torch.cat([o.reshape(-1) for o in occ]).to(DEV)
        self.weight = torch.cat(list(weights)).contiguous().to(DEV)
        self.ctx.bind_cache(self.tags, self.weight)
        self.host = [h.clone().pin_memory() for h in host]
        self.ctx.bind_host_tables([h.data_ptr() for h in self.host])
        self.cache_sizes, self.ways = list(cache_sizes), ways

    def occ(self, k):
        c = self.ctx
        return self.tags[c.tag_base[k]:c.tag_base[k + 1]].view(c.cache_sets[k], c.ways).cpu()

    def w(self, k):
        c = self.ctx
        return self.weight[c.row_base[k]:c.row_base[k + 1]].cpu()


def load_windows(g):
    ln_emb = [int(x) for x in g["ln_emb"]]
    T, ways, B = len(ln_emb), int(g["ways"]), int(g["B"])
    cache_sizes = [int(x) for x in g["cache_sizes"]]
    return ln_emb, T, ways, B, cache_sizes, int(g["m_spa"])


@pytest.mark.parametrize("name", ["cache_windows_small", "cache_windows_uniform"])
def test_probe_and_gather_forward(ops, golden, name):
    g = golden(name)
    ln_emb, T, ways, B, cache_sizes, D = load_windows(g)
    for w in range(int(g["nwin"])):
        occ = [t(g[f"w{w}_occ_{k}"]) for k in range(T)]
        weights = [t(g[f"w{w}_weight_{k}"]) for k in range(T)]
        host = [t(g[f"w{w}_host_{k}"]) for k in range(T)]
        st = DevState(ops, ln_emb, cache_sizes, D, ways, B, occ, weights, host)
        lS_i = t(g[f"w{w}_fwd_lS_i"]).to(DEV)
        slots, miss_pos, miss_count = ops.embbag_probe(st.ctx, lS_i)
        feat = torch.zeros(B, T + 1, D, device=DEV)
        ops.embbag_fwd(st.ctx, slots, None, feat[:, 1:, :], (T + 1) * D, D)
        st.ctx.check()
        for k in range(T):
            assert torch.equal(slots[k].cpu(), t(g[f"w{w}_fwd_idx_{k}"])), (w, k)          # bit-exact slot ids
            assert torch.equal(feat[:, k + 1, :].cpu(), t(g[f"w{w}_fwd_ly_{k}"])), (w, k)  # exact row copies
            assert torch.equal(st.w(k), t(g[f"w{w}_fwd_weight_{k}"])), (w, k)             # aux rows written
            nm = int(miss_count[k])
            want_miss = (t(g[f"w{w}_fwd_idx_{k}"]) >= cache_sizes[k] * ways).nonzero().flatten()
            assert torch.equal(miss_pos[k, :nm].cpu().long(), want_miss)
        assert torch.all(feat[:, 0, :] == 0)


@pytest.mark.parametrize("with_victims", [False, True])
def test_window_resolve_take_equals_probe(ops, with_victims):
    """The window-resident probe (cdlrm_window_resolve once per window + cdlrm_embbag_take per batch) leaves exactly what
    cdlrm_embbag_probe leaves for every batch: the same slot ids (misses numbered per batch in position order, in the
    requested aux region) and the same aux rows -- with the misses' rows coming from the host tables or, when a window's
    victim rows are bound, from there (and from the host for misses the list does not hold)."""
    from cdlrm_amd.model_no_ddp import Embedding_Table_Cache_Group, Embedding_Table_Group
    from cdlrm_amd.engine import WindowPipeline
    rng = np.random.RandomState(12)
    ln_emb, D, ways, B, nb, cache = np.array([5000, 64, 9, 1300, 20000]), 16, 4, 96, 7, 50
    T = len(ln_emb)
    np.random.seed(3)
    torch.manual_seed(3)
    host = Embedding_Table_Group(D, ln_emb).pin()
    cg = Embedding_Table_Cache_Group(D, ln_emb, cache, B, ways).to(DEV)
    ctx = cg.ctx
    ctx.bind_host_tables(host.device_pointers())
    win = torch.stack([torch.from_numpy((rng.zipf(1.15, size=nb * B).astype(np.int64) * 2654435761 % n)) for n in ln_emb]).to(DEV)
    pipe = WindowPipeline(cg, host, nb * B, parity_rng=False, seed=5, victim_rows=(400 if with_victims else 0))
    prev = torch.stack([torch.from_numpy(rng.randint(0, n, size=nb * B).astype(np.int64)) for n in ln_emb]).to(DEV)
    for w in (prev, win):           # two windows: the second insert meets full sets -> victims
        pipe.plan_window(w)
        pipe.commit()
        pipe.wait_writeback()
    torch.cuda.synchronize()
    if with_victims:
        nv = int(pipe.victims[pipe._vnext ^ 1].off.cpu()[-2])       # off[T]: entries listed (off[T + 1]: before the cap)
        assert 0 < nv, "the fixture must produce victims"
    # lookups: the window's own batches plus ids the window never saw (host fallback)
    look = win.clone()
    for k, n in enumerate(ln_emb):
        look[k, ::7] = torch.from_numpy(rng.randint(0, n, size=look[k, ::7].numel())).to(DEV)
    ws = torch.empty(T, nb * B, dtype=torch.int32, device=DEV)
    wsrc = torch.empty_like(ws)
    ops.window_resolve(ctx, look, B, ws, wsrc)
    for phase in (0, 1):
        for j in range(nb):
            idx = look[:, j * B:(j + 1) * B]
            cg.weight.data[:] = cg.weight.data         # no-op; keeps the buffer identity
            slots_p, _, mc = ops.embbag_probe(ctx, idx.contiguous(), aux_phase=phase)
            torch.cuda.synchronize()
            rows_p = cg.weight.data.clone()
            # wipe the aux rows, then take
            for k in range(T):
                a0 = cg.row_base[k] + ways * cg.cache_sizes[k]
                cg.weight.data[a0:a0 + 2 * B] = -7.0
            slots_t = torch.empty(T, B, dtype=torch.int32, device=DEV)
            ops.embbag_take(ctx, idx, ws[:, j * B:(j + 1) * B], wsrc[:, j * B:(j + 1) * B], slots_t, aux_phase=phase)
            torch.cuda.synchronize()
            assert torch.equal(slots_t, slots_p), (phase, j)
            assert int(mc.sum()) > 0
            for k in range(T):
                a0 = cg.row_base[k] + ways * cg.cache_sizes[k] + phase * B
                m = int(mc[k])
                assert torch.equal(cg.weight.data[a0:a0 + m], rows_p[a0:a0 + m]), (phase, j, k)
    ctx.check()


def test_appendix_a(ops, golden):
    g = golden("appendix_a")
    P = int(g["P"][0])
    st = DevState(ops, [50], [P], 2 + 2, 2, 4, [torch.full((P, 2), -1, dtype=torch.int64)],
                  [torch.zeros(2 * P + 4, 4)], [torch.cat([t(g["host"]), t(g["host"])], 1)])
    plan = ops.WindowPlan(st.ctx, 64)
    for w in range(2):
        raw = t(g[f"w{w}_raw"]).view(1, -1).to(DEV)
        plan.unique(raw)
        plan.probe()
        uo, ko, _ = plan.offsets()
        assert torch.equal(plan.uniq[:uo[1]].cpu(), t(g[f"w{w}_uniq"]))
        q = t(g[f"w{w}_q"]).to(DEV)
        assert ko[1] == q.shape[0]
        plan.assign(q)
        plan.fetch([h.data_ptr() for h in st.host], False)
        plan.commit()
        plan.writeback([h.data_ptr() for h in st.host], False)
        torch.cuda.synchronize()
        st.ctx.check()
        assert torch.equal(plan.way[:ko[1]].cpu().long(), t(g[f"w{w}_way"]))
        assert torch.equal(st.occ(0), t(g[f"w{w}_occ"]))
        assert torch.equal(st.w(0)[:, :2], t(g[f"w{w}_weight"]))
    assert st.occ(0).tolist() == [[-1, -1], [16, 26], [-1, -1], [8, 3], [-1, 4]]


@pytest.mark.parametrize("by_position", [False, True])
@pytest.mark.parametrize("name", ["cache_windows_small", "cache_windows_uniform"])
def test_window_insert_evict(ops, golden, name, by_position):
    """unique scan -> probe -> way choice (reference's own q) -> fetch -> commit -> write-back over
    consecutive windows: tags bit-exact, rows exact copies, evictions set-equal, host tables equal."""
    g = golden(name)
    ln_emb, T, ways, B, cache_sizes, D = load_windows(g)
    occ = O.new_occupancy_tables(cache_sizes, ways)
    weights = [t(g[f"weight0_{k}"]) for k in range(T)]
    host = [t(g[f"host0_{k}"]) for k in range(T)]
    st = DevState(ops, ln_emb, cache_sizes, D, ways, B, occ, weights, host)
    plan = ops.WindowPlan(st.ctx, int(g["L"]) * B)
    for w in range(int(g["nwin"])):
        win = t(g[f"w{w}_win"]).to(DEV)
        plan.unique(win)
        plan.probe()
        uo, ko, _ = plan.offsets()
        q_parts = []
        for k in range(T):
            assert torch.equal(plan.uniq[uo[k]:uo[k + 1]].cpu(), t(g[f"w{w}_uniq_{k}"])), (w, k)
            qk = t(g[f"w{w}_q_{k}"])
            assert ko[k + 1] - ko[k] == qk.shape[0], (w, k)
            q_parts.append(qk.reshape(-1, ways))
        plan.assign(torch.cat(q_parts).contiguous().to(DEV))
        if by_position:
            rows = [ops.gather_rows(st.host[k].data_ptr(), plan.uniq[uo[k]:uo[k + 1]], D) for k in range(T)]
            for k in range(T):
                assert torch.equal(rows[k].cpu(), t(g[f"w{w}_rows_{k}"]))
            # empty tables still need a valid pointer
            plan.fetch([r.data_ptr() if r.numel() else st.weight.data_ptr() for r in rows], True)
        else:
            plan.fetch([h.data_ptr() for h in st.host], False)
        plan.commit()
        plan.writeback([h.data_ptr() for h in st.host], False)
        _, _, wo = plan.offsets()
        torch.cuda.synchronize()
        st.ctx.check()
        for k in range(T):
            assert torch.equal(plan.way[ko[k]:ko[k + 1]].cpu().long(), t(g[f"w{w}_way_{k}"])), (w, k)
            assert torch.equal(st.occ(k), t(g[f"w{w}_occ_{k}"])), (w, k)
            nslots = cache_sizes[k] * ways
            assert torch.equal(st.w(k)[:nslots], t(g[f"w{w}_weight_{k}"])[:nslots]), (w, k)
            ev_tag = plan.ev_tag[wo[k]:wo[k + 1]].cpu()
            ev_rows = plan.stage[wo[k]:wo[k + 1]].cpu()
            valid = ev_tag != -1
            gi, gr = O.dedup_evictions(t(g[f"w{w}_ev_idx_{k}"]), t(g[f"w{w}_ev_rows_{k}"]))
            mi, mr = O.dedup_evictions(ev_tag[valid], ev_rows[valid])
            assert torch.equal(mi, gi) and torch.equal(mr, gr), (w, k)
            assert torch.equal(st.host[k], t(g[f"w{w}_host_{k}"])), (w, k)
        # scratch must be clean for the next window
        assert int((plan.winner != -1).sum()) == 0 and int(plan.prot.abs().sum()) == 0
        assert int(plan.bitmap.abs().sum()) == 0


def test_device_rng_mode_is_valid_insert(ops, golden):
    """Perf mode (Philox on device) is not bit-comparable, but must produce a legal insert: no protected
    way overwritten, tags unique per set, every resident tag maps to its set, rows match the host."""
    g = golden("cache_windows_small")
    ln_emb, T, ways, B, cache_sizes, D = load_windows(g)
    occ = O.new_occupancy_tables(cache_sizes, ways)
    st = DevState(ops, ln_emb, cache_sizes, D, ways, B, occ, [t(g[f"weight0_{k}"]) for k in range(T)],
                  [t(g[f"host0_{k}"]) for k in range(T)])
    plan = ops.WindowPlan(st.ctx, int(g["L"]) * B)
    for w in range(int(g["nwin"])):
        win = t(g[f"w{w}_win"]).to(DEV)
        pre = [st.occ(k).clone() for k in range(T)]
        plan.unique(win); plan.probe(); plan.assign(None, seed=1234 + w)
        plan.fetch([h.data_ptr() for h in st.host], False); plan.commit()
        plan.writeback([h.data_ptr() for h in st.host], False)
        torch.cuda.synchronize(); st.ctx.check()
        for k in range(T):
            o = st.occ(k)
            u = torch.unique(win[k].cpu())
            P = cache_sizes[k]
            # hits stay where they were
            was = (pre[k][u % P] == u.view(-1, 1))
            assert torch.equal(o[u % P][was], pre[k][u % P][was])
            res = o[o != -1]
            assert res.numel() == torch.unique(res).numel()
            sets = (o != -1).nonzero()[:, 0]
            assert torch.equal(res % P, sets)
            slots_rows = st.w(k)[: P * ways].view(ways, P, D).permute(1, 0, 2)[o != -1]
            # rows of tags inserted in THIS window equal the host rows (host is written back synchronously)
            new = ~torch.isin(res, pre[k][pre[k] != -1])
            assert torch.equal(slots_rows[new], st.host[k][res[new]])


@pytest.mark.parametrize("cap_mode", ["full", "cut"])
def test_window_victims_serve_misses_from_hbm(ops, golden, cap_mode):
    """cdlrm_plan_victims lists exactly the window's indices that stay outside the cache (unique - hits - winners)
    with their host rows; with the list bound, the per-iteration probe fills aux rows with the same values the host
    tables hold -- also for indices outside the list (not from the window / list cut at cap): host fallback."""
    g = golden("cache_windows_small")
    ln_emb, T, ways, B, cache_sizes, D = load_windows(g)
    occ = O.new_occupancy_tables(cache_sizes, ways)
    st = DevState(ops, ln_emb, cache_sizes, D, ways, 4 * B, occ, [torch.zeros(ways * cache_sizes[k] + 4 * B, D) for k in range(T)],
                  [t(g[f"host0_{k}"]) for k in range(T)])
    plan = ops.WindowPlan(st.ctx, int(g["L"]) * B)
    rng = np.random.RandomState(5)
    saw_victims = 0
    for w in range(int(g["nwin"])):
        win = t(g[f"w{w}_win"]).to(DEV)
        pre = [st.occ(k).clone() for k in range(T)]
        plan.unique(win); plan.probe(); plan.assign(None, seed=77 + w)
        plan.fetch([h.data_ptr() for h in st.host], False)
        uo, ko, wo = plan.offsets()
        expect = []
        for k in range(T):
            u = torch.unique(win[k].cpu())
            resident = pre[k][pre[k] != -1]
            winners = plan.win_idx[wo[k]:wo[k + 1]].cpu()
            expect.append(u[~torch.isin(u, resident) & ~torch.isin(u, winners)])
        total = sum(int(e.numel()) for e in expect)
        cap = max(1, total if cap_mode == "full" else total // 2)
        vic = ops.Victims(st.ctx, cap)
        plan.victims(vic)
        plan.commit()
        plan.writeback([h.data_ptr() for h in st.host], False)
        torch.cuda.synchronize(); st.ctx.check()
        off = vic.off.cpu().tolist()
        flat = torch.cat(expect)[:cap]
        assert off[T] == min(total, cap) and off[T + 1] == total       # listed / what the window has (a cut list says so)
        assert torch.equal(vic.idx[:off[T]].cpu(), flat)
        lo = 0
        for k in range(T):      # per-table boundaries of the (possibly cut) list
            hi = min(cap, lo + int(expect[k].numel()))
            assert off[k] == lo
            assert torch.equal(vic.rows[lo:hi].cpu(), st.host[k][flat[lo:hi]])
            lo = hi
        saw_victims += total
        # per-iteration probe: window ids + ids from outside the window, victims bound vs host reads
        n = 3 * B
        idx = torch.stack([torch.from_numpy(np.concatenate([
            rng.choice(win[k].cpu().numpy(), n - n // 4), rng.randint(0, ln_emb[k], n // 4)])) for k in range(T)]).to(DEV)
        st.ctx.bind_victims(vic)
        s1, mp1, mc1 = ops.embbag_probe(st.ctx, idx)
        torch.cuda.synchronize()
        aux1 = [st.w(k)[ways * cache_sizes[k]:].clone() for k in range(T)]
        st.weight.zero_()
        st.ctx.bind_victims(None)
        s2, mp2, mc2 = ops.embbag_probe(st.ctx, idx)
        torch.cuda.synchronize(); st.ctx.check()
        assert torch.equal(s1, s2) and torch.equal(mc1, mc2)
        for k in range(T):
            m = int(mc1[k])
            ids = idx[k].cpu()[mp1[k, :m].long().cpu()]
            assert torch.equal(aux1[k][:m], st.host[k][ids]), k
            assert torch.equal(st.w(k)[ways * cache_sizes[k]:][:m], st.host[k][ids]), k
        st.weight.zero_()
    assert saw_victims > 0, "the fixture must exercise the victim path"


@pytest.mark.parametrize("name", ["embsgd_onehot", "embsgd_multihot"])
def test_embbag_bwd_sgd_golden(ops, golden, name):
    g = golden(name)
    w0 = t(g["w0"])
    rows, D = w0.shape
    st = DevState(ops, [rows], [rows // 2], D, 2, 0, [torch.full((rows // 2, 2), -1, dtype=torch.int64)], [w0],
                  [torch.zeros(rows, D)])
    slots = t(g["slots"]).to(torch.int32).view(1, -1).to(DEV)
    offs = t(g["offsets"]).view(1, -1).to(DEV)
    nb = offs.shape[1]
    onehot = name.endswith("onehot")
    out = torch.zeros(nb, 1, D, device=DEV)
    ops.embbag_fwd(st.ctx, slots, None if onehot else offs, out, D, D)
    np.testing.assert_allclose(out[:, 0].cpu().numpy(), g["V"], rtol=1e-6, atol=1e-7)
    grad = t(g["grad"]).view(nb, 1, D).contiguous().to(DEV)
    work = ops.embbag_bwd_work(st.ctx, slots.shape[1], DEV)
    touched = torch.zeros(st.ctx.total_rows, dtype=torch.uint8, device=DEV)
    ops.embbag_bwd_sgd(st.ctx, slots, None if onehot else offs, grad, D, D, float(g["lr"]), work, touched)
    torch.cuda.synchronize()
    np.testing.assert_allclose(st.w(0).numpy(), g["w1"], rtol=1e-6, atol=1e-7)
    assert torch.equal(touched.cpu().nonzero().flatten(), torch.unique(t(g["slots"])))


@pytest.mark.parametrize("T,n,D,nslots,hot", [(3, 1000, 16, 64, 0.0), (2, 9000, 32, 5000, 0.6), (1, 20000, 128, 300, 0.3),
                                            (26, 2048, 64, 100000, 0.2)])
def test_embbag_bwd_sgd_vs_oracle(ops, T, n, D, nslots, hot):
    """Heavy repeats (long segments, multi-chunk sort/merge) against the oracle; result must also be
    bitwise reproducible run to run (no atomics)."""
    rng = np.random.RandomState(T * 1000 + n)
    P = (nslots + 1) // 2
    st = DevState(ops, [2 * P + 8] * T, [P] * T, D, 2, 8, [torch.full((P, 2), -1, dtype=torch.int64)] * T,
                  [torch.from_numpy(rng.randn(2 * P + 8, D).astype(np.float32)) for _ in range(T)],
                  [torch.zeros(2 * P + 8, D)] * T)
    slots = rng.randint(0, 2 * P, (T, n))
    if hot > 0:
        m = rng.rand(T, n) < hot
        slots[m] = 7
    slots_t = torch.from_numpy(slots.astype(np.int32)).to(DEV)
    grad = torch.from_numpy(rng.randn(n, T, D).astype(np.float32)).to(DEV)
    w_before = st.weight.clone()
    work = ops.embbag_bwd_work(st.ctx, n, DEV)
    results = []
    for rep in range(2):
        st.weight.copy_(w_before)
        ops.embbag_bwd_sgd(st.ctx, slots_t, None, grad, T * D, D, 0.25, work, None)
        torch.cuda.synchronize()
        results.append(st.weight.clone())
    assert torch.equal(results[0], results[1])
    for k in range(T):
        w = w_before[st.ctx.row_base[k]:st.ctx.row_base[k + 1]].cpu().clone()
        O.embbag_bwd_sgd(w, torch.from_numpy(slots[k]), torch.arange(n), grad[:, k, :].cpu(), 0.25)
        np.testing.assert_allclose(st.w(k).numpy(), w.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("itself", [0, 1])
def test_dense_golden(ops, golden, itself):
    """DLRM_Net forward/backward from the reference: R, Z, loss and every gradient."""
    g = golden("dense_itself%d" % itself)
    nb, nt = len(g["ln_bot"]) - 1, len(g["ln_top"]) - 1
    X, Tt = t(g["X"]).to(DEV), t(g["T"]).to(DEV)
    B = X.shape[0]
    ly = [t(g[f"ly_{k}"]) for k in range(5)]
    D = ly[0].shape[1]
    F = 6
    feat = torch.zeros(B, F, D, device=DEV)
    for k in range(5):
        feat[:, k + 1] = ly[k].to(DEV)

    def run_mlp(x, pre, n, last_sigmoid, out_last=None):
        acts, cur = [x], x
        for i in range(n):
            W, b = t(g[f"{pre}_w{i}"]).to(DEV), t(g[f"{pre}_b{i}"]).to(DEV)
            if out_last is not None and i == n - 1:
                y = out_last
            else:
                y = torch.empty(B, W.shape[0], device=DEV)
            ops.linear_fwd(cur, W, b, y, 2 if (last_sigmoid and i == n - 1) else 1)
            acts.append(y)
            cur = y
        return acts

    bot = run_mlp(X, "bot", nb, False, out_last=feat[:, 0, :])
    npairs = F * (F + 1) // 2 if itself else F * (F - 1) // 2
    R = torch.empty(B, D + npairs, device=DEV)
    ops.interact_fwd(feat, bool(itself), R)
    top = run_mlp(R, "top", nt, True)
    Z = top[-1]
    lossbuf = torch.zeros(65, device=DEV)
    dZ = torch.empty_like(Z)
    ops.bce_fwd_bwd(Z, Tt, lossbuf, dZ)
    torch.cuda.synchronize()
    np.testing.assert_allclose(R.cpu().numpy(), g["R"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(Z.cpu().numpy(), g["Z"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(lossbuf[0]), float(g["loss"]), rtol=1e-6)

    def back_mlp(acts, pre, n, last_sigmoid, dY, need_dx):
        for i in reversed(range(n)):
            W = t(g[f"{pre}_w{i}"]).to(DEV)
            Xi, Yi = acts[i], acts[i + 1]
            dW, db = torch.empty_like(W), torch.empty(W.shape[0], device=DEV)
            dX = torch.empty(B, W.shape[1], device=DEV) if (i > 0 or need_dx) else None
            work = ops.linear_bwd_work(B, W.shape[0], W.shape[1], DEV)
            ops.linear_bwd(Xi, W, Yi, dY, dX, dW, db, 2 if (last_sigmoid and i == n - 1) else 1, work)
            np.testing.assert_allclose(dW.cpu().numpy(), g[f"{pre}_gw{i}"], rtol=2e-4, atol=1e-7)
            np.testing.assert_allclose(db.cpu().numpy(), g[f"{pre}_gb{i}"], rtol=2e-4, atol=1e-7)
            dY = dX
        return dY

    dR = back_mlp(top, "top", nt, True, dZ, True)
    dfeat = torch.empty_like(feat)
    ops.interact_bwd(feat, dR, bool(itself), dfeat)
    torch.cuda.synchronize()
    for k in range(5):
        np.testing.assert_allclose(dfeat[:, k + 1].cpu().numpy(), g[f"ly_grad_{k}"], rtol=2e-4, atol=1e-8)
    back_mlp(bot, "bot", nb, False, dfeat[:, 0, :].contiguous(), False)


@pytest.mark.parametrize("M,N,K,act", [(8192, 512, 13, 1), (1030, 200, 5, 2), (33, 132, 32, 0), (1000, 256, 512, 1), (777, 1, 256, 2), (4096, 512, 479, 1),
                                       (130, 70, 33, 0), (1024, 512, 480, 1), (2048, 128, 256, 1), (8192, 1, 256, 2),
                                       (5000, 512, 512, 1), (3, 5, 2, 0), (1024, 512, 13, 1),
                                       # long batches on 16-byte-loadable layers: the LDS-DMA kernel (gemm_glds.h), 128x64 and
                                       # 64x64 tiles, partial tiles in both directions, K = 96 (3 K tiles), sigmoid epilogue
                                       (8192, 512, 512, 1), (8192, 256, 512, 1), (8192, 128, 256, 1), (6400, 512, 480, 2),
                                       (8256, 264, 96, 0), (8200, 264, 64, 1),
                                       # the wide kernel (gemm_wide.h: one workgroup per CU on 128x128 tiles, taken where they fill
                                       # the chip): forward N = 512 with K = 480 (15 K tiles, sigmoid) and its dgrad onto 480
                                       # columns (edge tiles in N, the generic epilogue); K = 256 / 64 (ring run-out: 8 and 2 K
                                       # tiles); a partial last row panel (16300 = 127 * 128 + 44: clamped source rows)
                                       (8192, 512, 480, 2), (8192, 512, 256, 0), (16384, 512, 64, 1), (16300, 512, 96, 1)])
def test_linear_vs_torch_fp32(ops, M, N, K, act):
    """FP32-MFMA Linear fwd/bwd against a plain torch fp32 reference (CPU, float64 accumulate for the bound).  Long batches run
    a second time with the caller's CDLRM_GEMM_ALONE hint, which routes eligible shapes to the wide kernel."""
    _linear_vs_torch_fp32(ops, M, N, K, act, False)
    if M >= 8192:
        _linear_vs_torch_fp32(ops, M, N, K, act, True)


def _linear_vs_torch_fp32(ops, M, N, K, act, alone):
    rng = np.random.RandomState(M + N + K)
    X = torch.from_numpy(rng.randn(M, K).astype(np.float32))
    W = torch.from_numpy((rng.randn(N, K) / np.sqrt(K)).astype(np.float32))
    b = torch.from_numpy(rng.randn(N).astype(np.float32))
    Xd, Wd, bd = X.to(DEV), W.to(DEV), b.to(DEV)
    Y = torch.empty(M, N, device=DEV)
    ops.linear_fwd(Xd, Wd, bd, Y, act, alone=alone)
    pre = X.double() @ W.double().t() + b.double()
    ref = {0: pre, 1: torch.relu(pre), 2: torch.sigmoid(pre)}[act]
    np.testing.assert_allclose(Y.cpu().numpy(), ref.float().numpy(), rtol=2e-5, atol=2e-5)
    dY = torch.from_numpy(rng.randn(M, N).astype(np.float32))
    dYd = dY.clone().to(DEV)
    dX, dW, db = torch.empty(M, K, device=DEV), torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    work = ops.linear_bwd_work(M, N, K, DEV)
    ops.linear_bwd(Xd, Wd, Y, dYd, dX, dW, db, act, work, alone=alone)
    Yc = Y.cpu().double()
    dZ = {0: dY.double(), 1: dY.double() * (Yc > 0), 2: dY.double() * Yc * (1 - Yc)}[act]
    scale = float(np.sqrt(M))
    np.testing.assert_allclose(dYd.cpu().numpy(), dZ.float().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dX.cpu().numpy(), (dZ @ W.double()).float().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dW.cpu().numpy(), (dZ.t() @ X.double()).float().numpy(), rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(db.cpu().numpy(), dZ.sum(0).float().numpy(), rtol=1e-4, atol=1e-4 * scale)


@pytest.mark.parametrize("M", [96, 1024, 8192])
def test_fused_activation_backward_chain(ops, M):
    """The training step's backward never runs a stand-alone activation-backward pass: loss kernel (sigmoid_bwd),
    dgrad epilogue (x_act) and interaction backward (x_act) each apply the derivative of the activation BELOW
    them.  Compare the chained result with torch autograd on the same two-layer head + interaction."""
    rng = np.random.RandomState(M)
    F, D, H = 5, 16, 40
    npairs = F * (F - 1) // 2
    feat = torch.from_numpy(rng.randn(M, F, D).astype(np.float32))
    feat[:, 0, :] = torch.relu(feat[:, 0, :])                  # feature 0 = a ReLU output
    W1 = torch.from_numpy((rng.randn(H, D + npairs) / 5).astype(np.float32))
    b1 = torch.from_numpy(rng.randn(H).astype(np.float32))
    W2 = torch.from_numpy((rng.randn(1, H) / 5).astype(np.float32))
    b2 = torch.from_numpy(rng.randn(1).astype(np.float32))
    T = torch.from_numpy((rng.rand(M, 1) > 0.5).astype(np.float32))
    # torch reference: pre-activation gradient of the layer that produced feature 0
    pre0 = feat[:, 0, :].clone().double().requires_grad_(True)     # stands for the pre-activation (positive part)
    f = torch.cat([torch.relu(pre0).unsqueeze(1), feat[:, 1:, :].double()], dim=1)
    f_rest = feat[:, 1:, :].clone().double().requires_grad_(True)
    f = torch.cat([torch.relu(pre0).unsqueeze(1), f_rest], dim=1)
    Zm = torch.bmm(f, f.transpose(1, 2))
    li = [i for i in range(F) for j in range(i)]
    lj = [j for i in range(F) for j in range(i)]
    R = torch.cat([f[:, 0, :], Zm[:, li, lj]], dim=1)
    W1d, b1d, W2d, b2d = [x.double().requires_grad_(True) for x in (W1, b1, W2, b2)]
    h = torch.relu(R @ W1d.t() + b1d)
    z = torch.sigmoid(h @ W2d.t() + b2d)
    loss = torch.nn.functional.binary_cross_entropy(z, T.double())
    loss.backward()
    # HIP chain
    fd = feat.to(DEV)
    Rd = torch.empty(M, D + npairs, device=DEV)
    ops.interact_fwd(fd, False, Rd)
    hd = torch.empty(M, H, device=DEV)
    zd = torch.empty(M, 1, device=DEV)
    ops.linear_fwd(Rd, W1.to(DEV), b1.to(DEV), hd, 1)
    ops.linear_fwd(hd, W2.to(DEV), b2.to(DEV), zd, 2)
    lossb = torch.zeros(65, device=DEV)
    dz = torch.empty(M, 1, device=DEV)
    ops.bce_fwd_bwd(zd, T.to(DEV), lossb, dz, sigmoid_bwd=True)
    dh = torch.empty(M, H, device=DEV)
    gW2, gb2 = torch.empty(1, H, device=DEV), torch.empty(1, device=DEV)
    ops.linear_bwd(hd, W2.to(DEV), None, dz, dh, gW2, gb2, 0, ops.linear_bwd_work(M, 1, H, DEV), x_act=1)
    dR = torch.empty(M, D + npairs, device=DEV)
    gW1, gb1 = torch.empty(H, D + npairs, device=DEV), torch.empty(H, device=DEV)
    ops.linear_bwd(Rd, W1.to(DEV), None, dh, dR, gW1, gb1, 0, ops.linear_bwd_work(M, H, D + npairs, DEV), x_act=0)
    dfeat = torch.empty_like(fd)
    ops.interact_bwd(fd, dR, False, dfeat, x_act=1)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(lossb[0]), float(loss), rtol=1e-5)
    tol = dict(rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(gW2.cpu().numpy(), W2d.grad.float().numpy(), **tol)
    np.testing.assert_allclose(gb2.cpu().numpy(), b2d.grad.float().numpy(), **tol)
    np.testing.assert_allclose(gW1.cpu().numpy(), W1d.grad.float().numpy(), **tol)
    np.testing.assert_allclose(gb1.cpu().numpy(), b1d.grad.float().numpy(), **tol)
    np.testing.assert_allclose(dfeat[:, 1:].cpu().numpy(), f_rest.grad.float().numpy(), **tol)
    np.testing.assert_allclose(dfeat[:, 0].cpu().numpy(), pre0.grad.float().numpy(), **tol)


@pytest.mark.parametrize("B,F,D,itself,pad", [(3000, 27, 128, 0, 1), (2049, 9, 32, 1, 3), (700, 32, 64, 0, 0),
                                               (5000, 27, 128, 1, 2), (33, 27, 256, 0, 1), (257, 4, 16, 0, 0),
                                               (1, 27, 128, 0, 1),
                                               # slab kernels with several samples per wave at the other widths; a row pitch
                                               # that is no multiple of 4 (whole-row forward, generic backward)
                                               (2100, 27, 256, 1, 3), (4100, 20, 32, 0, 2), (300, 27, 128, 0, 0)])
def test_interaction_kernels_vs_torch(ops, B, F, D, itself, pad):
    """Pairwise-dot interaction forward/backward (software-pipelined kernels for D = 32/64/128/256, generic otherwise)
    against torch autograd on the oracle's interact_features; R/dR with a padded row pitch as the engine uses."""
    rng = np.random.RandomState(B + F + D)
    feat = torch.from_numpy(rng.randn(B, F, D).astype(np.float32))
    npairs = F * (F + 1) // 2 if itself else F * (F - 1) // 2
    width = D + npairs + pad
    f = feat.clone().double().requires_grad_(True)
    ref = O.interact_features(f[:, 0, :], [f[:, k, :] for k in range(1, F)], "dot", bool(itself))
    G = torch.from_numpy(rng.randn(B, width).astype(np.float32))
    ref.backward(G[:, :D + npairs].double())
    fd = feat.to(DEV)
    R = torch.full((B, width), 7.0, device=DEV)
    ops.interact_fwd(fd, bool(itself), R)
    np.testing.assert_allclose(R[:, :D + npairs].cpu().numpy(), ref.detach().float().numpy(), rtol=2e-5, atol=2e-5)
    padc = R[:, D + npairs:]                                # pad columns: untouched, or zero (whole-float4 output rows)
    assert bool(((padc == 7.0) | (padc == 0.0)).all())
    dfeat = torch.empty_like(fd)
    ops.interact_bwd(fd, G.to(DEV), bool(itself), dfeat)
    np.testing.assert_allclose(dfeat.cpu().numpy(), f.grad.float().numpy(), rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("B,T,D,itself,x_act,n_extra", [(3000, 26, 128, 0, 1, 0), (1, 26, 128, 0, 1, 0), (5, 16, 32, 1, 0, 3),
                                                         (1023, 31, 64, 0, 2, 0), (2100, 26, 256, 1, 1, 0),
                                                         (4100, 19, 32, 0, 0, 28), (8192, 26, 128, 0, 1, 0),
                                                         (20000, 26, 128, 0, 1, 0)])
def test_fused_gather_interaction_equals_the_two_operators(ops, B, T, D, itself, x_act, n_extra):
    """cdlrm_gather_interact_fwd / _bwd (cached EmbeddingBag forward of one-index bags + dot interaction in one launch, the
    backward reading the rows again from the cache) against cdlrm_embbag_fwd + cdlrm_interact_fwd / cdlrm_interact_bwd on the
    same slot ids: bit for bit, and against torch autograd on the oracle's interact_features over the gathered rows.  Ragged
    batch sizes (one sample, fewer samples than waves, several samples per wave), every row width of the slab kernels, a slot
    pitch larger than the batch, auxiliary rows behind the cache rows as slot targets; the feature block handed to the fused
    kernels holds NaN outside feature 0 -- it must never be read."""
    rng = np.random.RandomState(B + T + D)
    ln = [int(v) for v in rng.randint(50, 90000, size=T)]
    P, ways, aux = 61, 4, 512
    cs = [min(P, v) for v in ln]
    ctx = ops.CacheCtx(ln, cs, D, ways, aux, torch.device(DEV))
    assert ops.gather_interact_supported(ctx)
    tags = torch.full((ctx.total_tags,), -1, dtype=torch.int64, device=DEV)
    weight = torch.from_numpy(rng.randn(ctx.total_rows, D).astype(np.float32)).to(DEV)
    ctx.bind_cache(tags, weight)
    n = B + n_extra
    rows_of = [cs[k] * ways + aux for k in range(T)]            # cache rows + the auxiliary rows behind them
    slots = torch.stack([torch.from_numpy(rng.randint(0, rows_of[k], size=n).astype(np.int32)) for k in range(T)]).to(DEV)
    F = T + 1
    npairs = F * (F + 1) // 2 if itself else F * (F - 1) // 2
    ld = (D + npairs + 3) // 4 * 4 + 4
    feat = torch.zeros(B, F, D, device=DEV)
    x = torch.from_numpy(rng.rand(B, D).astype(np.float32)).to(DEV)     # (0, 1): a valid ReLU / sigmoid output
    feat[:, 0, :] = x
    poisoned = torch.full((B, F, D), float("nan"), device=DEV)
    poisoned[:, 0, :] = x
    ops.embbag_fwd(ctx, slots[:, :B].contiguous(), None, feat[:, 1:, :], F * D, D)
    R, R2 = torch.full((B, ld), 7.0, device=DEV), torch.full((B, ld), 7.0, device=DEV)
    ops.interact_fwd(feat, bool(itself), R)
    ops.gather_interact_fwd(ctx, slots, poisoned[:, 0, :], bool(itself), R2)
    assert torch.equal(R, R2)
    dR = torch.from_numpy(rng.randn(B, ld).astype(np.float32)).to(DEV)
    dfeat, dfeat2 = torch.empty_like(feat), torch.empty_like(feat)
    ops.interact_bwd(feat, dR, bool(itself), dfeat, x_act=x_act)
    ops.gather_interact_bwd(ctx, slots, poisoned[:, 0, :], dR, bool(itself), dfeat2, x_act=x_act)
    assert torch.equal(dfeat, dfeat2)
    if B <= 4100:
        rb = torch.tensor(ctx.row_base[:T], device=DEV).view(T, 1)
        rows = weight[(slots[:, :B].to(torch.int64) + rb).reshape(-1)].view(T, B, D)
        f = torch.cat([x.unsqueeze(1), rows.permute(1, 0, 2)], dim=1).cpu().double().requires_grad_(True)
        ref = O.interact_features(f[:, 0, :], [f[:, k, :] for k in range(1, F)], "dot", bool(itself))
        np.testing.assert_allclose(R2[:, :D + npairs].cpu().numpy(), ref.detach().float().numpy(), rtol=2e-5, atol=2e-5)
        if x_act == 0:
            ref.backward(dR[:, :D + npairs].cpu().double())
            np.testing.assert_allclose(dfeat2.cpu().numpy(), f.grad.float().numpy(), rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("B,T,D,itself,x_act,span", [(3000, 26, 128, 0, 1, 20000), (1, 26, 128, 0, 1, 50), (5, 16, 32, 1, 0, 4),
                                                     (1023, 31, 64, 0, 2, 700), (2100, 26, 256, 1, 1, 90000),
                                                     (8192, 26, 128, 0, 1, 6000), (20000, 26, 128, 0, 1, 3)])
def test_once_only_slots_updated_by_the_interaction_backward(ops, B, T, D, itself, x_act, span):
    """cdlrm_gather_interact_bwd_sgd + cdlrm_embbag_bwd_apply_rest (the SGD step of the slots a batch reads once folded into
    the interaction backward, the sorted path left with the repeated slots) against cdlrm_gather_interact_bwd +
    cdlrm_embbag_bwd_apply on the same prepared work buffer: the cache rows, the touched flags and the dense feature's gradient
    bit for bit; the gradient rows of the repeated slots bit for bit, those of the once-only slots never written (the poison
    stays; the last table's are undefined).  Slot ranges from "almost every slot once" to "three slots per table"; aux rows are slot targets (never flagged)."""
    rng = np.random.RandomState(B + T + D + span)
    ln = [int(v) for v in rng.randint(50, 90000, size=T)]
    P, ways, aux = 6007, 4, 512
    cs = [min(P, v) for v in ln]
    dev = torch.device(DEV)
    ctx = ops.CacheCtx(ln, cs, D, ways, aux, dev)
    tags = torch.full((ctx.total_tags,), -1, dtype=torch.int64, device=DEV)
    w0 = torch.from_numpy(rng.randn(ctx.total_rows, D).astype(np.float32)).to(DEV)
    weight = w0.clone()
    ctx.bind_cache(tags, weight)
    rows_of = [cs[k] * ways + aux for k in range(T)]
    slots = torch.stack([torch.from_numpy((rows_of[k] - 1 - rng.randint(0, min(span, rows_of[k]), size=B)).astype(np.int32))
                         for k in range(T)]).to(DEV)
    F = T + 1
    npairs = F * (F + 1) // 2 if itself else F * (F - 1) // 2
    ld = (D + npairs + 3) // 4 * 4
    x = torch.from_numpy(rng.rand(B, D).astype(np.float32)).to(DEV)
    dR = torch.from_numpy(rng.randn(B, ld).astype(np.float32)).to(DEV)
    work = ops.embbag_bwd_work(ctx, B, dev)
    lr = 0.37
    # the two operators
    dfeat = torch.empty(B, F, D, device=DEV)
    touched = torch.zeros(ctx.total_rows, dtype=torch.uint8, device=DEV)
    ops.embbag_bwd_prepare(ctx, slots, work)
    ops.gather_interact_bwd(ctx, slots, x, dR, bool(itself), dfeat, x_act=x_act)
    ops.embbag_bwd_apply(ctx, B, None, dfeat[:, 1:, :], F * D, D, lr, work, touched)
    torch.cuda.synchronize()
    w_ref = weight.clone()
    # folded
    weight.copy_(w0)
    POISON = 12345.0
    dfeat2 = torch.full((B, F, D), POISON, device=DEV)
    touched2 = torch.zeros_like(touched)
    ops.embbag_bwd_prepare(ctx, slots, work)
    ops.gather_interact_bwd_sgd(ctx, slots, x, dR, bool(itself), dfeat2, ops.embbag_bwd_once_flags(ctx, work, B), B, lr,
                                x_act=x_act)
    ops.embbag_bwd_apply_rest(ctx, B, None, dfeat2[:, 1:, :], F * D, D, lr, work, touched2)
    torch.cuda.synchronize()
    assert torch.equal(weight, w_ref)
    assert torch.equal(touched, touched2)
    assert torch.equal(dfeat[:, 0, :], dfeat2[:, 0, :])
    sl = slots.cpu().numpy()
    once = np.zeros((T, B), dtype=bool)
    for k in range(T):
        _, inv, cnt = np.unique(sl[k], return_inverse=True, return_counts=True)
        once[k] = cnt[inv] == 1
    once_t = torch.from_numpy(once.T.copy()).to(DEV)                    # [B, T]
    g1, g2 = dfeat[:, 1:, :], dfeat2[:, 1:, :]
    assert torch.equal(g1[~once_t], g2[~once_t])
    # ... of tables 0 .. T-2: the store's spare lanes repeat the LAST row's last word (an unread duplicate), whatever its flag
    assert bool((g2[:, :T - 1][once_t[:, :T - 1]] == POISON).all())
    assert once.any() or span <= 4
    # an aux row (transient copy of a host row) is updated but never flagged
    rb = torch.tensor(ctx.row_base[:T], device=DEV).view(T, 1)
    aux_first = torch.tensor([cs[k] * ways for k in range(T)], device=DEV).view(T, 1)
    is_aux = (slots.to(torch.int64) >= aux_first)
    assert not bool(touched2[(slots.to(torch.int64) + rb)[is_aux]].any())


@pytest.mark.parametrize("T,D,n,nb,batch_len,col0,span", [(26, 128, 2048, 5, 2048, 0, 3000), (26, 128, 1000, 3, 2500, 700, 40),
                                                          (19, 32, 8192, 2, 8192, 0, 100000), (26, 64, 3000, 4, 3000, 0, 5),
                                                          (26, 128, 20000, 2, 20000, 0, 9000)])
def test_window_sorted_chunk_equals_per_batch_sort(ops, T, D, n, nb, batch_len, col0, span):
    """cdlrm_embbag_bwd_prepare_window (nb batches x T tables sorted by one set of launches, out of the resolver's [T, nb *
    batch_len] phase-0 slot ids, a rank's slice starting at col0) + cdlrm_embbag_bwd_apply_sorted against the per-batch
    cdlrm_embbag_bwd_prepare + _apply on the slot ids cdlrm_embbag_take would hand that batch (aux slots moved to the batch's aux
    region): cache rows and touched flags bit for bit, with the once-only slots in the sorted path (rest = 0) and folded into the
    interaction backward (rest = 1); the chunk's once-only flags equal the per-batch sort's."""
    rng = np.random.RandomState(T + D + n + nb + span)
    ln = [int(v) for v in rng.randint(50, 90000, size=T)]
    P, ways, aux = 6007, 4, 1024
    cs = [min(P, v) for v in ln]
    dev = torch.device(DEV)
    ctx = ops.CacheCtx(ln, cs, D, ways, aux, dev, aux_phases=2)
    tags = torch.full((ctx.total_tags,), -1, dtype=torch.int64, device=DEV)
    w0 = torch.from_numpy(rng.randn(ctx.total_rows, D).astype(np.float32)).to(DEV)
    weight = w0.clone()
    ctx.bind_cache(tags, weight)
    width = nb * batch_len
    first_aux = [cs[k] * ways for k in range(T)]
    # phase-0 slots: cache rows [first_aux - span, first_aux) and aux rows [first_aux, first_aux + aux)
    wsl = torch.stack([torch.from_numpy((first_aux[k] + rng.randint(-min(span, first_aux[k]), min(aux, max(span // 8, 2)), size=width))
                                        .astype(np.int32)) for k in range(T)]).to(DEV)
    sorted_buf = ops.embbag_bwd_sorted(ctx, nb, n, dev)
    if nb > 2:          # in slices of two batches (the trainer spreads a chunk's sort over several steps), last slice first
        for j0 in reversed(range(0, nb, 2)):
            ops.embbag_bwd_prepare_window(ctx, wsl[:, col0:], batch_len, nb, n, sorted_buf, j0=j0, count=min(2, nb - j0))
    else:
        ops.embbag_bwd_prepare_window(ctx, wsl[:, col0:], batch_len, nb, n, sorted_buf)
    F = T + 1
    npairs = F * (F - 1) // 2
    ld = (D + npairs + 3) // 4 * 4
    lr = 0.21
    work = ops.embbag_bwd_work(ctx, n, dev)
    work2 = ops.embbag_bwd_work(ctx, n, dev)
    fa = torch.tensor(first_aux, dtype=torch.int32, device=DEV).view(T, 1)
    fused = ops.gather_interact_supported(ctx)
    for j in range(nb):
        phase = j & 1
        sl = wsl[:, j * batch_len + col0: j * batch_len + col0 + n].clone()
        sl = torch.where(sl >= fa, sl + phase * aux, sl).contiguous()
        x = torch.from_numpy(rng.rand(n, D).astype(np.float32)).to(DEV)
        dR = torch.from_numpy(rng.randn(n, ld).astype(np.float32)).to(DEV)
        dfeat = torch.empty(n, F, D, device=DEV)
        touched = torch.zeros(ctx.total_rows, dtype=torch.uint8, device=DEV)
        weight.copy_(w0)
        ops.embbag_bwd_prepare(ctx, sl, work)
        if fused:
            ops.gather_interact_bwd(ctx, sl, x, dR, False, dfeat, x_act=1)
        else:
            dfeat.copy_(torch.from_numpy(rng.randn(n, F, D).astype(np.float32)))
        ops.embbag_bwd_apply(ctx, n, None, dfeat[:, 1:, :], F * D, D, lr, work, touched)
        torch.cuda.synchronize()
        w_ref = weight.clone()
        keys, meta, once = ops.embbag_bwd_sorted_views(ctx, sorted_buf, nb, n, j)
        # the flags: [T, n] at pitch nb * n inside the chunk against the per-batch sort's
        o0 = once - sorted_buf.data_ptr()
        got = torch.stack([sorted_buf[o0 + k * nb * n: o0 + k * nb * n + n] for k in range(T)])
        r0 = ops.embbag_bwd_once_flags(ctx, work, n) - work.data_ptr()
        assert torch.equal(got, work[r0:r0 + T * n].view(T, n))
        # rest = 0: every run in the sorted path
        weight.copy_(w0)
        t2 = torch.zeros_like(touched)
        ops.embbag_bwd_apply_sorted(ctx, n, dfeat[:, 1:, :], F * D, D, lr, work2, keys, meta, nb * n, phase, False, t2)
        torch.cuda.synchronize()
        assert torch.equal(weight, w_ref) and torch.equal(touched, t2)
        if fused:
            weight.copy_(w0)
            t3 = torch.zeros_like(touched)
            dfeat3 = torch.full((n, F, D), 777.0, device=DEV)
            ops.gather_interact_bwd_sgd(ctx, sl, x, dR, False, dfeat3, once, nb * n, lr, x_act=1)
            ops.embbag_bwd_apply_sorted(ctx, n, dfeat3[:, 1:, :], F * D, D, lr, work2, keys, meta, nb * n, phase, True, t3)
            torch.cuda.synchronize()
            assert torch.equal(weight, w_ref) and torch.equal(touched, t3)
            assert torch.equal(dfeat[:, 0, :], dfeat3[:, 0, :])


def test_fused_gather_interaction_refuses_other_shapes(ops):
    """Shapes outside the slab kernels (F <= 16, an embedding width that is none of 32 / 64 / 128 / 256) are refused with a
    message -- the caller issues the two operators -- and a timed launch leaves its stamps in the armed events."""
    for T, D in ((8, 128), (26, 48), (32, 128)):
        ctx = ops.CacheCtx([1000] * T, [16] * T, D, 4, 64, torch.device(DEV))
        assert not ops.gather_interact_supported(ctx)
        w = torch.zeros(ctx.total_rows, D, device=DEV)
        ctx.bind_cache(torch.full((ctx.total_tags,), -1, dtype=torch.int64, device=DEV), w)
        slots = torch.zeros(T, 64, dtype=torch.int32, device=DEV)
        x = torch.zeros(64, D, device=DEV)
        R = torch.zeros(64, 2048, device=DEV)
        with pytest.raises(RuntimeError, match="unsupported shape"):
            ops.gather_interact_fwd(ctx, slots, x, False, R)
        with pytest.raises(RuntimeError, match="unsupported shape"):
            ops.gather_interact_bwd(ctx, slots, x, R, False, torch.zeros(64, T + 1, D, device=DEV))
    T, D, n = 26, 128, 4096
    ctx = ops.CacheCtx([5000] * T, [64] * T, D, 4, n, torch.device(DEV))
    weight = torch.randn(ctx.total_rows, D, device=DEV)
    ctx.bind_cache(torch.full((ctx.total_tags,), -1, dtype=torch.int64, device=DEV), weight)
    slots = torch.randint(0, 256, (T, n), dtype=torch.int32, device=DEV)
    x = torch.rand(n, D, device=DEV)
    R0, R1 = torch.zeros(n, 480, device=DEV), torch.zeros(n, 480, device=DEV)
    ops.gather_interact_fwd(ctx, slots, x, False, R0)
    e0, e1 = ops.TimingEvent(), ops.TimingEvent()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ops.time_next_gather(ctx, e0, e1)
    t0.record()
    ops.gather_interact_fwd(ctx, slots, x, False, R1)
    t1.record()
    torch.cuda.synchronize()
    us, around = e0.elapsed_us(e1), t0.elapsed_time(t1) * 1e3
    assert 0.5 < us <= around + 1.0, (us, around)
    assert torch.equal(R0, R1)


@pytest.mark.parametrize("M", [64, 1000, 2048, 4100, 4096, 8192, 16384, 20011])
def test_mlp_wgrad_group(ops, M):
    """All layers' weight + bias gradients in one call (grouped LDS-free launch up to M = 2048, tiled split-M path
    above, up to M = 20011: a ragged last slab) against fp64 torch."""
    rng = np.random.RandomState(M)
    shapes = [(512, 13), (256, 512), (128, 256), (512, 480), (512, 512), (256, 512), (1, 256), (70, 33), (5, 3)]
    Xs, dZs, dWs, dbs = [], [], [], []
    for n, (N, K) in enumerate(shapes):
        ld = K + (4 if n == 3 else 0)                   # one input with a row pitch wider than K
        xb = torch.from_numpy(rng.randn(M, ld).astype(np.float32)).to(DEV)
        Xs.append(xb[:, :K])
        dZs.append(torch.from_numpy(rng.randn(M, N).astype(np.float32)).to(DEV))
        dWs.append(torch.empty(N, K, device=DEV))
        dbs.append(None if n == 2 else torch.empty(N, device=DEV))
    plan = ops.WgradPlan(Xs, dZs, dWs, dbs, ops.mlp_wgrad_work(M, [s[0] for s in shapes], [s[1] for s in shapes], DEV))
    ops.mlp_wgrad(plan)
    ops.mlp_wgrad(plan)          # idempotent: no accumulation into the outputs
    torch.cuda.synchronize()
    scale = float(np.sqrt(M))
    for X, dZ, dW, db in zip(Xs, dZs, dWs, dbs):
        ref = dZ.cpu().double().t() @ X.cpu().double()
        np.testing.assert_allclose(dW.cpu().numpy(), ref.float().numpy(), rtol=1e-4, atol=1e-5 * scale)
        if db is not None:
            np.testing.assert_allclose(db.cpu().numpy(), dZ.cpu().double().sum(0).float().numpy(), rtol=1e-4,
                                       atol=1e-5 * scale)


def test_wide_gemm_weight_gradient_layout(ops):
    """The wide kernel's third layout (gemm_wide.h: A = dZ^T and B = X both contraction-strided, split over the batch, bias
    gradient fused): not taken by the training step (beside the other queues' GEMMs it loses, profiles/r06_ab_gemm3_in_step.txt),
    reachable through the development selector cdlrm_debug_set(6, 512) -- same answers as the default path, against fp64."""
    from cdlrm_amd import _lib
    M = 8192
    rng = np.random.RandomState(7)
    shapes = [(512, 512), (512, 480), (256, 512)]
    Xs, dZs, dWs, dbs = [], [], [], []
    for N, K in shapes:
        Xs.append(torch.from_numpy(rng.randn(M, K).astype(np.float32)).to(DEV))
        dZs.append(torch.from_numpy(rng.randn(M, N).astype(np.float32)).to(DEV))
        dWs.append(torch.empty(N, K, device=DEV))
        dbs.append(torch.empty(N, device=DEV))
    plan = ops.WgradPlan(Xs, dZs, dWs, dbs, ops.mlp_wgrad_work(M, [s[0] for s in shapes], [s[1] for s in shapes], DEV))
    torch.cuda.synchronize()
    assert _lib.raw().cdlrm_debug_set(6, 512) == 0
    try:
        ops.mlp_wgrad(plan)
        torch.cuda.synchronize()
    finally:
        assert _lib.raw().cdlrm_debug_set(6, 0) == 0
    scale = float(np.sqrt(M))
    for X, dZ, dW, db in zip(Xs, dZs, dWs, dbs):
        ref = dZ.cpu().double().t() @ X.cpu().double()
        np.testing.assert_allclose(dW.cpu().numpy(), ref.float().numpy(), rtol=1e-4, atol=1e-5 * scale)
        np.testing.assert_allclose(db.cpu().numpy(), dZ.cpu().double().sum(0).float().numpy(), rtol=1e-4, atol=1e-5 * scale)
    first = [w.clone() for w in dWs]
    ops.mlp_wgrad(plan)                 # the default kernel on the same inputs: equal to rounding
    torch.cuda.synchronize()
    for a, b in zip(first, dWs):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)


@pytest.mark.parametrize("M", [64, 1000, 2048, 4100, 8192, 16500])
def test_mlp_wgrad_with_fused_sgd_step(ops, M):
    """cdlrm_mlp_wgrad_sgd: the dense SGD step inside the weight-gradient launches (slab reduction; elementwise pass for
    layers without slabs) = cdlrm_mlp_wgrad followed by p -= lr * g, bit for bit, gradients still left in dW / db."""
    rng = np.random.RandomState(M + 1)
    shapes = [(512, 13), (256, 512), (128, 256), (512, 480), (1, 256), (70, 33)]
    lr = 0.37
    Xs, dZs, dWs, dbs, Ws, bs = [], [], [], [], [], []
    for n, (N, K) in enumerate(shapes):
        Xs.append(torch.from_numpy(rng.randn(M, K).astype(np.float32)).to(DEV))
        dZs.append(torch.from_numpy(rng.randn(M, N).astype(np.float32)).to(DEV))
        dWs.append(torch.empty(N, K, device=DEV))
        dbs.append(None if n == 2 else torch.empty(N, device=DEV))
        Ws.append(torch.from_numpy(rng.randn(N, K).astype(np.float32)).to(DEV))
        bs.append(None if n == 2 else torch.from_numpy(rng.randn(N).astype(np.float32)).to(DEV))
    W0, b0 = [w.clone() for w in Ws], [None if b is None else b.clone() for b in bs]
    work = ops.mlp_wgrad_work(M, [s[0] for s in shapes], [s[1] for s in shapes], DEV)
    plan = ops.WgradPlan(Xs, dZs, dWs, dbs, work)
    plan.set_params(Ws, bs)
    ops.mlp_wgrad(plan, lr=lr)
    torch.cuda.synchronize()
    gW, gb = [w.clone() for w in dWs], [None if b is None else b.clone() for b in dbs]
    ops.mlp_wgrad(plan)                                  # gradients alone
    for i in range(len(shapes)):
        assert torch.equal(dWs[i], gW[i])
        ops.sgd_step(W0[i].view(-1), dWs[i].view(-1), lr)
        assert torch.equal(W0[i], Ws[i]), i
        if dbs[i] is not None:
            assert torch.equal(dbs[i], gb[i])
            ops.sgd_step(b0[i], dbs[i], lr)
            assert torch.equal(b0[i], bs[i]), i


def test_qr_embedding_bag_golden(ops, golden):
    """QREmbeddingBag forward + gradients vs the reference's module, incl. the float32-division quirk."""
    from cdlrm_amd.tricks.qr_embedding_bag import QREmbeddingBag
    from cdlrm_amd import _lib
    g = golden("qr")
    c = int(g["c"])
    for op in ("mult", "add", "concat"):
        wq, wr = t(g[f"{op}_wq"]).to(DEV), t(g[f"{op}_wr"]).to(DEV)
        E = QREmbeddingBag(103, wq.shape[1], c, operation=op, mode="sum", sparse=True, _weight=[wq.clone(), wr.clone()])
        V = E(t(g[f"{op}_idx"]).to(DEV), t(g[f"{op}_offs"]).to(DEV))
        np.testing.assert_allclose(V.detach().cpu().numpy(), g[f"{op}_V"], rtol=1e-6, atol=1e-7)
        V.backward(t(g[f"{op}_G"]).to(DEV))
        np.testing.assert_allclose(E.weight_q.grad.cpu().numpy(), g[f"{op}_gq"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(E.weight_r.grad.cpu().numpy(), g[f"{op}_gr"], rtol=1e-5, atol=1e-6)
    # quotient arithmetic above 2**24: rows addressed = (idx / c).long() in float32
    big = t(g["big_idx"]).to(DEV)
    rows_q = int(g["big_q"].max()) + 1
    Wq = torch.arange(rows_q, dtype=torch.float32, device=DEV).view(-1, 1).repeat(1, 4).contiguous()
    Wr = torch.zeros(c, 4, device=DEV)
    out = torch.empty(big.numel(), 4, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    off = torch.arange(big.numel(), device=DEV)
    _lib.check(_lib.lib().cdlrm_qr_embbag_fwd(big.data_ptr(), off.data_ptr(), big.numel(), big.numel(), Wq.data_ptr(),
                                              Wr.data_ptr(), rows_q, c, 4, 1, out.data_ptr(), None, None, err.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
    assert int(err) == 0
    got = out[:, 0].cpu().double().numpy()
    # float32 row ids are exact up to 2**24; compare where representable, and the quirk case explicitly
    want = g["big_q"].astype(np.float64)
    assert np.array_equal(np.float32(got), np.float32(want))
    assert int(g["big_q"][4]) == 10_000_000


def test_qr_embedding_bag_c4_table_size(ops, golden):
    """The QR operator at BASELINE configs[3]'s table size: 39 884 406 categories, 4 collisions (a 9 971 102-row, 10 GB
    quotient table), embed-dim 256, lookups above 2**24 -- against the reference's module run at that size
    (tools/make_golden.py:g_qr_c4; the table contents are the generator's formula, restated here on the device)."""
    from cdlrm_amd.tricks.qr_embedding_bag import QREmbeddingBag
    g = golden("qr_c4")
    n, c, D = int(g["n"]), int(g["c"]), int(g["D"])
    rows_q = -(-n // c)

    def table(rows, salt):
        out = torch.empty(rows, D, dtype=torch.float32, device=DEV)
        d = torch.arange(D, dtype=torch.int64, device=DEV).view(1, -1)
        for r0 in range(0, rows, 1 << 20):
            i = torch.arange(r0, min(rows, r0 + (1 << 20)), dtype=torch.int64, device=DEV).view(-1, 1)
            out[r0:r0 + i.shape[0]] = ((i * 37 + d * 11 + salt) & 1023).to(torch.float32) / 1024.0 - 0.5
        return out

    wq, wr = table(rows_q, 5), table(c, 901)
    idx, offs, G = t(g["idx"]).to(DEV), t(g["offs"]).to(DEV), t(g["G"]).to(DEV)
    rows = t(g["mult_gq_rows"]).to(DEV)
    assert torch.equal(torch.unique(t(g["q"])).to(DEV), rows)
    for op in ("mult", "add"):
        E = QREmbeddingBag(n, D, c, operation=op, mode="sum", sparse=True, _weight=[wq, wr])
        V = E(idx, offs)
        np.testing.assert_allclose(V.detach().cpu().numpy(), g[f"{op}_V"], rtol=1e-6, atol=1e-6)
        V.backward(G)
        gq = E.weight_q.grad
        # the rows the float32 quotients address (39884403 / 4 -> 9971101, not 9971100), and nothing else
        touched = torch.zeros(rows_q, dtype=torch.bool, device=DEV)
        touched[rows] = True
        assert float(gq[~touched].abs().max()) == 0.0
        if op == "mult":
            np.testing.assert_allclose(gq[rows].cpu().numpy(), g["mult_gq_vals"], rtol=1e-5, atol=1e-6)
        else:
            np.testing.assert_allclose(gq[rows].double().sum(dim=1).cpu().numpy(), g["add_gq_rowsum"], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(E.weight_r.grad.cpu().numpy(), g[f"{op}_gr"], rtol=1e-5, atol=1e-4)
        E.weight_q.grad = E.weight_r.grad = None
        del E, gq, V
    del wq, wr
    torch.cuda.empty_cache()


def test_md_embedding_bag_golden(ops, golden):
    """PrEmbeddingBag (mixed-dimension trick) forward + gradients vs the reference's module: widths 1, 2, 4, 8, with and
    without the projection, empty bags; md_solver's widths vs the reference's."""
    from cdlrm_amd.tricks.md_embedding_bag import PrEmbeddingBag, md_solver
    g = golden("md")
    for name in ("criteo", "b_budget", "noround", "alpha0"):
        d0, B = int(g[f"solver_{name}_d0"]), float(g[f"solver_{name}_B"])
        d = md_solver(t(g[f"solver_{name}_n"]), float(g[f"solver_{name}_alpha"]), d0=None if d0 < 0 else d0,
                      B=None if B < 0 else B, round_dim=bool(g[f"solver_{name}_round"]))
        assert np.array_equal(d.double().numpy(), g[f"solver_{name}_d"].astype(np.float64)), name
    d = md_solver(torch.tensor([100, 5000, 70, 900000]), 0.25, d0=16, k=t(g["solver_k_k"]))
    assert np.array_equal(d.double().numpy(), g["solver_k_d"].astype(np.float64))
    for name in ("proj", "ident", "w1", "w2"):
        W = t(g[f"{name}_W"])
        E = PrEmbeddingBag(W.shape[0], W.shape[1], int(g[f"{name}_base"])).to(DEV)
        with torch.no_grad():
            E.embs.weight.copy_(W)
            if f"{name}_P" in g.files:
                E.proj.weight.copy_(t(g[f"{name}_P"]))
        V = E(t(g[f"{name}_idx"]).to(DEV), t(g[f"{name}_offs"]).to(DEV))
        np.testing.assert_allclose(V.detach().cpu().numpy(), g[f"{name}_V"], rtol=1e-5, atol=1e-6)
        V.backward(t(g[f"{name}_G"]).to(DEV))
        np.testing.assert_allclose(E.embs.weight.grad.cpu().numpy(), g[f"{name}_gW"], rtol=1e-5, atol=1e-6)
        if f"{name}_P" in g.files:
            np.testing.assert_allclose(E.proj.weight.grad.cpu().numpy(), g[f"{name}_gP"], rtol=1e-5, atol=1e-5)
    with pytest.raises(IndexError):
        E(torch.tensor([W.shape[0]], device=DEV), torch.tensor([0], device=DEV))
    with pytest.raises(ValueError):
        PrEmbeddingBag(10, 16, 8)


def test_sgd_and_agg(ops):
    p = torch.randn(100003, device=DEV)
    gr = torch.randn(100003, device=DEV)
    want = (p.cpu() + (-0.37) * gr.cpu())
    ops.sgd_step(p, gr, 0.37)
    np.testing.assert_allclose(p.cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-7)
    st = DevState(ops, [500, 40], [101, 40], 8, 4, 16, [torch.full((101, 4), -1, dtype=torch.int64),
                                                       torch.full((40, 4), -1, dtype=torch.int64)],
                  [torch.randn(4 * 101 + 16, 8), torch.randn(4 * 40 + 16, 8)], [torch.zeros(500, 8), torch.zeros(40, 8)])
    total = st.ctx.total_rows
    touched = torch.zeros(total, dtype=torch.uint8, device=DEV)
    rows = torch.randperm(total)[:57].sort().values
    touched[rows.to(DEV)] = 1
    out = torch.empty(total, dtype=torch.int64, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.agg_compact(st.ctx, touched, out, cnt)
    assert int(cnt) == 57 and torch.equal(out[:57].cpu(), rows) and int(touched.sum()) == 0
    buf = torch.empty(total, 8, device=DEV)
    before = st.weight.clone()
    ops.agg_gather(st.ctx, out, cnt, 2.0, buf, total)
    assert torch.equal(buf[:57].cpu(), before[rows.to(DEV)].cpu() / 2.0)
    ops.agg_scatter(st.ctx, out, cnt, buf, total)
    want = before.clone()
    want[rows.to(DEV)] = before[rows.to(DEV)] / 2.0
    assert torch.equal(st.weight, want)


DENSE_VARIANTS = ["cat_bce", "dot_mse", "dot_wbce", "dot_bce_thr", "cat_wbce_thr"]


@pytest.mark.parametrize("name", DENSE_VARIANTS)
def test_dense_variants_golden(ops, golden, name):
    """The non-default arms of the dense path against the reference's DLRM_Net + loss_fn_wrap: "cat" interaction,
    MSE / weighted BCE, --loss-threshold -- through the fused output head (last layer + loss + its input gradient)."""
    g = golden("dense_" + name)
    op, kind, thr = str(g["op"]), ops.LOSS[str(g["loss_kind"])], float(g["loss_threshold"])
    ws = [float(x) for x in g["loss_weights"]]
    nb, nt = len(g["ln_bot"]) - 1, len(g["ln_top"]) - 1
    X, Tt = t(g["X"]).to(DEV), t(g["T"]).to(DEV)
    B, F = X.shape[0], 6
    ly = [t(g[f"ly_{k}"]) for k in range(5)]
    D = ly[0].shape[1]
    feat = torch.zeros(B, F, D, device=DEV)
    for k in range(5):
        feat[:, k + 1] = ly[k].to(DEV)
    Wb = [t(g[f"bot_w{i}"]).to(DEV) for i in range(nb)]
    bb = [t(g[f"bot_b{i}"]).to(DEV) for i in range(nb)]
    Wt = [t(g[f"top_w{i}"]).to(DEV) for i in range(nt)]
    bt = [t(g[f"top_b{i}"]).to(DEV) for i in range(nt)]
    bot = [X]
    for i in range(nb):
        y = feat[:, 0, :] if i == nb - 1 else torch.empty(B, Wb[i].shape[0], device=DEV)
        ops.linear_fwd(bot[-1], Wb[i], bb[i], y, 1)
        bot.append(y)
    if op == "dot":
        R = torch.empty(B, D + F * (F - 1) // 2, device=DEV)
        ops.interact_fwd(feat, False, R)
    else:
        R = feat.view(B, F * D)
    top = [R]
    for i in range(nt - 1):
        y = torch.empty(B, Wt[i].shape[0], device=DEV)
        ops.linear_fwd(top[-1], Wt[i], bt[i], y, 1)
        top.append(y)
    Z, Zc, dZ = torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV)
    dY = torch.empty_like(top[-1])
    lossbuf = torch.zeros(65, device=DEV)
    scratch = ops.head_scratch(DEV)
    for _ in range(2):      # twice: the arrival counter in scratch must be left at zero
        ops.head_fwd_bwd(top[-1], Wt[-1], bt[-1], Tt, Z, dZ, dY, lossbuf, scratch, x_act=1, kind=kind, weights=ws,
                         threshold=thr, Zc=Zc)
    torch.cuda.synchronize()
    np.testing.assert_allclose(Zc.cpu().numpy(), g["Z"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(lossbuf[0]), float(g["loss"]), rtol=2e-6)
    # the same numbers from the stand-alone loss kernel on the head's Z
    lb2, dZ2, Zc2 = torch.zeros(65, device=DEV), torch.empty_like(dZ), torch.empty_like(Z)
    ops.loss_fwd_bwd(Z, Tt, lb2, dZ2, kind=kind, weights=ws, threshold=thr, Zc=Zc2, sigmoid_bwd=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(lb2[0]), float(g["loss"]), rtol=2e-6)
    np.testing.assert_allclose(dZ2.cpu().numpy(), dZ.cpu().numpy(), rtol=1e-6, atol=1e-12)
    assert torch.equal(Zc2, Zc)
    # backward: weight / bias gradients of every layer from the pre-activation gradients, feature gradients
    dzs_top = [None] * nt
    dzs_top[-1] = dZ
    cur = dY
    for i in reversed(range(nt - 1)):
        dzs_top[i] = cur
        dX = torch.empty(B, Wt[i].shape[1], device=DEV)
        work = ops.linear_bwd_work(B, Wt[i].shape[0], Wt[i].shape[1], DEV)
        ops.linear_bwd(top[i], Wt[i], top[i + 1], cur, dX, None, None, 0, work, x_act=(1 if i > 0 else 0))
        cur = dX
    dR = cur
    dfeat = torch.empty_like(feat)
    if op == "dot":
        ops.interact_bwd(feat, dR, False, dfeat, x_act=1)
    else:
        dfeat.copy_(dR.view(B, F, D))
        ops.act_bwd(dfeat[:, 0, :], feat[:, 0, :], 1)
    dzs_bot = [None] * nb
    cur = dfeat[:, 0, :]
    for i in reversed(range(nb)):
        dzs_bot[i] = cur
        if i > 0:
            dX = torch.empty(B, Wb[i].shape[1], device=DEV)
            work = ops.linear_bwd_work(B, Wb[i].shape[0], Wb[i].shape[1], DEV)
            ops.linear_bwd(bot[i], Wb[i], bot[i + 1], cur, dX, None, None, 0, work, x_act=1)
            cur = dX
    torch.cuda.synchronize()
    for k in range(5):
        np.testing.assert_allclose(dfeat[:, k + 1].cpu().numpy(), g[f"ly_grad_{k}"], rtol=2e-4, atol=1e-8)
    for pre, acts, dzs, n in (("top", top, dzs_top, nt), ("bot", bot, dzs_bot, nb)):
        for i in range(n):
            dW = (dzs[i].t() @ acts[i]).cpu().numpy()       # plumbing check of the dZ buffers (wgrad kernels: own tests)
            np.testing.assert_allclose(dW, g[f"{pre}_gw{i}"], rtol=3e-4, atol=1e-7)
            np.testing.assert_allclose(dzs[i].sum(0).cpu().numpy(), g[f"{pre}_gb{i}"], rtol=3e-4, atol=1e-7)


@pytest.mark.parametrize("n", [1, 777, 40000])
@pytest.mark.parametrize("kind,thr", [("bce", 0.0), ("mse", 0.0), ("wbce", 0.0), ("bce", 0.3), ("wbce", 0.2), ("mse", 0.1)])
def test_loss_kernel_vs_oracle(ops, n, kind, thr):
    rng = np.random.RandomState(n + len(kind))
    z = torch.from_numpy(rng.rand(n, 1).astype(np.float32))
    z[0, 0] = 1e-9 if n > 0 else z[0, 0]            # log clamp at -100 / denominator floor
    tt = torch.from_numpy(np.round(rng.rand(n, 1)).astype(np.float32))
    ws = torch.tensor([0.3, 1.7], dtype=torch.float64)
    zr = z.clone().requires_grad_(True)
    zc = torch.clamp(zr, min=thr, max=1 - thr) if thr > 0 else zr
    E = O.loss_fn(zc, tt, kind, ws)
    E.backward()
    lb, dZ, Zc = torch.zeros(65, device=DEV), torch.empty(n, 1, device=DEV), torch.empty(n, 1, device=DEV)
    ops.loss_fwd_bwd(z.to(DEV), tt.to(DEV), lb, dZ, kind=ops.LOSS[kind], weights=(0.3, 1.7), threshold=thr, Zc=Zc)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(lb[0]), float(E), rtol=2e-5)
    np.testing.assert_allclose(dZ.cpu().numpy(), zr.grad.numpy(), rtol=2e-5, atol=1e-12)
    assert torch.equal(Zc.cpu(), zc.detach())


@pytest.mark.parametrize("B,K,x_act", [(1, 8, 0), (1030, 256, 1), (8192, 256, 1), (333, 479, 2), (64, 5, 1)])
def test_head_vs_torch_fp32(ops, B, K, x_act):
    """Fused head against plain torch fp32: sigmoid(Y w + b), BCE, dZ and dY."""
    g = torch.Generator().manual_seed(B + K)
    Y = torch.rand(B, K, generator=g) - (0.3 if x_act == 1 else 0.0)
    if x_act == 1:
        Y = torch.relu(Y)
    w, b = torch.randn(1, K, generator=g) * 0.2, torch.randn(1, generator=g)
    tt = torch.round(torch.rand(B, 1, generator=g))
    Yr = Y.clone().requires_grad_(True)
    pre = torch.nn.functional.linear(Yr, w, b)
    Zr = torch.sigmoid(pre)
    E = torch.nn.functional.binary_cross_entropy(Zr, tt)
    pre.retain_grad()
    E.backward()
    mask = (Y > 0).float() if x_act == 1 else ((1 - Y) * Y if x_act == 2 else torch.ones_like(Y))
    Z, dZ = torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV)
    pitch = (K + 3) // 4 * 4
    Yd = torch.zeros(B, pitch, device=DEV)
    Yd[:, :K] = Y.to(DEV)
    dYd = torch.zeros(B, pitch, device=DEV)
    lb, scratch = torch.zeros(65, device=DEV), ops.head_scratch(DEV)
    ops.head_fwd_bwd(Yd[:, :K], w.to(DEV), b.to(DEV), tt.to(DEV), Z, dZ, dYd[:, :K], lb, scratch, x_act=x_act)
    torch.cuda.synchronize()
    np.testing.assert_allclose(Z.cpu().numpy(), Zr.detach().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(float(lb[0]), float(E), rtol=1e-5)
    np.testing.assert_allclose(dZ.cpu().numpy(), pre.grad.numpy(), rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(dYd[:, :K].cpu().numpy(), (Yr.grad * mask).numpy(), rtol=1e-4, atol=1e-9)
    # the two-call form: partial sums first, the loss when cdlrm_head_finish has run
    lb2 = torch.full((65,), -1.0, device=DEV)
    ops.head_fwd_bwd(Yd[:, :K], w.to(DEV), b.to(DEV), tt.to(DEV), Z, dZ, dYd[:, :K], lb2, scratch, x_act=x_act, finish=False)
    torch.cuda.synchronize()
    assert float(lb2[0]) == -1.0
    acc = torch.zeros(2, dtype=torch.float64, device=DEV)
    ops.head_finish(scratch, B, lb2, acc=acc)
    torch.cuda.synchronize()
    assert torch.equal(lb2[:3], lb[:3])
    # ... and the running float64 sums of [correct predictions, loss * B] over steps (main_no_ddp.py:427-433)
    ops.head_finish(scratch, B, lb2, acc=acc)
    assert acc.tolist() == [2.0 * float(lb[1]), 2.0 * float(lb[2])]


def test_gather_launch_timing_events(ops):
    """cdlrm_ctx_time_next_gather: the next gather leaves its own start / stop timestamps in the caller's events (attached
    to the launch), once; the result of the gather is unchanged and agrees with events recorded around it."""
    T, P, ways, D, n = 3, 64, 4, 128, 4096
    ln = [5000, 300, 70000]
    ctx = ops.CacheCtx(ln, [P] * T, D, ways, n, DEV)
    tags = torch.full((ctx.total_tags,), -1, dtype=torch.int64, device=DEV)
    weight = torch.randn(ctx.total_rows, D, device=DEV)
    ctx.bind_cache(tags, weight)
    g = torch.Generator().manual_seed(3)
    slots = torch.stack([torch.randint(0, P * ways, (n,), generator=g) for _ in range(T)]).to(torch.int32).to(DEV)
    out0 = torch.empty(n, T + 1, D, device=DEV)
    out1 = torch.empty_like(out0)
    ops.embbag_fwd(ctx, slots, None, out0[:, 1:, :], (T + 1) * D, D)
    e0, e1 = ops.TimingEvent(), ops.TimingEvent()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ops.time_next_gather(ctx, e0, e1)
    t0.record()
    ops.embbag_fwd(ctx, slots, None, out1[:, 1:, :], (T + 1) * D, D)
    t1.record()
    torch.cuda.synchronize()
    us, around = e0.elapsed_us(e1), t0.elapsed_time(t1) * 1e3
    assert 0.5 < us <= around + 1.0, (us, around)
    assert torch.equal(out0[:, 1:, :], out1[:, 1:, :])
    ops.embbag_fwd(ctx, slots, None, out1[:, 1:, :], (T + 1) * D, D)        # not timed again: the events keep their stamps
    torch.cuda.synchronize()
    assert abs(e0.elapsed_us(e1) - us) < 1e-3


def test_streamed_window_unique_equals_one_shot(ops):
    """cdlrm_window_unique_add x k + _finish (a window fed in chunks) gives the sorted unique lists of the one-shot
    scan over the concatenated window, and leaves the bitmap ready for the next window."""
    rng = np.random.RandomState(12)
    ln_emb = [5000, 64, 70000, 3]
    D, ways, aux, cache_sizes = 8, 4, 64, [100, 64, 100, 3]
    st = DevState(ops, ln_emb, cache_sizes, D, ways, aux,
                  [torch.full((p, ways), -1, dtype=torch.int64) for p in cache_sizes],
                  [torch.zeros(ways * p + aux, D) for p in cache_sizes], [torch.zeros(n, D) for n in ln_emb])
    plan = ops.WindowPlan(st.ctx, 6000)
    for rep in range(2):
        chunks = [torch.stack([torch.from_numpy(rng.randint(0, n, size=m).astype(np.int64)) for n in ln_emb]).to(DEV)
                  for m in (700, 1, 2048, 333)]
        for c in chunks:
            plan.unique_add(c)
        plan.unique_finish()
        uo, _, _ = plan.offsets()
        got = plan.uniq[:uo[-1]].cpu().clone()
        plan.unique(torch.cat(chunks, dim=1))
        uo2, _, _ = plan.offsets()
        assert uo == uo2 and torch.equal(got, plan.uniq[:uo2[-1]].cpu())
        full = torch.cat(chunks, dim=1).cpu()
        for k in range(len(ln_emb)):
            assert torch.equal(got[uo[k]:uo[k + 1]], torch.unique(full[k]))
    st.ctx.check()


@pytest.mark.parametrize("T,nb,D", [(3, 33, 16), (2, 129, 128)])
def test_embbag_multihot_empty_bags_and_scratch_bag(ops, T, nb, D):
    """The layout engine.square_bags() hands the kernels: per-table offsets with EMPTY bags in the middle, a trailing
    extra bag that holds padding lookups and a zero gradient row -- forward against torch's embedding_bag, fused
    backward + SGD against the oracle (the padding must leave every row bit-identical to the unpadded update)."""
    rng = np.random.RandomState(T * 100 + nb)
    P = 200
    rows = 2 * P + 8
    w0 = [torch.from_numpy(rng.randn(rows, D).astype(np.float32)) for _ in range(T)]
    st = DevState(ops, [rows] * T, [P] * T, D, 2, 8, [torch.full((P, 2), -1, dtype=torch.int64)] * T, w0,
                  [torch.zeros(rows, D)] * T)
    real_b = nb - 1
    lens = rng.randint(0, 6, size=(T, real_b))
    lens[:, 0] = np.maximum(lens[:, 0], 1)                      # every table has at least one lookup
    n_real = lens.sum(1)
    n = int((n_real.max() + 31) // 32 * 32 + 32)                # squared-off width: every table gets padding
    slots = np.zeros((T, n), dtype=np.int32)
    offs = np.zeros((T, nb), dtype=np.int64)
    for k in range(T):
        s = rng.randint(0, 2 * P, size=n_real[k])
        slots[k, :n_real[k]] = s
        slots[k, n_real[k]:] = s[0]                             # padding repeats the table's first lookup
        offs[k, :real_b] = np.concatenate([[0], np.cumsum(lens[k])[:-1]])
        offs[k, real_b] = n_real[k]                             # the scratch bag
    slots_t, offs_t = torch.from_numpy(slots).to(DEV), torch.from_numpy(offs).to(DEV)
    out = torch.zeros(nb, T, D, device=DEV)
    ops.embbag_fwd(st.ctx, slots_t, offs_t, out, T * D, D)
    grad = torch.from_numpy(rng.randn(nb, T, D).astype(np.float32))
    grad[real_b] = 0.0                                          # the scratch bag's gradient row is zero
    work = ops.embbag_bwd_work(st.ctx, n, DEV)
    ops.embbag_bwd_sgd(st.ctx, slots_t, offs_t, grad.to(DEV), T * D, D, 0.3, work, None)
    torch.cuda.synchronize()
    for k in range(T):
        real_slots = torch.from_numpy(slots[k, :n_real[k]].astype(np.int64))
        real_offs = torch.from_numpy(offs[k, :real_b])
        want = torch.nn.functional.embedding_bag(real_slots, w0[k], real_offs, mode="sum")
        np.testing.assert_allclose(out[:real_b, k].cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
        assert bool((want[lens[k] == 0] == 0).all())            # empty bags pool to zero
        w = w0[k].clone()
        O.embbag_bwd_sgd(w, real_slots, real_offs, grad[:real_b, k, :], 0.3)
        np.testing.assert_allclose(st.w(k).numpy(), w.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("alpha", [1.05, 1.0, 0.0])
def test_synthetic_index_stream_device_vs_host(ops, alpha):
    """cdlrm_synth_indices (csrc/synth.hip): the counter-based index stream of cdlrm_amd.synth.CriteoSynth generated on the
    device against its numpy restatement -- the same integers except where float64 pow rounds differently on host and device
    (a handful of lookups in a million, each landing on a neighbouring rank) --, inside every table's range, and independent of
    how the stream is cut into windows / chunks (the look-ahead plan and the trainer ask for different chunkings)."""
    from cdlrm_amd.synth import CriteoSynth
    ln = [39884406, 17289, 3, 2953546, 1]
    B = 1000
    dev_s = CriteoSynth(ln, 13, B, seed=77, alpha=alpha, device=DEV)
    cpu_s = CriteoSynth(ln, 13, B, seed=77, alpha=alpha, device="cpu")
    a = dev_s.window(3, 7).cpu()
    b = cpu_s.window(3, 7)
    assert a.shape == b.shape == (5, 7000) and a.dtype == torch.int64
    for k, n in enumerate(ln):
        assert int(a[k].min()) >= 0 and int(a[k].max()) < n
    assert float((a != b).double().mean()) < 1e-4
    # chunking independence: batches 21 .. 27 as one window of 7, or as seven windows of 1
    parts = torch.cat([dev_s.window(21 + j, 1) for j in range(7)], dim=1).cpu()
    assert torch.equal(parts, a)
    # another seed, another stream
    assert not torch.equal(CriteoSynth(ln, 13, B, seed=78, alpha=alpha, device=DEV).window(3, 7).cpu(), a)
    if alpha > 0:       # skew: the hottest id of the big table takes far more than its uniform share
        big = dev_s.window(0, 50)[0]
        assert int(torch.bincount(big % 1000003).max()) > 50 * big.numel() / 1000003


@pytest.mark.parametrize("act", [0, 1, 2])
def test_smallk_rows_kernel_equals_the_tiled_one(ops, act):
    """The 13-wide first layer on the weights-in-registers kernel (M >= 256, N % 256 == 0) computes every output with the
    tiled kernel's arithmetic (k ascending from 0, bias last): the same rows through the tiled kernel (a batch below 256)
    are bit-identical; a ragged batch and a row pitch wider than N included."""
    rng = np.random.RandomState(7 + act)
    M, N, K = 8192 + 37, 512, 13
    X = torch.from_numpy(rng.randn(M, K).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rng.randn(N, K) / 3.6).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rng.randn(N).astype(np.float32)).to(DEV)
    Yb = torch.full((M, N + 8), 7.0, device=DEV)
    Y = Yb[:, :N]
    ops.linear_fwd(X, W, b, Y, act)
    assert bool((Yb[:, N:] == 7.0).all())
    for lo in (0, 4000, M - 200):
        Yt = torch.empty(200, N, device=DEV)
        ops.linear_fwd(X[lo:lo + 200], W, b, Yt, act)          # 200 rows: the tiled kernel
        assert torch.equal(Yt, Y[lo:lo + 200].contiguous()), lo
    Yn = torch.empty(M, N, device=DEV)
    ops.linear_fwd(X, W, None, Yn, act)                         # no bias
    pre = X.cpu().double() @ W.cpu().double().t()
    ref = {0: pre, 1: torch.relu(pre), 2: torch.sigmoid(pre)}[act]
    np.testing.assert_allclose(Yn.cpu().numpy(), ref.float().numpy(), rtol=2e-5, atol=2e-5)


def test_victim_writeback_last_occurrence_wins(ops):
    """cdlrm_victim_writeback on an EMPTY cache (every lookup misses): every distinct index's host row becomes the aux row of
    its LAST occurrence in the batch (position order), rows of indices not in the batch stay; no victim rows bound (the
    victim-list search is skipped), wsrc = NULL."""
    from cdlrm_amd.model_no_ddp import Embedding_Table_Cache_Group, Embedding_Table_Group
    ln = np.array([500, 37])
    D, n = 16, 96
    rng = np.random.RandomState(5)
    host = Embedding_Table_Group(D, ln, init="empty_meta")
    for k in range(2):
        host.emb_l[k].weight.data = torch.from_numpy(rng.randn(int(ln[k]), D).astype(np.float32))
    host.pin()
    cg = Embedding_Table_Cache_Group(D, ln, 16, n, 2, device=DEV).to(DEV)
    cg.ctx.bind_host_tables(host.device_pointers())
    idx = torch.stack([torch.from_numpy(rng.randint(0, 40, size=n)), torch.from_numpy(rng.randint(0, 37, size=n))]).to(DEV)
    slots, miss_pos, miss_count = ops.embbag_probe(cg.ctx, idx)
    torch.cuda.synchronize()
    assert miss_count.tolist() == [n, n]                    # empty cache: all misses, aux row i = position i
    before = [host.emb_l[k].weight.data.clone() for k in range(2)]
    for k in range(2):                                      # "training": every aux row gets a value that names its position
        first_aux = cg.num_ways * int(cg.cache_sizes[k])
        cg.emb_l[k].weight.data[first_aux:first_aux + n] += torch.arange(1, n + 1, device=DEV, dtype=torch.float32).view(-1, 1)
    aux_rows = [cg.emb_l[k].weight.data[cg.num_ways * int(cg.cache_sizes[k]):][:n].clone().cpu() for k in range(2)]
    work = ops.victim_writeback_work(cg.ctx, n)
    ops.victim_writeback(cg.ctx, idx, slots, None, 0, work)
    torch.cuda.synchronize()
    for k in range(2):
        want = before[k].clone()
        for p, i in enumerate(idx[k].cpu().tolist()):
            want[i] = aux_rows[k][p]
        assert torch.equal(host.emb_l[k].weight.data, want), k
