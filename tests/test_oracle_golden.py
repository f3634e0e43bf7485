"""Pins the CPU oracle (oracle/cdlrm_oracle.py) against golden vectors captured from the imported
reference (tools/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import cdlrm_oracle as O

torch.set_num_threads(1)


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_isprime_and_next_prime(golden):
    g = golden("isprime")
    tab = np.array([O.is_prime_ref(n) for n in range(1, 5000)], dtype=np.uint8)
    assert np.array_equal(tab, g["isprime_1_4999"])
    # the quirk count SURVEY.md quotes: disagreements with true primality below 5000
    true_p = np.array([n > 1 and all(n % d for d in range(2, int(n ** 0.5) + 1)) for n in range(1, 5000)])
    assert int((tab.astype(bool) != true_p).sum()) == 388
    for c, p in zip(g["next_prime_in"], g["next_prime_out"]):
        assert O.find_next_prime(int(c)) == int(p)
    assert O.find_next_prime(2000) == 2003 and O.find_next_prime(150000) == 150001


def test_appendix_a_known_answer(golden):
    g = golden("appendix_a")
    host = [t(g["host"])]
    P = int(g["P"][0])
    occ = O.new_occupancy_tables([P], 2)
    weights = [torch.zeros(2 * P + 4, 2)]
    for w in range(2):
        raw = t(g[f"w{w}_raw"]).view(1, -1)
        rows, uniqs, maps = O.process_batch_slice(raw, host)
        assert torch.equal(uniqs[0], t(g[f"w{w}_uniq"]))
        assert list(maps[0].shape) == list(g[f"w{w}_map_shape"])
        torch.manual_seed(w)
        ev, det = O.cache_embeddings(rows, uniqs, occ, weights, [P])
        assert torch.equal(det[0]["q"], t(g[f"w{w}_q"]))
        assert torch.equal(det[0]["way"], t(g[f"w{w}_way"]))
        assert torch.equal(occ[0], t(g[f"w{w}_occ"]))
        assert torch.equal(weights[0], t(g[f"w{w}_weight"]))
        assert torch.equal(ev[0][0], t(g[f"w{w}_ev_idx"]))
        assert torch.equal(ev[0][1], t(g[f"w{w}_ev_rows"]))
    # the numbers SURVEY.md Appendix A prints
    assert occ[0].tolist() == [[-1, -1], [16, 26], [-1, -1], [8, 3], [-1, 4]]
    ly, cg = O.cache_forward(occ, weights, [P], t(g["fwd_lS_o"]), t(g["fwd_lS_i"]), host)
    assert cg[0].tolist() == [10, 1, 11, 6] and cg[0].dtype == torch.int32
    assert torch.equal(cg[0], t(g["fwd_idx"]))
    assert torch.equal(ly[0], t(g["fwd_ly"]))
    assert torch.equal(weights[0], t(g["fwd_weight"]))


@pytest.mark.parametrize("name", ["cache_windows_small", "cache_windows_uniform"])
def test_cache_windows(golden, name):
    g = golden(name)
    ln_emb = [int(x) for x in g["ln_emb"]]
    T, ways, B = len(ln_emb), int(g["ways"]), int(g["B"])
    P, cache_sizes, rows_n = O.cache_geometry(ln_emb, int(g["cache_size"]), ways, B)
    assert P == int(g["P"]) and cache_sizes == [int(x) for x in g["cache_sizes"]]
    host = [t(g[f"host0_{k}"]).clone() for k in range(T)]
    weights = [t(g[f"weight0_{k}"]).clone() for k in range(T)]
    assert [w.shape[0] for w in weights] == rows_n
    occ = O.new_occupancy_tables(cache_sizes, ways)
    saw_evict = saw_contest = saw_hit = False
    for w in range(int(g["nwin"])):
        win = t(g[f"w{w}_win"])
        rows, uniqs, _ = O.process_batch_slice(win, host)
        for k in range(T):
            assert torch.equal(uniqs[k], t(g[f"w{w}_uniq_{k}"]))
            assert torch.equal(rows[k], t(g[f"w{w}_rows_{k}"]))
        pre_occ = [o.clone() for o in occ]
        torch.manual_seed(int(g[f"w{w}_qseed"]))
        ev, det = O.cache_embeddings(rows, uniqs, occ, weights, cache_sizes)
        for k in range(T):
            assert torch.equal(det[k]["q"], t(g[f"w{w}_q_{k}"])), (w, k)
            assert torch.equal(det[k]["way"], t(g[f"w{w}_way_{k}"])), (w, k)
            assert torch.equal(occ[k], t(g[f"w{w}_occ_{k}"])), (w, k)
            assert torch.equal(weights[k], t(g[f"w{w}_weight_{k}"])), (w, k)
            assert torch.equal(ev[k][0], t(g[f"w{w}_ev_idx_{k}"])), (w, k)
            assert torch.equal(ev[k][1], t(g[f"w{w}_ev_rows_{k}"])), (w, k)
            saw_evict |= ev[k][0].numel() > 0
            s = det[k]["slots"]
            saw_contest |= s.numel() != torch.unique(s).numel()
            saw_hit |= bool((pre_occ[k][uniqs[k] % cache_sizes[k]] == uniqs[k].view(-1, 1)).any())
        O.eviction_writeback(host, ev, False)
        for k in range(T):
            assert torch.equal(host[k], t(g[f"w{w}_host_{k}"])), (w, k)
        lS_i = t(g[f"w{w}_fwd_lS_i"])
        lS_o = torch.arange(B).repeat(T, 1)
        ly, cg = O.cache_forward(occ, weights, cache_sizes, lS_o, lS_i, host)
        for k in range(T):
            assert torch.equal(cg[k], t(g[f"w{w}_fwd_idx_{k}"])), (w, k)
            assert torch.equal(ly[k], t(g[f"w{w}_fwd_ly_{k}"])), (w, k)
            assert torch.equal(weights[k], t(g[f"w{w}_fwd_weight_{k}"])), (w, k)
    assert saw_evict and saw_contest and saw_hit


@pytest.mark.parametrize("avg", [0, 1])
def test_eviction_writeback(golden, avg):
    g = golden("writeback_avg%d" % avg)
    host = [t(g[f"before_{k}"]).clone() for k in range(2)]
    ev = [(t(g[f"idx_{k}"]), t(g[f"emb_{k}"])) for k in range(2)]
    O.eviction_writeback(host, ev, bool(avg))
    for k in range(2):
        assert torch.equal(host[k], t(g[f"after_{k}"]))


def test_init(golden):
    g = golden("init")
    seed = int(g["seed"])
    ln_emb, m_spa = [int(x) for x in g["ln_emb"]], int(g["m_spa"])
    np.random.seed(seed)
    host = O.init_host_tables(ln_emb, m_spa)
    for k in range(len(ln_emb)):
        assert torch.equal(host[k][:4], t(g[f"host_head_{k}"]))
        assert float(host[k].double().sum()) == float(g[f"host_sum_{k}"])
    tr = O.OracleTrainer(ln_emb, m_spa, g["ln_bot"], g["ln_top"], cache_size=100, num_ways=4, mini_batch_size=32,
                         seed=seed)
    for k in range(len(ln_emb)):
        assert list(tr.weights[0][k].shape) == list(g[f"cache_shape_{k}"])
        assert torch.equal(tr.weights[0][k][:4], t(g[f"cache_head_{k}"]))
    for i in range(len(tr.bot[0][0])):
        assert torch.equal(tr.bot[0][0][i], t(g[f"bot_w{i}"])) and torch.equal(tr.bot[0][1][i], t(g[f"bot_b{i}"]))
    for i in range(len(tr.top[0][0])):
        assert torch.equal(tr.top[0][0][i], t(g[f"top_w{i}"])) and torch.equal(tr.top[0][1][i], t(g[f"top_b{i}"]))


@pytest.mark.parametrize("itself", [0, 1])
def test_dense_fwd_bwd(golden, itself):
    g = golden("dense_itself%d" % itself)
    nb, nt = len(g["ln_bot"]) - 1, len(g["ln_top"]) - 1
    bw = [t(g[f"bot_w{i}"]).requires_grad_(True) for i in range(nb)]
    bb = [t(g[f"bot_b{i}"]).requires_grad_(True) for i in range(nb)]
    tw = [t(g[f"top_w{i}"]).requires_grad_(True) for i in range(nt)]
    tb = [t(g[f"top_b{i}"]).requires_grad_(True) for i in range(nt)]
    ly = [t(g[f"ly_{k}"]).requires_grad_(True) for k in range(5)]
    X, Tt = t(g["X"]), t(g["T"])
    x = O.mlp_forward(X, bw, bb)
    R = O.interact_features(x, ly, "dot", bool(itself))
    assert torch.equal(R, t(g["R"]))
    Z = O.dlrm_forward(X, ly, (bw, bb), (tw, tb), "dot", bool(itself))
    E = O.loss_fn(Z, Tt, "bce")
    E.backward()
    assert torch.equal(Z, t(g["Z"])) and float(E) == float(g["loss"])
    for k in range(5):
        assert torch.equal(ly[k].grad, t(g[f"ly_grad_{k}"]))
    for i in range(nb):
        assert torch.equal(bw[i].grad, t(g[f"bot_gw{i}"])) and torch.equal(bb[i].grad, t(g[f"bot_gb{i}"]))
    for i in range(nt):
        assert torch.equal(tw[i].grad, t(g[f"top_gw{i}"])) and torch.equal(tb[i].grad, t(g[f"top_gb{i}"]))


DENSE_VARIANTS = ["cat_bce", "dot_mse", "dot_wbce", "dot_bce_thr", "cat_wbce_thr"]


@pytest.mark.parametrize("name", DENSE_VARIANTS)
def test_dense_variants(golden, name):
    """"cat" interaction, MSE / weighted BCE, the --loss-threshold clamp (model_no_ddp.py:297-316,
    main_no_ddp.py:212-221): the oracle against the reference's DLRM_Net + loss_fn_wrap."""
    g = golden("dense_" + name)
    op, kind, thr = str(g["op"]), str(g["loss_kind"]), float(g["loss_threshold"])
    ws = torch.tensor(g["loss_weights"])                    # float64, as main_no_ddp.py:370 builds it
    nb, nt = len(g["ln_bot"]) - 1, len(g["ln_top"]) - 1
    bw = [t(g[f"bot_w{i}"]).requires_grad_(True) for i in range(nb)]
    bb = [t(g[f"bot_b{i}"]).requires_grad_(True) for i in range(nb)]
    tw = [t(g[f"top_w{i}"]).requires_grad_(True) for i in range(nt)]
    tb = [t(g[f"top_b{i}"]).requires_grad_(True) for i in range(nt)]
    ly = [t(g[f"ly_{k}"]).requires_grad_(True) for k in range(5)]
    X, Tt = t(g["X"]), t(g["T"])
    Z = O.dlrm_forward(X, ly, (bw, bb), (tw, tb), op, False, thr)
    E = O.loss_fn(Z, Tt, kind, ws)
    E.backward()
    assert torch.equal(Z.detach(), t(g["Z"])) and float(E) == float(g["loss"])
    for k in range(5):
        assert torch.equal(ly[k].grad, t(g[f"ly_grad_{k}"]))
    for i in range(nb):
        assert torch.equal(bw[i].grad, t(g[f"bot_gw{i}"])) and torch.equal(bb[i].grad, t(g[f"bot_gb{i}"]))
    for i in range(nt):
        assert torch.equal(tw[i].grad, t(g[f"top_gw{i}"])) and torch.equal(tb[i].grad, t(g[f"top_gb{i}"]))


@pytest.mark.parametrize("name", ["embsgd_onehot", "embsgd_multihot"])
def test_embbag_bwd_sgd(golden, name):
    g = golden(name)
    w = t(g["w0"]).clone()
    slots, offs = t(g["slots"]), t(g["offsets"])
    V = torch.nn.functional.embedding_bag(slots, w, offs, mode="sum")
    assert torch.equal(V, t(g["V"]))
    O.embbag_bwd_sgd(w, slots, offs, t(g["grad"]), float(g["lr"]))
    np.testing.assert_allclose(w.numpy(), g["w1"], rtol=1e-6, atol=1e-7)
    touched = torch.unique(slots)
    mask = torch.ones(w.shape[0], dtype=torch.bool)
    mask[touched] = False
    assert torch.equal(w[mask], t(g["w1"])[mask])


def make_batches(g):
    """The batch stream tools/make_golden.py:ref_train draws (numpy RandomState(seed+1))."""
    ln_emb = [int(x) for x in g["ln_emb"]]
    B, seed, alpha = int(g["B"]), int(g["seed"]), float(g["alpha"])
    rng = np.random.RandomState(seed + 1)
    T = len(ln_emb)
    out = []
    for j in range(int(g["nbatch"])):
        X = torch.from_numpy(rng.rand(B, int(g["ln_bot"][0])).astype(np.float32))
        lS_i = torch.stack([torch.from_numpy(((rng.zipf(alpha, size=B).astype(np.int64)) * 2654435761 % ln_emb[k])
                                             .astype(np.int64)) for k in range(T)])
        lS_o = torch.arange(B, dtype=torch.int64).repeat(T, 1)
        Tt = torch.from_numpy(np.round(rng.rand(B, 1)).astype(np.float32))
        out.append((X, lS_o, lS_i, Tt))
    return out


def run_oracle_training(g, world):
    ln_emb = [int(x) for x in g["ln_emb"]]
    L = int(g["L"])
    tr = O.OracleTrainer(ln_emb, int(g["m_spa"]), g["ln_bot"], g["ln_top"] if "ln_top" in g.files else
                         np.array([int(g["m_spa"]) + (len(ln_emb) + 1) * len(ln_emb) // 2] + list(g["top"])),
                         cache_size=int(g["cache_size"]), num_ways=int(g["ways"]), mini_batch_size=int(g["B"]),
                         world_size=world, lr=float(g["lr"]), lr_embeds=float(g["lr_emb"]), lookahead=L,
                         table_agg_freq=int(g["agg_freq"]) if "agg_freq" in g.files else 10 ** 9,
                         table_agg_op=str(g["agg_op"]) if "agg_op" in g.files else "mean", seed=int(g["seed"]))
    batches = make_batches(g)
    for j, (X, lS_o, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            win = torch.cat([b[2] for b in batches[j:j + L]], dim=1)
            if "reseed" not in g.files or bool(g["reseed"]):
                torch.manual_seed(5000 + j)      # else: ONE generator stream since the trainer's seeding, as Run runs
            tr.refill(win)
        tr.step(j, X, lS_o, lS_i, Tt)
    return tr


@pytest.mark.parametrize("name", ["train_small", "train_c1", "train_stream"])
def test_loss_trajectory_w1(golden, name):
    g = golden(name)
    tr = run_oracle_training(g, 1)
    losses = np.array([l[0] for l in tr.losses])
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    for k in range(len(g["ln_emb"])):
        assert torch.equal(tr.occ[k], t(g[f"occ_{k}"]))          # tag state bit-exact
        np.testing.assert_allclose(float(tr.host[k].double().sum()), float(g[f"host_sum_{k}"]), rtol=1e-6)
    for i in range(len(tr.top[0][0])):
        np.testing.assert_allclose(tr.top[0][0][i].numpy(), g[f"top_w{i}"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["train_c3shape", "train_c2shape", "train_c5shape", "train_c4shape"])
def test_loss_trajectory_config_shapes(golden, name):
    """The oracle at the bench configurations' SHAPES (26 Criteo-cardinality tables, D = 128 / 32, 16- / 8-way, the
    configs' MLP widths; tools/make_golden.py:g_train_shapes) against the imported reference's run."""
    g = golden(name)
    torch.set_num_threads(4)
    try:
        tr = run_oracle_training(g, 1)
    finally:
        torch.set_num_threads(1)
    losses = np.array([l[0] for l in tr.losses])
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    for k in range(len(g["ln_emb"])):
        assert torch.equal(tr.occ[k], t(g[f"occ_{k}"]).to(torch.int64)), k         # tag state bit-exact
        # (sums of ~1e6 values of magnitude 1e-2 that nearly cancel: absolute tolerance)
        np.testing.assert_allclose(float(tr.host[k].double().sum()), float(g[f"host_sum_{k}"]), rtol=1e-6, atol=1e-4)
        nrow = int(g["ways"]) * tr.cache_sizes[k]
        np.testing.assert_allclose(float(tr.weights[0][k][:nrow].double().sum()), float(g[f"weight_sum_{k}"]),
                                   rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("name", ["train_w2_mean", "train_w2_freq1", "train_w2_max", "train_w2_sum"])
def test_loss_trajectory_w2(golden, name):
    g = golden(name)
    tr = run_oracle_training(g, 2)
    for r in range(2):
        losses = np.array([l[r] for l in tr.losses])
        np.testing.assert_allclose(losses, g[f"r{r}_losses"], rtol=1e-5)
        for i in range(len(tr.top[r][0])):
            np.testing.assert_allclose(tr.top[r][0][i].numpy(), g[f"r{r}_top_w{i}"], rtol=1e-4, atol=1e-6)
            # bias grads are NOT all-reduced (main_no_ddp.py:237-245): per-rank biases drift apart
            np.testing.assert_allclose(tr.top[r][1][i].numpy(), g[f"r{r}_top_b{i}"], rtol=1e-4, atol=1e-6)
    assert not np.array_equal(g["r0_top_b0"], g["r1_top_b0"])
    for k in range(len(g["ln_emb"])):
        assert torch.equal(tr.occ[k], t(g[f"occ_{k}"]))


def test_qr_operator(golden):
    g = golden("qr")
    c = int(g["c"])
    big = t(g["big_idx"])
    assert torch.equal((big / c).long(), t(g["big_q"]))
    assert int((big / c).long()[4]) == 10_000_000                  # exact floor is 9_999_999
    for op in ("mult", "add", "concat"):
        wq, wr = t(g[f"{op}_wq"]).requires_grad_(True), t(g[f"{op}_wr"]).requires_grad_(True)
        V = O.qr_embedding_bag(t(g[f"{op}_idx"]), t(g[f"{op}_offs"]), wq, wr, c, op)
        assert torch.equal(V, t(g[f"{op}_V"]))
        V.backward(t(g[f"{op}_G"]))
        np.testing.assert_allclose(wq.grad.numpy(), g[f"{op}_gq"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(wr.grad.numpy(), g[f"{op}_gr"], rtol=1e-6, atol=1e-7)


def qr_c4_rows(rows, D, salt):
    """tools/make_golden.py:qr_c4_weights for a list of rows (the 10 GB table itself is not stored)."""
    i = torch.as_tensor(rows, dtype=torch.int64).view(-1, 1)
    d = torch.arange(D, dtype=torch.int64).view(1, -1)
    return ((i * 37 + d * 11 + salt) & 1023).to(torch.float32) / 1024.0 - 0.5


def test_qr_operator_c4_table_size(golden):
    """BASELINE configs[3]'s table size (39.9 M categories, D = 256): the oracle's quotient arithmetic on ids above 2**24
    against the reference's, and its outputs / gradients on the rows those quotients address (compact copy of the touched
    rows: idx' = compact_row * c + r keeps both the quotient and the remainder of every lookup)."""
    g = golden("qr_c4")
    c, D = int(g["c"]), int(g["D"])
    idx, offs = t(g["idx"]), t(g["offs"])
    q = (idx / c).long()
    assert torch.equal(q, t(g["q"])) and int((q != t(g["q_exact"])).sum()) > 30
    rows, inv = torch.unique(q, return_inverse=True)
    idx_c = inv * c + torch.remainder(idx, c)
    for op in ("mult", "add"):
        wq = qr_c4_rows(rows, D, 5).requires_grad_(True)
        wr = qr_c4_rows(torch.arange(c), D, 901).requires_grad_(True)
        V = O.qr_embedding_bag(idx_c, offs, wq, wr, c, op)
        np.testing.assert_allclose(V.detach().numpy(), g[f"{op}_V"], rtol=1e-6, atol=1e-6)
        V.backward(t(g["G"]))
        assert torch.equal(rows, t(g[f"{op}_gq_rows"]))
        if op == "mult":
            np.testing.assert_allclose(wq.grad.numpy(), g[f"{op}_gq_vals"], rtol=1e-5, atol=1e-6)
        else:
            np.testing.assert_allclose(wq.grad.double().sum(dim=1).numpy(), g[f"{op}_gq_rowsum"], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(wr.grad.numpy(), g[f"{op}_gr"], rtol=1e-5, atol=1e-4)


def test_md_operator(golden):
    """Mixed-dimension trick: the solver's widths and PrEmbeddingBag forward/gradients vs the reference's module."""
    g = golden("md")
    for name in ("criteo", "b_budget", "noround", "alpha0"):
        d0, B = int(g[f"solver_{name}_d0"]), float(g[f"solver_{name}_B"])
        d = O.md_solver(g[f"solver_{name}_n"], float(g[f"solver_{name}_alpha"]), d0=None if d0 < 0 else d0,
                        B=None if B < 0 else B, round_dim=bool(g[f"solver_{name}_round"]))
        assert np.array_equal(np.asarray(d, dtype=np.float64), g[f"solver_{name}_d"].astype(np.float64)), name
    d = O.md_solver([100, 5000, 70, 900000], 0.25, d0=16, k=g["solver_k_k"])
    assert np.array_equal(np.asarray(d, dtype=np.float64), g["solver_k_d"].astype(np.float64))
    for name in ("proj", "ident", "w1", "w2"):
        W = t(g[f"{name}_W"]).requires_grad_(True)
        P = t(g[f"{name}_P"]).requires_grad_(True) if f"{name}_P" in g.files else None
        V = O.pr_embedding_bag(t(g[f"{name}_idx"]), t(g[f"{name}_offs"]), W, P)
        assert torch.equal(V, t(g[f"{name}_V"]))
        V.backward(t(g[f"{name}_G"]))
        np.testing.assert_allclose(W.grad.numpy(), g[f"{name}_gW"], rtol=1e-6, atol=1e-7)
        if P is not None:
            np.testing.assert_allclose(P.grad.numpy(), g[f"{name}_gP"], rtol=1e-6, atol=1e-7)


def test_window_groups(golden):
    g = golden("window_groups")
    ci = 0
    while f"case{ci}_cfg" in g.files:
        nb, L, cw = [int(x) for x in g[f"case{ci}_cfg"]]
        want = [[int(x) for x in row if x >= 0] for row in g[f"case{ci}_groups"]]
        assert O.window_groups(nb, L, cw) == want, (nb, L, cw)
        ci += 1
    assert ci >= 5
