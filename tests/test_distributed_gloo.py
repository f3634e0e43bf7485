"""world_size-2 run of the data-parallel engine over gloo on CPU, checked against what the REFERENCE's own
2-process run produced (tests/golden/train_w2_*.npz: its aggregate_gradients / broadcast_and_aggregate /
load_caches_and_broadcast call sites over gloo, tools/make_golden.py).

The HIP kernels cannot run here, so cdlrm_amd.ops is replaced by tests/fake_ops.py (the CPU oracle behind the same
call surface); what is under test is the engine's multi-rank control flow: the flat weight-grad all-reduce with
un-reduced biases, the touched-row merge, sync-to-rank-0 at a refill, replicated deterministic inserts and rank-0
write-back."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batches(g):
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


def _worker(rank, world, port, name, host_shared, ret, defer=False):
    try:
        _worker_body(rank, world, port, name, host_shared, ret, defer)
    except BaseException as e:      # a dead worker must fail the test, not hang it
        import traceback
        ret.put((rank, {"error": traceback.format_exc()}))
        raise


def _worker_body(rank, world, port, name, host_shared, ret, defer=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    import fake_ops
    import cdlrm_amd.engine as engine
    import cdlrm_amd.model_no_ddp as M
    engine.ops = fake_ops
    M.ops = fake_ops
    M.Embedding_Table_Group.device_pointers = lambda self: self._fake_ptrs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    ln_emb = np.array([int(x) for x in g["ln_emb"]])
    m_spa, seed, B, L = int(g["m_spa"]), int(g["seed"]), int(g["B"]), int(g["L"])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2] + [int(x) for x in g["top"]])
    eg = M.Embedding_Table_Group(m_spa, ln_emb, init="empty_meta")
    for k in range(len(ln_emb)):
        eg.emb_l[k].weight.data = host_shared[k]
    eg._fake_ptrs = fake_ops.register_host(host_shared)
    eg._pinned = True
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = M.Embedding_Table_Cache_Group(m_spa, ln_emb, int(g["cache_size"]), B, int(g["ways"]))
    dl = M.DLRM_Net(np.array(g["ln_bot"]), ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0)
    eng = engine.TrainEngine(cg, dl, eg, lr=float(g["lr"]), lr_embeds=float(g["lr_emb"]), world_size=world, rank=rank,
                             table_agg_freq=int(g["agg_freq"]), table_agg_op=str(g["agg_op"]), defer_top_update=defer)
    pipe = engine.WindowPipeline(cg, eg, L * B, parity_rng=True, rank=rank, world_size=world)
    lbs = -(-B // world)            # ceil (main_no_ddp.py:344); the last rank's slice is shorter when world does not divide B
    losses = []
    batches = _batches(g)
    for j, (X, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            eng.sync_touched_to_rank0()
            torch.manual_seed(5000 + j)          # every replica consumes the q stream rank 0 consumed in the reference
            pipe.plan_window(torch.cat([b[1] for b in batches[j:j + L]], dim=1))
            pipe.commit()
            pipe.wait_writeback()
        sl = slice(rank * lbs, (rank + 1) * lbs)
        loss = eng.step(X[sl], lS_i[:, sl].contiguous(), Tt[sl], j=j)
        losses.append(float(loss[0]))
    eng.finish()
    lin = M._linears(dl.top_l)
    ret.put((rank, dict(losses=np.array(losses), occ=[o.clone().numpy() for o in cg.occupancy_tables],
                        top_w=[l.weight.data.clone().numpy() for l in lin],
                        top_b=[l.bias.data.clone().numpy() for l in lin])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,port,defer", [("train_w2_mean", 29811, False), ("train_w2_freq1", 29812, False),
                                             ("train_w2_max", 29813, False), ("train_w2_mean", 29814, True),
                                             ("train_w2_sum", 29815, True)])
def test_two_rank_training_matches_reference(golden, name, port, defer):
    """defer: the top / bottom MLP gradients travel as two exchanges (TrainEngine(defer_top_update=True))."""
    from oracle import cdlrm_oracle as O
    g = golden(name)
    np.random.seed(int(g["seed"]))
    host = [h.share_memory_() for h in O.init_host_tables([int(x) for x in g["ln_emb"]], int(g["m_spa"]))]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, host, ret, defer)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, payload = ret.get(timeout=240)
        assert "error" not in payload, payload["error"]
        got[r] = payload
    for p in procs:
        p.join(timeout=60)
    for r in range(2):
        np.testing.assert_allclose(got[r]["losses"], g[f"r{r}_losses"], rtol=1e-5)
        for k in range(len(g["ln_emb"])):
            assert np.array_equal(got[r]["occ"][k], g[f"occ_{k}"]), (r, k)            # tag replicas == reference's shared tags
        for i in range(len(got[r]["top_w"])):
            np.testing.assert_allclose(got[r]["top_w"][i], g[f"r{r}_top_w{i}"], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(got[r]["top_b"][i], g[f"r{r}_top_b{i}"], rtol=1e-4, atol=1e-6)
    # weights are all-reduced (identical replicas), biases are not (main_no_ddp.py:237-245)
    np.testing.assert_allclose(got[0]["top_w"][0], got[1]["top_w"][0], rtol=0, atol=0)
    assert not np.array_equal(got[0]["top_b"][0], got[1]["top_b"][0])
    for k in range(len(g["ln_emb"])):
        np.testing.assert_allclose(float(host[k].double().sum()), float(g[f"host_sum_{k}"]), rtol=1e-6)


@pytest.mark.parametrize("world,port,defer,name", [(3, 29816, True, "train_w2_mean"), (8, 29817, False, "train_w2_freq1"),
                                                   (8, 29818, True, "train_w2_max")])
def test_three_and_eight_rank_training_matches_oracle(golden, world, port, defer, name):
    """World sizes the reference's goldens do not cover (they are 2-process runs): 8 ranks -- the node the bench targets --
    and 3, where `lbs = ceil(B / world)` leaves the last rank a SHORT slice (32 = 11 + 11 + 10).  Checked against the
    oracle's W-rank emulation (pinned on the reference's 1- and 2-process goldens; same code at any W): per-rank losses,
    shared tag state, identical weight replicas, per-rank biases, host tables after rank 0's write-backs."""
    from oracle import cdlrm_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_golden import make_batches as oracle_batches
    g = golden(name)
    ln_emb = [int(x) for x in g["ln_emb"]]
    L, m_spa = int(g["L"]), int(g["m_spa"])
    ln_top = np.array([m_spa + (len(ln_emb) + 1) * len(ln_emb) // 2] + list(g["top"]))
    tr = O.OracleTrainer(ln_emb, m_spa, g["ln_bot"], ln_top, cache_size=int(g["cache_size"]), num_ways=int(g["ways"]),
                         mini_batch_size=int(g["B"]), world_size=world, lr=float(g["lr"]), lr_embeds=float(g["lr_emb"]),
                         lookahead=L, table_agg_freq=int(g["agg_freq"]), table_agg_op=str(g["agg_op"]), seed=int(g["seed"]))
    ob = oracle_batches(g)
    for j, (X, lS_o, lS_i, Tt) in enumerate(ob):
        if j % L == 0:
            torch.manual_seed(5000 + j)
            tr.refill(torch.cat([b[2] for b in ob[j:j + L]], dim=1))
        tr.step(j, X, lS_o, lS_i, Tt)
    np.random.seed(int(g["seed"]))
    host = [h.share_memory_() for h in O.init_host_tables(ln_emb, m_spa)]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, host, ret, defer)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, payload = ret.get(timeout=400)
        assert "error" not in payload, payload["error"]
        got[r] = payload
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        np.testing.assert_allclose(got[r]["losses"], np.array([l[r] for l in tr.losses]), rtol=1e-5)
        for k in range(len(ln_emb)):
            assert np.array_equal(got[r]["occ"][k], tr.occ[k].numpy()), (r, k)
        for i in range(len(got[r]["top_w"])):
            np.testing.assert_allclose(got[r]["top_w"][i], tr.top[r][0][i].numpy(), rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(got[r]["top_b"][i], tr.top[r][1][i].numpy(), rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(got[r]["top_w"][i], got[0]["top_w"][i], rtol=0, atol=0)     # replicas agree bit for bit
    for k in range(len(ln_emb)):
        np.testing.assert_allclose(float(host[k].double().sum()), float(tr.host[k].double().sum()), rtol=1e-6)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("n", [0, 1, 5, 7, 8, 9, 1000, 1001, 4099])
def test_shard_ranges_tile_a_plan_list(world, n):
    """WindowPipeline._shard_range, the slice of a plan list rank r fetches: the ranges tile [0, n) without overlap, rank r's
    starts at r * chunk -- where the in-place all_gather_into_tensor of commit() expects it -- and a list whose padded length
    world * chunk does not fit the buffer is fetched whole (None).  Remainder cases: n not a multiple of world, ranks whose
    range is empty."""
    import cdlrm_amd.engine as engine
    cap = 4100
    got = np.full(cap, -1)
    for r in range(world):
        p = engine.WindowPipeline.__new__(engine.WindowPipeline)
        p.shard, p.world, p.rank = True, world, r
        sh = p._shard_range(n, cap)
        chunk = -(-n // world) if n else 0
        if n == 0 or chunk * world > cap:
            assert sh is None
            continue
        c, lo, hi = sh
        assert c == chunk and lo == min(r * chunk, n) and hi == min(lo + chunk, n) and lo <= hi
        assert (got[lo:hi] == -1).all()
        got[lo:hi] = r
        assert hi - lo <= chunk and r * chunk + (hi - lo) <= cap
    if n and -(-n // world) * world <= cap:
        assert (got[:n] >= 0).all() and (got[n:] == -1).all()
