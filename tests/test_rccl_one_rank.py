"""The multi-rank code path over a REAL RCCL communicator -- of size 1, the only size a one-GPU box can host.

Every world > 1 branch of the engine runs (`force_collectives=True`): `init_process_group("nccl", device_id=...)`, the
flat / split weight-gradient exchange with `ReduceOp.AVG` (gloo takes the scale + SUM branch instead), the uint8 MAX
all-reduce of the touched-row flags, the chunked touched-row merge on the exchange stream, `sync_touched_to_rank0`'s
broadcast, the in-place `all_gather_into_tensor` of the sharded window fetch at `WindowPipeline.commit()`, and the
barrier behind the write-back (reference call sites: main_no_ddp.py:234-292, 309-321, 343).  On one rank every one of
these collectives is the identity (a mean over one rank, a maximum over one rank), so the run must equal the world-1
fast path BIT FOR BIT -- losses, tags, every dense parameter, every cache row -- and the reference's golden trajectory.
RCCL is asynchronous on its own stream (gloo is host-synchronous), so a missing stream dependency around a collective
shows up here as a mismatch.

The run lives in a spawned child: RCCL state stays out of the pytest process, and a hang ends in a traceback.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _train(g, *, force, defer, chunk, long_batch, device_rng, agg_freq, agg_op):
    """One run of golden `g`'s batch stream; returns everything the comparison needs (CPU tensors)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_engine_parity import DEV, make_batches
    import cdlrm_amd.engine as engine
    import cdlrm_amd.model_no_ddp as M
    ln_emb = np.array([int(x) for x in g["ln_emb"]])
    m_spa, seed, B, L = int(g["m_spa"]), int(g["seed"]), int(g["B"]), int(g["L"])
    nf = len(ln_emb) + 1
    ln_top = np.array([m_spa + nf * (nf - 1) // 2] + [int(x) for x in g["top"]])
    np.random.seed(seed)
    torch.manual_seed(seed)
    host = M.Embedding_Table_Group(m_spa, ln_emb).pin()
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = M.Embedding_Table_Cache_Group(m_spa, ln_emb, int(g["cache_size"]), B, int(g["ways"])).to(DEV)
    dl = M.DLRM_Net(np.array(g["ln_bot"]), ln_top, "dot", False, True, -1, ln_top.size - 2, 0.0).to(DEV)
    eng = engine.TrainEngine(cg, dl, host, lr=float(g["lr"]), lr_embeds=float(g["lr_emb"]), world_size=1, rank=0,
                             table_agg_freq=agg_freq, table_agg_op=agg_op, defer_top_update=defer, force_collectives=force)
    pipe = engine.WindowPipeline(cg, host, L * B, parity_rng=not device_rng, seed=seed, rank=0, world_size=1,
                                 host_gather=device_rng, gather_threads=2, shard_fetch=True, force_collectives=force)
    assert pipe.shard == (force and device_rng) and eng.multi == force
    if chunk:
        eng.agg_chunk_rows = chunk
    if long_batch:
        eng.gather_alone_min = 1
    batches = make_batches(g)
    dev_idx = [b[1].to(DEV) for b in batches]
    losses, merges = [], 0
    for j, (X, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            eng.sync_touched_to_rank0()
            torch.manual_seed(5000 + j)
            win = torch.cat([b[1] for b in batches[j:j + L]], dim=1).to(DEV)
            pipe.plan_window(win)
            n_exchange = None
            if device_rng:
                pipe._worker.join()
                n_exchange = len(pipe._exchange)
            pipe.commit()
            pipe.wait_writeback()
            if device_rng and force:
                assert n_exchange == 2, "winner and victim lists are both exchanged"
            rs = engine.WindowResolver(eng, win, B, chunk=2) if long_batch else None
        nxt = dev_idx[j + 1] if j + 1 < len(batches) and (j + 1) % L != 0 else None
        merges += int(force and j > 0 and j % agg_freq == 0)
        loss = eng.step(X.to(DEV), dev_idx[j], Tt.to(DEV), j=j, next_idx=nxt,
                        res=rs.batch(j % L) if rs is not None else None,
                        next_res=rs.batch(j % L + 1) if (rs is not None and nxt is not None) else None)
        if rs is not None:
            rs.ensure(j % L + rs.CH + 2)
        losses.append(loss[0:1].clone())
    eng.finish()
    cg.ctx.check()
    torch.cuda.synchronize()
    return dict(losses=torch.cat(losses).cpu(), tags=cg.tags.cpu().clone(), params=eng.param_flat.cpu().clone(),
                weight=cg.weight.data.cpu().clone(), host=[E.weight.data.clone() for E in host.emb_l], merges=merges,
                lanes=sorted({t["native"].lanes for t in eng._tapes.values() if t["native"] is not None}),
                avg=bool(eng._reduce_avg()) if force else None)


def _child(port, name, case, ret):
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        # as bench.py and Run: the trainer on a high-priority stream of its own -- launch tapes then replay in two lanes (the side
        # queues' calls from the library's helper thread) around the collectives
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
        forced = _train(g, force=True, **case)
        plain = _train(g, force=False, **case)
        out = dict(merges=forced["merges"], avg=forced["avg"], losses=forced["losses"].numpy(), lanes=forced["lanes"])
        for key in ("losses", "tags", "params", "weight"):
            out["same_" + key] = bool(torch.equal(forced[key], plain[key]))
        out["same_host"] = all(torch.equal(a, b) for a, b in zip(forced["host"], plain["host"]))
        out["tags"] = forced["tags"].numpy()
        ret.put(out)
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        ret.put({"error": traceback.format_exc()})
        raise


CASES = {
    # flat exchange (one AVG all-reduce), a merge every other step in one piece
    "flat": dict(defer=False, chunk=0, long_batch=False, device_rng=False, agg_freq=2, agg_op="mean"),
    # split exchange with the deferred top-MLP update, a merge EVERY step in chunks of 8 rows on the exchange stream
    "split_chunked": dict(defer=True, chunk=8, long_batch=False, device_rng=False, agg_freq=1, agg_op="mean"),
    # the long-batch schedule (chained take, window-resident probe) with MAX merges
    "long_batch_max": dict(defer=True, chunk=16, long_batch=True, device_rng=False, agg_freq=2, agg_op="max"),
    # device-RNG plan with CPU-gathered rows, sharded fetch + in-place all-gather at commit()
    "sharded_fetch": dict(defer=True, chunk=0, long_batch=False, device_rng=True, agg_freq=3, agg_op="sum"),
}


@pytest.mark.parametrize("case,port", [("flat", 29861), ("split_chunked", 29862), ("long_batch_max", 29863),
                                       ("sharded_fetch", 29864)])
def test_multi_rank_path_over_one_rank_rccl_is_the_fast_path(golden, case, port):
    name = "train_small"
    g = golden(name)
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    p = ctx.Process(target=_child, args=(port, name, CASES[case], ret))
    p.start()
    out = ret.get(timeout=400)
    p.join(timeout=60)
    assert "error" not in out, out["error"]
    assert out["avg"] is True, "the ReduceOp.AVG branch is the one RCCL runs"
    assert out["merges"] >= 2
    assert out["lanes"] and min(out["lanes"]) >= 2, "the steps replayed as multi-lane native tapes"
    for key in ("losses", "tags", "params", "weight", "host"):
        assert out["same_" + key], "%s differs between the forced multi-rank path and the one-rank fast path" % key
    if not CASES[case]["device_rng"]:       # parity RNG: also the reference's own trajectory and tag state
        np.testing.assert_allclose(out["losses"], g["losses"], rtol=1e-5)
        want = np.concatenate([np.asarray(g[f"occ_{k}"]).reshape(-1) for k in range(len(g["ln_emb"]))])
        assert np.array_equal(out["tags"], want)
