"""Stream ordering of the multi-rank step around ASYNCHRONOUS collectives, checked on one GPU.

What a one-GPU box cannot show with real backends: gloo is host-synchronous (every collective drains the stream it
follows, so a missing event dependency cannot bite), and a 1-rank RCCL communicator runs every in-place collective as
the identity (a consumer that does not wait for it reads the right bytes anyway).  This test replaces
`torch.distributed` inside the engine by a double with ProcessGroupNCCL's stream semantics and W identical peers:

* a collective runs on the group's OWN stream; that stream first waits for the caller's CURRENT stream (an event), the
  caller's current stream then waits for the collective -- the host never blocks (`work.wait()` of a synchronous-API
  NCCL collective is a stream wait);
* while it runs -- a `torch.cuda._sleep` of configurable length lets the host issue far ahead -- a floating-point
  operand holds NaN, and it only then receives the result (SUM over W identical peers = x * W, AVG / MAX / broadcast
  from rank 0 = x).  A kernel that reads the operand without being ordered behind the collective sees NaN; a
  collective that is not ordered behind its producer reduces the previous step's bytes.

With identical peers the averaged gradient, the merged cache rows and the rank-0 broadcast equal the local ones, so a
world-2 engine fed the batches of `train_small` must end on the one-rank engine's bits -- losses, tags, dense
parameters, cache rows -- with the collectives delayed or not, over the AVG branch (backend "nccl") and the
scale + SUM branch (any other backend), flat and split exchanges, whole and chunked merges.
Reference call sites: main_no_ddp.py:234-292 (aggregate_gradients, broadcast_and_aggregate), :309-321, :417-423.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class AsyncPeers:
    """Stand-in for the `torch.distributed` names the engine uses."""

    class ReduceOp:
        SUM, AVG, MAX = "sum", "avg", "max"

    def __init__(self, world, backend, delay_cycles, forget_wait=False):
        self.world, self.backend, self.delay = int(world), backend, int(delay_cycles)
        self.forget_wait = forget_wait                # negative control: the caller's stream does NOT wait for the collective
        self.stream = torch.cuda.Stream()
        self.calls = {"all_reduce": 0, "broadcast": 0, "barrier": 0, "on_streams": set()}

    def get_backend(self, group=None):
        return self.backend

    def get_world_size(self, group=None):
        return self.world

    def barrier(self, group=None):
        self.calls["barrier"] += 1

    def _run(self, t, fn):
        cur = torch.cuda.current_stream()
        self.calls["on_streams"].add(cur.cuda_stream)
        self.stream.wait_stream(cur)                  # the collective reads its operand as the caller's stream left it
        with torch.cuda.stream(self.stream):
            if t.is_floating_point():
                keep = t.clone()
                t.fill_(float("nan"))                 # ... and the operand is unusable while the collective is in flight
                if self.delay:
                    torch.cuda._sleep(self.delay)
                t.copy_(fn(keep))
            else:
                if self.delay:
                    torch.cuda._sleep(self.delay)
                t.copy_(fn(t.clone()))
        if not self.forget_wait:
            cur.wait_stream(self.stream)              # work.wait(): a stream dependency, the host goes on

    def all_reduce(self, t, op="sum", group=None, async_op=False):
        assert not async_op
        self.calls["all_reduce"] += 1
        W = self.world
        if op == "sum":
            self._run(t, lambda x: x * W)             # W identical peers
        elif op in ("avg", "max"):
            self._run(t, lambda x: x)
        else:
            raise ValueError(op)

    def broadcast(self, t, src=0, group=None, async_op=False):
        assert src == 0 and not async_op
        self.calls["broadcast"] += 1
        self._run(t, lambda x: x)

    def all_gather_into_tensor(self, out, inp, group=None):
        raise AssertionError("the sharded window fetch is not part of this test (shard_fetch=False)")


def _train(g, *, world, defer, chunk, agg_freq, agg_op, long_batch=False, budget=0, skip_pump_wait=False, stats=None):
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
    eng = engine.TrainEngine(cg, dl, host, lr=float(g["lr"]), lr_embeds=float(g["lr_emb"]), world_size=world, rank=0,
                             table_agg_freq=agg_freq, table_agg_op=agg_op, defer_top_update=defer)
    pipe = engine.WindowPipeline(cg, host, L * B * world, parity_rng=True, seed=seed, rank=0, world_size=world, shard_fetch=False)
    assert eng.multi == (world > 1)
    if chunk:
        eng.agg_chunk_rows = chunk
    if long_batch:
        eng.gather_alone_min = 1
    if budget:
        eng.merge_budget_rows, eng.merge_budget_auto = budget, False
    if skip_pump_wait:              # negative control: steps do not wait for the merge rows they use
        eng._pump_wait = lambda everything=False: None if not everything else engine.TrainEngine._pump_wait(eng, True)
    batches = make_batches(g)
    dev_idx = [b[1].to(DEV) for b in batches]
    losses = []
    for j, (X, lS_i, Tt) in enumerate(batches):
        if j % L == 0:
            eng.sync_touched_to_rank0()
            torch.manual_seed(5000 + j)
            # the window holds GLOBAL batches: this rank's slice first, then the (identical) peers' -- the same set of
            # unique indices as the one-rank window, so the same plan
            win = torch.cat([b[1] for b in batches[j:j + L] for _ in range(world)], dim=1).to(DEV)
            pipe.plan_window(win)
            pipe.commit()
            pipe.wait_writeback()
            rs = engine.WindowResolver(eng, win, B * world, chunk=2) if long_batch else None
        nxt = dev_idx[j + 1] if j + 1 < len(batches) and (j + 1) % L != 0 else None
        loss = eng.step(X.to(DEV), dev_idx[j], Tt.to(DEV), j=j, next_idx=nxt,
                        res=rs.batch(j % L) if rs is not None else None,
                        next_res=rs.batch(j % L + 1) if (rs is not None and nxt is not None) else None)
        if rs is not None:
            rs.ensure(j % L + rs.CH + 2)
        if stats is not None and eng._pump is not None:
            stats["steps_with_rows_on_their_way"] = stats.get("steps_with_rows_on_their_way", 0) + 1
        losses.append(loss[0:1].clone())
    eng.finish()
    cg.ctx.check()
    torch.cuda.synchronize()
    return dict(losses=torch.cat(losses).cpu(), tags=cg.tags.cpu().clone(), params=eng.param_flat.cpu().clone(),
                weight=cg.weight.data.cpu().clone(), host=[E.weight.data.clone() for E in host.emb_l])


CASES = {
    # one flat weight-gradient exchange, a merge every other step in one piece
    "flat": dict(defer=False, chunk=0, agg_freq=2, agg_op="mean"),
    # split exchange with the deferred top-MLP update (its all-reduce on the weight-gradient stream, beside the backward),
    # a merge EVERY step in chunks of 8 rows on the exchange stream
    "split_chunked": dict(defer=True, chunk=8, agg_freq=1, agg_op="mean"),
    # the long-batch schedule (gather alone on the training queue, chained take, window-resident probe), MAX merges
    "long_batch_max": dict(defer=True, chunk=16, agg_freq=2, agg_op="max", long_batch=True),
}


LAZY = dict(defer=True, chunk=4, agg_freq=5, agg_op="mean", long_batch=True, budget=4)


@pytest.mark.parametrize("delay", [0, 400000])
def test_merge_rows_on_their_way_across_steps(golden, monkeypatch, delay):
    """engine.MergePump under delayed asynchronous collectives: windows of 32 batches, a merge every 5 steps whose rows travel
    in chunks of 4 over the following steps (exchange stream: gather -> all-reduce -> scatter per chunk) while the steps
    train; a step waits for the chunk that holds the last row it uses.  Two identical peers: the one-rank engine's bits."""
    import cdlrm_amd.engine as engine
    g = golden("train_c1")
    main = torch.cuda.Stream(priority=-1)
    stats = {}
    with torch.cuda.stream(main):
        plain = _train(g, world=1, **LAZY)
        peers = AsyncPeers(2, "nccl", delay)
        monkeypatch.setattr(engine, "dist", peers)
        multi = _train(g, world=2, stats=stats, **LAZY)
    assert stats.get("steps_with_rows_on_their_way", 0) >= 10, stats
    for key in ("losses", "tags", "params", "weight"):
        assert torch.equal(multi[key], plain[key]), "%s differs from the one-rank run (delay %d)" % (key, delay)
    np.testing.assert_allclose(multi["losses"].numpy(), g["losses"], rtol=1e-5)


def test_the_double_notices_a_step_that_does_not_wait_for_its_merge_rows(golden, monkeypatch):
    """Negative control of the case above: without the per-step wait a step reads rows whose chunk is still in flight (NaN
    while the collective owns the buffer, or the un-merged value) and the run leaves the one-rank bits."""
    import cdlrm_amd.engine as engine
    g = golden("train_c1")
    main = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(main):
        plain = _train(g, world=1, **LAZY)
        monkeypatch.setattr(engine, "dist", AsyncPeers(2, "nccl", 400000))
        multi = _train(g, world=2, skip_pump_wait=True, **LAZY)
    assert not torch.equal(multi["weight"], plain["weight"])


@pytest.mark.parametrize("backend", ["nccl", "other"])
@pytest.mark.parametrize("delay", [0, 400000])
@pytest.mark.parametrize("case", sorted(CASES))
def test_world2_step_is_ordered_around_asynchronous_collectives(golden, monkeypatch, case, delay, backend):
    import cdlrm_amd.engine as engine
    g = golden("train_small")
    # as bench.py and Run: the trainer on a high-priority stream of its own (multi-lane tape replay around the collectives)
    main = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(main):
        plain = _train(g, world=1, **CASES[case])
        peers = AsyncPeers(2, backend, delay)
        monkeypatch.setattr(engine, "dist", peers)
        multi = _train(g, world=2, **CASES[case])
    assert peers.calls["all_reduce"] >= 2 * len(plain["losses"])
    if CASES[case]["agg_freq"] > 1:             # rows touched since the last merge travel from rank 0 at the window boundary
        assert peers.calls["broadcast"] >= 1
    if CASES[case]["defer"]:
        assert len(peers.calls["on_streams"]) >= 2, "the top MLP's exchange is issued from the weight-gradient stream"
    assert torch.isfinite(multi["losses"]).all(), multi["losses"]
    for key in ("losses", "tags", "params", "weight"):
        assert torch.equal(multi[key], plain[key]), "%s differs from the one-rank run (%s, delay %d)" % (key, backend, delay)
    assert all(torch.equal(a, b) for a, b in zip(multi["host"], plain["host"]))
    np.testing.assert_allclose(multi["losses"].numpy(), g["losses"], rtol=1e-5)


def test_the_double_notices_a_collective_nobody_waits_for(golden, monkeypatch):
    """Negative control: the same run with the wait behind every collective left out must NOT reproduce the one-rank bits
    (the consumers read operands the collective still owns)."""
    import cdlrm_amd.engine as engine
    g = golden("train_small")
    main = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(main):
        plain = _train(g, world=1, **CASES["flat"])
        monkeypatch.setattr(engine, "dist", AsyncPeers(2, "nccl", 400000, forget_wait=True))
        multi = _train(g, world=2, **CASES["flat"])
    assert not torch.equal(multi["params"], plain["params"])
