"""Host-side mirror of the reference's main_no_ddp.py: CLI, cache control plane and trainer loop.

    ProcessArgs                       main_no_ddp.py:34-145    (every flag accepted; the live set is implemented)
    CacheEmbeddings                   :148-209  set-associative insert/evict of one window (HIP plan/commit)
    loss_fn_wrap / time_wrap / wait_wrap                       :212-231
    aggregate_gradients               :234-247  dense weight-grad mean all-reduce (RCCL), biases NOT reduced
    broadcast_and_aggregate           :250-292  touched-row merge across ranks
    share_occupancy_tables            :295-306
    load_caches_and_broadcast         :309-321
    Run                               :324-502  per-GPU trainer

Launch: `python -m cdlrm_amd.main_no_ddp <reference flags> --world-size=N` starts its N trainer processes itself, one per
GPU (cdlrm_amd/launch.py: the reference's mp.spawn, main_no_ddp.py:638-643, as a torch.distributed.run child started before
the first HIP call); under an external `torchrun` it is one of the ranks.  The reference's Manager queues are replaced by
in-process threads/streams; tensors never cross a process boundary.
"""
from __future__ import annotations

import argparse
import math
import os
import queue
import sys
import threading
import time
from timeit import default_timer as timer

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .cache_manager import Prefetcher
from .model_no_ddp import (CacheSGD, DLRM_Net, Embedding_Table_Cache_Group, Embedding_Table_Group, HipBCELoss, _linears)


def ProcessArgs(argv=None):
    parser = argparse.ArgumentParser(description="Train Deep Learning Recommendation Model (DLRM)")
    # ---- model ----
    parser.add_argument("--arch-sparse-feature-size", type=int, default=2)
    parser.add_argument("--arch-embedding-size", type=str, default="4-3-2")
    parser.add_argument("--arch-mlp-bot", type=str, default="4-3-2")
    parser.add_argument("--arch-mlp-top", type=str, default="4-2-1")
    parser.add_argument("--arch-interaction-op", type=str, default="dot")
    parser.add_argument("--arch-interaction-itself", action="store_true", default=False)
    # ---- activation and loss ----
    parser.add_argument("--activation-function", type=str, default="relu")
    parser.add_argument("--loss-function", type=str, default="mse")  # or bce or wbce
    parser.add_argument("--loss-weights", type=str, default="1.0-1.0")
    parser.add_argument("--loss-threshold", type=float, default=0.0)
    parser.add_argument("--round-targets", type=bool, default=False)
    # ---- data ----
    parser.add_argument("--data-size", type=int, default=1)
    parser.add_argument("--num-batches", type=int, default=0)
    parser.add_argument("--data-generation", type=str, default="random")
    parser.add_argument("--data-trace-file", type=str, default="./input/dist_emb_j.log")
    parser.add_argument("--data-set", type=str, default="kaggle")
    parser.add_argument("--raw-data-file", type=str, default="")
    parser.add_argument("--processed-data-file", type=str, default="")
    parser.add_argument("--data-randomize", type=str, default="total")
    parser.add_argument("--data-trace-enable-padding", type=bool, default=False)
    parser.add_argument("--max-ind-range", type=int, default=-1)
    parser.add_argument("--data-sub-sample-rate", type=float, default=0.0)
    parser.add_argument("--num-indices-per-lookup", type=int, default=10)
    parser.add_argument("--num-indices-per-lookup-fixed", type=bool, default=False)
    parser.add_argument("--num-workers", type=int, default=0)
    parser.add_argument("--memory-map", action="store_true", default=False)
    # ---- embedding table args ----
    parser.add_argument("--md-flag", action="store_true", default=False)
    parser.add_argument("--md-threshold", type=int, default=200)
    parser.add_argument("--md-temperature", type=float, default=0.3)
    parser.add_argument("--md-round-dims", action="store_true", default=False)
    parser.add_argument("--qr-flag", action="store_true", default=False)
    parser.add_argument("--qr-threshold", type=int, default=200)
    parser.add_argument("--qr-operation", type=str, default="mult")
    parser.add_argument("--qr-collisions", type=int, default=4)
    # ---- training ----
    parser.add_argument("--mini-batch-size", type=int, default=1)
    parser.add_argument("--nepochs", type=int, default=1)
    parser.add_argument("--learning-rate", type=float, default=0.1)
    parser.add_argument("--lr-embeds", type=float, default=0.3)
    parser.add_argument("--print-precision", type=int, default=5)
    parser.add_argument("--numpy-rand-seed", type=int, default=123)
    parser.add_argument("--sync-dense-params", type=bool, default=True)
    parser.add_argument("--lookahead", type=int, default=2)
    parser.add_argument("--cache-workers", type=int, default=2)
    parser.add_argument("--cache-size", type=int, default=10240)
    parser.add_argument("--num-ways", type=int, default=4)
    parser.add_argument("--average-on-writeback", action="store_true", default=False)
    parser.add_argument("--evict-victim-cache", action="store_true", default=False)
    # ---- debugging and profiling ----
    parser.add_argument("--print-freq", type=int, default=1)
    parser.add_argument("--test-freq", type=int, default=-1)
    parser.add_argument("--test-mini-batch-size", type=int, default=-1)
    parser.add_argument("--test-num-workers", type=int, default=-1)
    parser.add_argument("--print-time", action="store_true", default=False)
    parser.add_argument("--debug-mode", action="store_true", default=False)
    parser.add_argument("--enable-profiling", action="store_true", default=False)
    parser.add_argument("--plot-compute-graph", action="store_true", default=False)
    # ---- store/load model ----
    parser.add_argument("--save-model", type=str, default="")
    parser.add_argument("--load-model", type=str, default="")
    # ---- mlperf ----
    parser.add_argument("--mlperf-logging", action="store_true", default=False)
    parser.add_argument("--mlperf-acc-threshold", type=float, default=0.0)
    parser.add_argument("--mlperf-auc-threshold", type=float, default=0.0)
    parser.add_argument("--mlperf-bin-loader", action="store_true", default=False)
    parser.add_argument("--mlperf-bin-shuffle", action="store_true", default=False)
    parser.add_argument("--large-batch", action="store_true", default=False)
    # ---- distributed training ----
    parser.add_argument("--world-size", type=int, default=2)
    parser.add_argument("--master-port", type=int, default=12345)
    parser.add_argument("--trainer-start-core", type=int, default=7)
    parser.add_argument("--main-start-core", type=int, default=0)
    parser.add_argument("--dense-threshold", type=int, default=1000)
    parser.add_argument("--table-agg-op", type=str, default="mean")
    parser.add_argument("--table-agg-freq", type=int, default=1)
    parser.add_argument("--batch-fifo-size", type=int, default=8)
    parser.add_argument("--eviction-fifo-size", type=int, default=8)
    parser.add_argument("--eviction-fifo-timeout", type=int, default=300)
    # ---- misc ----
    parser.add_argument("--inference-only", action="store_true", default=False)
    parser.add_argument("--save-onnx", action="store_true", default=False)
    parser.add_argument("--use-gpu", action="store_true", default=False)
    # ---- this build (not in the reference) ----
    parser.add_argument("--synthetic-alpha", type=float, default=1.05,
                        help="Zipf exponent of --data-generation=criteo-synthetic indices (0 = uniform)")
    parser.add_argument("--device-rng", action="store_true", default=False,
                        help="way choice by counter-based Philox on the GPU (perf mode; not bit-comparable with the "
                             "reference's torch-CPU Categorical draw)")
    return parser.parse_args(argv)


# --------------------------------------------------------------------------------------------------
# cache control plane
# --------------------------------------------------------------------------------------------------


def _compat_plan(cache_group: Embedding_Table_Cache_Group, n_uniq: int) -> ops.WindowPlan:
    plan = getattr(cache_group, "_compat_plan", None)
    if plan is None or plan.cap_uniq < n_uniq or plan.ctx is not cache_group.ctx:
        cap = max(1024, int(n_uniq * 1.25))
        plan = ops.WindowPlan(cache_group.ctx, cap, cap_uniq=cap, cap_win=min(cap, cache_group.ctx.total_tags) + 16)
        cache_group._compat_plan = plan
    return plan


@torch.no_grad()
def CacheEmbeddings(cached_entries_per_table, lists_of_unique_idxs, unique_indices_maps, cache_group, eviction_fifo, rank):
    """main_no_ddp.py:148-209 with the reference's signature.  Tag probe of the window's unique ids, full-set
    filter, way choice (the Exp(1) draw comes from the torch CPU generator, table by table, exactly like
    Categorical.sample()), contested-slot resolution, eviction gather, tag + row update -- all on the GPU."""
    dev = cache_group.weight.device
    T = len(cached_entries_per_table)
    plan = _compat_plan(cache_group, sum(int(u.numel()) for u in lists_of_unique_idxs))
    plan.set_unique(lists_of_unique_idxs)
    plan.probe()
    uo, ko, _ = plan.offsets()
    ways = cache_group.num_ways
    qs = []
    for k in range(T):
        M = ko[k + 1] - ko[k]
        qs.append(torch.empty(M, ways).exponential_(1) if M > 0 else torch.empty(0, ways))
    q = torch.cat(qs).contiguous().to(dev) if ko[T] > 0 else torch.empty(1, ways, device=dev)
    plan.assign(q)
    rows = [r.to(dev, torch.float32).contiguous() for r in cached_entries_per_table]
    plan.fetch([r.data_ptr() if r.numel() else cache_group.weight.data_ptr() for r in rows], True)
    plan.commit()
    _, _, wo = plan.offsets()
    cache_group.ctx.check()
    eviction_data = []
    for k in range(T):
        tag = plan.ev_tag[wo[k]:wo[k + 1]]
        valid = tag != -1
        eviction_data.append((tag[valid].clone(), plan.stage[wo[k]:wo[k + 1]][valid].clone()))
    if rank == 0:
        eviction_fifo.put(eviction_data)


def loss_fn_wrap(Z, T, loss_fn, args, loss_ws=None):
    if args.loss_function == "mse" or args.loss_function == "bce":
        return loss_fn(Z, T)
    elif args.loss_function == "wbce":
        loss_ws_ = loss_ws[T.data.view(-1).long()].view_as(T)
        loss_fn_ = loss_fn(Z, T)
    loss_sc_ = loss_ws_ * loss_fn_
    return loss_sc_.mean()


def time_wrap(rank):
    torch.cuda.synchronize(rank)
    return time.time()


def wait_wrap(req_objs):
    for obj in req_objs:
        obj.wait()


def aggregate_gradients(dlrm):
    """main_no_ddp.py:234-247: every Linear's weight.grad /= world, async all-reduce SUM; bias grads are not
    reduced (reference behaviour, kept)."""
    request_objs_mlp = []
    for seq in (dlrm.bot_l, dlrm.top_l):
        for layer in seq:
            if isinstance(layer, nn.modules.linear.Linear):
                ops.scale_div(layer.weight.grad, float(dist.get_world_size()))
                request_objs_mlp.append(dist.all_reduce(layer.weight.grad, async_op=True))
    return request_objs_mlp


@torch.no_grad()
def broadcast_and_aggregate(cache_group, cache_group_idxs, rank, reduce_op="mean"):
    """main_no_ddp.py:250-292.  cache_group_idxs: int32 [T, m] slot ids this rank looked up since the last merge.
    The union over ranks (the reference all-gathers the lists and calls torch.unique per table) is the union of
    touched-row flags; the rows travel as one compacted [U, D] buffer instead of T separate all-reduces."""
    ctx = cache_group.ctx
    dev = cache_group.weight.device
    W = dist.get_world_size()
    flags = torch.zeros(ctx.total_rows, dtype=torch.uint8, device=dev)
    ops.mark_rows(ctx, cache_group_idxs.to(dev, torch.int32).contiguous(), flags)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    rows = torch.empty(ctx.total_rows, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    ops.agg_compact(ctx, flags, rows, count)
    U = int(count.item())
    if U == 0:
        return
    buf = torch.empty(U, ctx.D, dtype=torch.float32, device=dev)
    if reduce_op == "sum":
        ops.agg_gather(ctx, rows, count, 1.0, buf, U)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    elif reduce_op == "mean":
        ops.agg_gather(ctx, rows, count, float(W), buf, U)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    elif reduce_op == "max":
        ops.agg_gather(ctx, rows, count, 1.0, buf, U)
        dist.all_reduce(buf, op=dist.ReduceOp.MAX)
    ops.agg_scatter(ctx, rows, count, buf, U)


def share_occupancy_tables(cache_group, occupancy_tables_fifos, rank):
    """main_no_ddp.py:295-306.  The reference shares ONE CPU tag table between the ranks; here every GPU keeps its
    own replica in HBM and all replicas evolve identically (same window payload, same way draws), so there is
    nothing to hand over -- the function only checks that the replicas start out equal."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        chk = cache_group.tags.to(torch.float64).sum().view(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert float(lo) == float(hi), "cache tag replicas differ between ranks"


def _multi_rank() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


@torch.no_grad()
def _take_rank0_rows(cache_group):
    """Every cache row some rank updated since the last refill becomes rank 0's copy on every rank -- what the
    reference's whole-cache broadcast from rank 0 (main_no_ddp.py:318-319) does to the rows that CAN differ between the
    replicas.  The union of the ranks' touched-row flags (set by the fused backward) is compacted into one row list;
    only those rows travel.  The flags are consumed."""
    ctx, dev = cache_group.ctx, cache_group.weight.device
    flags = cache_group.touched
    dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    rows = torch.empty(ctx.total_rows, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    ops.agg_compact(ctx, flags, rows, count)           # clears the flags it lists
    U = int(count.item())
    if U == 0:
        return
    buf = torch.empty(U, ctx.D, dtype=torch.float32, device=dev)
    ops.agg_gather(ctx, rows, count, 1.0, buf, U)
    dist.broadcast(buf, src=0)
    ops.agg_scatter(ctx, rows, count, buf, U)


@torch.no_grad()
def load_caches_and_broadcast(cache_group, batch_fifo, eviction_fifo, rank):
    """main_no_ddp.py:309-321.  Reference: rank 0 takes the window from batch_fifo and inserts it, then EVERY cache
    table is broadcast from rank 0 (10.9 GB at the README config).  Here: rank 0 alone takes the window from the FIFO
    (as in the reference: the FIFO is one shared queue) and broadcasts the window's payload -- the unique lists and
    their host rows, [U, D] instead of whole cache tables --, every rank first takes rank 0's copy of the rows that can
    differ between the replicas, then performs the same deterministic insert (rank 0's Exp(1) generator state is shared
    so the way choices agree).  Same end state on every rank as after the reference's broadcast."""
    multi = _multi_rank()
    dev = cache_group.weight.device
    T = len(cache_group.cache_sizes)
    if not multi:
        cached_entries_per_table, lists_of_unique_idxs, unique_indices_maps = batch_fifo.get()
    else:
        payload = None
        if rank == 0:
            payload = batch_fifo.get()
            sizes = [int(u.numel()) for u in payload[1]]
        else:
            sizes = None
        box = [sizes]
        dist.broadcast_object_list(box, src=0)
        sizes = box[0]
        n = sum(sizes)
        D = cache_group.m_spa
        uniq_flat = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        rows_flat = torch.empty(max(n, 1), D, dtype=torch.float32, device=dev)
        if rank == 0 and n:
            uniq_flat[:n] = torch.cat([u.reshape(-1).to(dev, torch.int64) for u in payload[1]])
            rows_flat[:n] = torch.cat([r.to(dev, torch.float32).reshape(-1, D) for r in payload[0]])
        dist.broadcast(uniq_flat, src=0)
        dist.broadcast(rows_flat, src=0)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        lists_of_unique_idxs = [uniq_flat[offs[k]:offs[k + 1]] for k in range(T)]
        cached_entries_per_table = [rows_flat[offs[k]:offs[k + 1]] for k in range(T)]
        from .cache_manager import UniqueIndexMap
        unique_indices_maps = [UniqueIndexMap(u) for u in lists_of_unique_idxs]
        _take_rank0_rows(cache_group)
        # ranks other than 0 must consume the same Exp(1) stream as rank 0: share its generator state
        state = torch.get_rng_state().to(dev)
        dist.broadcast(state, src=0)
        torch.set_rng_state(state.cpu())
    CacheEmbeddings(cached_entries_per_table, lists_of_unique_idxs, unique_indices_maps, cache_group, eviction_fifo, rank)
    return []


# --------------------------------------------------------------------------------------------------
# trainer
# --------------------------------------------------------------------------------------------------


def Run(rank, m_spa, ln_emb, ln_bot, ln_top, train_ld, test_ld, batch_fifo, eviction_fifo, occupancy_tables_fifos,
        emb_tables, args):
    """main_no_ddp.py:324-502 on the fused engine.  One process per GPU; `rank` is the device index and the
    distributed rank.  train_ld yields (X, lS_o, lS_i, T) global batches; every rank takes its slice."""
    from .engine import TrainEngine, WindowPipeline, WindowResolver, pad_window, square_bags
    try:
        from setproctitle import setproctitle
        setproctitle("DlrmTrainer:" + str(rank))
    except ImportError:
        pass
    np.random.seed(args.numpy_rand_seed)
    torch.cuda.manual_seed(args.numpy_rand_seed)
    torch.manual_seed(args.numpy_rand_seed)
    np.set_printoptions(precision=args.print_precision)
    torch.set_printoptions(precision=args.print_precision)
    # `rank` is the device index AND the distributed rank (main_no_ddp.py:324-344); `args.device_index` (set by main() when
    # ranks are emulated on one GPU) separates the two
    dev_index = int(getattr(args, "device_index", rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    world = args.world_size
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if world > 1 and dist.get_world_size() != world:
        sys.exit("ERROR: --world-size=%d but torch.distributed reports %d ranks" % (world, dist.get_world_size()))
    local_batch_size = math.ceil(args.mini_batch_size / world)
    # the trainer's queue outranks the engine's side queues (see bench.py)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))

    # multi-hot bags (--data-generation=random): a batch has up to mini_batch_size * num_indices_per_lookup lookups per
    # table, each miss takes its own aux row (model_no_ddp.py:176-179) -- the reference sizes the aux region for one
    # lookup per sample and would index past it; here the region covers the squared-off width (slot ids of the first
    # mini_batch_size misses are the reference's)
    multi_hot = getattr(train_ld, "multi_hot", False)
    aux_rows = args.mini_batch_size
    if multi_hot:
        if world > 1:
            sys.exit("ERROR: multi-hot bags (--data-generation=random) run on one rank; the rank slice "
                     "(main_no_ddp.py:388-391) is defined for one lookup per sample")
        aux_rows = (args.mini_batch_size * max(1, args.num_indices_per_lookup) + 255) // 256 * 256
    cache_group = Embedding_Table_Cache_Group(m_spa, ln_emb, max_cache_size=args.cache_size,
                                              aux_table_size=aux_rows, num_ways=args.num_ways).to(dev)
    dlrm = DLRM_Net(ln_bot, ln_top, arch_interaction_op=args.arch_interaction_op,
                    arch_interaction_itself=args.arch_interaction_itself, sync_dense_params=args.sync_dense_params,
                    sigmoid_bot=-1, sigmoid_top=ln_top.size - 2, loss_threshold=args.loss_threshold).to(dev)
    share_occupancy_tables(cache_group, occupancy_tables_fifos, rank)
    if args.loss_function not in ("mse", "bce", "wbce"):
        sys.exit("ERROR: --loss-function=" + args.loss_function + " is not supported")
    # --loss-weights "w0-w1" (main_no_ddp.py:370); only read by wbce
    loss_ws = [float(x) for x in str(args.loss_weights).split("-")] if args.loss_function == "wbce" else [1.0, 1.0]
    emb_tables.pin() if not getattr(emb_tables, "_pinned", False) else None
    if args.load_model:
        # counterpart of --save-model: MLPs + host tables of a previous run (the cache starts empty)
        ld = torch.load(args.load_model, map_location="cpu")
        dlrm.load_state_dict({k: v.to(dev) for k, v in ld["dlrm"].items()})
        if rank == 0 or world == 1:
            for E, w in zip(emb_tables.emb_l, ld["emb"]):
                E.weight.data.copy_(w)
        if world > 1:
            dist.barrier()
    eng = TrainEngine(cache_group, dlrm, emb_tables, lr=args.learning_rate, lr_embeds=args.lr_embeds, world_size=world,
                      rank=rank, table_agg_freq=args.table_agg_freq, table_agg_op=args.table_agg_op,
                      loss=args.loss_function, loss_weights=loss_ws, defer_top_update=True)
    for kv in filter(None, os.environ.get("CDLRM_ENGINE_ATTR", "").split(",")):
        # development (as bench.py --engine-attr): TrainEngine schedule knobs for same-box A/B runs of the CLI, 'name=value,...'
        k_, v_ = kv.split("=")
        assert hasattr(eng, k_), k_
        setattr(eng, k_, eval(v_))
    L = args.lookahead
    # --device-rng (performance mode): the plan of window w+1 is made while window w trains -- what the reference's
    # Prefetcher process is for (cache_manager.py:66-115) -- with its rows gathered by CPU threads in the background.  The
    # plan reads the tag state and the host rows as the previous commit left them, nothing the training changes, so the
    # result is the synchronous plan's.  Parity mode draws the way choices in line at the boundary, as the reference does.
    # (`args.plan_at_boundary`: not a CLI flag -- tests set it to compare against planning every window at its boundary)
    lookahead_plan = bool(args.device_rng) and not getattr(args, "plan_at_boundary", False)
    if args.evict_victim_cache:
        # --evict-victim-cache (main_no_ddp.py:96: parsed and never used by the reference; here the trained rows of MISSED
        # lookups are written back to the host tables behind every step, engine.TrainEngine.evict_victim): one rank, and the
        # next window is planned at the boundary -- its row gather has to see the write-backs of the window's last step
        if world > 1:
            sys.exit("ERROR: --evict-victim-cache is defined for --world-size=1 (every rank would write its own misses' rows "
                     "to the one host table)")
        lookahead_plan = False
        eng.evict_victim = True
    pipe = WindowPipeline(cache_group, emb_tables, L * aux_rows * 2, parity_rng=not args.device_rng,
                          seed=args.numpy_rand_seed, average_on_writeback=args.average_on_writeback, rank=rank,
                          world_size=world, host_gather=lookahead_plan)

    # Print statistics (main_no_ddp.py:427-476) stay on the device between print boundaries: the head's finish launch adds
    # [#correct, loss * mbs] of every step to the engine's float64 accumulator (`eng.stat_acc`, on the side stream: no launch on
    # the training queue for it), and the host reads (and, at world > 1, all-reduces) it every print_freq iterations only.  The reference synchronises the device
    # twice per step for its timer (time_wrap, :401, 425) and moves Z and T to the host every step (:428-431); here the
    # wall time between two print boundaries, minus the refills and the test loop inside it, is the ms/it it prints --
    # the same quantity (iteration time without caching overhead) without stopping the pipeline every step.
    acc = eng.stat_acc                                           # [sum of correct, sum of loss * mbs]
    acc.zero_()
    total_iter = total_samp = 0
    caching_overhead = []
    print_interval_stats = args.print_freq > 0
    stop_training = False               # --mlperf-acc-threshold / --mlperf-auc-threshold reached (decided by rank 0's test loop)
    for epoch in range(args.nepochs):
        if stop_training:
            break
        it = iter(train_ld)
        window = []
        next_window = None          # look-ahead plan: the window whose plan is already in flight
        next_win_idx = cur_win_idx = None
        resolver, wj = None, 0
        carried_idx = None          # device indices of the batch whose probe the previous step already issued
        j = 0
        torch.cuda.synchronize(dev)
        t_mark = time.time()        # start of the current print interval
        t_excluded = 0.0            # refill + test time inside it

        def read_window():
            win = []
            for _ in range(L):
                try:
                    win.append(next(it))
                except StopIteration:
                    break
            return win

        def window_indices(win):
            if multi_hot:           # ragged per-table lists: the plan only needs each table's set of indices
                return pad_window([torch.cat([torch.as_tensor(b[2][k]).reshape(-1) for b in win])
                                   for k in range(len(ln_emb))]).to(dev)
            return torch.cat([torch.as_tensor(b[2]) if not isinstance(b[2], (list, tuple)) else
                              torch.stack([torch.as_tensor(s).reshape(-1) for s in b[2]]) for b in win], dim=1).to(dev)

        while True:
            if not window:
                # look-ahead: the next L batches; their insert plan is made now, or was made while the last window trained
                planned = next_window is not None
                window, next_window = (next_window, None) if planned else (read_window(), None)
                if not window:
                    break
                torch.cuda.synchronize(dev)     # the steps issued so far belong to the iteration time, not to the refill
                start = timer()
                cur_win_idx = next_win_idx if planned else None
                if not planned:
                    cur_win_idx = window_indices(window)
                    pipe.plan_window(cur_win_idx)
                if world > 1:
                    eng.sync_touched_to_rank0()
                pipe.commit()
                pipe.wait_writeback()
                # window-resident probe (one lookup per bag): the window's lookups are resolved against the new tags once
                resolver, wj = None, 0
                if not multi_hot:
                    if cur_win_idx is None:
                        cur_win_idx = window_indices(window)
                    if cur_win_idx.shape[1] == len(window) * args.mini_batch_size:      # whole batches only
                        # (chunks of 32 batches at world > 1: the row merge orders its rows by the batches resolved ahead)
                        resolver = WindowResolver(eng, cur_win_idx, args.mini_batch_size, chunk=16 if world == 1 else 32)
                caching_overhead.append(timer() - start)
                t_excluded += caching_overhead[-1]
                if lookahead_plan:
                    next_window = read_window() or None
                    if next_window is not None:     # evictions are in the host tables: the next plan may read them
                        next_win_idx = window_indices(next_window)
                        pipe.plan_window(next_win_idx)
            X, lS_o, lS_i, T = window.pop(0)
            Or = nxt = None
            sl = slice(rank * local_batch_size, (rank + 1) * local_batch_size)

            def rank_indices(li):       # this rank's slice of a batch's indices, on the device
                li = torch.as_tensor(li) if not isinstance(li, (list, tuple)) else torch.stack(
                    [torch.as_tensor(s).reshape(-1) for s in li])
                v = li[:, sl]
                if v.device.type != "cuda" or v.stride(1) != 1:     # (the day-file loader hands out X_cat^T, a strided view)
                    v = v.contiguous()
                # a batch that already lies on the device as columns of its window (the synthetic front end) is used in place:
                # the take and the probe read rows at a pitch -- packing it cost a 1.7 MB copy kernel on the training queue per step
                return v.to(dev)

            if multi_hot:
                Or, Ir = square_bags([lS_o[k] for k in range(len(ln_emb))], lS_i, dev)
                Xr, Tr = X.to(dev), T.to(dev)
            else:
                Xr = X[sl, :].to(dev)
                Ir = carried_idx if carried_idx is not None else rank_indices(lS_i)
                Tr = T[sl, :].to(dev)
                # the next batch of the SAME window (the look-ahead already holds it): its tag probe and aux-row fill run
                # during this step instead of at the head of the next one
                nxt = rank_indices(window[0][2]) if window else None
                carried_idx = nxt
            rs = resolver if (resolver is not None and Xr.shape[0] == resolver.width) else None
            lossbuf = eng.step(Xr, Ir, Tr, lS_o=Or, j=j, next_idx=nxt,
                               res=rs.batch(wj) if rs is not None else None,
                               next_res=rs.batch(wj + 1) if (rs is not None and nxt is not None) else None,
                               loss_sync=False)
            if rs is not None:
                rs.ensure(wj + rs.CH + 2)
            wj += 1
            mbs = Tr.shape[0]
            total_iter += 1
            total_samp += mbs
            last = (j == len(train_ld) - 1)
            should_print = print_interval_stats and j > 0 and j % args.print_freq == 0
            should_test = test_ld is not None and ((j > 0 and args.test_freq > 0 and j % args.test_freq == 0) or last)
            if should_print:
                eng.finish()                    # the statistics of the steps issued so far are complete behind this
                if world > 1:
                    dist.all_reduce(acc)
                torch.cuda.synchronize(dev)
                total_time = time.time() - t_mark - t_excluded
                if rank == 0:
                    s = acc.tolist()
                    gT = 1000.0 * total_time / total_iter
                    gA = (s[0] / world) / total_samp
                    gL = (s[1] / world) / total_samp
                    avg_caching_overhead = np.mean(caching_overhead) / args.lookahead if caching_overhead else 0.0
                    print('Epoch {}: Finished {}/{} in {} ms/it. Caching overhead = {}. Loss = {}, Train Acc = {}'.format(
                        epoch, j, len(train_ld), gT, 1000 * avg_caching_overhead, gL, gA), flush=True)
                acc.zero_()
                total_iter = total_samp = 0
                caching_overhead = []
                t_mark, t_excluded = time.time(), 0.0
            if should_test and world > 1:
                # rows of a merge that are still travelling land through collectives: every rank issues them here, at the same
                # step, BEFORE the part only rank 0 runs (the test loop issues none)
                eng.drain_merge()
            if rank == 0 and should_test:
                # Testing -- only rank 0 tests (main_no_ddp.py:478-494).  The reference's `j % args.test_freq == 0`
                # with its default test_freq = -1 is true for every j; here test_freq <= 0 means "at the end only".
                t_test = time.time()
                print('Testing at {}/{}....'.format(j, len(train_ld)), flush=True)
                test_samp = 0
                total_test_acc = 0
                test_scores, test_targets = [], []
                for Xt, lS_ot, lS_it, Tt in test_ld:
                    if getattr(test_ld, "multi_hot", False):        # ragged multi-hot test batches
                        Ot, It = square_bags([lS_ot[k] for k in range(len(ln_emb))], lS_it, dev)
                        Zt = eng.evaluate(Xt.to(dev), It, Ot)
                    else:
                        lS_it = torch.as_tensor(lS_it) if not isinstance(lS_it, (list, tuple)) else torch.stack(
                            [torch.as_tensor(s).reshape(-1) for s in lS_it])
                        Zt = eng.evaluate(Xt.to(dev), lS_it.contiguous().to(dev))
                    S_ = Zt.cpu().numpy()
                    Tn = Tt.cpu().numpy()
                    total_test_acc += np.sum((np.round(S_, 0) == Tn).astype(np.uint32))
                    test_samp += Tn.shape[0]
                    test_scores.append(Zt.reshape(-1).clone())
                    test_targets.append(Tt.reshape(-1).to(dev))
                print('Test accuracy = {}%'.format(100 * (total_test_acc / test_samp)), flush=True)
                # the second figure the reference's MLPerf flags speak of (main_no_ddp.py:117-120) and never compute: the AUC of
                # the test scores, rank-sum on the device (ops.roc_auc; an extra line behind the reference's own)
                test_auc = ops.roc_auc(torch.cat(test_scores), torch.cat(test_targets)) if test_scores else float("nan")
                print('Test AUC = {}'.format(test_auc), flush=True)
                if (args.mlperf_acc_threshold > 0 and total_test_acc / test_samp >= args.mlperf_acc_threshold) or \
                        (args.mlperf_auc_threshold > 0 and test_auc >= args.mlperf_auc_threshold):
                    print('MLPerf threshold reached at {}/{}: training stops'.format(j, len(train_ld)), flush=True)
                    stop_training = True
                t_excluded += time.time() - t_test
            if should_test and world > 1 and (args.mlperf_acc_threshold > 0 or args.mlperf_auc_threshold > 0):
                # rank 0 alone tests: every rank learns whether it said stop (one flag, only when a threshold is set)
                flag = torch.tensor([1 if stop_training else 0], device=dev)
                dist.broadcast(flag, src=0)
                stop_training = bool(int(flag[0]))
            j += 1
            if stop_training:
                break
    eng.finish()
    pipe.close()            # a look-ahead plan past the last window may still be in flight
    torch.cuda.synchronize()
    if args.save_model:
        # --save-model is parsed but never acted on by the reference (main_no_ddp.py:111); here: flush every valid
        # cache row to its host row (rank 0's replica, as for evictions), then save the MLPs and the host tables
        if rank == 0:
            cache_group.flush_to_host(emb_tables, average=args.average_on_writeback)
            torch.save({"dlrm": {k: v.detach().cpu() for k, v in dlrm.state_dict().items()},
                        "emb": [E.weight.data.clone() for E in emb_tables.emb_l],
                        "ln_emb": [int(n) for n in ln_emb], "m_spa": int(m_spa)}, args.save_model)
        if world > 1:
            dist.barrier()
    return eng


class _SyntheticLoader:
    """Criteo-layout synthetic loader (X, lS_o, lS_i, T) on the host side of the reference's loop.  The index stream is
    generated on the device in chunks of up to 64 batches (one generator launch per table and chunk instead of one per
    table and batch) and handed out as per-batch views: a loader that keeps up with a sub-millisecond step."""

    CHUNK = 64

    def __init__(self, syn, num_batches, B):
        self.syn, self.n, self.B = syn, num_batches, B

    def __len__(self):
        return self.n

    def __iter__(self):
        T = len(self.syn.ln_emb)
        lS_o = torch.arange(self.B, device=self.syn.device).repeat(T, 1)
        C = self.CHUNK
        chunk, c0 = None, -1
        for j in range(self.n):
            if j // C != c0:
                c0 = j // C
                chunk = self.syn.window(c0, C)          # batches c0*C .. c0*C + C - 1
            idx = chunk[:, (j - c0 * C) * self.B:(j - c0 * C + 1) * self.B]
            X, Tt = self.syn.dense(j)
            yield X, lS_o, idx, Tt


def main(argv=None):
    # The step's streams are tuned for the HIP runtime's default of 4 hardware queues per process (with 6 or 8 the c3
    # step measured 1.39-1.42 ms instead of 0.79): pin the default before the first HIP call, as bench.py does.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
    args = ProcessArgs(argv)
    from . import launch
    if args.world_size > 1 and not launch.under_launcher():
        # the reference's `mp.spawn(Run, nprocs=args.world_size)` (main_no_ddp.py:638-643): this process -- before its first
        # HIP call -- starts one trainer process per GPU as children and exits with their return code
        sys.exit(launch.spawn_ranks(args.world_size, sys.argv[1:] if argv is None else list(argv),
                                    module="cdlrm_amd.main_no_ddp", port=launch.free_port() if launch.emulated() else
                                    args.master_port))
    np.random.seed(args.numpy_rand_seed)
    torch.manual_seed(args.numpy_rand_seed)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if launch.under_launcher() and int(os.environ["WORLD_SIZE"]) != args.world_size:
        sys.exit("ERROR: --world-size=%d does not match the launcher's WORLD_SIZE %s"
                 % (args.world_size, os.environ["WORLD_SIZE"]))
    if rank != local_rank:
        # `Run(rank, ...)` keeps the reference's contract -- ONE integer that is both the device index and the
        # distributed rank (main_no_ddp.py:324-344, single node, MASTER_ADDR=localhost) -- and the host tables are one
        # /dev/shm mapping shared by the ranks of a node
        sys.exit("ERROR: cdlrm_amd.main_no_ddp runs on one node (RANK %d != LOCAL_RANK %d), like the reference"
                 % (rank, local_rank))
    ln_bot = np.fromstring(args.arch_mlp_bot, dtype=int, sep="-")
    if args.data_generation not in ("random", "synthetic", "criteo-synthetic", "dataset"):
        sys.exit("ERROR: --data-generation=%s is not supported (dataset | criteo-synthetic | random | synthetic)"
                 % args.data_generation)
    train_ld = test_ld = None
    if args.data_generation == "dataset":
        # the pre-processed day files of the reference's terabyte path (dlrm_data_pytorch.py:440-492): table sizes
        # from <raw>_fea_count.npz (:180-181), batches from <raw>_<day>_reordered.npz
        from .data_loader_terabyte import DataLoader
        d_dir, d_name = os.path.dirname(args.raw_data_file) or ".", os.path.basename(args.raw_data_file)
        with np.load(args.raw_data_file + "_fea_count.npz") as data:
            counts = data["counts"]
        args.arch_embedding_size = "-".join(str(int(c)) for c in counts)
        days = [d for d in range(24) if os.path.exists(os.path.join(d_dir, "%s_%d_reordered.npz" % (d_name, d)))]
        if not days:
            sys.exit("ERROR: no %s_<day>_reordered.npz under %s" % (d_name, d_dir))
        train_days, test_days = (days[:-1], days[-1:]) if len(days) > 1 else (days, days)
        train_ld = DataLoader(d_name, d_dir, train_days, args.mini_batch_size, args.max_ind_range, "train", True)
        tb = args.test_mini_batch_size if args.test_mini_batch_size > 0 else args.mini_batch_size
        test_ld = DataLoader(d_name, d_dir, test_days, tb, args.max_ind_range, "test")
    ln_emb = np.fromstring(args.arch_embedding_size, dtype=int, sep="-")
    if args.max_ind_range > 0:
        ln_emb = np.minimum(ln_emb, args.max_ind_range)
    m_den = ln_bot[0]
    m_spa = args.arch_sparse_feature_size
    num_fea = ln_emb.size + 1
    m_den_out = ln_bot[ln_bot.size - 1]
    if args.arch_interaction_op == "dot":
        num_int = (num_fea * (num_fea + 1)) // 2 + m_den_out if args.arch_interaction_itself else \
            (num_fea * (num_fea - 1)) // 2 + m_den_out
    elif args.arch_interaction_op == "cat":
        num_int = num_fea * m_den_out
    else:
        sys.exit("ERROR: --arch-interaction-op=" + args.arch_interaction_op + " is not supported")
    ln_top = np.fromstring(str(num_int) + "-" + args.arch_mlp_top, dtype=int, sep="-")
    if m_spa != m_den_out:
        sys.exit("ERROR: arch-sparse-feature-size " + str(m_spa) + " does not match last dim of bottom mlp " + str(m_den_out))
    if args.md_flag or args.qr_flag:
        # the reference computes md_solver's width list here and then builds plain tables from it (main_no_ddp.py:612-621),
        # which raises in nn.EmbeddingBag; --qr-flag only changes a width check (:579-596), the tables stay plain.  A cache row has one width and one source
        # row, so neither trick has cached semantics: the operators are available stand-alone
        # (cdlrm_amd.tricks, Embedding_Table_Group(qr_flag= / md_flag=)).
        sys.exit("ERROR: --md-flag / --qr-flag tables cannot feed the embedding cache (stand-alone operators only)")
    from . import _lib
    try:
        _lib.require_gpu("cdlrm_amd.main_no_ddp")
    except _lib.CdlrmLibraryError as e:
        sys.exit("ERROR: " + str(e))
    # development: CDLRM_BENCH_EMULATE=1 places every rank on device 0 with gloo collectives (tests on a one-GPU box)
    emulate = launch.emulated() and args.world_size > 1
    dev_index = 0 if emulate else local_rank
    if not emulate and args.world_size > torch.cuda.device_count():
        sys.exit("ERROR: --world-size=%d, but this node has %d GPUs (one trainer process per GPU)"
                 % (args.world_size, torch.cuda.device_count()))
    args.device_index = dev_index
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if args.world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        if emulate:
            dist.init_process_group("gloo", rank=rank, world_size=args.world_size)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=args.world_size, device_id=dev)
    launch.check_world(args.world_size)
    from . import synth
    from .hostmem import make_host_tables
    emb_tables = make_host_tables(ln_emb, m_spa, device=dev, seed=args.numpy_rand_seed, rank=rank, world=args.world_size,
                                  shm_name="cdlrm_run_%d" % args.master_port,
                                  barrier=(dist.barrier if args.world_size > 1 else (lambda: None)))
    if train_ld is None and args.data_generation in ("random", "synthetic"):
        # the reference's random front ends (dlrm_data_pytorch.py:658-684): uniform multi-hot bags, or bags with the reuse
        # profile of a recorded trace (--data-trace-file, "j" = table number); ragged tables either way
        from .dlrm_data_pytorch import make_random_data_and_loader
        _, train_ld = make_random_data_and_loader(args, ln_emb, m_den)
        train_ld.multi_hot = True
    if train_ld is None:
        nb = args.num_batches if args.num_batches > 0 else max(1, args.data_size // args.mini_batch_size)
        syn = synth.CriteoSynth(ln_emb, int(m_den), args.mini_batch_size, seed=args.numpy_rand_seed,
                                alpha=args.synthetic_alpha, device=dev)
        train_ld = _SyntheticLoader(syn, nb, args.mini_batch_size)
    Run(local_rank, m_spa, ln_emb, ln_bot, ln_top, train_ld, test_ld, None, None, None, emb_tables, args)
    if args.world_size > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
