"""CPU-only tests of the host side: the reference-shaped surface (names, geometry, init, CLI, window
grouping), the C-ABI library's symbol table, and that the product path refuses to run without the HIP
library / device instead of falling back to anything."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="session")
def built_lib():
    from cdlrm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.lib()


def test_library_exports_every_declared_symbol(built_lib):
    """include/cdlrm_hip.h is the boundary: every function it declares must be exported by the .so and bound in
    cdlrm_amd/_lib.py (no compute call is made here: there is no GPU)."""
    from cdlrm_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "cdlrm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cdlrm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(built_lib, name), "libcdlrm_hip.so does not export " + name
        assert name in _lib.PROTOTYPES, "cdlrm_amd/_lib.py does not bind " + name
    assert set(_lib.PROTOTYPES) <= declared, set(_lib.PROTOTYPES) - declared
    assert built_lib.cdlrm_abi_version() == 1


def test_launch_tape_generic_call(built_lib):
    """csrc/tape.hip: the typed call reaches interleaved float / int32 / pointer / int64 arguments and cells (the
    library's own self-test), and a tape built from Python replays a recorded call with a patched cell -- no GPU needed:
    the probe entry point only adds its arguments up."""
    import ctypes as C
    from cdlrm_amd import _lib
    assert built_lib.cdlrm_tape_selftest() == 0
    assert _lib.native_tape_ok()
    L = _lib.raw()
    probe = L.cdlrm_tape_probe
    probe.restype = C.c_int
    probe.argtypes = [C.c_float, C.c_int64, C.c_float, C.c_int32, C.c_void_p, C.c_int64, C.c_float, C.c_int32, C.c_int64,
                      C.c_int64, C.c_void_p, C.c_int32, C.c_float, C.c_int64]
    cell = C.c_void_p(1 << 33)
    tape = _lib.NativeTape([(probe, (0.5, 1, 0.25, -2, cell, 3, 2.0, 4, 5, 6, None, 7, 1.0, 8), True)], {"x": cell})
    assert int(L.cdlrm_tape_length(tape._h)) == 1
    assert tape.replay() == 0
    cell.value = 12345
    assert tape.replay() == 0          # (the sums live in the library; the self-test above checks their values)
    with pytest.raises(_lib.TapeUnsupported):
        _lib.NativeTape([(print, ("x",), False)], {})
    # an entry point that is not registered with its type in csrc/tape.hip is refused, not called through a guessed one
    with pytest.raises(_lib.TapeUnsupported, match="not a registered tape entry point"):
        _lib.NativeTape([(L.cdlrm_ctx_destroy, (None,), True)], {})


def test_no_cpu_fallback():
    from cdlrm_amd import ops
    from cdlrm_amd.model_no_ddp import Embedding_Table_Cache_Group, Embedding_Table_Group
    with pytest.raises(RuntimeError):
        ops.CacheCtx([100], [10], 8, 2, 4, torch.device("cpu"))
    np.random.seed(0)
    cg = Embedding_Table_Cache_Group(8, np.array([100, 7]), 10, 4, 2, cache_init="zeros")
    eg = Embedding_Table_Group(8, np.array([100, 7]))
    with pytest.raises(RuntimeError):
        cg(torch.arange(4).repeat(2, 1), torch.zeros(2, 4, dtype=torch.int64), eg, "cpu")
    if not torch.cuda.is_available():
        from cdlrm_amd.cache_manager import Prefetcher
        with pytest.raises(RuntimeError):
            Prefetcher.process_batch_slice(torch.zeros(2, 4, dtype=torch.int64), eg)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from cdlrm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.CdlrmLibraryError):
        _lib.lib()


def test_isprime_and_geometry(golden):
    from cdlrm_amd.model_no_ddp import Embedding_Table_Cache_Group, isPrime
    g = golden("isprime")
    assert np.array_equal(np.array([isPrime(n) for n in range(1, 5000)], dtype=np.uint8), g["isprime_1_4999"])
    cg = Embedding_Table_Cache_Group(4, np.array([3000, 50, 7, 1200]), 40, 16, 4, cache_init="zeros")
    for c, p in zip(g["next_prime_in"], g["next_prime_out"]):
        assert cg.find_next_prime(int(c)) == int(p)
    gw = golden("cache_windows_small")
    assert cg.max_cache_size == int(gw["P"]) and cg.cache_sizes == [int(x) for x in gw["cache_sizes"]]
    occ = cg.occupancy_tables
    for k, P in enumerate(cg.cache_sizes):
        assert tuple(occ[k].shape) == (P, 4) and occ[k].dtype == torch.int64 and int((occ[k] != -1).sum()) == 0
        assert tuple(cg.emb_l[k].weight.shape) == (4 * P + 16, 4)
    assert torch.equal(cg.compute_set_indices(0, torch.tensor([0, 41, 82, 100])), torch.tensor([0, 0, 0, 100 % 41]))


def test_init_matches_reference(golden):
    """numpy-seeded init of host tables and MLPs (model_no_ddp.py:70-73, 255-261) and the cache rows' default
    N(0,1) init from the torch CPU generator (:138)."""
    from cdlrm_amd.model_no_ddp import DLRM_Net, Embedding_Table_Cache_Group, Embedding_Table_Group, _linears
    g = golden("init")
    seed = int(g["seed"])
    ln_emb, m_spa = g["ln_emb"], int(g["m_spa"])
    np.random.seed(seed)
    torch.manual_seed(seed)
    eg = Embedding_Table_Group(m_spa, ln_emb)
    for k in range(len(ln_emb)):
        assert torch.equal(eg.emb_l[k].weight.data[:4], t(g[f"host_head_{k}"]))
        assert float(eg.emb_l[k].weight.data.double().sum()) == float(g[f"host_sum_{k}"])
    np.random.seed(seed)
    torch.manual_seed(seed)
    cg = Embedding_Table_Cache_Group(m_spa, ln_emb, 100, 32, 4)
    dl = DLRM_Net(g["ln_bot"], g["ln_top"], "dot", False, True, -1, len(g["ln_top"]) - 2, 0.0)
    for k in range(len(ln_emb)):
        assert list(cg.emb_l[k].weight.shape) == list(g[f"cache_shape_{k}"])
        assert torch.equal(cg.emb_l[k].weight[:4], t(g[f"cache_head_{k}"]))
    for i, l in enumerate(_linears(dl.bot_l)):
        assert torch.equal(l.weight.data, t(g[f"bot_w{i}"])) and torch.equal(l.bias.data, t(g[f"bot_b{i}"]))
    for i, l in enumerate(_linears(dl.top_l)):
        assert torch.equal(l.weight.data, t(g[f"top_w{i}"])) and torch.equal(l.bias.data, t(g[f"top_b{i}"]))
    assert isinstance(dl.top_l[-1], torch.nn.Sigmoid) and isinstance(dl.bot_l[-1], torch.nn.ReLU)


def test_window_grouping_matches_reference(golden):
    """The order in which Prefetcher.run groups loader batches into FIFO entries (cache_manager.py:85-110),
    captured by running the reference's Prefetcher on a fake loader."""
    import types
    from cdlrm_amd.cache_manager import Prefetcher, window_groups
    g = golden("window_groups")
    ci = 0
    while f"case{ci}_cfg" in g.files:
        nb, L, cw = [int(x) for x in g[f"case{ci}_cfg"]]
        want = [[int(x) for x in row if x >= 0] for row in g[f"case{ci}_groups"]]
        assert window_groups(nb, L, cw) == want
        B = 4
        ld = [(None, None, torch.stack([torch.arange(j * B, (j + 1) * B), torch.arange(j * B, (j + 1) * B) % 7]), None)
              for j in range(nb)]
        args = types.SimpleNamespace(lookahead=L, cache_workers=cw, nepochs=1)
        pf = Prefetcher(args, None, None, None, None, ld)
        got = [sorted(set((s[0] // B).tolist())) for s in pf.window_slices()]
        assert got == want
        ci += 1


def test_process_args_accepts_reference_cli():
    from cdlrm_amd.main_no_ddp import ProcessArgs
    a = ProcessArgs([])
    assert (a.lookahead, a.cache_workers, a.cache_size, a.num_ways, a.world_size, a.table_agg_freq) == (2, 2, 10240, 4, 2, 1)
    assert a.loss_function == "mse" and a.numpy_rand_seed == 123 and a.lr_embeds == 0.3
    readme = ("--arch-sparse-feature-size=128 --arch-mlp-bot=13-512-256-128 --arch-mlp-top=512-512-256-1 "
              "--data-generation=dataset --data-set=terabyte --loss-function=bce --round-targets=True "
              "--learning-rate=0.8 --lr-embeds=0.8 --mini-batch-size=8192 --print-freq=1024 --test-freq=20000 "
              "--lookahead=3000 --cache-size=150000 --num-ways=16 --table-agg-freq=100 --batch-fifo-size=8 "
              "--cache-workers=4 --world-size=8 --data-sub-sample-rate=0.875 --large-batch --memory-map "
              "--master-port=29500 --average-on-writeback --mlperf-bin-loader").split()
    a = ProcessArgs(readme)
    assert a.lookahead == 3000 and a.cache_size == 150000 and a.num_ways == 16 and a.world_size == 8
    assert a.round_targets is True and a.average_on_writeback and a.large_batch


def test_unique_index_map():
    from cdlrm_amd.cache_manager import UniqueIndexMap
    u = torch.tensor([1, 3, 6, 8, 11, 13])
    m = UniqueIndexMap(u)
    assert list(m.shape) == [14, 1]
    assert m[torch.tensor([6, 1, 13, 2])].flatten().tolist() == [2, 0, 5, -1]


def test_synthetic_is_counter_based():
    from cdlrm_amd.synth import CriteoSynth, KAGGLE_COUNTS, TERABYTE_COUNTS
    assert len(KAGGLE_COUNTS) == 26 and len(TERABYTE_COUNTS) == 26 and sum(TERABYTE_COUNTS) == 187767399
    s1 = CriteoSynth([1000, 7, 50000], 13, 16, seed=5, device="cpu")
    s2 = CriteoSynth([1000, 7, 50000], 13, 16, seed=5, device="cpu")
    w = s1.window(3, 4)
    assert torch.equal(w, s2.window(3, 4)) and w.shape == (3, 64) and w.dtype == torch.int64
    assert not torch.equal(w, s1.window(4, 4))
    for k, n in enumerate([1000, 7, 50000]):
        assert int(w[k].min()) >= 0 and int(w[k].max()) < n
    X, T = s1.dense(0)
    assert X.shape == (16, 13) and set(T.unique().tolist()) <= {0.0, 1.0}
    # skew: the hottest id of the big table covers far more than 1/n of the lookups
    big = CriteoSynth([50000], 13, 4096, seed=1, alpha=1.05, device="cpu").window(0, 4)[0]
    assert torch.bincount(big).max().item() > 50 * big.numel() / 50000


def test_square_bags_and_pad_window():
    """engine.square_bags: ragged per-table index lists -> rectangular [T, n] + offsets [T, B + 1]; the pooled sums of
    the real bags are unchanged, the padding lands in the extra bag and only repeats indices the batch already has."""
    import torch
    from cdlrm_amd.engine import pad_window, square_bags
    lS_i = [torch.tensor([5, 9, 9, 2, 7]), torch.tensor([1, 4]), torch.tensor([3, 3, 3])]
    lS_o = [torch.tensor([0, 2, 3]), torch.tensor([0, 1, 1]), torch.tensor([0, 1, 2])]     # table 1: bag 1 is empty
    off, idx = square_bags(lS_o, lS_i, multiple=4)
    assert off.shape == (3, 4) and idx.shape == (3, 8) and idx.dtype == torch.int64
    for k in range(3):
        n = lS_i[k].numel()
        assert torch.equal(idx[k, :n], lS_i[k]) and bool((idx[k, n:] == lS_i[k][0]).all())
        assert torch.equal(off[k, :3], lS_o[k]) and int(off[k, 3]) == n
        W = torch.arange(40, dtype=torch.float32).view(10, 4)
        want = torch.nn.functional.embedding_bag(lS_i[k], W, lS_o[k], mode="sum")
        got = torch.nn.functional.embedding_bag(idx[k], W, off[k], mode="sum")
        assert torch.equal(got[:3], want)                      # the real bags
        assert set(idx[k].tolist()) == set(lS_i[k].tolist())   # no new index enters the batch / the window
    # rectangular inputs pass through
    o2, i2 = square_bags(torch.zeros(2, 3, dtype=torch.int64), torch.ones(2, 3, dtype=torch.int64))
    assert o2.shape == (2, 3) and i2.shape == (2, 3)
    win = pad_window([torch.tensor([4, 4, 1]), torch.tensor([7])])
    assert win.shape == (2, 3) and win[1].tolist() == [7, 7, 7] and win[0].tolist() == [4, 4, 1]


def test_tape_recording_is_per_thread_and_destructors_bypass_it():
    """A step's launch tape must only see the recording thread's own calls: the window plan's background thread and
    the garbage collector (CacheCtx.__del__ -> cdlrm_ctx_destroy) make library calls at arbitrary times; a destroy
    recorded on a tape would be replayed on a dangling handle at every step."""
    import threading
    from cdlrm_amd import _lib, ops
    tape = []
    _lib.start_recording(tape)
    try:
        seen = []
        t = threading.Thread(target=lambda: seen.append(_lib.record(lambda: 7)))
        t.start()
        t.join()
        assert seen == [7] and tape == [], "another thread's call landed on this thread's tape"
        assert _lib.record(lambda: 3) == 3 and len(tape) == 1
        # the proxy records, the raw library does not
        assert _lib.lib().cdlrm_abi_version() == 1 and len(tape) == 2
        assert _lib.raw().cdlrm_abi_version() == 1 and len(tape) == 2
    finally:
        _lib.stop_recording()
    import inspect
    assert "_lib.raw()" in inspect.getsource(ops.CacheCtx.__del__)


@pytest.mark.parametrize("CH,nb,skip_next_at", [(16, 200, 63), (4, 37, 7), (1, 9, 3), (3, 10, None)])
def test_window_resolver_ring_never_recycles_a_live_chunk(monkeypatch, CH, nb, skip_next_at):
    """engine.WindowResolver keeps chunk results in a ring of three engine-owned buffers (no per-chunk allocation: the
    caching allocator knows nothing about the prefetch / side streams that write and read them).  Host-side invariant,
    checked here with the library call stubbed out: when chunk c's resolve is issued into slot c % 3, every batch of the
    chunk that held the slot has already been handed out AND its step issued; bench.py's call pattern (batch j, batch
    j + 1, ensure) including the iteration at which it hands no next batch over (the plan launch) runs to the end."""
    import cdlrm_amd.engine as engine
    issued = []

    class Ctx:
        T = 2

    class Eng:
        ctx, world, rank, dev = Ctx(), 1, 0, torch.device("cpu")
        pref = side = engine.S._NullStream()
        _bufs = {}

    monkeypatch.setattr(engine.ops, "window_resolve",
                        lambda ctx, cols, lbs, ws, wsrc, stream=None, batch_len=0: issued.append((ws.data_ptr(), cols.shape[1])))
    B = 8
    win = torch.zeros(2, nb * B, dtype=torch.int64)
    eng = Eng()
    rs = engine.WindowResolver(eng, win, B, chunk=CH)
    stepped = -1                    # last batch whose step has been issued
    seen_issue = len(issued)
    for j in range(nb):
        r = rs.batch(j)
        assert r[0].shape == (2, B)
        if j + 1 < nb and j != skip_next_at:
            rs.batch(j + 1)
        stepped = j                 # eng.step(...) would be issued here
        rs.ensure(j + rs.CH + 2)
        for c in range(seen_issue, len(issued)):        # chunk c was issued: the slot's previous holder is chunk c - 3
            assert c < 3 or (c - 2) * CH - 1 <= stepped, (c, stepped)
            assert issued[c][0] == issued[c % 3][0]     # ... and it is the ring slot, not a new allocation
        seen_issue = len(issued)
    assert len(issued) == -(-nb // CH)
    assert set(eng._bufs) == {("wres", CH * B), ("wres_ev",)}      # one ring of buffers, one ring of events: nothing per chunk


@pytest.mark.parametrize("CH,SL,nb,B", [(16, 2, 100, 8192), (4, 4, 37, 8192), (3, 2, 20, 8192), (16, 2, 70, 1024), (2, 1, 9, 8192)])
def test_window_resolver_sorts_slices_ahead_of_their_steps(monkeypatch, CH, SL, nb, B):
    """engine.WindowResolver.ensure_sorted (the embedding backward's slot sort per look-ahead chunk slice): host-side protocol with
    the library calls stubbed out.  After the ensure() that follows step j the lists of batch j + 2 are sorted (the next step
    issues that batch's take, whose stream waits for the slice); a slice is issued after its chunk's resolve and into the
    chunk's ring slot only when every batch of the slot's previous holder has been stepped; a batch handed to a step has
    sorted lists unless the engine's switch was off when its slice fell due (then it has none and the step sorts its own)."""
    import cdlrm_amd.engine as engine
    S = engine.S
    monkeypatch.setattr(S, "is_hip", lambda dev: True)
    monkeypatch.setattr(S, "new_event", lambda dev, timing=False: S._NullEvent())
    monkeypatch.setattr(S, "current_stream", lambda dev: S._NullStream())
    log = []

    class Ctx:
        T = 2

    class Eng:
        ctx, world, rank, dev = Ctx(), 1, 0, torch.device("cpu")
        pref = side = S._NullStream()
        sort_chunks, sort_slice, sort_after, _emb_done, _events = True, SL, "emb_done", None, {}

        def __init__(self):
            self._bufs, self._pending_resolve, self.mark_next = {}, None, False

        def sort_stream(self, local_batch=0):
            return S._NullStream()

    monkeypatch.setattr(engine.ops, "window_resolve",
                        lambda ctx, cols, lbs, ws, wsrc, stream=None, batch_len=0: log.append(("resolve", ws.data_ptr())))
    monkeypatch.setattr(engine.ops, "embbag_bwd_sorted", lambda ctx, nbc, n, dev: torch.zeros(8, dtype=torch.uint8))
    monkeypatch.setattr(engine.ops, "embbag_bwd_prepare_window",
                        lambda ctx, ws, batch_len, nbc, n, buf, stream=None, j0=0, count=None:
                        log.append(("sort", ws.data_ptr(), buf.data_ptr(), nbc, j0, count, batch_len, n)))
    monkeypatch.setattr(engine.ops, "embbag_bwd_sorted_views", lambda ctx, buf, nbc, n, jl: (buf.data_ptr(), nbc, jl))
    win = torch.zeros(2, nb * B, dtype=torch.int64)
    eng = Eng()
    rs = engine.WindowResolver(eng, win, B, chunk=CH)
    sl = rs.SL
    assert sl == max(1, min(max(SL, SL * 8192 // B), CH)), "slice length: ~16 k lookups per table at short batches"
    first_sorted = {}                                       # batch -> index in the log of the sort that covers it
    seen = [0]

    def check_new_sorts(stepped):
        before, seen[0] = seen[0], len(log)
        for k in range(before, len(log)):
            if log[k][0] != "sort":
                continue
            _, wsp, bufp, nbc, j0, cnt, batch_len, n = log[k]
            c = next(cc for cc in range(-(-nb // CH)) if rs._ring[cc % 3][0].data_ptr() == wsp and cc * CH + j0 > stepped)
            assert ("resolve", wsp) in log[:k], "a slice is sorted after its chunk's resolve was issued"
            assert c < 3 or (c - 3) * CH + CH - 1 <= stepped, "the ring slot's previous lists may still be read"
            assert batch_len == B and n == B and 1 <= cnt <= sl and j0 % sl == 0 and j0 + cnt <= nbc
            for b in range(c * CH + j0, c * CH + j0 + cnt):
                first_sorted[b] = k

    check_new_sorts(-1)                                     # the window's first slices are issued by the constructor
    assert 0 in first_sorted and 1 in first_sorted or nb < 2
    off_at = nb // 2                                        # the engine's switch goes off for one slice in the middle
    for j in range(nb):
        r = rs.batch(j)
        if j + 1 < nb:
            rs.batch(j + 1)
        v = r[3][0].sorted_views(r[3][1])
        if j in first_sorted:
            assert v is not None and v[2] == j % CH, (j, v)
        else:
            assert v is None
        eng.sort_chunks = not (off_at <= j < off_at + 1)
        rs.ensure(j + rs.CH + 2)
        check_new_sorts(j)
        if eng.sort_chunks and j + 2 < nb and not (off_at - 3 * sl <= j <= off_at + 3 * sl):
            assert j + 2 in first_sorted, "the batch after next has its lists before the step that issues its take"
    assert len(first_sorted) >= nb - 2 * sl - 2
    assert rs.sorted_views(nb) is None


def test_window_resolver_hands_due_chunks_to_the_next_step(monkeypatch):
    """Long batches: a chunk that falls due is handed to the engine (`_pending_resolve` + `mark_next`) and issued by the NEXT
    step behind its interaction forward, one chunk per step; the first chunks of a window and a chunk whose batches are
    asked for before a step could place it are issued at once.  Host-side protocol with an engine stand-in (no GPU)."""
    import cdlrm_amd.engine as engine
    S = engine.S
    monkeypatch.setattr(S, "is_hip", lambda dev: True)
    monkeypatch.setattr(S, "new_event", lambda dev, timing=False: S._NullEvent())
    monkeypatch.setattr(S, "current_stream", lambda dev: S._NullStream())
    log = []

    class Ctx:
        T = 2

    class Eng:
        ctx, world, rank, dev = Ctx(), 1, 0, torch.device("cpu")
        pref = side = S._NullStream()

        def __init__(self):
            self._bufs, self._pending_resolve, self.mark_next = {}, None, False

        def _side_gather(self, B):
            return False

        def _issue_resolve(self, pr, rec, main, placed):
            log.append(("placed" if placed else "now", pr["cols"].data_ptr()))

        def flush_pending_resolve(self):
            pr, self._pending_resolve, self.mark_next = self._pending_resolve, None, False
            self._issue_resolve(pr, None, None, placed=False)

        def step(self):                     # what TrainEngine.step does with the hand-over
            if self.mark_next and self._pending_resolve is not None:
                self._issue_resolve(self._pending_resolve, None, None, placed=True)
                self._pending_resolve = None
            self.mark_next = False

    B, CH, nb = 8, 4, 40
    win = torch.zeros(2, nb * B, dtype=torch.int64)
    first_col = lambda c: win[:, c * CH * B:].data_ptr()
    eng = Eng()
    rs = engine.WindowResolver(eng, win, B, chunk=CH)
    assert log == [("now", first_col(0)), ("now", first_col(1))]        # window start: at once
    for j in range(nb - 1):
        before = len(log)
        r0, r1 = rs.batch(j), rs.batch(j + 1)
        assert len(log) == before, "nothing is issued while the batches of resolved chunks are handed out"
        eng.step()
        placed = log[before:]
        assert len(placed) <= 1 and all(kind == "placed" for kind, _ in placed)
        rs.ensure(j + rs.CH + 2)
        assert len(log) == before + len(placed), "a due chunk waits for the next step"
        assert r0[0].shape == r1[0].shape == (2, B)
    assert [p for _, p in log] == [first_col(c) for c in range(nb // CH)]
    assert sum(kind == "placed" for kind, _ in log) == nb // CH - 2
    # a chunk asked for before any step ran: flushed at once
    log.clear()
    eng2 = Eng()
    rs2 = engine.WindowResolver(eng2, win, B, chunk=CH)
    rs2.ensure(3 * CH)                      # chunk 2 falls due: handed over
    assert eng2._pending_resolve is not None and len(log) == 2
    rs2.batch(2 * CH)                       # ... and needed now
    assert eng2._pending_resolve is None and log[-1] == ("now", first_col(2))


def test_tape_registry_matches_the_ctypes_prototypes(built_lib):
    """Every entry point a recorded step can issue is registered in csrc/tape.hip with its true C type, and that type has
    the same number of integer-class and float parameters as the ctypes prototype the host marshals with (a mismatch would
    make the native tape fall back to Python replay silently)."""
    import ctypes as C
    from cdlrm_amd import _lib
    L = _lib.raw()
    step_calls = {"cdlrm_embbag_probe", "cdlrm_embbag_take", "cdlrm_embbag_fwd", "cdlrm_embbag_bwd_prepare",
                  "cdlrm_embbag_bwd_apply", "cdlrm_interact_fwd", "cdlrm_interact_bwd", "cdlrm_linear_fwd", "cdlrm_linear_bwd",
                  "cdlrm_mlp_wgrad", "cdlrm_mlp_wgrad_sgd", "cdlrm_loss_fwd_bwd", "cdlrm_head_fwd_bwd", "cdlrm_head_finish",
                  "cdlrm_act_bwd", "cdlrm_sgd_step", "cdlrm_sgd_step2", "cdlrm_scale_div", "cdlrm_ctx_time_next_gather",
                  "cdlrm_event_record", "cdlrm_stream_wait_event"}
    accepted = set()
    for name, (res, argtypes) in _lib.PROTOTYPES.items():
        if res is not C.c_int or name.startswith("cdlrm_tape_"):
            continue
        args = tuple(0.0 if ty is C.c_float else None if ty is C.c_void_p else 0 for ty in argtypes)
        if any(hasattr(ty, "contents") or ty is C.c_double for ty in argtypes):     # struct pointers, doubles: not step calls
            continue
        try:
            _lib.NativeTape([(getattr(L, name), args, True)], {})
            accepted.add(name)
        except _lib.TapeUnsupported as e:
            assert "not a registered tape entry point" in str(e), (name, str(e))
    assert step_calls <= accepted, sorted(step_calls - accepted)


def test_two_lane_tape_replay_orders_the_lanes(built_lib):
    """csrc/tape.hip, multi-lane replay: lane-0 ops are issued by the replaying thread, the ops of lanes 1 .. 3 by the library's
    helper threads at the same time; an op with a dependency is held back until the named op of the other lane has been issued (the
    host-side order of record / wait calls on one event), and each lane keeps its own tape order.  A logging probe stands in
    for the runtime calls: no GPU involved."""
    import ctypes as C
    from cdlrm_amd import _lib
    L = _lib.raw()
    log = L.cdlrm_tape_probe_log
    log.restype, log.argtypes = C.c_int, [C.c_int64, C.c_int64]
    take = L.cdlrm_tape_probe_log_take
    take.restype, take.argtypes = C.c_int64, [C.c_void_p, C.c_int64]
    buf = (C.c_int64 * 4096)()
    take(buf, 4096)
    n = 40
    # (a) a chain across the lanes: op k waits for op k - 1 of the other lane -> the global order is 0, 1, 2, ...
    tape = _lib.NativeTape([(log, (k, 2000 * (k % 3)), True) for k in range(n)], {})
    lane = (C.c_int32 * n)(*[k % 2 for k in range(n)])
    dep = (C.c_int32 * n)(*[k - 1 for k in range(n)])
    _lib.check(L.cdlrm_tape_set_lanes(tape._h, lane, dep, n))
    for _ in range(50):
        assert tape.replay() == 0
        assert take(buf, 4096) == n and list(buf[:n]) == list(range(n))
    # (b) no dependencies: each lane in its own tape order, the two free against each other
    tape2 = _lib.NativeTape([(log, (k, 500), True) for k in range(n)], {})
    lane2 = [0 if k % 3 else 1 for k in range(n)]
    _lib.check(L.cdlrm_tape_set_lanes(tape2._h, (C.c_int32 * n)(*lane2), (C.c_int32 * n)(*([-1] * n)), n))
    for _ in range(50):
        assert tape2.replay() == 0
        assert take(buf, 4096) == n
        got = list(buf[:n])
        for ln in (0, 1):
            assert [k for k in got if lane2[k] == ln] == [k for k in range(n) if lane2[k] == ln]
    # (a') the same chain over THREE lanes (two helper threads): still the global order 0, 1, 2, ...
    tape3 = _lib.NativeTape([(log, (k, 1500 * (k % 4)), True) for k in range(n)], {})
    _lib.check(L.cdlrm_tape_set_lanes(tape3._h, (C.c_int32 * n)(*[k % 3 for k in range(n)]), dep, n))
    for _ in range(50):
        assert tape3.replay() == 0
        assert take(buf, 4096) == n and list(buf[:n]) == list(range(n))
    assert L.cdlrm_tape_set_lanes(tape3._h, (C.c_int32 * n)(*([4] + [0] * (n - 1))), (C.c_int32 * n)(*([-1] * n)), n) != 0   # lanes 0 .. 3
    # (c) a dependency on a LATER op or on the same lane is refused
    bad = (C.c_int32 * n)(*([-1] * (n - 1) + [n - 1]))
    assert L.cdlrm_tape_set_lanes(tape2._h, (C.c_int32 * n)(*lane2), bad, n) != 0
    same = (C.c_int32 * n)(*([-1, -1, 1] + [-1] * (n - 3)))     # op 2 (lane 0) on op 1 (lane 0)
    assert L.cdlrm_tape_set_lanes(tape2._h, (C.c_int32 * n)(*lane2), same, n) != 0


def test_event_consumers_reject_bad_arguments_through_their_scope_guard(built_lib):
    """cdlrm_linear_bwd / cdlrm_interact_bwd / cdlrm_gather_interact_bwd open the scope that flushes an attached completion event on EVERY exit path; the
    argument-check exits are reachable without a GPU (no launch happens) and must return the error code, not crash."""
    from cdlrm_amd import _lib
    L = _lib.raw()
    assert L.cdlrm_interact_bwd(None, None, 0, 0, 1, 4, 0, 0, None, None) == -22
    assert b"cdlrm_interact_bwd" in L.cdlrm_last_error()
    assert L.cdlrm_linear_bwd(None, 0, None, None, 0, None, 0, None, 0, None, None, 0, 0, 0, 0, 0, None, None) == -22
    # the fused gather + interaction operators (the backward opens the same scope)
    assert L.cdlrm_gather_interact_bwd(None, None, 0, None, 0, None, 0, 0, 0, 0, None, None) == -22
    assert b"cdlrm_gather_interact_bwd" in L.cdlrm_last_error()
    assert L.cdlrm_gather_interact_fwd(None, None, 0, None, 0, 0, 0, None, 0, None) == -22
    assert L.cdlrm_gather_interact_supported(None) == 0


def test_roofline_kernel_choice_and_bytes():
    """tools/roofkernel.py: which kernel of a trace is the roofline kernel (the fused gather + interaction forward when the step
    ran it, the stand-alone gather otherwise) and the bytes a launch is priced at -- the numbers bench.py and DESIGN.md quote."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import roofkernel as rk
    fused = "void k_interact_fwd_s<32, 4, true>(float const*, IaGather, long, int, int, float*, long)"
    block = "void k_interact_fwd_s<32, 4, false>(float const*, IaGather, long, int, int, float*, long)"
    gather = "void k_embbag_fwd_arange_p<32, 4>(TableDesc const*, int, int, HIP_vector_type<float, 4u> const*, int const*, long, float*, long, long, int)"
    assert rk.kind(fused) == "fused" and rk.kind(block) is None and rk.kind(gather) == "gather"
    assert rk.kind("void k_interact_bwd_s<32, 4, true>(float const*)") is None
    assert rk.pick([block, gather, "k_take<32>"]) == "gather"
    assert rk.pick([block, gather, fused]) == "fused"          # (a fused trace also holds bench.py's stand-alone operator timing)
    assert rk.pick([block]) is None
    survey, own = rk.bytes_per_launch("gather", 8192, 26, 128)
    assert survey == 221511680 and own == 8192 * 26 * (8 * 128 + 4)
    survey, own = rk.bytes_per_launch("fused", 8192, 26, 128)
    # rows + slot ids, the dense feature, the 479-wide interaction row on its 480-float pitch
    assert survey == 221511680 and own == 8192 * 26 * 516 + 8192 * 512 + 8192 * 480 * 4 == 129826816
    assert rk.bytes_per_launch("fused", 65536, 26, 128)[1] == 1038614528


def test_asm_mfma_hazards_in_built_library():
    """k_gemm3's MFMAs are asm statements the compiler's hazard recognizer does not see (cdlrm_amd/csrc/gemm_wide.h): the built
    code object must not touch an accumulator within the MFMA's write-back window (tools/mfma_hazard_check.py; the first
    version of the kernel failed exactly this way -- accumulator copies 8 wait states behind the MFMA)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "cdlrm_amd", "csrc", "libcdlrm_hip.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mfma_hazard_check.py"), so], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "k_gemm3" in r.stdout and "0 finding(s)" in r.stdout
