/*
 * libcdlrm_hip.so -- C ABI of the MI355X (gfx950) look-ahead embedding-cache training path.
 *
 * The reference (lkp411/cDLRM) is pure Python over stock ATen ops: it has NO FFI / plugin interface
 * for this path.  This header is therefore the build's own thin lower boundary; every
 * entry point names the reference call site (file:line in the reference tree) whose work it takes
 * over.  The upper boundary (same names / argument meaning as the reference) is the Python package
 * cdlrm_amd/{model_no_ddp,cache_manager,main_no_ddp}.py, which binds this library through ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - every function returns 0 on success, a positive hipError_t, or a negative CDLRM_E* code;
 *     nothing throws across the ABI; cdlrm_last_error() returns a thread-local message.
 *   - all tensor memory is CALLER-owned (torch tensors' data_ptr()); the library allocates only the
 *     opaque context (descriptor copies, small scan scratch, one device error word).
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); calls are
 *     asynchronous on it unless the name ends in _sync.
 *   - "device-visible" pointers may point to HBM or to pinned / registered host memory.
 *   - slot ids follow the reference: slot = P_k * way + set, aux slots start at P_k * ways
 *     (model_no_ddp.py:174, 177); tags are int64 [P_k, ways] row-major, -1 = empty (:144-147).
 */
#ifndef CDLRM_HIP_H
#define CDLRM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDLRM_ABI_VERSION 1

#define CDLRM_EINVAL (-22)   /* bad argument / shape the kernels do not support            */
#define CDLRM_ERANGE (-34)   /* an index was outside its table, or a capacity was exceeded */
#define CDLRM_ENOMEM (-12)

typedef struct cdlrm_ctx cdlrm_ctx;

/* Geometry of one cache group (Embedding_Table_Cache_Group.__init__, model_no_ddp.py:101-147). */
typedef struct {
    int32_t num_tables;          /* T                                                     */
    int32_t dim;                 /* D, fp32 elements per row (multiple of 4)               */
    int32_t num_ways;            /* 1..64                                                  */
    int32_t aux_rows;            /* aux_table_size (main_no_ddp.py:348)                    */
    const int64_t* table_rows;   /* [T] n_k  rows of the host master table                 */
    const int64_t* cache_sets;   /* [T] P_k = min(n_k, find_next_prime(cache_size))        */
    int32_t device;              /* HIP device ordinal                                     */
    int32_t aux_phases;          /* 0/1: one aux region per table; 2: two (aux rows double-  */
                                 /* buffered: the NEXT batch's misses are filled while the    */
                                 /* current batch still trains on its own aux rows)           */
} cdlrm_geometry;

int cdlrm_abi_version(void);
/* DEVELOPMENT EXPORT, not part of the drop-in surface (no reference counterpart; a maintainer binds nothing to it): key 0 .. 7
 * selects a kernel variant the current one replaced, or a grid size, so that two builds' worth of behaviour can be timed against
 * each other on one box in one process (tools/ab_step.py --attr debug:<key>, bench.py --debug, which prints what it set in
 * config.debug).  All zero unless a tool sets them.  No key skips work in the shipped library (the two timing experiments that
 * do -- key 6, bits 1 and 2 -- are compiled in by -DCDLRM_DEV only and refused here otherwise); variants are bit-identical
 * except the GEMM kernel selectors of key 6, whose kernels differ in contraction order (equal to fp32 rounding); grid sizes
 * only change placement.  The list of keys: csrc/common.h. */
int cdlrm_debug_set(int32_t key, int32_t value);
const char* cdlrm_last_error(void);

int cdlrm_ctx_create(const cdlrm_geometry* geo, cdlrm_ctx** out);
int cdlrm_ctx_destroy(cdlrm_ctx* ctx);

/* Bind the caller-owned cache state.  tags: int64 [sum_k P_k*ways] (table k at element offset
 * sum_{j<k} P_j*ways); weight: fp32 [sum_k rows_k, D], rows_k = ways*P_k + aux*max(1, aux_phases) (table k at row
 * sum_{j<k} rows_j; its first ways*P_k + aux rows are the reference's cache table).
 * Replaces the per-table nn.EmbeddingBag weights + CPU occupancy tables of model_no_ddp.py:130-147. */
int cdlrm_ctx_bind_cache(cdlrm_ctx* ctx, int64_t* tags, float* weight);

/* Bind the host master tables (Embedding_Table_Group.emb_l[k].weight, model_no_ddp.py:61-74):
 * host_rows[k] is a device-visible pointer to fp32 [n_k, D] (pinned/registered host memory, or HBM). */
int cdlrm_ctx_bind_host_tables(cdlrm_ctx* ctx, float* const* host_rows);

/* Pin an existing host allocation so kernels can read/write it over PCIe; returns its device alias. */
int cdlrm_host_register(void* host_ptr, uint64_t bytes, void** device_alias);
int cdlrm_host_unregister(void* host_ptr);

/* Read (and clear) the device error word set by kernels on out-of-range input (CDLRM_ERANGE). */
int cdlrm_ctx_check_sync(cdlrm_ctx* ctx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-iteration path: Embedding_Table_Cache_Group.forward (model_no_ddp.py:149-212)
 * ------------------------------------------------------------------------------------------- */

/* Tag probe + slot translation + aux-row fill (model_no_ddp.py:163-187).
 *   idx        device int64 [T, n] (row stride ld_idx): the lookups of every table
 *   slots_out  device int32 [T, n] contiguous: cache_group_idxs (model_no_ddp.py:204); misses get
 *              aux slots P_k*ways + i in position order and aux row i is overwritten with the host
 *              row (:176-179).
 *   miss_pos   device int32 [T, n]: positions of table k's misses in order, first miss_count[k]
 *              entries of row k valid (victim_cache_entries, model_no_ddp.py:187)
 *   miss_count device int32 [T]: number of misses per table.
 *   aux_phase  which aux region receives the misses (0 unless the geometry has aux_phases = 2): aux slots are
 *              then P_k*ways + aux_phase*aux + i.                                                  */
int cdlrm_embbag_probe(cdlrm_ctx* ctx, const int64_t* idx, int64_t n, int64_t ld_idx,
                       int32_t* slots_out, int32_t* miss_pos, int32_t* miss_count, int32_t aux_phase,
                       void* stream);

/* Window-resident form of the same (round 2).  Tags only change at a refill (main_no_ddp.py:393-399), so the tag match
 * of model_no_ddp.py:163-174, the ordered miss numbering (:176-177) and the lookup of a miss in the window's victim list
 * need to run ONCE per look-ahead window, not once per iteration:
 *   cdlrm_window_resolve   idx [T, n] = the lookups of n/batch_len consecutive GLOBAL batches of batch_len lookups each;
 *                          every batch is cut into segments of seg_len lookups (= one rank's slice, ceil(B/world): the
 *                          last segment of a batch is shorter when world does not divide B, main_no_ddp.py:344, 388-391) and
 *                          the numbering restarts at every segment AND every batch boundary.  batch_len = 0: one run of
 *                          seg_len-long segments.  wslots int32 [T, n]: the slot of every lookup, misses numbered per segment in position
 *                          order as P_k*ways + i (aux phase 0); wsrc int32 [T, n]: for a miss, the position of its row in
 *                          the bound victim rows (cdlrm_ctx_bind_victims), -1 = read the host table; undefined for hits.
 *                          Call after the window's cdlrm_plan_commit + cdlrm_ctx_bind_victims.
 *   cdlrm_embbag_take      per iteration: slots_out int32 [T, n] = the batch's columns of wslots (misses moved to aux
 *                          region aux_phase) and the misses' rows copied into their aux rows -- exactly what
 *                          cdlrm_embbag_probe leaves behind for that batch.  wslots / wsrc / idx point at the batch's
 *                          first column; ld_w / ld_idx are the row strides. */
int cdlrm_window_resolve(cdlrm_ctx* ctx, const int64_t* idx, int64_t n, int64_t ld_idx, int64_t batch_len, int64_t seg_len,
                         int32_t* wslots, int32_t* wsrc, void* stream);
int cdlrm_embbag_take(cdlrm_ctx* ctx, const int64_t* idx, int64_t n, int64_t ld_idx, const int32_t* wslots,
                      const int32_t* wsrc, int64_t ld_w, int32_t* slots_out, int32_t aux_phase, void* stream);

/* --evict-victim-cache (main_no_ddp.py:96; victim_cache_entries, model_no_ddp.py:187 -- parsed / recorded by the reference
 * and never used: the update a MISSED row received is lost there).  Behind the step's embedding update: the trained aux row
 * of every miss of the batch is written to its host row and to its copy among the bound window victim rows; of several
 * misses of one index in the batch the LAST (position order) is written -- `emb_tables[k].weight[missing] = cache[k].weight[aux]`.
 *   idx [T, n] / slots [T, n] the batch's lookups and their slot ids as cdlrm_embbag_probe / _take left them (aux_phase: the
 *   aux region they used); wsrc [T, n] (row stride ld_w) the resolver's victim positions or NULL (the victim list is searched);
 *   work: device scratch of cdlrm_victim_writeback_work_bytes(T, n) bytes, 256-byte aligned. */
uint64_t cdlrm_victim_writeback_work_bytes(int32_t T, int64_t n);
int cdlrm_victim_writeback(cdlrm_ctx* ctx, const int64_t* idx, int64_t n, int64_t ld_idx, const int32_t* slots,
                           const int32_t* wsrc, int64_t ld_w, int32_t aux_phase, void* work, void* stream);

/* Fused multi-table sum-pool gather: nn.EmbeddingBag(mode="sum") forward on the cache rows for all
 * T tables in one launch (model_no_ddp.py:200-203).
 *   slots    device int32 [T, n]
 *   offsets  device int64 [T, n_bags] bag starts (row stride ld_off), or NULL for the Criteo layout
 *            offsets == arange (one index per bag, n_bags == n; data_loader_terabyte.py:83-87)
 *   out      device fp32: out[(b * ld_bag) + (t * ld_table) + c], c < D.  ld_bag = (T+1)*D and
 *            ld_table = D with out pointing at feature 1 writes straight into the [B, T+1, D]
 *            interaction operand (model_no_ddp.py:276).                                         */
int cdlrm_embbag_fwd(cdlrm_ctx* ctx, const int32_t* slots, const int64_t* offsets, int64_t n,
                     int64_t n_bags, int64_t ld_off, float* out, int64_t ld_bag, int64_t ld_table,
                     void* stream);

/* EmbeddingBag backward + sparse SGD on the cache rows, fused and atomics-free
 * (main_no_ddp.py:376, 409, 413): W[slot] -= lr * sum_{i: slot_i = slot} grad[bag(i)], repeated
 * slots summed in position order.  Also marks the touched rows for the table-agg merge.
 *   grad     device fp32, same addressing as `out` above (ld_bag / ld_table)
 *   work     device scratch, cdlrm_embbag_bwd_work_bytes(T, n) bytes
 *   touched  device uint8 [total cache rows] or NULL: set to 1 for every updated row of the cache proper (aux rows --
 *            transient copies of host rows, rewritten by every forward -- are never flagged)
 *            (replaces cache_group_idxs_window, main_no_ddp.py:417-423)                          */
uint64_t cdlrm_embbag_bwd_work_bytes(int32_t num_tables, int64_t n, int32_t dim);
/* The same in two halves: `prepare` (sort of the slot ids + run metadata into `work`) depends only on
 * the probe result, so a trainer issues it during the forward pass; `apply` consumes the gradient.
 * Every `apply` follows a `prepare` on the SAME work buffer, in stream order (prepare also empties the buffer's long-run list);
 * backward passes on different work buffers share no state and may be in flight together. */
int cdlrm_embbag_bwd_prepare(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, void* work, void* stream);
int cdlrm_embbag_bwd_apply(cdlrm_ctx* ctx, const int64_t* offsets, int64_t n, int64_t n_bags,
                           int64_t ld_off, const float* grad, int64_t ld_bag, int64_t ld_table, float lr,
                           void* work, uint8_t* touched, void* stream);
int cdlrm_embbag_bwd_sgd(cdlrm_ctx* ctx, const int32_t* slots, const int64_t* offsets, int64_t n,
                         int64_t n_bags, int64_t ld_off, const float* grad, int64_t ld_bag,
                         int64_t ld_table, float lr, void* work, uint8_t* touched, void* stream);
/* `apply` behind cdlrm_gather_interact_bwd_sgd (below): the rows of slots that the batch reads ONCE were updated by that launch
 * and their gradient rows never written; this call sums and applies the runs of two and more lookups (and sets the touched
 * flags of all of them).  Same arguments as cdlrm_embbag_bwd_apply; together the two calls leave the cache rows bit for bit as
 * cdlrm_gather_interact_bwd + cdlrm_embbag_bwd_apply do (nn.EmbeddingBag backward + optim.SGD, main_no_ddp.py:376, 409, 413). */
int cdlrm_embbag_bwd_apply_rest(cdlrm_ctx* ctx, const int64_t* offsets, int64_t n, int64_t n_bags,
                                int64_t ld_off, const float* grad, int64_t ld_bag, int64_t ld_table, float lr,
                                void* work, uint8_t* touched, void* stream);
/* The once-only flags `prepare` left in `work`: uint8 [T, n], 1 where the lookup's slot occurs once in the batch (host-side
 * pointer arithmetic, no launch). */
int cdlrm_embbag_bwd_once_flags(cdlrm_ctx* ctx, void* work, int64_t n, const uint8_t** once);
/* The slot sort of a whole look-ahead CHUNK at once (round 6).  Slot ids of a window do not change while it trains
 * (cdlrm_window_resolve), so the `prepare` of its batches need not wait for their steps: nb batches x T tables are sorted by
 * one set of launches, on the resolver's stream, into the caller-owned `sorted` (cdlrm_embbag_bwd_sorted_bytes; 256-byte
 * aligned).  wslots: the resolver's int32 [T, ld_w]; batch j's n slot ids of table t at wslots[t * ld_w + j * batch_len ..]
 * (a rank passes wslots + its first column and n = its slice).  Aux slots are phase-0 slots, as the resolver writes them.
 * A call sorts `count` of the chunk's batches from batch j0 on (a SLICE: the trainer spreads a chunk's sort over several steps);
 * the lists of the other batches are left alone.
 *   _sorted_views: batch j's sorted keys / run distances / once-only flags; table t's at + t * nb * n elements.
 *   _apply_sorted: cdlrm_embbag_bwd_apply (rest = 0) or _apply_rest (rest = 1) over those lists instead of `work`'s own sort:
 *     `work` lends its scratch only (a buffer no `prepare` ever touched must be zero-filled); aux_phase: the aux region the
 *     batch trains on (what cdlrm_embbag_take was given).  One lookup per bag and table.  Same rows, bit for bit. */
uint64_t cdlrm_embbag_bwd_sorted_bytes(int32_t num_tables, int32_t nb, int64_t n);
int cdlrm_embbag_bwd_prepare_window(cdlrm_ctx* ctx, const int32_t* wslots, int64_t ld_w, int64_t batch_len, int32_t nb,
                                    int64_t n, int32_t j0, int32_t count, void* sorted, void* stream);
int cdlrm_embbag_bwd_sorted_views(cdlrm_ctx* ctx, void* sorted, int32_t nb, int64_t n, int32_t j, const uint64_t** keys,
                                  const int32_t** meta, const uint8_t** once);
int cdlrm_embbag_bwd_apply_sorted(cdlrm_ctx* ctx, int64_t n, const float* grad, int64_t ld_bag, int64_t ld_table, float lr,
                                  void* work, const uint64_t* keys, const int32_t* meta, int64_t tstride, int32_t aux_phase,
                                  int32_t rest, uint8_t* touched, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Look-ahead window path: Prefetcher.process_batch_slice (cache_manager.py:28-46) and
 * CacheEmbeddings (main_no_ddp.py:148-209), split into plan (side stream, overlaps training of the
 * previous window) and commit (at the window boundary).
 * ------------------------------------------------------------------------------------------- */

/* Caller-owned plan buffers.  Capacities are in elements; a kernel that would exceed one sets the
 * device error word (CDLRM_ERANGE) instead of writing out of bounds. */
typedef struct {
    /* lookahead scan */
    uint64_t* bitmap;        /* [sum_k ceil(n_k/64)] zero on entry, zero again on return          */
    int64_t*  uniq;          /* [cap_uniq] sorted unique indices, table-major                      */
    int64_t*  uniq_off;      /* [T+1] start of table k in uniq                                     */
    int64_t   cap_uniq;
    /* insert plan */
    uint64_t* prot;          /* [sum_k P_k] ways hit by this window, per set (bit w)               */
    uint8_t*  hit;           /* [cap_uniq] 1 = already cached                                      */
    int32_t*  kept;          /* [cap_uniq] positions (into uniq) of the claimants, ascending       */
    int64_t*  kept_off;      /* [T+1]                                                              */
    uint8_t*  way;           /* [cap_uniq] way chosen by claimant m                                */
    uint8_t*  flags;         /* [cap_uniq] scratch for the compactions, 16-byte aligned            */
    int32_t*  winner;        /* [total cache rows] scratch, -1 on entry and on return              */
    int32_t*  win_claim;     /* [cap_win] claimant id m of winner w (ascending)                    */
    int64_t*  win_idx;       /* [cap_win] index being inserted                                     */
    int64_t*  win_row;       /* [cap_win] global cache row = row_base_k + P_k*way + set            */
    int64_t*  win_tag;       /* [cap_win] global tag element = tag_base_k + set*ways + way         */
    int64_t*  win_off;       /* [T+1]                                                              */
    int64_t   cap_win;
    float*    stage;         /* [cap_win, D] incoming rows; after commit: the rows they replaced   */
    int64_t*  ev_tag;        /* [cap_win] after commit: tag that was evicted, -1 if slot was empty */
} cdlrm_plan;

/* K1: sorted unique of every table's window indices (torch.unique, cache_manager.py:32) by bitmap +
 * popcount scan.  idx: device int64 [T, n] (row stride ld_idx).  Fills plan->uniq / uniq_off. */
int cdlrm_window_unique(cdlrm_ctx* ctx, const cdlrm_plan* plan, const int64_t* idx, int64_t n,
                        int64_t ld_idx, void* stream);
/* K1 streamed (cache_manager.py:85-110 collects a window batch by batch): _add folds one chunk [T, n] of the window
 * into the plan's bitmap, any number of times; _finish emits uniq / uniq_off for everything added since the last
 * finish and leaves the bitmap zero.  cdlrm_window_unique == one _add + _finish. */
int cdlrm_window_unique_add(cdlrm_ctx* ctx, const cdlrm_plan* plan, const int64_t* idx, int64_t n,
                            int64_t ld_idx, void* stream);
int cdlrm_window_unique_finish(cdlrm_ctx* ctx, const cdlrm_plan* plan, void* stream);

/* K2: tag probe of the unique list + full-set filter (main_no_ddp.py:155-180).
 * Fills hit / prot / kept / kept_off.  uniq and uniq_off must be valid (from cdlrm_window_unique or
 * written by the caller for the CacheEmbeddings drop-in entry). */
int cdlrm_plan_probe(cdlrm_ctx* ctx, const cdlrm_plan* plan, void* stream);

/* Copy uniq_off / kept_off / win_off ([T+1] each, any may be NULL) to host memory and wait. */
int cdlrm_plan_offsets_sync(cdlrm_ctx* ctx, const cdlrm_plan* plan, int64_t* uniq_off,
                            int64_t* kept_off, int64_t* win_off, void* stream);

/* K3: way choice + contested-slot resolution (main_no_ddp.py:171-185, 203-204).
 *   q  device fp32 [M_total, ways] Exp(1) draws in claimant order (parity mode: drawn by the torch
 *      CPU generator exactly as Categorical.sample() does), or NULL: counter-based Philox keyed by
 *      `seed` on the device (perf mode, not bit-comparable with the reference).
 * way = argmax_w((avail_w / sum avail) / q_w), first maximum; the claimant latest in ascending-index
 * order wins a contested (set, way).  Fills way / win_* . */
int cdlrm_plan_assign(cdlrm_ctx* ctx, const cdlrm_plan* plan, const float* q, uint64_t seed,
                      void* stream);

/* K5a: fetch the winners' rows into plan->stage.
 *   src_rows [T] host array of device-visible base pointers; by_position = 0: row = src[k][index]
 *   (host master tables, Embedding_Table_Group.fetch_unique_idx_slices, model_no_ddp.py:80-87);
 *   by_position = 1: row = src[k][position in uniq_k] (the `cached_entries_per_table` argument of
 *   CacheEmbeddings, main_no_ddp.py:205-206). */
int cdlrm_plan_fetch(cdlrm_ctx* ctx, const cdlrm_plan* plan, const float* const* src_rows,
                     int by_position, void* stream);

/* K4+K5b: at the window boundary swap stage <-> cache rows and write the tags
 * (main_no_ddp.py:190-206).  Afterwards stage[w] / ev_tag[w] hold the evicted row / tag. */
int cdlrm_plan_commit(cdlrm_ctx* ctx, const cdlrm_plan* plan, void* stream);

/* K14: eviction write-back (Prefetcher.eviction_manager, cache_manager.py:57-62):
 * host[k][ev_tag] = row, or (host + row) / 2 with average != 0.  dst_rows as src_rows above. */
int cdlrm_plan_writeback(cdlrm_ctx* ctx, const cdlrm_plan* plan, float* const* dst_rows,
                         int average, void* stream);

/* Window victims: the window's unique indices that stay OUTSIDE the cache after this plan commits (not cached
 * before, no way won).  The reference serves each of their lookups from the host table (aux rows,
 * model_no_ddp.py:176-179) -- the same host row every time while the window trains.  cdlrm_plan_victims (after
 * cdlrm_plan_assign, same stream) lists them and fetches their rows once into caller-owned HBM;
 * cdlrm_ctx_bind_victims (at the commit of that window) makes cdlrm_embbag_probe fill aux rows from there, falling
 * back to the host table for any index not in the list (beyond cap, or not from this window).  NULL unbinds. */
typedef struct {
    int32_t* pos;            /* [cap] positions into plan->uniq, ascending                                   */
    int64_t* idx;            /* [cap] the indices, ascending per table                                       */
    int64_t* off;            /* [T+2] start of table k; off[T] = entries listed (<= cap); off[T+1] = entries the    */
                             /*       window HAS (> cap: the list was cut -- a caller may come back with a larger one) */
    float*   rows;           /* [cap, D] their host rows                                                     */
    int64_t  cap;
} cdlrm_victims;
int cdlrm_plan_victims(cdlrm_ctx* ctx, const cdlrm_plan* plan, const cdlrm_victims* victims, void* stream);
/* The list only (pos / idx / off); the caller fills victims->rows itself (cdlrm_host_gather_rows + one DMA). */
int cdlrm_plan_victims_list(cdlrm_ctx* ctx, const cdlrm_plan* plan, const cdlrm_victims* victims, void* stream);

/* HOST function (no GPU work): dst[j, :] = tables[t(j)][idx[j], :] for j < off[T], t(j) = the table whose range
 * [off[t], off[t+1]) holds j; tables / idx / off / dst are host pointers (dst typically pinned), nthreads CPU threads.
 * The CPU-side counterpart of cdlrm_plan_fetch / the victims' row fetch -- what the reference's Prefetcher process does
 * in `emb_tables.fetch_unique_idx_slices` (model_no_ddp.py:80-87): gathered rows then reach HBM by ONE DMA copy, which
 * disturbs the training kernels far less than GPU waves reading host memory. */
int cdlrm_host_gather_rows(const float* const* tables, const int64_t* idx, const int64_t* off, int32_t T,
                           int32_t D, float* dst, int32_t nthreads);
int cdlrm_ctx_bind_victims(cdlrm_ctx* ctx, const cdlrm_victims* victims);

/* Generic row gather used by the drop-in process_batch_slice (rows = W_host[uniq],
 * model_no_ddp.py:84): out[i, :] = src[index[i], :]. */
int cdlrm_gather_rows(const float* src, const int64_t* index, int64_t count, int32_t dim,
                      float* out, void* stream);

/* Generic row scatter used by the drop-in Prefetcher.eviction_manager (cache_manager.py:57-62):
 * dst[index[i], :] = rows[i, :], or (dst + rows) / 2 with average != 0.  dst is device-visible.
 * average != 0 reads and writes dst in one pass and therefore needs DISTINCT indices; an eviction list in the
 * reference's format may repeat an index (one entry per claimant of an occupied slot, all carrying the same row,
 * SURVEY App. A): blend it first with cdlrm_blend_rows -- every entry computed from the OLD destination row, as
 * `W[idx] = (W[idx] + emb) / 2` (cache_manager.py:62) does -- then scatter the blended rows with average = 0. */
int cdlrm_scatter_rows(float* dst, const int64_t* index, const float* rows, int64_t count,
                       int32_t dim, int average, void* stream);
/* out[i, :] = (dst[index[i], :] + rows[i, :]) / 2   (the gather half of the averaging write-back) */
int cdlrm_blend_rows(const float* dst, const int64_t* index, const float* rows, int64_t count,
                     int32_t dim, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Quotient-remainder embedding bag, stand-alone operator (QREmbeddingBag.forward,
 * tricks/qr_embedding_bag.py:156-174; the reference never wires it into the cached path).
 *   q = (idx / collisions).long() as a float32 true division + truncation (quirk kept), r = idx % collisions;
 *   out[b] = sum_bag Wq[q] (op 0: * , 1: + , 2: concat) sum_bag Wr[r];  out is [n_bags, D] (2D for concat).
 *   eq_out / er_out (optional, [n_bags, D]) keep the pooled operands for the mult backward.
 *   err_word: device int32, set non-zero when a quotient falls outside Wq.
 * bwd accumulates DENSE gradients gWq [rows_q, D], gWr [collisions, D] (caller zeroes them).
 * ------------------------------------------------------------------------------------------- */
int cdlrm_qr_embbag_fwd(const int64_t* idx, const int64_t* offsets, int64_t n, int64_t n_bags,
                        const float* Wq, const float* Wr, int64_t rows_q, int32_t collisions, int32_t dim,
                        int32_t op, float* out, float* eq_out, float* er_out, int32_t* err_word, void* stream);
int cdlrm_qr_embbag_bwd(const int64_t* idx, const int64_t* offsets, int64_t n, int64_t n_bags,
                        const float* eq, const float* er, const float* grad_out, int64_t rows_q,
                        int32_t collisions, int32_t dim, int32_t op, float* gWq, float* gWr, void* stream);

/* Plain EmbeddingBag(mode="sum") over one stand-alone table of ANY width, forward and dense backward: the `embs` of the
 * mixed-dimension trick's PrEmbeddingBag (tricks/md_embedding_bag.py:60-78; its widths are powers of two down to 1).
 * Stand-alone operator like the QR bag: the reference never wires it into the cached path (main_no_ddp.py:612-621).
 *   out [n_bags, dim] = sum over the bag's indices of W[idx];  gW [rows, dim] += grad_out[bag] (float atomics).
 *   err_word: device int32, bit 0 set when an index lies outside [0, rows). */
int cdlrm_bag_fwd(const int64_t* idx, const int64_t* offsets, int64_t n, int64_t n_bags, const float* W, int64_t rows,
                  int32_t dim, float* out, int32_t* err_word, void* stream);
int cdlrm_bag_bwd(const int64_t* idx, const int64_t* offsets, int64_t n, int64_t n_bags, const float* grad_out,
                  int64_t rows, int32_t dim, float* gW, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Table aggregation (broadcast_and_aggregate, main_no_ddp.py:250-292)
 * ------------------------------------------------------------------------------------------- */

/* touched[row_base_t + slots[t, i]] = 1 for slots int32 [T, n]: turns gathered cache_group_idxs
 * (main_no_ddp.py:254-268) into the touched-row flags cdlrm_agg_compact consumes. */
int cdlrm_mark_rows(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, uint8_t* touched, void* stream);

/* Compact the touched-row flags into a sorted row list (torch.unique over the gathered
 * cache_group_idxs, :270) and clear them.  rows_out device int64 [cap]; count_out device int64 [1]. */
int cdlrm_agg_compact(cdlrm_ctx* ctx, uint8_t* touched, int64_t total_rows, int64_t* rows_out,
                      int64_t cap, int64_t* count_out, void* stream);
/* buf[i,:] = weight[rows[i],:] / scale   (:273-281)   /   weight[rows[i],:] = buf[i,:]   (:288-292)
 * for i < min(*count - first, cap): `rows` / `buf` point at entry `first` of a list whose total length is the device
 * word *count, so a merge can be cut into chunks (gather chunk i+1 while chunk i is reduced over xGMI) without the
 * host knowing more than an upper bound. */
int cdlrm_agg_gather(cdlrm_ctx* ctx, const int64_t* rows, const int64_t* count, float scale,
                     float* buf, int64_t cap, int64_t first, void* stream);
int cdlrm_agg_scatter(cdlrm_ctx* ctx, const int64_t* rows, const int64_t* count, const float* buf,
                      int64_t cap, int64_t first, void* stream);
/* Deadlines for the merge: a row only has to be merged before the first later step that USES it on any rank, and the look-ahead
 * window knows which rows the next batches use.
 * cdlrm_agg_mark_tier: tier[row] = value for every cache slot named by a [T, n] view (row pitch ld) of RESOLVED slot ids
 *   (cdlrm_window_resolve's wslots; aux slots are skipped).  Called class by class from the latest deadline to the earliest.
 * cdlrm_agg_split: stable counting sort of a merge's sorted row list (cdlrm_agg_compact; `count` entries, known to the host)
 *   by tier byte (values >= n_classes count as the last class): rows_out = the rows grouped by class, ascending inside a class
 *   -- identical on every rank --, class_off[n_classes + 1] (device) = where each class starts. */
int cdlrm_agg_mark_tier(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, int64_t ld, int32_t value, uint8_t* tier, void* stream);
int cdlrm_agg_split(cdlrm_ctx* ctx, const int64_t* rows, int64_t count, const uint8_t* tier, int32_t n_classes,
                    int64_t* rows_out, int64_t* class_off, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense model: DLRM_Net (model_no_ddp.py:215-316), loss (main_no_ddp.py:212-221), SGD (:375, 415)
 * ------------------------------------------------------------------------------------------- */

/* interact_features "dot" (model_no_ddp.py:272-293): feat fp32 [B, F, D] (feature 0 = bottom-MLP
 * output x); R[b] = [x_b, <f_i, f_j> for i in 0..F-1 for j in 0..i-1(+itself)] fp32 [B, ld_r].  Columns of the row
 * pitch beyond the row's width are left alone or, up to the next multiple of 4, set to 0 (rows leave as whole float4
 * words when ld_r % 4 == 0 and R is 16-byte aligned). */
int cdlrm_interact_fwd(const float* feat, int64_t B, int32_t F, int32_t D, int32_t itself,
                       float* R, int64_t ld_r, void* stream);
/* dfeat[b] = (G + G^T) feat_b with G the strictly-lower (or lower) triangle filled from dR, plus dR's
 * first D columns added to feature 0.  x_act != 0: feature 0 is the output of an activation (1 ReLU,
 * 2 sigmoid; the bottom MLP's last layer, model_no_ddp.py:262-266) and its gradient row is multiplied by
 * act'(feat[b,0,:]) here, so the bottom MLP's backward starts from the pre-activation gradient. */
int cdlrm_interact_bwd(const float* feat, const float* dR, int64_t ld_r, int64_t B, int32_t F,
                       int32_t D, int32_t itself, int32_t x_act, float* dfeat, void* stream);
/* Cached EmbeddingBag forward + interact_features "dot" in ONE launch (model_no_ddp.py:200-203 + :272-293), Criteo layout
 * (one index per bag: the bag's sum-pool is its cache row).  The rows the slot ids name are read straight into the
 * interaction -- the [B, T, D] block cdlrm_embbag_fwd would write and cdlrm_interact_fwd read back is never moved.
 *   slots  device int32 [T, n] (cdlrm_embbag_probe / cdlrm_embbag_take), sample b of table t at slots[t * n + b], n >= B
 *   x      device fp32 [B, D] rows of pitch ld_x: the bottom MLP's output (feature 0)
 *   R      as cdlrm_interact_fwd's, F = T + 1; whole float4 rows (ld_r % 4 == 0, 16-byte aligned)
 * Bit-identical to cdlrm_embbag_fwd + cdlrm_interact_fwd.  Shapes: cdlrm_gather_interact_supported() (D in 32 / 64 / 128 /
 * 256, 16 < T + 1 <= 32); anything else is refused -- the caller issues the two operators.  A pair of events armed with
 * cdlrm_ctx_time_next_gather times this launch; a completion event attached with cdlrm_event_attach_next completes with it
 * (as the launch's stop event; recorded behind the launch when the launch is being timed). */
int cdlrm_gather_interact_supported(cdlrm_ctx* ctx);
int cdlrm_gather_interact_fwd(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, const float* x, int64_t ld_x, int64_t B,
                              int32_t itself, float* R, int64_t ld_r, void* stream);
/* Its backward: cdlrm_interact_bwd with the rows read again from the cache (call it BEFORE the batch's embedding update:
 * cdlrm_embbag_bwd_apply rewrites them).  dfeat fp32 [B, T + 1, D] as cdlrm_interact_bwd's; carries an attached completion
 * event (cdlrm_event_attach_next) like cdlrm_interact_bwd. */
int cdlrm_gather_interact_bwd(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, const float* x, int64_t ld_x,
                              const float* dR, int64_t ld_r, int64_t B, int32_t itself, int32_t x_act, float* dfeat,
                              void* stream);
/* The same with the sparse SGD step of the once-only slots folded in (main_no_ddp.py:376 backward + :413 optimizer_embeds.step()
 * for those rows).  once: uint8 flags of the batch's lookups, table t's at once + t * ld_once (cdlrm_embbag_bwd_once_flags of
 * the batch's prepared work buffer, ld_once = n; or cdlrm_embbag_bwd_sorted_views, ld_once = nb * n) -- written by a sort that has
 * COMPLETED (the caller orders the streams); lr: the embedding learning rate.  A lookup whose slot no other lookup of the batch
 * shares gets W[slot] -= lr * g in this launch (one addend: no order to keep) and its gradient row in dfeat is left UNDEFINED
 * (not written, but for stray words of the last table's); every other lookup's gradient row is written as before.  Follow
 * with cdlrm_embbag_bwd_apply_rest / _apply_sorted(rest = 1) on the same sort. */
int cdlrm_gather_interact_bwd_sgd(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, const float* x, int64_t ld_x,
                                  const float* dR, int64_t ld_r, int64_t B, int32_t itself, int32_t x_act, float* dfeat,
                                  const uint8_t* once, int64_t ld_once, float lr, void* stream);
/* Linear + activation (create_mlp, model_no_ddp.py:244-270): Y = act(X W^T + b).
 * X [M, K] ld_x, W [N, K] row-major (nn.Linear.weight), Y [M, N] ld_y. act: 0 none, 1 ReLU, 2 sigmoid. */
/* CDLRM_GEMM_ALONE, or-ed into `act` of cdlrm_linear_fwd / cdlrm_linear_bwd: the caller's promise that no other GEMM runs beside
 * this launch (the training step's top-MLP forward and dgrad chain).  A scheduling hint: long batches whose 128x128 tiles fill
 * the chip then take the one-workgroup-per-CU kernel (csrc/gemm_wide.h), which must not share its CU.  Same fp32 fma arithmetic;
 * the contraction order inside a 16-deep group differs from the default kernel's, so results may differ in the last bits. */
#define CDLRM_GEMM_ALONE 0x100
int cdlrm_linear_fwd(const float* X, int64_t ld_x, const float* W, const float* bias, float* Y,
                     int64_t ld_y, int64_t M, int32_t N, int32_t K, int32_t act, void* stream);
/* Backward of the same layer.  act != 0: dY is the gradient w.r.t. the layer's OUTPUT and is overwritten
 * in place by dZ = dY * act'(Y); act == 0: dY already is dZ (Y may be NULL).
 * dX [M, K] ld_dx (may be NULL for the first layer), dW [N, K] (NULL: the weight gradient is taken later
 * by cdlrm_mlp_wgrad from the dZ left in dY), db [N] (may be NULL).
 * x_act != 0: X is itself the output of an activation (1 ReLU, 2 sigmoid) and dX is multiplied by act'(X)
 * in the GEMM epilogue, i.e. dX is then the dZ of the layer below (pass act = 0 there): the training
 * step chains the layers this way so no stand-alone activation-backward pass touches HBM.
 * work: device scratch of cdlrm_linear_bwd_work_bytes(M, N, K) bytes (split-M partial slabs of dW and of
 * db, summed in a fixed order: bitwise reproducible). */
uint64_t cdlrm_linear_bwd_work_bytes(int64_t M, int32_t N, int32_t K);
int cdlrm_linear_bwd(const float* X, int64_t ld_x, const float* W, const float* Y, int64_t ld_y,
                     float* dY, int64_t ld_dy, float* dX, int64_t ld_dx, float* dW, float* db,
                     int64_t M, int32_t N, int32_t K, int32_t act, int32_t x_act, void* work,
                     void* stream);

/* Weight and bias gradients of n_layers Linear layers from the pre-activation gradients dZ[i] [M, N[i]] that
 * cdlrm_linear_bwd(dW = NULL) left behind: dW[i] [N[i], K[i]] = dZ[i]^T X[i], db[i] [N[i]] (entries may be NULL)
 * = column sums of dZ[i].  The arrays are HOST arrays of device pointers / sizes.  Replaces the per-parameter
 * autograd accumulation of the reference's loss.backward() (main_no_ddp.py:409) for the MLP weights: at small
 * per-GPU batches all layers run as one grouped launch (+ one grouped reduction of the batch slabs).
 * work: device scratch of cdlrm_mlp_wgrad_work_bytes(...) bytes, 256-byte aligned. */
uint64_t cdlrm_mlp_wgrad_work_bytes(int32_t n_layers, int64_t M, const int32_t* N, const int32_t* K);
int cdlrm_mlp_wgrad(int32_t n_layers, const float* const* X, const int64_t* ld_x, const float* const* dZ,
                    const int64_t* ld_dz, float* const* dW, float* const* db, int64_t M, const int32_t* N,
                    const int32_t* K, void* work, void* stream);
/* cdlrm_mlp_wgrad followed by the dense SGD step of the same layers -- W[i] -= lr * dW[i], b[i] -= lr * db[i]
 * (optimizer_mlps.step(), main_no_ddp.py:415) -- inside the same launches (the slab reduction writes the gradient and steps
 * the parameter): for callers with nothing between the two, i.e. one rank (no gradient exchange, :412-414).  W[i] [N[i], K[i]]
 * and b[i] [N[i]] (b[i] may be NULL where db[i] is); gradients are still left in dW / db.  Same arithmetic as
 * cdlrm_mlp_wgrad + cdlrm_sgd_step. */
int cdlrm_mlp_wgrad_sgd(int32_t n_layers, const float* const* X, const int64_t* ld_x, const float* const* dZ,
                        const int64_t* ld_dz, float* const* dW, float* const* db, float* const* W, float* const* b,
                        float lr, int64_t M, const int32_t* N, const int32_t* K, void* work, void* stream);

/* BCELoss(mean) forward + backward on the sigmoid output (torch clamps log at -100):
 * loss_out device fp32 [65]: [0] = loss, [1..64] = partial sums (fixed-order, reproducible);
 * dZ[i] = (z - t) / (max((1 - z) z, 1e-12) * n), may be NULL.  sigmoid_bwd != 0: Z came out of a
 * sigmoid (sigmoid_top, main_no_ddp.py:358) and dZ is additionally multiplied by (1 - z) z, i.e. it is
 * the gradient w.r.t. the last layer's pre-activation. */
int cdlrm_bce_fwd_bwd(const float* Z, const float* target, int64_t n, float* loss_out, float* dZ,
                      int32_t sigmoid_bwd, void* stream);

/* The loss with all its arms (main_no_ddp.py:212-221, 364-372) and the --loss-threshold clamp of the prediction
 * (model_no_ddp.py:311-314): kind 0 BCELoss(mean), 1 MSELoss(mean), 2 weighted BCE (w0 / w1 = --loss-weights for
 * target 0 / 1, mean of w[t] * BCE(none)); 0 < threshold < 1: z = clamp(Z, threshold, 1 - threshold), gradient only
 * where Z lies inside.  loss_out (>= 3 floats): [0] = loss, [1] = number of samples with round(prediction) == target
 * (the train-accuracy count of main_no_ddp.py:431, round half to even as np.round), [2] = loss * n in fp32 (the term
 * the reference adds to its running loss, :433) -- so a trainer keeps its print statistics on the device and reads them
 * at print boundaries.  dZ (may be NULL) = dL/dZ, times (1 - Z) Z when sigmoid_bwd != 0;
 * Zc (may be NULL) = the clamped prediction the reference's DLRM_Net.forward returns. */
int cdlrm_loss_fwd_bwd(const float* Z, const float* target, int64_t n, int32_t kind, float w0, float w1,
                       float threshold, float* loss_out, float* dZ, float* Zc, int32_t sigmoid_bwd, void* stream);

/* Output head in one launch: last top-MLP layer (out_features 1 + sigmoid, main_no_ddp.py:358; w [K], bias [1] or
 * NULL), the loss above, and the layer's input gradient.  Y [B, K] (row pitch ldy) is the layer's input, produced
 * by activation x_act (0 none, 1 ReLU, 2 sigmoid).  Outputs: Z [B] = sigmoid(Y w + b), Zc [B] (may be NULL) clamped
 * prediction, dZ [B] = dL/d(pre-activation of the last layer), dY [B, K] (pitch lddy; may be NULL) = dZ w^T times
 * the derivative of x_act, loss_out[0..2] as for cdlrm_loss_fwd_bwd.  scratch: cdlrm_head_scratch_floats() floats,
 * zeroed once by the caller (per-workgroup partial sums of the loss + one arrival counter the kernel leaves zero).  finish != 0: loss_out is complete when the call's work on `stream` is;
 * finish == 0: the partial sums are left in `scratch` and cdlrm_head_finish(scratch, B, loss_out, acc, any stream ordered
 * behind this call) turns them into loss_out -- the training step runs it beside the backward, not in front of it.
 * acc (may be NULL): device float64 [2], acc[0] += correct predictions, acc[1] += loss * B of this batch -- the running sums
 * the reference keeps on the host between two print boundaries (main_no_ddp.py:427-433), kept on the device instead. */
int64_t cdlrm_head_scratch_floats(void);
int cdlrm_head_fwd_bwd(const float* Y, int64_t ldy, const float* w, const float* bias, const float* target,
                       int64_t B, int32_t K, int32_t kind, float w0, float w1, float threshold, int32_t x_act,
                       float* Z, float* Zc, float* dZ, float* dY, int64_t lddy, float* loss_out, float* scratch,
                       int32_t finish, void* stream);
int cdlrm_head_finish(const float* scratch, int64_t B, float* loss_out, double* acc, void* stream);

/* p -= lr * g over a flat fp32 buffer (optim.SGD without momentum, main_no_ddp.py:375, 415). */
int cdlrm_sgd_step(float* param, const float* grad, int64_t n, float lr, void* stream);

/* dX *= act'(X) element-wise over an [M, N] block (row pitches lddx / ldx; act 1 ReLU output, 2 sigmoid output): the
 * "cat" interaction (model_no_ddp.py:297-299) passes the top MLP's input gradient straight to the bottom MLP's
 * output, whose activation derivative no other kernel applies. */
int cdlrm_act_bwd(float* dX, int64_t lddx, const float* X, int64_t ldx, int64_t M, int32_t N, int32_t act, void* stream);

/* The same update over two ranges [off0, off0 + n0) and [off1, off1 + n1) of one flat buffer in one launch (one
 * sub-network's weights and its biases: the bottom and the top MLP are updated at different points of the step). */
int cdlrm_sgd_step2(float* param, const float* grad, int64_t off0, int64_t n0, int64_t off1, int64_t n1, float lr,
                    void* stream);

/* x /= divisor (aggregate_gradients: layer.weight.grad /= world_size, main_no_ddp.py:239, 244). */
int cdlrm_scale_div(float* x, int64_t n, float divisor, void* stream);

/* ---- synthetic input (cdlrm_amd/synth.py; the reference has no Criteo-shaped generator: dlrm_data_pytorch.py:763-805 is
 * uniform multi-hot) ---------------------------------------------------------------------------------------------------------
 * out[i] = index of lookup first + i of one table's infinite, counter-based lookup stream: Zipf-like ranks (exponent alpha;
 * <= 0: uniform) scattered over [0, n_rows).  A pure function of (key, position): the trainer's batches and the look-ahead's
 * second pass over the same indices (cache_manager.py:87-90) regenerate identical data independently, one launch per table. */
int cdlrm_synth_indices(int64_t* out, int64_t count, int64_t first, int64_t n_rows, double alpha, uint64_t key, void* stream);

/* ---- launch tapes -------------------------------------------------------------------------------------------------
 * A training step's call sequence (this library's entry points + event records / stream waits), recorded once per control
 * path by the host and re-issued by ONE call per step (the reference issues the same ops from Python every iteration,
 * main_no_ddp.py:404-415; at a per-rank batch of 1024 the interpreter alone costs more than the GPU work).  Calls are
 * stored as (function, integer-class arguments, float arguments); an integer argument may be a "cell" -- a slot of the
 * tape's cell array that the host patches before a replay (the batch's tensors).  Every entry point a tape may hold is
 * registered in csrc/tape.hip with its true type and called through a pointer of that type (cdlrm_tape_add refuses
 * anything else); cdlrm_tape_selftest() checks the argument unpacking. */
typedef struct cdlrm_tape cdlrm_tape;
cdlrm_tape* cdlrm_tape_create(int32_t n_cells);
void cdlrm_tape_destroy(cdlrm_tape* t);
/* fn: any entry point of this header with scalar arguments (<= 24 integer-class, <= 8 float); iargs/fargs in parameter
 * order per class; cell[i] >= 0 takes iargs[i] from that cell at replay time. */
int cdlrm_tape_add(cdlrm_tape* t, void* fn, int32_t n_int, const int64_t* iargs, const int32_t* cell, int32_t n_flt,
                   const float* fargs);
int64_t* cdlrm_tape_cells(cdlrm_tape* t);
int64_t cdlrm_tape_length(cdlrm_tape* t);
/* position, in fn's parameter list, of the stream the call issues on (the library's convention is "last parameter"; the
 * event / stream calls below differ); -1: the call issues nothing itself; -2: fn is not a registered tape entry point */
int32_t cdlrm_tape_stream_arg(void* fn);
/* development (tools/host_time.py): op k's entry-point name (NULL: no such op) and, in out[5], its lane, the replays timed,
 * the host nanoseconds spent inside the call and waiting for another lane's op, the longest single call; the clocks run under cdlrm_debug_set(3, 1) */
const char* cdlrm_tape_op_info(cdlrm_tape* t, int64_t k, int64_t* out);
int cdlrm_tape_replay(cdlrm_tape* t);      /* first non-zero return code of a replayed call, else 0 */
/* Multi-lane replay: at short per-rank batches the HOST thread that issues a step's ~45 runtime calls, not the GPU, sets the
 * step time.  lane[k] = 1 .. 3 hands op k to that helper thread of the process (one per side queue), lane[k] = 0 stays with
 * the thread that calls cdlrm_tape_replay (the training queue's calls); dep[k] >= 0 = the tape index of an earlier op of
 * ANOTHER lane that has to be issued first -- the host-side order of the record / wait calls on one event.
 * n = cdlrm_tape_length. */
int cdlrm_tape_set_lanes(cdlrm_tape* t, const int32_t* lane, const int32_t* dep, int64_t n);
int cdlrm_tape_selftest(void);
/* Self-test hooks (no GPU involved; used by cdlrm_tape_selftest and tests/test_host_logic.py): a call with interleaved
 * float / int32 / pointer / int64 parameters that adds its arguments up, and a logging call (spins `spin` iterations, then
 * appends `tag` to a process-wide log that cdlrm_tape_probe_log_take drains; returns the number of entries logged). */
int cdlrm_tape_probe(float f0, int64_t a0, float f1, int32_t a1, void* a2, int64_t a3, float f2, int32_t a4, int64_t a5,
                     int64_t a6, void* a7, int32_t a8, float f3, int64_t a9);
int cdlrm_tape_probe_log(int64_t tag, int64_t spin);
int64_t cdlrm_tape_probe_log_take(int64_t* out, int64_t cap);
/* hipEventRecord / hipStreamWaitEvent as tape-able entry points (raw hipEvent_t / hipStream_t handles) */
int cdlrm_event_record(void* event, void* stream);
int cdlrm_stream_wait_event(void* stream, void* event);
/* `event` completes with the next kernel the calling thread launches on `stream` through cdlrm_linear_bwd,
 * cdlrm_interact_bwd, cdlrm_gather_interact_bwd or cdlrm_gather_interact_fwd -- attached to that launch as its stop event instead of recorded behind it: a record is a marker packet
 * and a 6-8 us bubble on the training queue, an attached event is free.  If the call cannot attach it (several launches, a
 * kernel path without the plumbing) it records the event behind its launches: same guarantees either way. */
int cdlrm_event_attach_next(void* event, void* stream);
/* a stream at an explicit priority (lower = more urgent, clamped into the device's range): the look-ahead plan's stream is
 * created at the least urgent level so that the training step goes first wherever the two compete for CUs */
void* cdlrm_stream_create(int32_t priority);
int cdlrm_stream_destroy(void* stream);
/* events of the library's own (handles exist from creation on); elapsed time in microseconds, waits for `stop` */
void* cdlrm_event_create(int32_t timing);
int cdlrm_event_destroy(void* event);
int cdlrm_event_elapsed_us(void* start, void* stop, float* us);
/* Measurement: the NEXT cdlrm_embbag_fwd (or cdlrm_gather_interact_fwd) on this context leaves its start / stop timestamps in the two (timing) events --
 * attached to the launch itself (hipExtLaunchKernel), no event records around it -- so bench.py can price the gather
 * (the roofline kernel) live, per launch.  A timed launch costs its queue ~7 us (completion signal); the elapsed time reads
 * 0.5-2 us above the profiler's duration of the same kernel. */
int cdlrm_ctx_time_next_gather(cdlrm_ctx* ctx, void* start_event, void* stop_event);

#ifdef __cplusplus
}
#endif
#endif /* CDLRM_HIP_H */
