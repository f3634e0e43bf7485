// Per-iteration embedding path on gfx950: tag probe (K6), fused multi-table sum-pool gather (K7, the
// HBM-roofline kernel) and atomics-free backward + sparse SGD (K8).
//
// Data layout in HBM: one flat fp32 weight buffer [sum_k rows_k, D] (row = 4*D bytes,
// 16-byte lanes), one flat int64 tag buffer, tags of a set contiguous (ways*8 bytes = one 128-B line
// at 16 ways).  Every kernel covers all T tables in one launch: blockIdx.y = table.
#include "common.h"
#include <hip/hip_ext.h>

// ---------------------------------------------------------------------------------------------
// K6a: tag probe.  LPL lanes cooperate on one lookup: lane g reads ways g, g+LPL, ... (8-byte tags,
// consecutive lanes -> consecutive tags -> one coalesced segment per set).
// model_no_ddp.py:166-174
// ---------------------------------------------------------------------------------------------
template <int LPL, int PU>
__global__ void __launch_bounds__(256) k_probe(const TableDesc* __restrict__ tab, int ways,
                                               const int64_t* __restrict__ tags,
                                               const int64_t* __restrict__ idx, int64_t n, int64_t ld_idx,
                                               int32_t* __restrict__ slots, int* err) {
    // PU lookups per lane group and pass, phased (all ids, then all tag words, then the reductions): a group's set is one
    // random 128-B line, so what bounds a LONG probe (a window chunk: millions of lookups) is the number of lines in flight
    // per wave; a single batch has fewer lookups than the grid has lane groups and uses PU = 1
    const int t = blockIdx.y;
    const TableDesc d = tab[t];
    const int g = threadIdx.x % LPL;
    const int64_t grp0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPL;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / LPL;
    const int64_t n_round = cdiv_dev(n, ngrp * PU) * ngrp * PU;
    for (int64_t i0 = grp0; i0 < n_round; i0 += ngrp * PU) {
        int64_t v[PU], set[PU];
        bool valid[PU], bad[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int64_t i = i0 + u * ngrp;
            valid[u] = i < n;
            v[u] = idx[(int64_t)t * ld_idx + min(i, n - 1)];
        }
        int found[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            bad[u] = valid[u] && (v[u] < 0 || v[u] >= d.n_rows);
            if (bad[u] || !valid[u]) v[u] = 0;
            set[u] = mod_sets(v[u], d.P);
            found[u] = 0x7fffffff;
        }
        if (LPL >= 16 || ways <= LPL) {         // one tag word per lane (the common geometry): loads of the PU sets back to back
            int64_t tw[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) tw[u] = g < ways ? tags[d.tag_base + set[u] * ways + g] : -2;
#pragma unroll
            for (int u = 0; u < PU; ++u)
                if (g < ways && tw[u] == v[u]) found[u] = g;
            if (ways > LPL) {                   // more ways than lanes (ways > 16): the remaining words
#pragma unroll
                for (int u = 0; u < PU; ++u)
                    for (int w = g + LPL; w < ways; w += LPL)
                        if (tags[d.tag_base + set[u] * ways + w] == v[u]) found[u] = min(found[u], w);
            }
        } else {
#pragma unroll
            for (int u = 0; u < PU; ++u)
                for (int w = g; w < ways; w += LPL)
                    if (tags[d.tag_base + set[u] * ways + w] == v[u]) found[u] = min(found[u], w);
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
#pragma unroll
            for (int m = LPL >> 1; m >= 1; m >>= 1) found[u] = min(found[u], __shfl_xor(found[u], m, LPL));
            if (valid[u] && g == 0) {
                slots[(int64_t)t * n + i0 + u * ngrp] = (found[u] == 0x7fffffff) ? -1 : (int32_t)(d.P * found[u] + set[u]);
                if (bad[u]) atomicOr(err, 1);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K6b: ordered miss resolution, one workgroup per table: the i-th miss (in position order) gets aux
// slot P*ways + i (model_no_ddp.py:176-177).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_resolve(const TableDesc* __restrict__ tab, int ways, int aux, int aux_first,
                                                  int32_t* __restrict__ slots, int64_t n,
                                                  int32_t* __restrict__ miss_pos, int32_t* __restrict__ miss_count,
                                                  int* err) {
    __shared__ int smem[32];
    const int t = blockIdx.x;
    const TableDesc d = tab[t];
    int32_t* row = slots + (int64_t)t * n;
    int32_t* mp = miss_pos + (int64_t)t * n;
    int running = 0;
    for (int64_t base = 0; base < n; base += blockDim.x) {
        const int64_t i = base + threadIdx.x;
        const int miss = (i < n && row[i] < 0) ? 1 : 0;
        int total;
        const int ex = block_excl_scan(miss, smem, &total);
        if (miss) {
            const int r = running + ex;
            if (r < aux) {
                row[i] = (int32_t)(d.P * ways + aux_first + r);
                mp[r] = (int32_t)i;
            } else {
                row[i] = (int32_t)(d.P * ways + aux_first);   // keep addresses legal; the call reports the error
                atomicOr(err, 2);
            }
        }
        running += total;
    }
    if (threadIdx.x == 0) miss_count[t] = min(running, aux);
}

// ---------------------------------------------------------------------------------------------
// K6c: aux-row fill: cache.weight[aux_i] = W_host[idx_miss_i]  (model_no_ddp.py:179): from the window's
// HBM-resident victim rows when bound (VICT: binary search of the index in the table's sorted victim list), else
// zero-copy reads of the pinned host table over PCIe; 16 bytes per lane.
// ---------------------------------------------------------------------------------------------
template <int FILL_U, bool VICT>
__global__ void __launch_bounds__(256) k_fill_aux(const TableDesc* __restrict__ tab, int T, int ways, int aux_first,
                                                  int D4,
                                                  float4* __restrict__ weight, float* const* __restrict__ host_rows,
                                                  const int64_t* __restrict__ idx, int64_t n, int64_t ld_idx,
                                                  const int32_t* __restrict__ miss_pos,
                                                  const int32_t* __restrict__ miss_count,
                                                  const int64_t* __restrict__ v_idx, const int64_t* __restrict__ v_off,
                                                  const float* __restrict__ v_rows) {
    // PCIe-bound (without victims): what matters is the number of host reads in flight (the link has a bounded tag pool), not the
    // number of waves -- and every read beyond what the link can carry sits in the L2/fabric request queues in
    // front of the HBM traffic of whatever runs beside this kernel (measured: concurrent GEMMs 2-3x slower under a
    // 200-workgroup fill).  So: ONE small grid over the misses of all tables (most tables have none), FILL_U
    // independent 16-B reads per lane.
    __shared__ int64_t first[65];           // first[t] = 16-B elements of the tables before t
    __shared__ int64_t aux_base[64];        // first aux row of table t
    __shared__ uint64_t hrow[64];           // host table of t (address)
    __shared__ int64_t voff[65];            // victim list of table t: v_idx[voff[t] .. voff[t+1])
    __shared__ int search_steps;
    if (VICT) {
        if (threadIdx.x <= T) voff[threadIdx.x] = v_off[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) {
            int64_t longest = 0;
            for (int t = 0; t < T; ++t) longest = max(longest, voff[t + 1] - voff[t]);
            int steps = 0;
            while (((int64_t)1 << steps) <= longest) ++steps;       // lower_bound over `longest` entries
            search_steps = steps;
        }
    }
    if (threadIdx.x == 0) {
        int64_t acc = 0;
        for (int t = 0; t < T; ++t) {
            first[t] = acc;
            acc += (int64_t)miss_count[t] * D4;
        }
        first[T] = acc;
    }
    if (threadIdx.x < T) {
        const TableDesc d = tab[threadIdx.x];
        aux_base[threadIdx.x] = d.row_base + d.P * ways + aux_first;
        hrow[threadIdx.x] = (uint64_t)(uintptr_t)host_rows[threadIdx.x];
    }
    __syncthreads();
    const int64_t total = first[T];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += stride * FILL_U) {
        // PHASED on purpose: vmcnt is an in-order counter, so a dependent miss_pos -> idx -> host-row chain written
        // per u would make chain u+1 wait for the slow host read of chain u (one PCIe round trip per read,
        // measured: 5 GB/s).  All positions first, then all ids, then all host reads back to back.
        int t[FILL_U], r[FILL_U], c[FILL_U], pos[FILL_U];
        int64_t id[FILL_U];
#pragma unroll
        for (int u = 0; u < FILL_U; ++u) {
            const int64_t e = min(e0 + u * stride, total - 1);
            int tt = 0;
            while (e >= first[tt + 1]) ++tt;            // LDS only
            const int64_t el = e - first[tt];
            t[u] = tt; r[u] = (int)(el / D4); c[u] = (int)(el % D4);
        }
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) f32x4* host_ptr;      // global, not flat, loads
        host_ptr src[FILL_U];
        f32x4 v[FILL_U];
        int64_t dst[FILL_U];
#pragma unroll
        for (int u = 0; u < FILL_U; ++u) {
            src[u] = (host_ptr)hrow[t[u]];
            dst[u] = (aux_base[t[u]] + r[u]) * D4 + c[u];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < FILL_U; ++u) pos[u] = miss_pos[(int64_t)t[u] * n + r[u]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < FILL_U; ++u) id[u] = idx[(int64_t)t[u] * ld_idx + pos[u]];
        __builtin_amdgcn_sched_barrier(0);
        int64_t at[FILL_U];
#pragma unroll
        for (int u = 0; u < FILL_U; ++u) at[u] = id[u] * D4 + c[u];
        if (VICT) {
            // lower_bound of id in the table's victim list, all FILL_U searches in lock step (loads of one step in
            // flight together); lanes of one row search redundantly (same addresses: one request per wave)
            int64_t lo[FILL_U], hi[FILL_U];
#pragma unroll
            for (int u = 0; u < FILL_U; ++u) { lo[u] = voff[t[u]]; hi[u] = voff[t[u] + 1]; }
            const int steps = search_steps;
            for (int it = 0; it < steps; ++it) {
                int64_t probe[FILL_U];
#pragma unroll
                for (int u = 0; u < FILL_U; ++u) probe[u] = v_idx[lo[u] < hi[u] ? (lo[u] + hi[u]) >> 1 : lo[u] - (lo[u] > 0)];
#pragma unroll
                for (int u = 0; u < FILL_U; ++u) {
                    const int64_t mid = (lo[u] + hi[u]) >> 1;
                    if (lo[u] < hi[u]) {
                        if (probe[u] < id[u]) lo[u] = mid + 1; else hi[u] = mid;
                    }
                }
            }
            int64_t found[FILL_U];
#pragma unroll
            for (int u = 0; u < FILL_U; ++u) found[u] = v_idx[lo[u] < voff[t[u] + 1] ? lo[u] : voff[0]];
#pragma unroll
            for (int u = 0; u < FILL_U; ++u) {
                if (lo[u] < voff[t[u] + 1] && found[u] == id[u]) {
                    src[u] = (host_ptr)(uintptr_t)v_rows;
                    at[u] = lo[u] * D4 + c[u];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < FILL_U; ++u) v[u] = src[u][at[u]];
        __builtin_amdgcn_sched_barrier(0);
        // unconditional stores (elements past the end were clamped to the last one: same value, same address) --
        // a condition here lets the compiler sink each host read into its store's block, serialising them again
#pragma unroll
        for (int u = 0; u < FILL_U; ++u) *reinterpret_cast<f32x4*>(weight + dst[u]) = v[u];
    }
}

// ---------------------------------------------------------------------------------------------
// K7: fused multi-table EmbeddingBag(sum) forward.  LPR lanes x 16 B cover one row chunk; a wave
// holds 64/LPR bags per step and UNROLL steps in flight (independent slot -> row -> store chains).
// Algorithmic bytes per lookup (Criteo layout): 4D row read + 4D output write + 8 index + 8 offset.
// ---------------------------------------------------------------------------------------------
// Persistent grid: a fixed number of workgroups walks (table, bag-chunk) work items; the slot ids of the NEXT item are loaded
// before the current item's rows, so the slot -> row dependency costs one memory latency per item instead of two,
// and there is no per-block launch ramp.  Loads are unpredicated (tail bags clamp to the last bag), only the
// stores are masked.
typedef float v4f __attribute__((ext_vector_type(4)));
template <int LPR, int U>
__global__ void __launch_bounds__(256) k_embbag_fwd_arange_p(const TableDesc* __restrict__ tab, int T, int D4,
                                                             const float4* __restrict__ weight,
                                                             const int32_t* __restrict__ slots, int64_t n,
                                                             float* __restrict__ out, int64_t ld_bag, int64_t ld_table,
                                                             int nt) {
    const int c = threadIdx.x % LPR;
    const int gpb = blockDim.x / LPR;
    const int gid = threadIdx.x / LPR;
    const int64_t chunks = cdiv_dev(n, (int64_t)gpb * U);
    const int64_t total = chunks * T;
    int64_t w = blockIdx.x;
    if (w >= total) return;
    int32_t s[U];
    {
        const int t = (int)(w / chunks);
        const int64_t b0 = ((w % chunks) * gpb + gid) * U;
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = slots[(int64_t)t * n + min(b0 + u, n - 1)];
    }
    while (true) {
        const int t = (int)(w / chunks);
        const int64_t b0 = ((w % chunks) * gpb + gid) * U;
        const int64_t wn = w + gridDim.x;
        int32_t sn[U];
        if (wn < total) {
            const int tn = (int)(wn / chunks);
            const int64_t bn = ((wn % chunks) * gpb + gid) * U;
#pragma unroll
            for (int u = 0; u < U; ++u) sn[u] = slots[(int64_t)tn * n + min(bn + u, n - 1)];
        }
        const int64_t row_base = tab[t].row_base;
        float* o = out + (int64_t)t * ld_table;
        for (int cc = c; cc < D4; cc += LPR) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = weight[(row_base + s[u]) * D4 + cc];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (b0 + u < n) {
                    float* p = o + (b0 + u) * ld_bag + cc * 4;
                    if (nt) {
                        v4f x = {v[u].x, v[u].y, v[u].z, v[u].w};
                        __builtin_nontemporal_store(x, reinterpret_cast<v4f*>(p));
                    } else {
                        *reinterpret_cast<float4*>(p) = v[u];
                    }
                }
        }
        if (wn >= total) break;
        w = wn;
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = sn[u];
    }
}

// general bags (multi-hot): sum in lookup order
template <int LPR>
__global__ void __launch_bounds__(256) k_embbag_fwd_bags(const TableDesc* __restrict__ tab, int D4,
                                                         const float4* __restrict__ weight,
                                                         const int32_t* __restrict__ slots,
                                                         const int64_t* __restrict__ offsets, int64_t n,
                                                         int64_t n_bags, int64_t ld_off, float* __restrict__ out,
                                                         int64_t ld_bag, int64_t ld_table) {
    const int t = blockIdx.y;
    const int64_t row_base = tab[t].row_base;
    const int c = threadIdx.x % LPR;
    const int gpb = blockDim.x / LPR;
    const int gid = threadIdx.x / LPR;
    const int32_t* sl = slots + (int64_t)t * n;
    const int64_t* off = offsets + (int64_t)t * ld_off;
    float* o = out + (int64_t)t * ld_table;
    for (int64_t b = (int64_t)blockIdx.x * gpb + gid; b < n_bags; b += (int64_t)gridDim.x * gpb) {
        const int64_t lo = off[b];
        const int64_t hi = (b + 1 < n_bags) ? off[b + 1] : n;
        for (int cc = c; cc < D4; cc += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t i = lo; i < hi; ++i) {
                const float4 v = weight[(row_base + sl[i]) * D4 + cc];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            *reinterpret_cast<float4*>(o + b * ld_bag + cc * 4) = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int lanes_per_row(int D4) { int l = pow2ceil(D4); return l > 64 ? 64 : (l < 4 ? 4 : l); }

#define DISPATCH_LPR(lpr, CALL)                 \
    switch (lpr) {                              \
        case 4: { CALL(4); break; }             \
        case 8: { CALL(8); break; }             \
        case 16: { CALL(16); break; }           \
        case 32: { CALL(32); break; }           \
        default: { CALL(64); break; }           \
    }

extern "C" int cdlrm_embbag_probe(cdlrm_ctx* ctx, const int64_t* idx, int64_t n, int64_t ld_idx,
                                  int32_t* slots_out, int32_t* miss_pos, int32_t* miss_count, int32_t aux_phase,
                                  void* stream) {
    CDLRM_CLEAR_STALE();
    CDLRM_REQUIRE(ctx && idx && slots_out && miss_pos && miss_count, "null argument");
    CDLRM_REQUIRE(aux_phase >= 0 && aux_phase < ctx->aux_phases, "aux_phase outside the geometry's aux_phases");
    const int aux_first = aux_phase * ctx->aux;
    CDLRM_REQUIRE(ctx->tags && ctx->weight, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(n >= 0 && ld_idx >= n && n < ((int64_t)1 << 31), "bad n / ld_idx");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        CDLRM_HIP_CHECK(hipMemsetAsync(miss_count, 0, sizeof(int32_t) * ctx->T, s));
        return 0;
    }
    int lpl = pow2ceil(ctx->ways);
    if (lpl > 16) lpl = 16;
    const int64_t groups_per_block = 256 / lpl;
    int64_t gx = cdiv(n, groups_per_block);
    if (gx > 4096) gx = 4096;
    dim3 grid((unsigned)gx, (unsigned)ctx->T);
#define PROBE_CALL(L) hipLaunchKernelGGL((k_probe<L, 1>), grid, dim3(256), 0, s, ctx->d_tab, ctx->ways, ctx->tags, idx, n, ld_idx, slots_out, ctx->d_err)
    switch (lpl) {
        case 1: PROBE_CALL(1); break;
        case 2: PROBE_CALL(2); break;
        case 4: PROBE_CALL(4); break;
        case 8: PROBE_CALL(8); break;
        default: PROBE_CALL(16); break;
    }
#undef PROBE_CALL
    hipLaunchKernelGGL(k_resolve, dim3(ctx->T), dim3(1024), 0, s, ctx->d_tab, ctx->ways, ctx->aux, aux_first, slots_out, n,
                       miss_pos, miss_count, ctx->d_err);
    if (ctx->aux > 0) {
        CDLRM_REQUIRE(ctx->h_host_rows[0] != nullptr, "cdlrm_ctx_bind_host_tables first");
        const int D4 = ctx->D / 4;
        CDLRM_REQUIRE(ctx->T <= 64, "more than 64 tables");
        // grid x reads-per-lane x 4 KB per workgroup-read = bytes in flight.  Measured on the MI355X box (random 512-B
        // rows of a 96 GB pinned table): a host read takes ~17 us round trip, so the ~50 GB/s link needs ~0.5-1 MB in
        // flight; 32 workgroups x 4 reads is the knee (16x8: 1.04 ms/step, 32x4: 0.96, 64x4: 0.98)
        const int fill_grid = 32, fill_u = 4;
        int64_t fx = cdiv((int64_t)(n < ctx->aux ? n : ctx->aux) * D4 * ctx->T, 256);
        const bool vict = ctx->vict_idx != nullptr;
        // with the window's victim rows resident in HBM the fill is a latency-bound search + HBM copy: wide grid
        const int64_t cap_grid = vict ? 512 : fill_grid;
        if (fx > cap_grid) fx = cap_grid;
#define FILL_CALL(U, V)                                                                                                \
    hipLaunchKernelGGL((k_fill_aux<U, V>), dim3((unsigned)fx), dim3(256), 0, s, ctx->d_tab, ctx->T, ctx->ways,         \
                       aux_first, D4, reinterpret_cast<float4*>(ctx->weight), ctx->d_host_rows, idx, n, ld_idx,        \
                       miss_pos, miss_count, ctx->vict_idx, ctx->vict_off, ctx->vict_rows)
        if (vict) FILL_CALL(4, true);
        else if (fill_u >= 8) FILL_CALL(8, false);
        else if (fill_u >= 4) FILL_CALL(4, false);
        else if (fill_u >= 2) FILL_CALL(2, false);
        else FILL_CALL(1, false);
#undef FILL_CALL
    }
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Window-resident probe.  Tags only change at a refill (main_no_ddp.py:393-399), so every lookup of a look-ahead window
// resolves to the same slot whenever it is probed while that window trains: the tag match, the ordered miss numbering
// and the search of a miss in the window's victim list are done ONCE per window (in chunks of batches, ahead of the
// training position) instead of once per iteration; the per-iteration work that remains is k_take below -- copy the
// batch's slot ids and the rows of its misses.  Results are identical to cdlrm_embbag_probe on each batch (same slot
// ids, same aux rows): tests/test_hip_kernels.py::test_window_resolve_take_equals_probe.
// ---------------------------------------------------------------------------------------------
// ordered miss numbering per SEGMENT (= one batch of one rank: seg_len lookups): the i-th miss of a segment, in
// position order, gets aux slot P*ways + i (phase 0; k_take adds the aux phase)
__global__ void __launch_bounds__(1024) k_resolve_seg(const TableDesc* __restrict__ tab, int ways, int aux,
                                                      int32_t* __restrict__ wslots, int64_t n, int64_t batch_len,
                                                      int64_t seg_len, int64_t segs_per_batch, int* err) {
    __shared__ int smem[32];
    const int t = blockIdx.y;
    const TableDesc d = tab[t];
    // segment = (batch b, rank r): starts at b * batch_len + r * seg_len and ends with the rank's slice or the batch
    const int64_t b = (int64_t)blockIdx.x / segs_per_batch, r = (int64_t)blockIdx.x % segs_per_batch;
    const int64_t s0 = b * batch_len + r * seg_len;
    const int64_t cnt = min(min(seg_len, batch_len - r * seg_len), n - s0);
    int32_t* row = wslots + (int64_t)t * n + s0;
    int running = 0;
    for (int64_t base = 0; base < cnt; base += blockDim.x) {
        const int64_t i = base + threadIdx.x;
        const int miss = (i < cnt && row[i] < 0) ? 1 : 0;
        int total;
        const int ex = block_excl_scan(miss, smem, &total);
        if (miss) {
            const int r = running + ex;
            if (r < aux) {
                row[i] = (int32_t)(d.P * ways + r);
            } else {
                row[i] = (int32_t)(d.P * ways);     // keep addresses legal; the call reports the error
                atomicOr(err, 2);
            }
        }
        running += total;
    }
}

// where a miss's row comes from: its position in the window's victim rows (sorted list per table: binary search), or
// -1 = not listed (the host table)
__global__ void __launch_bounds__(256) k_victim_pos(const TableDesc* __restrict__ tab, int T, int ways,
                                                    const int64_t* __restrict__ idx, int64_t n, int64_t ld_idx,
                                                    const int32_t* __restrict__ wslots, int32_t* __restrict__ wsrc,
                                                    const int64_t* __restrict__ v_idx, const int64_t* __restrict__ v_off) {
    const int t = blockIdx.y;
    const int64_t first_aux = tab[t].P * ways;
    const int64_t lo0 = v_idx ? v_off[t] : 0, hi0 = v_idx ? v_off[t + 1] : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (wslots[(int64_t)t * n + i] < first_aux) continue;
        const int64_t id = idx[(int64_t)t * ld_idx + i];
        int64_t lo = lo0, hi = hi0;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (v_idx[mid] < id) lo = mid + 1; else hi = mid;
        }
        wsrc[(int64_t)t * n + i] = (lo < hi0 && v_idx[lo] == id) ? (int32_t)lo : -1;
    }
}

extern "C" int cdlrm_window_resolve(cdlrm_ctx* ctx, const int64_t* idx, int64_t n, int64_t ld_idx, int64_t batch_len,
                                    int64_t seg_len, int32_t* wslots, int32_t* wsrc, void* stream) {
    CDLRM_CLEAR_STALE();
    CDLRM_REQUIRE(ctx && idx && wslots && wsrc, "null argument");
    CDLRM_REQUIRE(ctx->tags && ctx->weight, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(n >= 1 && ld_idx >= n && n < ((int64_t)1 << 31) && seg_len >= 1 && batch_len >= 0, "bad n / ld_idx / seg_len");
    if (batch_len == 0) batch_len = cdiv(n, seg_len) * seg_len;     // one run of seg_len-long segments
    CDLRM_REQUIRE(ctx->vict_idx == nullptr || ctx->total_rows < ((int64_t)1 << 31), "victim positions must fit 31 bits");
    hipStream_t s = (hipStream_t)stream;
    int lpl = pow2ceil(ctx->ways);
    if (lpl > 16) lpl = 16;
    // ~4096 workgroups over all tables, every lane group walking its share four lookups at a time.  Until round 3 the grid had
    // one lane group per lookup -- 213 k workgroups for a c3 chunk, three of the four lookups "in flight" clamped dummies,
    // launch-rate-bound: 403 us per chunk beside the training step, which lost 0.34-0.45 ms to it.  On 4096 workgroups the probe
    // takes 133 us (2048 / 8192: 162 / 136; eight / sixteen lookups in flight: 138-142 / 159-199).  A probe this fast doubles a
    // gather it runs beside (60-70 us for that launch) -- which is where it landed while it was queued behind the step's weight
    // gradients on the prefetch stream (one gather in sixteen; throttled to 512 workgroups: gather mean 33.4 -> 30.9 us, 427 us
    // per chunk); the long-batch step now issues it right behind its interaction forward (TrainEngine._issue_resolve), beside
    // the top MLP's GEMMs, which leave HBM idle.
    int64_t gx = cdiv(n, (int64_t)(256 / lpl) * 4);
    const int64_t gcap = 4096 / ctx->T > 0 ? 4096 / ctx->T : 1;
    if (gx > gcap) gx = gcap;
    dim3 grid((unsigned)gx, (unsigned)ctx->T);
#define PROBE_CALL(L) hipLaunchKernelGGL((k_probe<L, 4>), grid, dim3(256), 0, s, ctx->d_tab, ctx->ways, ctx->tags, idx, n, ld_idx, wslots, ctx->d_err)
    switch (lpl) {
        case 1: PROBE_CALL(1); break;
        case 2: PROBE_CALL(2); break;
        case 4: PROBE_CALL(4); break;
        case 8: PROBE_CALL(8); break;
        default: PROBE_CALL(16); break;
    }
#undef PROBE_CALL
    const int64_t spb = cdiv(batch_len, seg_len);
    const int64_t nseg = cdiv(n, batch_len) * spb;
    CDLRM_REQUIRE(nseg <= 65535 * 16, "too many segments");
    // grid.x is the segment index (up to 2^31 - 1 on HIP)
    hipLaunchKernelGGL(k_resolve_seg, dim3((unsigned)nseg, (unsigned)ctx->T), dim3(1024), 0, s, ctx->d_tab, ctx->ways, ctx->aux,
                       wslots, n, batch_len, seg_len, spb, ctx->d_err);
    int64_t gv = cdiv(n, 256);
    if (gv > 4096) gv = 4096;
    hipLaunchKernelGGL(k_victim_pos, dim3((unsigned)gv, (unsigned)ctx->T), dim3(256), 0, s, ctx->d_tab, ctx->T, ctx->ways, idx, n,
                       ld_idx, wslots, wsrc, ctx->vict_idx, ctx->vict_off);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// the per-iteration remainder: slot ids of one batch out of the window's resolved ids (misses move to the aux region of
// `aux_phase`), rows of its misses from the victim rows (or the host table) into their aux rows
// One lane per lookup for the slot ids (a wave takes 64 consecutive lookups of one table: coalesced 256-byte reads and
// writes), then the wave's misses, found by ballot, are copied 64 / LPR rows at a time by LPR lanes each.  (The first
// version gave every lookup LPR lanes: 6.8 M threads for 213 k lookups at c3, 62 us beside the first top-MLP GEMM,
// which it slowed by a third.)
template <int LPR>
__global__ void __launch_bounds__(256) k_take(const TableDesc* __restrict__ tab, int ways, int aux_first, int D4,
                                              float4* __restrict__ weight, float* const* __restrict__ host_rows,
                                              const int64_t* __restrict__ idx, int64_t ld_idx,
                                              const int32_t* __restrict__ wslots, const int32_t* __restrict__ wsrc,
                                              int64_t ld_w, int64_t n, const float* __restrict__ v_rows,
                                              int32_t* __restrict__ slots_out) {
    constexpr int RPW = 64 / LPR;                   // rows copied per pass of a wave
    const int t = blockIdx.y;
    const TableDesc d = tab[t];
    const int32_t first_aux = (int32_t)(d.P * ways);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane % LPR, gid = lane / LPR;
    const float4* vr = reinterpret_cast<const float4*>(v_rows);
    const float4* hr = reinterpret_cast<const float4*>(host_rows[t]);
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64; base < n; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + lane;
        int32_t sl = 0, src = 0;
        int64_t id = 0;
        bool aux = false;
        if (i < n) {
            sl = wslots[(int64_t)t * ld_w + i];
            aux = sl >= first_aux;
            if (aux) {
                sl += aux_first;
                src = wsrc[(int64_t)t * ld_w + i];
                id = idx[(int64_t)t * ld_idx + i];
            }
            slots_out[(int64_t)t * n + i] = sl;
        }
        unsigned long long mask = __ballot(aux);
        while (mask) {                              // wave-uniform
            int mine = -1;
#pragma unroll
            for (int g = 0; g < RPW; ++g) {
                if (mask) {
                    const int b = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    if (gid == g) mine = b;
                }
            }
            const int from_lane = mine < 0 ? 0 : mine;
            const int32_t r_sl = __shfl(sl, from_lane, 64), r_src = __shfl(src, from_lane, 64);
            const int64_t r_id = ((int64_t)__shfl((int)(id >> 32), from_lane, 64) << 32) |
                                 (uint32_t)__shfl((int)(id & 0xffffffff), from_lane, 64);
            if (mine >= 0) {
                const float4* from = r_src >= 0 ? vr + (int64_t)r_src * D4 : hr + r_id * D4;
                float4* to = weight + (d.row_base + r_sl) * D4;
                for (int cc = c; cc < D4; cc += LPR) to[cc] = from[cc];
            }
        }
    }
}

extern "C" int cdlrm_embbag_take(cdlrm_ctx* ctx, const int64_t* idx, int64_t n, int64_t ld_idx, const int32_t* wslots,
                                 const int32_t* wsrc, int64_t ld_w, int32_t* slots_out, int32_t aux_phase, void* stream) {
    CDLRM_CLEAR_STALE();
    CDLRM_REQUIRE(ctx && idx && wslots && wsrc && slots_out, "null argument");
    CDLRM_REQUIRE(aux_phase >= 0 && aux_phase < ctx->aux_phases, "aux_phase outside the geometry's aux_phases");
    CDLRM_REQUIRE(ctx->tags && ctx->weight, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(n >= 1 && ld_idx >= n && ld_w >= n, "bad n / ld");
    CDLRM_REQUIRE(ctx->h_host_rows[0] != nullptr, "cdlrm_ctx_bind_host_tables first");
    hipStream_t s = (hipStream_t)stream;
    const int D4 = ctx->D / 4;
    const int lpr = lanes_per_row(D4);
    int64_t gx = cdiv(n, 256);
    if (gx > 1024) gx = 1024;
    dim3 grid((unsigned)gx, (unsigned)ctx->T);
#define TAKE_CALL(L) hipLaunchKernelGGL(k_take<L>, grid, dim3(256), 0, s, ctx->d_tab, ctx->ways, aux_phase * ctx->aux, D4, reinterpret_cast<float4*>(ctx->weight), ctx->d_host_rows, idx, ld_idx, wslots, wsrc, ld_w, n, ctx->vict_rows, slots_out)
    DISPATCH_LPR(lpr, TAKE_CALL)
#undef TAKE_CALL
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Victim write-back (--evict-victim-cache; model_no_ddp.py:187 records victim_cache_entries = (aux rows, missing indices) per
// table and main_no_ddp.py:96 parses the flag, neither is ever used: the reference drops the update a MISSED row received).
// Defined here as what `emb_tables[k].weight[missing] = cache[k].weight[aux_rows]` does behind the step: the trained aux
// row of every miss goes back to its host row -- and to its copy among the window's resident victim rows, which later
// misses of the window read instead of the host table.  An index that misses several times in one batch has one aux row
// per occurrence, each trained on its own gradient; the LAST occurrence (position order) is the one written, as the
// sequential assignment leaves it.  Two launches: the misses' indices in miss order (miss i of a table sits in aux row i),
// then one wave per miss that looks for a later occurrence of its index and, finding none, copies the row.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_vwb_list(const TableDesc* __restrict__ tab, int ways, int aux_first,
                                                  const int64_t* __restrict__ idx, int64_t ld_idx,
                                                  const int32_t* __restrict__ slots, int64_t n,
                                                  int64_t* __restrict__ m_idx, int32_t* __restrict__ m_pos,
                                                  int32_t* __restrict__ m_count) {
    const int t = blockIdx.y;
    const int32_t first = (int32_t)(tab[t].P * ways) + aux_first;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t sl = slots[(int64_t)t * n + p];
        if (sl < first) continue;
        const int32_t i = sl - first;
        m_idx[(int64_t)t * n + i] = idx[(int64_t)t * ld_idx + p];
        m_pos[(int64_t)t * n + i] = (int32_t)p;
        atomicMax(m_count + t, i + 1);
    }
}

__global__ void __launch_bounds__(256) k_vwb_apply(const TableDesc* __restrict__ tab, int ways, int aux_first, int D4,
                                                   const float4* __restrict__ weight, float* const* __restrict__ host_rows,
                                                   const int64_t* __restrict__ m_idx, const int32_t* __restrict__ m_pos,
                                                   const int32_t* __restrict__ m_count, int64_t n,
                                                   const int32_t* __restrict__ wsrc, int64_t ld_w,
                                                   const int64_t* __restrict__ v_idx, const int64_t* __restrict__ v_off,
                                                   float* __restrict__ v_rows) {
    const int t = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cnt = m_count[t];
    const TableDesc d = tab[t];
    const int64_t* mi = m_idx + (int64_t)t * n;
    for (int i = blockIdx.x * 4 + wave; i < cnt; i += gridDim.x * 4) {
        const int64_t key = mi[i];
        bool later = false;
        for (int j = i + 1 + lane; j < cnt; j += 64) later |= mi[j] == key;
        if (__ballot(later)) continue;              // a later occurrence of this index writes the row (wave-uniform)
        // where the row's copy sits among the window's victim rows (-1: nowhere): the resolver's answer, or a search
        int64_t vp = -1;
        if (wsrc) vp = wsrc[(int64_t)t * ld_w + m_pos[(int64_t)t * n + i]];
        else if (v_idx) {
            const int64_t lo = v_off[t], hi = v_off[t + 1];
            const int64_t q = (int64_t)lower_bound_u64(reinterpret_cast<const uint64_t*>(v_idx), lo, hi, (uint64_t)key);
            if (q < hi && v_idx[q] == key) vp = q;
        }
        const float4* from = weight + (d.row_base + d.P * ways + aux_first + i) * D4;
        float4* host = reinterpret_cast<float4*>(host_rows[t]) + key * D4;
        float4* vict = vp >= 0 ? reinterpret_cast<float4*>(v_rows) + vp * D4 : nullptr;
        for (int cc = lane; cc < D4; cc += 64) {
            const float4 v = from[cc];
            host[cc] = v;
            if (vict) vict[cc] = v;
        }
    }
}

extern "C" uint64_t cdlrm_victim_writeback_work_bytes(int32_t T, int64_t n) {
    return (uint64_t)T * n * 12 + (((uint64_t)T * 4 + 255) & ~(uint64_t)255) + 512;
}

extern "C" int cdlrm_victim_writeback(cdlrm_ctx* ctx, const int64_t* idx, int64_t n, int64_t ld_idx, const int32_t* slots,
                                      const int32_t* wsrc, int64_t ld_w, int32_t aux_phase, void* work, void* stream) {
    CDLRM_REQUIRE(ctx && idx && slots && work, "null argument");
    CDLRM_REQUIRE(aux_phase >= 0 && aux_phase < ctx->aux_phases, "aux_phase outside the geometry's aux_phases");
    CDLRM_REQUIRE(ctx->tags && ctx->weight, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(ctx->h_host_rows[0] != nullptr, "cdlrm_ctx_bind_host_tables first");
    CDLRM_REQUIRE(n >= 0 && ld_idx >= n && (!wsrc || ld_w >= n) && ((uintptr_t)work & 255) == 0, "bad n / ld / work alignment");
    if (n == 0 || ctx->aux == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int T = ctx->T, D4 = ctx->D / 4;
    char* wp = (char*)work;
    int32_t* m_count = (int32_t*)wp; wp += (((uint64_t)T * 4 + 255) & ~(uint64_t)255);
    int64_t* m_idx = (int64_t*)wp; wp += (uint64_t)T * n * 8;
    int32_t* m_pos = (int32_t*)wp;
    CDLRM_HIP_CHECK(hipMemsetAsync(m_count, 0, (size_t)T * 4, s));
    int64_t gx = cdiv(n, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_vwb_list, dim3((unsigned)gx, (unsigned)T), dim3(256), 0, s, ctx->d_tab, ctx->ways, aux_phase * ctx->aux,
                       idx, ld_idx, slots, n, m_idx, m_pos, m_count);
    int64_t ga = cdiv(n < ctx->aux ? n : (int64_t)ctx->aux, 4);
    if (ga > 512) ga = 512;
    hipLaunchKernelGGL(k_vwb_apply, dim3((unsigned)ga, (unsigned)T), dim3(256), 0, s, ctx->d_tab, ctx->ways, aux_phase * ctx->aux, D4,
                       reinterpret_cast<const float4*>(ctx->weight), ctx->d_host_rows, m_idx, m_pos, m_count, n, wsrc, ld_w,
                       ctx->vict_idx, ctx->vict_off, const_cast<float*>(ctx->vict_rows));
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_ctx_time_next_gather(cdlrm_ctx* ctx, void* start_event, void* stop_event) {
    CDLRM_REQUIRE(ctx && start_event && stop_event, "null argument");
    ctx->ev_start = start_event;
    ctx->ev_stop = stop_event;
    return 0;
}

extern "C" int cdlrm_embbag_fwd(cdlrm_ctx* ctx, const int32_t* slots, const int64_t* offsets, int64_t n,
                                int64_t n_bags, int64_t ld_off, float* out, int64_t ld_bag, int64_t ld_table,
                                void* stream) {
    CDLRM_CLEAR_STALE();
    CDLRM_REQUIRE(ctx && slots && out, "null argument");
    CDLRM_REQUIRE(ctx->weight, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(((uintptr_t)out & 15) == 0 && ld_bag % 4 == 0 && ld_table % 4 == 0, "16-byte aligned output rows");
    CDLRM_REQUIRE(offsets != nullptr || n_bags == n, "Criteo layout needs n_bags == n");
    hipStream_t s = (hipStream_t)stream;
    if (n_bags == 0) return 0;
    const int D4 = ctx->D / 4;
    const int lpr = lanes_per_row(D4);
    const int gpb = 256 / lpr;
    const float4* w = reinterpret_cast<const float4*>(ctx->weight);
    hipEvent_t ev0 = (hipEvent_t)ctx->ev_start, ev1 = (hipEvent_t)ctx->ev_stop;
    ctx->ev_start = ctx->ev_stop = nullptr;
    // persistent grid of 4096 workgroups, 4 rows in flight per lane group: measured on MI355X (c3 shape) 33 us against 45 us
    // for a one-shot grid (8 in flight and non-temporal output stores: no better)
    const int pgrid = 4096, nt = 0;
    if (!offsets) {
        constexpr int U = 4;
        int64_t total = cdiv(n, (int64_t)gpb * U) * ctx->T;
        int64_t gx = total < pgrid ? total : pgrid;
        // the launch's own start / stop timestamps land in the caller's events (cdlrm_ctx_time_next_gather): no marker
        // packets on the queue, so timing a launch does not move it or its neighbours
#define PFWD_CALL(L)                                                                                                  \
    hipExtLaunchKernelGGL((k_embbag_fwd_arange_p<L, U>), dim3((unsigned)gx), dim3(256), 0, s, ev0, ev1, 0, ctx->d_tab, \
                          ctx->T, D4, w, slots, n, out, ld_bag, ld_table, nt)
        DISPATCH_LPR(lpr, PFWD_CALL)
#undef PFWD_CALL
    } else {
        if (ev0) CDLRM_HIP_CHECK(hipEventRecord(ev0, s));
        int64_t gx = cdiv(n_bags, gpb);
        if (gx > 65535) gx = 65535;
        dim3 grid((unsigned)gx, (unsigned)ctx->T);
#define FWD_CALL(L) hipLaunchKernelGGL(k_embbag_fwd_bags<L>, grid, dim3(256), 0, s, ctx->d_tab, D4, w, slots, offsets, n, n_bags, ld_off, out, ld_bag, ld_table)
        DISPATCH_LPR(lpr, FWD_CALL)
#undef FWD_CALL
    }
    if (ev1 && offsets) CDLRM_HIP_CHECK(hipEventRecord(ev1, s));
    CDLRM_LAUNCH_CHECK();
    return 0;
}

