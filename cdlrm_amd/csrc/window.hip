// Look-ahead window path on gfx950: unique-index scan (K1), set-associative insert plan (K2/K3),
// winners-only pinned-host -> HBM row fetch (K5a), commit with in-place eviction swap (K4/K5b) and
// eviction write-back (K14).  Reference: cache_manager.py:28-46, main_no_ddp.py:148-209.
//
// Everything is integer / byte work bound by HBM (and PCIe for the fetch): no MFMA here.  All T
// tables are handled by flat launches over table-major arrays; per-table boundaries are the
// [T+1] offset arrays (uniq_off, kept_off, win_off), looked up by a 5-step binary search.
#include "scan.h"

#define WIN_BLOCKS 2048
#define WIN_THREADS 256

// ---------------------------------------------------------------------------------------------
// K1: bitmap + popcount scan  ==  torch.unique(sorted)  (cache_manager.py:32)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WIN_THREADS) k_bm_set(const TableDesc* __restrict__ tab,
                                                        const int64_t* __restrict__ idx, int64_t n, int64_t ld_idx,
                                                        unsigned long long* __restrict__ bitmap, int* err) {
    const int t = blockIdx.y;
    const TableDesc d = tab[t];
    unsigned long long* bm = bitmap + d.bm_base;
    const int64_t* row = idx + (int64_t)t * ld_idx;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = row[i];
        if (v < 0 || v >= d.n_rows) { atomicOr(err, 1); continue; }
        const unsigned long long bit = 1ull << (v & 63);
        // a stale read can only miss a bit that is already set -> a redundant atomic, never a lost one
        if (!(bm[v >> 6] & bit)) atomicOr(&bm[v >> 6], bit);
    }
}

__global__ void __launch_bounds__(256) k_bm_count(const unsigned long long* __restrict__ bitmap, int64_t* sums) {
    __shared__ int smem[32];
    const int64_t w0 = (int64_t)blockIdx.x * BM_WPB + threadIdx.x * 4;
    const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(bitmap + w0);
    const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(bitmap + w0 + 2);
    int cnt = __popcll(a.x) + __popcll(a.y) + __popcll(b.x) + __popcll(b.y);
    int total;
    block_excl_scan(cnt, smem, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_bm_emit(const TableDesc* __restrict__ tab, int T,
                                                 unsigned long long* __restrict__ bitmap,
                                                 const int64_t* __restrict__ sums, int64_t* __restrict__ uniq,
                                                 int64_t* __restrict__ uniq_off, int64_t cap, int* err) {
    __shared__ int smem[32];
    const int64_t wblk = (int64_t)blockIdx.x * BM_WPB;
    // table of this block (table bitmaps are padded to whole blocks)
    int lo = 0, hi = T;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tab[mid].bm_base <= wblk) lo = mid; else hi = mid;
    }
    const int64_t bm_base = tab[lo].bm_base;
    if (threadIdx.x == 0 && wblk == bm_base) uniq_off[lo] = sums[blockIdx.x];
    const int64_t w0 = wblk + threadIdx.x * 4;
    unsigned long long w[4];
    {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(bitmap + w0);
        const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(bitmap + w0 + 2);
        w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y;
    }
    const int cnt = __popcll(w[0]) + __popcll(w[1]) + __popcll(w[2]) + __popcll(w[3]);
    int total;
    const int ex = block_excl_scan(cnt, smem, &total);
    if (cnt) {
        int64_t pos = sums[blockIdx.x] + ex;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned long long x = w[k];
            const int64_t base = (w0 + k - bm_base) << 6;
            while (x) {
                const int b = __ffsll((long long)x) - 1;
                x &= x - 1;
                if (pos < cap) uniq[pos] = base + b; else atomicOr(err, 4);
                ++pos;
            }
        }
        *reinterpret_cast<ulonglong2*>(bitmap + w0) = make_ulonglong2(0, 0);
        *reinterpret_cast<ulonglong2*>(bitmap + w0 + 2) = make_ulonglong2(0, 0);
    }
}

// Streamed form of K1 for windows that do not fit HBM as one tensor (config c5: 8000 batches x 65536 = 109 GB of
// indices): the bitmap accumulates any number of chunks, then ONE count + emit turns it into the sorted unique lists
// (and leaves the bitmap zero for the next window).
extern "C" int cdlrm_window_unique_add(cdlrm_ctx* ctx, const cdlrm_plan* plan, const int64_t* idx, int64_t n,
                                       int64_t ld_idx, void* stream) {
    CDLRM_REQUIRE(ctx && plan && idx, "null argument");
    CDLRM_REQUIRE(plan->bitmap, "plan buffers missing");
    CDLRM_REQUIRE(n >= 1 && ld_idx >= n, "bad n / ld_idx");
    CDLRM_REQUIRE(((uintptr_t)plan->bitmap & 15) == 0, "bitmap must be 16-byte aligned");
    int64_t gx = cdiv(n, WIN_THREADS);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_bm_set, dim3((unsigned)gx, (unsigned)ctx->T), dim3(WIN_THREADS), 0, (hipStream_t)stream, ctx->d_tab,
                       idx, n, ld_idx, (unsigned long long*)plan->bitmap, ctx->d_err);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_window_unique_finish(cdlrm_ctx* ctx, const cdlrm_plan* plan, void* stream) {
    CDLRM_REQUIRE(ctx && plan, "null argument");
    CDLRM_REQUIRE(plan->bitmap && plan->uniq && plan->uniq_off, "plan buffers missing");
    hipStream_t s = (hipStream_t)stream;
    const int64_t nblocks = ctx->total_bm_words / BM_WPB;
    int rc = cdlrm_scan_reserve(ctx, nblocks);
    if (rc) return rc;
    hipLaunchKernelGGL(k_bm_count, dim3((unsigned)nblocks), dim3(256), 0, s, (const unsigned long long*)plan->bitmap,
                       ctx->d_scan);
    hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(1024), 0, s, ctx->d_scan, nblocks, plan->uniq_off + ctx->T);
    hipLaunchKernelGGL(k_bm_emit, dim3((unsigned)nblocks), dim3(256), 0, s, ctx->d_tab, ctx->T,
                       (unsigned long long*)plan->bitmap, ctx->d_scan, plan->uniq, plan->uniq_off, plan->cap_uniq,
                       ctx->d_err);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_window_unique(cdlrm_ctx* ctx, const cdlrm_plan* plan, const int64_t* idx, int64_t n,
                                   int64_t ld_idx, void* stream) {
    int rc = cdlrm_window_unique_add(ctx, plan, idx, n, ld_idx, stream);
    if (rc) return rc;
    return cdlrm_window_unique_finish(ctx, plan, stream);
}

// ---------------------------------------------------------------------------------------------
// K2: probe the unique list against the tags (main_no_ddp.py:155-165), mark the ways this window
// hits (avail_tensor_sampler, :171-172), then keep the misses whose set still has a free way
// (:173-180).
// ---------------------------------------------------------------------------------------------
template <int LPL>
__global__ void __launch_bounds__(WIN_THREADS) k_uniq_probe(const TableDesc* __restrict__ tab, int T, int ways,
                                                            const int64_t* __restrict__ tags,
                                                            const int64_t* __restrict__ uniq,
                                                            const int64_t* __restrict__ uniq_off, int64_t cap,
                                                            uint8_t* __restrict__ hit,
                                                            unsigned long long* __restrict__ prot, int* err) {
    const int64_t U = min(uniq_off[T], cap);
    const int g = threadIdx.x % LPL;
    const int64_t grp0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPL;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / LPL;
    const int64_t u_round = cdiv_dev(U, ngrp) * ngrp;
    for (int64_t u = grp0; u < u_round; u += ngrp) {
        const bool valid = u < U;
        int found = 0x7fffffff;
        int64_t set = 0;
        int t = 0;
        bool bad = false;
        if (valid) {
            t = table_of(uniq_off, T, u);
            int64_t v = uniq[u];
            if (v < 0 || v >= tab[t].n_rows) { bad = true; v = 0; }
            set = mod_sets(v, tab[t].P);
            const int64_t* tg = tags + tab[t].tag_base + set * ways;
            for (int w = g; w < ways; w += LPL)
                if (tg[w] == v) found = w;
        }
#pragma unroll
        for (int m = LPL >> 1; m >= 1; m >>= 1) found = min(found, __shfl_xor(found, m, LPL));
        if (valid && g == 0) {
            const bool h = found != 0x7fffffff;
            hit[u] = (h || bad) ? 1 : 0;          // an out-of-range id is reported and never inserted
            if (bad) atomicOr(err, 1);
            else if (h) atomicOr(&prot[tab[t].set_base + set], 1ull << found);
        }
    }
}

__global__ void __launch_bounds__(WIN_THREADS) k_kept_flags(const TableDesc* __restrict__ tab, int T, int ways,
                                                            const int64_t* __restrict__ uniq,
                                                            const int64_t* __restrict__ uniq_off, int64_t cap,
                                                            const uint8_t* __restrict__ hit,
                                                            const unsigned long long* __restrict__ prot,
                                                            uint8_t* __restrict__ flags) {
    const int64_t U = min(uniq_off[T], cap);
    const unsigned long long full = ways == 64 ? ~0ull : ((1ull << ways) - 1);
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < U; u += (int64_t)gridDim.x * blockDim.x) {
        uint8_t f = 0;
        if (!hit[u]) {
            const int t = table_of(uniq_off, T, u);
            const int64_t set = mod_sets(uniq[u], tab[t].P);
            f = (prot[tab[t].set_base + set] & full) != full;
        }
        flags[u] = f;
    }
}

// off_out[k] = number of entries of the ascending list `list[0..count)` that are < bound[k], k = 0..T
__global__ void k_offsets_from_sorted(const int32_t* __restrict__ list, const int64_t* __restrict__ d_count, int64_t cap,
                                      const int64_t* __restrict__ bound, int T, int64_t* __restrict__ off_out,
                                      int64_t* __restrict__ total_out = nullptr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > T) return;
    const int64_t cnt = min(*d_count, cap);
    if (k == T) {
        off_out[T] = cnt;
        if (total_out) *total_out = *d_count;       // the length the list would have without the cap
        return;
    }
    int64_t lo = 0, hi = cnt;
    const int64_t b = bound[k];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)list[mid] < b) lo = mid + 1; else hi = mid;
    }
    off_out[k] = lo;
}

extern "C" int cdlrm_plan_probe(cdlrm_ctx* ctx, const cdlrm_plan* plan, void* stream) {
    CDLRM_REQUIRE(ctx && plan, "null argument");
    CDLRM_REQUIRE(ctx->tags, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(plan->uniq && plan->uniq_off && plan->prot && plan->hit && plan->kept && plan->kept_off && plan->way && plan->flags,
                  "plan buffers missing");
    CDLRM_REQUIRE(plan->cap_uniq < ((int64_t)1 << 31), "cap_uniq < 2^31");
    hipStream_t s = (hipStream_t)stream;
    int rc = cdlrm_scan_reserve(ctx, cdiv(plan->cap_uniq, 4096) + 1);
    if (rc) return rc;
    int lpl = pow2ceil(ctx->ways);
    if (lpl > 16) lpl = 16;
    int64_t gx = cdiv(plan->cap_uniq * lpl, WIN_THREADS);
    if (gx > WIN_BLOCKS) gx = WIN_BLOCKS;
    if (gx < 1) gx = 1;
#define UP_CALL(L) hipLaunchKernelGGL(k_uniq_probe<L>, dim3((unsigned)gx), dim3(WIN_THREADS), 0, s, ctx->d_tab, ctx->T, ctx->ways, ctx->tags, plan->uniq, plan->uniq_off, plan->cap_uniq, plan->hit, (unsigned long long*)plan->prot, ctx->d_err)
    switch (lpl) {
        case 1: UP_CALL(1); break;
        case 2: UP_CALL(2); break;
        case 4: UP_CALL(4); break;
        case 8: UP_CALL(8); break;
        default: UP_CALL(16); break;
    }
#undef UP_CALL
    int64_t gx2 = cdiv(plan->cap_uniq, WIN_THREADS);
    if (gx2 > WIN_BLOCKS) gx2 = WIN_BLOCKS;
    if (gx2 < 1) gx2 = 1;
    hipLaunchKernelGGL(k_kept_flags, dim3((unsigned)gx2), dim3(WIN_THREADS), 0, s, ctx->d_tab, ctx->T, ctx->ways,
                       plan->uniq, plan->uniq_off, plan->cap_uniq, plan->hit, (const unsigned long long*)plan->prot,
                       plan->flags);
    CDLRM_LAUNCH_CHECK();
    rc = cdlrm_compact_flags(ctx, plan->flags, plan->uniq_off + ctx->T, plan->cap_uniq, plan->kept, nullptr,
                             plan->cap_uniq, ctx->d_small + 0, 1, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_offsets_from_sorted, dim3(cdiv(ctx->T + 1, 64)), dim3(64), 0, s, plan->kept, ctx->d_small + 0,
                       plan->cap_uniq, plan->uniq_off, ctx->T, plan->kept_off);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_plan_offsets_sync(cdlrm_ctx* ctx, const cdlrm_plan* plan, int64_t* uniq_off, int64_t* kept_off,
                                       int64_t* win_off, void* stream) {
    CDLRM_REQUIRE(ctx && plan, "null argument");
    CDLRM_REQUIRE(3 * (ctx->T + 1) <= 3900, "too many tables for the pinned staging");
    hipStream_t s = (hipStream_t)stream;
    const size_t bytes = sizeof(int64_t) * (ctx->T + 1);
    int64_t* h = ctx->h_pinned;
    if (uniq_off) CDLRM_HIP_CHECK(hipMemcpyAsync(h, plan->uniq_off, bytes, hipMemcpyDeviceToHost, s));
    if (kept_off) CDLRM_HIP_CHECK(hipMemcpyAsync(h + (ctx->T + 1), plan->kept_off, bytes, hipMemcpyDeviceToHost, s));
    if (win_off) CDLRM_HIP_CHECK(hipMemcpyAsync(h + 2 * (ctx->T + 1), plan->win_off, bytes, hipMemcpyDeviceToHost, s));
    CDLRM_HIP_CHECK(hipStreamSynchronize(s));
    if (uniq_off) memcpy(uniq_off, h, bytes);
    if (kept_off) memcpy(kept_off, h + (ctx->T + 1), bytes);
    if (win_off) memcpy(win_off, h + 2 * (ctx->T + 1), bytes);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K3: way choice (Categorical(avail).sample(), main_no_ddp.py:183-185) and contested slots
// (:203-204: the claimant latest in ascending-index order wins).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void __launch_bounds__(WIN_THREADS) k_assign(const TableDesc* __restrict__ tab, int T, int ways,
                                                        const int64_t* __restrict__ uniq,
                                                        const int64_t* __restrict__ uniq_off,
                                                        const int32_t* __restrict__ kept,
                                                        const int64_t* __restrict__ kept_off, int64_t cap,
                                                        const unsigned long long* __restrict__ prot,
                                                        const float* __restrict__ q, uint64_t seed,
                                                        uint8_t* __restrict__ way_out, int32_t* __restrict__ winner) {
    const int64_t M = min(kept_off[T], cap);
    const unsigned long long full = ways == 64 ? ~0ull : ((1ull << ways) - 1);
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = kept[m];
        const int t = table_of(uniq_off, T, u);
        const int64_t v = uniq[u];
        const int64_t P = tab[t].P;
        const int64_t set = mod_sets(v, P);
        const unsigned long long avail = ~prot[tab[t].set_base + set] & full;
        // probs = avail / avail.sum(-1)  (torch.distributions.Categorical), float32
        const float p = 1.0f / (float)__popcll(avail);
        float best = 0.f;
        int bw = 0;
        for (int w = 0; w < ways; ++w) {
            float qv;
            if (q) {
                qv = q[m * ways + w];
            } else {
                uint32_t r[4];
                const uint64_t ctr = (uint64_t)m * ways + w;
                philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), 0x43444c52u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
                qv = -__logf(((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f));
            }
            const float a = ((avail >> w) & 1ull) ? p : 0.0f;
            const float val = a / qv;                // torch.multinomial: argmax(probs / q), first maximum
            if (w == 0 || val > best) { best = val; bw = w; }
        }
        way_out[m] = (uint8_t)bw;
        atomicMax(&winner[tab[t].row_base + P * bw + set], (int32_t)m);
    }
}

__global__ void __launch_bounds__(WIN_THREADS) k_winner_flags(const TableDesc* __restrict__ tab, int T,
                                                              const int64_t* __restrict__ uniq,
                                                              const int64_t* __restrict__ uniq_off,
                                                              const int32_t* __restrict__ kept,
                                                              const int64_t* __restrict__ kept_off, int64_t cap,
                                                              const uint8_t* __restrict__ way,
                                                              const int32_t* __restrict__ winner,
                                                              uint8_t* __restrict__ flags) {
    const int64_t M = min(kept_off[T], cap);
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = kept[m];
        const int t = table_of(uniq_off, T, u);
        const int64_t P = tab[t].P;
        const int64_t set = mod_sets(uniq[u], P);
        flags[m] = winner[tab[t].row_base + P * way[m] + set] == (int32_t)m;
    }
}

__global__ void __launch_bounds__(WIN_THREADS) k_winner_fill(const TableDesc* __restrict__ tab, int T, int ways,
                                                             const int64_t* __restrict__ uniq,
                                                             const int64_t* __restrict__ uniq_off,
                                                             const int32_t* __restrict__ kept,
                                                             const uint8_t* __restrict__ way,
                                                             const int32_t* __restrict__ win_claim,
                                                             const int64_t* __restrict__ win_off, int64_t cap,
                                                             int64_t* __restrict__ win_idx, int64_t* __restrict__ win_row,
                                                             int64_t* __restrict__ win_tag, int32_t* __restrict__ winner) {
    const int64_t Wn = min(win_off[T], cap);
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < Wn; w += (int64_t)gridDim.x * blockDim.x) {
        const int32_t m = win_claim[w];
        const int64_t u = kept[m];
        const int t = table_of(uniq_off, T, u);
        const int64_t v = uniq[u];
        const int64_t P = tab[t].P;
        const int64_t set = mod_sets(v, P);
        const int wy = way[m];
        const int64_t row = tab[t].row_base + P * wy + set;
        win_idx[w] = v;
        win_row[w] = row;
        win_tag[w] = tab[t].tag_base + set * ways + wy;
        winner[row] = -1;      // every claimed slot has exactly one winner: scratch is clean again
    }
}

__global__ void __launch_bounds__(WIN_THREADS) k_prot_clear(const TableDesc* __restrict__ tab, int T,
                                                            const int64_t* __restrict__ uniq,
                                                            const int64_t* __restrict__ uniq_off, int64_t cap,
                                                            const uint8_t* __restrict__ hit,
                                                            unsigned long long* __restrict__ prot) {
    const int64_t U = min(uniq_off[T], cap);
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < U; u += (int64_t)gridDim.x * blockDim.x) {
        if (hit[u]) {
            const int t = table_of(uniq_off, T, u);
            prot[tab[t].set_base + mod_sets(uniq[u], tab[t].P)] = 0ull;
        }
    }
}

extern "C" int cdlrm_plan_assign(cdlrm_ctx* ctx, const cdlrm_plan* plan, const float* q, uint64_t seed, void* stream) {
    CDLRM_REQUIRE(ctx && plan, "null argument");
    CDLRM_REQUIRE(plan->winner && plan->win_claim && plan->win_idx && plan->win_row && plan->win_tag && plan->win_off &&
                      plan->way && plan->hit && plan->kept && plan->flags,
                  "plan buffers missing");
    hipStream_t s = (hipStream_t)stream;
    int64_t gx = cdiv(plan->cap_uniq, WIN_THREADS);
    if (gx > WIN_BLOCKS) gx = WIN_BLOCKS;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_assign, dim3((unsigned)gx), dim3(WIN_THREADS), 0, s, ctx->d_tab, ctx->T, ctx->ways, plan->uniq,
                       plan->uniq_off, plan->kept, plan->kept_off, plan->cap_uniq,
                       (const unsigned long long*)plan->prot, q, seed, plan->way, plan->winner);
    uint8_t* flags = plan->flags;
    hipLaunchKernelGGL(k_winner_flags, dim3((unsigned)gx), dim3(WIN_THREADS), 0, s, ctx->d_tab, ctx->T, plan->uniq,
                       plan->uniq_off, plan->kept, plan->kept_off, plan->cap_uniq, plan->way, plan->winner, flags);
    CDLRM_LAUNCH_CHECK();
    int rc = cdlrm_compact_flags(ctx, flags, plan->kept_off + ctx->T, plan->cap_uniq, plan->win_claim, nullptr,
                                 plan->cap_win, ctx->d_small + 1, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_offsets_from_sorted, dim3(cdiv(ctx->T + 1, 64)), dim3(64), 0, s, plan->win_claim,
                       ctx->d_small + 1, plan->cap_win, plan->kept_off, ctx->T, plan->win_off);
    int64_t gw = cdiv(plan->cap_win, WIN_THREADS);
    if (gw > WIN_BLOCKS) gw = WIN_BLOCKS;
    if (gw < 1) gw = 1;
    hipLaunchKernelGGL(k_winner_fill, dim3((unsigned)gw), dim3(WIN_THREADS), 0, s, ctx->d_tab, ctx->T, ctx->ways,
                       plan->uniq, plan->uniq_off, plan->kept, plan->way, plan->win_claim, plan->win_off, plan->cap_win,
                       plan->win_idx, plan->win_row, plan->win_tag, plan->winner);
    hipLaunchKernelGGL(k_prot_clear, dim3((unsigned)gx), dim3(WIN_THREADS), 0, s, ctx->d_tab, ctx->T, plan->uniq,
                       plan->uniq_off, plan->cap_uniq, plan->hit, (unsigned long long*)plan->prot);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K5a / K4+K5b / K14 : row movement.  One LPR-lane group (16 B per lane) per row.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fetch(int T, int D4, int lpr, const int64_t* __restrict__ win_off, int64_t cap,
                                               const int64_t* __restrict__ win_idx, const int32_t* __restrict__ win_claim,
                                               const int32_t* __restrict__ kept, const int64_t* __restrict__ uniq_off,
                                               float* const* __restrict__ src, int by_position,
                                               float4* __restrict__ stage) {
    const int64_t Wn = min(win_off[T], cap);
    const int c = threadIdx.x % lpr;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / lpr;
    const int64_t ng = (int64_t)gridDim.x * blockDim.x / lpr;
    for (int64_t w = g0; w < Wn; w += ng) {
        const int t = table_of(win_off, T, w);
        const int64_t r = by_position ? ((int64_t)kept[win_claim[w]] - uniq_off[t]) : win_idx[w];
        const float4* s = reinterpret_cast<const float4*>(src[t]) + r * D4;
        for (int cc = c; cc < D4; cc += lpr) stage[w * D4 + cc] = s[cc];
    }
}

__global__ void __launch_bounds__(256) k_commit(int T, int D4, int lpr, const int64_t* __restrict__ win_off, int64_t cap,
                                                const int64_t* __restrict__ win_idx, const int64_t* __restrict__ win_row,
                                                const int64_t* __restrict__ win_tag, int64_t* __restrict__ tags,
                                                float4* __restrict__ weight, float4* __restrict__ stage,
                                                int64_t* __restrict__ ev_tag) {
    const int64_t Wn = min(win_off[T], cap);
    const int c = threadIdx.x % lpr;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / lpr;
    const int64_t ng = (int64_t)gridDim.x * blockDim.x / lpr;
    for (int64_t w = g0; w < Wn; w += ng) {
        const int64_t tp = win_tag[w];
        const int64_t old = tags[tp];                  // every lane of the group reads the same word
        const int64_t row = win_row[w];
        for (int cc = c; cc < D4; cc += lpr) {
            const float4 in = stage[w * D4 + cc];
            if (old != -1) stage[w * D4 + cc] = weight[row * D4 + cc];     // evicted row (main_no_ddp.py:197)
            weight[row * D4 + cc] = in;                                     // :206
        }
        // all lanes of the group are in one wave: the tag read above is complete for every lane
        // before any lane's store below is issued (in-order issue within the wave)
        if (c == 0) {
            ev_tag[w] = old;
            tags[tp] = win_idx[w];                                          // :204
        }
    }
}

__global__ void __launch_bounds__(256) k_writeback(int T, int D4, int lpr, const int64_t* __restrict__ win_off,
                                                   int64_t cap, const int64_t* __restrict__ ev_tag,
                                                   const float4* __restrict__ stage, float* const* __restrict__ dst,
                                                   int average) {
    const int64_t Wn = min(win_off[T], cap);
    const int c = threadIdx.x % lpr;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / lpr;
    const int64_t ng = (int64_t)gridDim.x * blockDim.x / lpr;
    for (int64_t w = g0; w < Wn; w += ng) {
        const int64_t old = ev_tag[w];
        if (old == -1) continue;
        const int t = table_of(win_off, T, w);
        float4* d = reinterpret_cast<float4*>(dst[t]) + old * D4;
        for (int cc = c; cc < D4; cc += lpr) {
            float4 v = stage[w * D4 + cc];
            if (average) {                              // cache_manager.py:61-62
                const float4 h = d[cc];
                v.x = (h.x + v.x) / 2; v.y = (h.y + v.y) / 2; v.z = (h.z + v.z) / 2; v.w = (h.w + v.w) / 2;
            }
            d[cc] = v;
        }
    }
}

static int lanes_per_row_w(int D4) { int l = pow2ceil(D4); return l > 64 ? 64 : (l < 4 ? 4 : l); }

// Rows of the pinned HOST tables -> HBM, for the plan's bulk fetches (winners, victims).  PCIe-bound: the link needs
// ~1 MB of reads in flight, not thousands of waves -- a wide grid of waves parked on PCIe round trips takes the wave
// slots of every CU and starves the training kernels running beside the plan (measured: GEMMs 2x slower for the
// 180 ms per window the old 2048-workgroup fetches ran).  So: a SMALL grid, HOSTROWS_U independent 16-B reads per
// lane, loads phased (ids first, then all host reads back to back -- see k_fill_aux).
#define HOSTROWS_U 8
template <bool VIA_POS>
__global__ void __launch_bounds__(256) k_host_rows(int T, int D4, const int64_t* __restrict__ off, int64_t cap,
                                                   const int64_t* __restrict__ ids, const int32_t* __restrict__ pos,
                                                   const int64_t* __restrict__ uniq, float* const* __restrict__ src,
                                                   float4* __restrict__ out, int64_t* __restrict__ idx_out) {
    __shared__ int64_t soff[1025];
    __shared__ uint64_t sptr[1024];
    for (int k = threadIdx.x; k <= T; k += blockDim.x) soff[k] = off[k];
    for (int k = threadIdx.x; k < T; k += blockDim.x) sptr[k] = (uint64_t)(uintptr_t)src[k];
    __syncthreads();
    const int64_t n = min(soff[T], cap);
    const int64_t total = n * D4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) f32x4* host_ptr;
    for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += stride * HOSTROWS_U) {
        int64_t j[HOSTROWS_U], id[HOSTROWS_U];
        int c[HOSTROWS_U];
        host_ptr sp[HOSTROWS_U];
        f32x4 v[HOSTROWS_U];
#pragma unroll
        for (int u = 0; u < HOSTROWS_U; ++u) {
            const int64_t e = min(e0 + u * stride, total - 1);
            j[u] = e / D4;
            c[u] = (int)(e % D4);
            sp[u] = (host_ptr)sptr[table_of(soff, T, j[u])];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (VIA_POS) {
            int32_t p[HOSTROWS_U];
#pragma unroll
            for (int u = 0; u < HOSTROWS_U; ++u) p[u] = pos[j[u]];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < HOSTROWS_U; ++u) id[u] = uniq[p[u]];
        } else {
#pragma unroll
            for (int u = 0; u < HOSTROWS_U; ++u) id[u] = ids[j[u]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < HOSTROWS_U; ++u) v[u] = sp[u][id[u] * D4 + c[u]];
        __builtin_amdgcn_sched_barrier(0);
        // unconditional stores: elements past the end were clamped to the last one (same value, same address)
#pragma unroll
        for (int u = 0; u < HOSTROWS_U; ++u) {
            *reinterpret_cast<f32x4*>(out + j[u] * D4 + c[u]) = v[u];
            if (VIA_POS) idx_out[j[u]] = id[u];
        }
    }
}

static int host_rows_grid() { return 32; }

extern "C" int cdlrm_plan_fetch(cdlrm_ctx* ctx, const cdlrm_plan* plan, const float* const* src_rows, int by_position,
                                void* stream) {
    CDLRM_REQUIRE(ctx && plan && src_rows, "null argument");
    CDLRM_REQUIRE(plan->stage && ((uintptr_t)plan->stage & 15) == 0, "stage missing / unaligned");
    hipStream_t s = (hipStream_t)stream;
    for (int k = 0; k < ctx->T; ++k) CDLRM_REQUIRE(src_rows[k] && ((uintptr_t)src_rows[k] & 15) == 0, "aligned sources");
    CDLRM_HIP_CHECK(hipMemcpyAsync(ctx->d_ptr_fetch, src_rows, sizeof(float*) * ctx->T, hipMemcpyHostToDevice, s));
    const int D4 = ctx->D / 4, lpr = lanes_per_row_w(D4);
    int64_t gx = cdiv(plan->cap_win * lpr, 256);
    if (gx > WIN_BLOCKS) gx = WIN_BLOCKS;
    if (gx < 1) gx = 1;
    if (!by_position) {
        CDLRM_REQUIRE(ctx->T <= 1024, "more than 1024 tables");
        hipLaunchKernelGGL(k_host_rows<false>, dim3((unsigned)host_rows_grid()), dim3(256), 0, s, ctx->T, D4, plan->win_off,
                           plan->cap_win, plan->win_idx, nullptr, nullptr, ctx->d_ptr_fetch,
                           reinterpret_cast<float4*>(plan->stage), nullptr);
        CDLRM_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_fetch, dim3((unsigned)gx), dim3(256), 0, s, ctx->T, D4, lpr, plan->win_off, plan->cap_win,
                       plan->win_idx, plan->win_claim, plan->kept, plan->uniq_off, ctx->d_ptr_fetch, by_position,
                       reinterpret_cast<float4*>(plan->stage));
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_plan_commit(cdlrm_ctx* ctx, const cdlrm_plan* plan, void* stream) {
    CDLRM_REQUIRE(ctx && plan, "null argument");
    CDLRM_REQUIRE(ctx->tags && ctx->weight, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(plan->stage && plan->ev_tag && plan->win_off, "plan buffers missing");
    hipStream_t s = (hipStream_t)stream;
    const int D4 = ctx->D / 4, lpr = lanes_per_row_w(D4);
    int64_t gx = cdiv(plan->cap_win * lpr, 256);
    if (gx > WIN_BLOCKS) gx = WIN_BLOCKS;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_commit, dim3((unsigned)gx), dim3(256), 0, s, ctx->T, D4, lpr, plan->win_off, plan->cap_win,
                       plan->win_idx, plan->win_row, plan->win_tag, ctx->tags, reinterpret_cast<float4*>(ctx->weight),
                       reinterpret_cast<float4*>(plan->stage), plan->ev_tag);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_plan_writeback(cdlrm_ctx* ctx, const cdlrm_plan* plan, float* const* dst_rows, int average,
                                    void* stream) {
    CDLRM_REQUIRE(ctx && plan && dst_rows, "null argument");
    hipStream_t s = (hipStream_t)stream;
    for (int k = 0; k < ctx->T; ++k) CDLRM_REQUIRE(dst_rows[k] && ((uintptr_t)dst_rows[k] & 15) == 0, "aligned tables");
    CDLRM_HIP_CHECK(hipMemcpyAsync(ctx->d_ptr_wb, dst_rows, sizeof(float*) * ctx->T, hipMemcpyHostToDevice, s));
    const int D4 = ctx->D / 4, lpr = lanes_per_row_w(D4);
    int64_t gx = cdiv(plan->cap_win * lpr, 256);
    if (gx > WIN_BLOCKS) gx = WIN_BLOCKS;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_writeback, dim3((unsigned)gx), dim3(256), 0, s, ctx->T, D4, lpr, plan->win_off, plan->cap_win,
                       plan->ev_tag, reinterpret_cast<const float4*>(plan->stage), ctx->d_ptr_wb, average);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Window victims: the window's unique indices that are neither cached already nor inserted by this plan (no free
// way in their set).  Every lookup of such an index is a miss (model_no_ddp.py:176-179 reads its HOST row into an
// aux slot each time).  Their host rows do not change while the window trains, so they are fetched ONCE per window
// into HBM (on the plan stream, during the previous window) and the per-iteration aux fill becomes an HBM copy.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WIN_THREADS) k_victim_flags(const int64_t* __restrict__ uniq_off, int T, int64_t cap,
                                                              const uint8_t* __restrict__ hit, uint8_t* __restrict__ flags) {
    const int64_t U = min(uniq_off[T], cap);
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < U; u += (int64_t)gridDim.x * blockDim.x)
        flags[u] = hit[u] ? 0 : 1;
}

__global__ void __launch_bounds__(WIN_THREADS) k_victim_unflag(const int64_t* __restrict__ win_off, int T, int64_t cap,
                                                               const int32_t* __restrict__ win_claim,
                                                               const int32_t* __restrict__ kept, uint8_t* __restrict__ flags) {
    const int64_t Wn = min(win_off[T], cap);
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < Wn; w += (int64_t)gridDim.x * blockDim.x)
        flags[kept[win_claim[w]]] = 0;
}

__global__ void __launch_bounds__(WIN_THREADS) k_victim_ids(const int64_t* __restrict__ v_off, int T, int64_t cap,
                                                            const int32_t* __restrict__ pos, const int64_t* __restrict__ uniq,
                                                            int64_t* __restrict__ idx) {
    const int64_t n = min(v_off[T], cap);
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        idx[j] = uniq[pos[j]];
}

static int plan_victims_impl(cdlrm_ctx* ctx, const cdlrm_plan* plan, const cdlrm_victims* v, int fetch, void* stream) {
    CDLRM_REQUIRE(ctx && plan && v, "null argument");
    CDLRM_REQUIRE(v->pos && v->idx && v->off && v->rows && v->cap >= 1 && ((uintptr_t)v->rows & 15) == 0, "victim buffers");
    CDLRM_REQUIRE(plan->uniq && plan->uniq_off && plan->hit && plan->kept && plan->win_claim && plan->win_off && plan->flags,
                  "plan buffers missing");
    CDLRM_REQUIRE(ctx->h_host_rows[0] != nullptr, "cdlrm_ctx_bind_host_tables first");
    hipStream_t s = (hipStream_t)stream;
    int64_t gx = cdiv(plan->cap_uniq, WIN_THREADS);
    if (gx > WIN_BLOCKS) gx = WIN_BLOCKS;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_victim_flags, dim3((unsigned)gx), dim3(WIN_THREADS), 0, s, plan->uniq_off, ctx->T, plan->cap_uniq,
                       plan->hit, plan->flags);
    int64_t gw = cdiv(plan->cap_win, WIN_THREADS);
    if (gw > WIN_BLOCKS) gw = WIN_BLOCKS;
    if (gw < 1) gw = 1;
    hipLaunchKernelGGL(k_victim_unflag, dim3((unsigned)gw), dim3(WIN_THREADS), 0, s, plan->win_off, ctx->T, plan->cap_win,
                       plan->win_claim, plan->kept, plan->flags);
    CDLRM_LAUNCH_CHECK();
    int rc = cdlrm_scan_reserve(ctx, cdiv(plan->cap_uniq, 4096) + 1);
    if (rc) return rc;
    // beyond v->cap the list is cut (those indices keep reading the host table): clear | soft cap
    rc = cdlrm_compact_flags(ctx, plan->flags, plan->uniq_off + ctx->T, plan->cap_uniq, v->pos, nullptr, v->cap,
                             ctx->d_small + 2, 1 | 2, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_offsets_from_sorted, dim3(cdiv(ctx->T + 1, 64)), dim3(64), 0, s, v->pos, ctx->d_small + 2, v->cap,
                       plan->uniq_off, ctx->T, v->off, v->off + ctx->T + 1);
    CDLRM_REQUIRE(ctx->T <= 1024, "more than 1024 tables");
    if (fetch) {
        const int D4 = ctx->D / 4;
        hipLaunchKernelGGL(k_host_rows<true>, dim3((unsigned)host_rows_grid()), dim3(256), 0, s, ctx->T, D4, v->off, v->cap,
                           nullptr, v->pos, plan->uniq, ctx->d_host_rows, reinterpret_cast<float4*>(v->rows), v->idx);
    } else {
        int64_t gv = cdiv(v->cap, WIN_THREADS);
        if (gv > WIN_BLOCKS) gv = WIN_BLOCKS;
        hipLaunchKernelGGL(k_victim_ids, dim3((unsigned)gv), dim3(WIN_THREADS), 0, s, v->off, ctx->T, v->cap, v->pos,
                           plan->uniq, v->idx);
    }
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_plan_victims(cdlrm_ctx* ctx, const cdlrm_plan* plan, const cdlrm_victims* v, void* stream) {
    return plan_victims_impl(ctx, plan, v, 1, stream);
}

// the list only (idx / off); the caller moves the rows itself (cdlrm_host_gather_rows + one DMA copy into v->rows)
extern "C" int cdlrm_plan_victims_list(cdlrm_ctx* ctx, const cdlrm_plan* plan, const cdlrm_victims* v, void* stream) {
    return plan_victims_impl(ctx, plan, v, 0, stream);
}

extern "C" int cdlrm_ctx_bind_victims(cdlrm_ctx* ctx, const cdlrm_victims* v) {
    CDLRM_REQUIRE(ctx, "null argument");
    if (!v) {
        ctx->vict_idx = nullptr; ctx->vict_off = nullptr; ctx->vict_rows = nullptr;
        return 0;
    }
    CDLRM_REQUIRE(v->idx && v->off && v->rows && ((uintptr_t)v->rows & 15) == 0, "victim buffers");
    ctx->vict_idx = v->idx; ctx->vict_off = v->off; ctx->vict_rows = v->rows;
    return 0;
}

__global__ void __launch_bounds__(256) k_gather_rows(const float4* __restrict__ src, const int64_t* __restrict__ index,
                                                     int64_t count, int D4, float4* __restrict__ out) {
    const int64_t total = count * D4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / D4;
        const int c = (int)(e % D4);
        out[e] = src[index[r] * D4 + c];
    }
}

extern "C" int cdlrm_gather_rows(const float* src, const int64_t* index, int64_t count, int32_t dim, float* out,
                                 void* stream) {
    CDLRM_REQUIRE(src && index && out && dim % 4 == 0, "bad argument");
    CDLRM_REQUIRE((((uintptr_t)src | (uintptr_t)out) & 15) == 0, "16-byte aligned rows");
    if (count == 0) return 0;
    int64_t gx = cdiv(count * (dim / 4), 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(src), index, count, dim / 4, reinterpret_cast<float4*>(out));
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K13: table aggregation helpers (broadcast_and_aggregate, main_no_ddp.py:250-292)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_agg_gather(const float4* __restrict__ weight, const int64_t* __restrict__ rows,
                                                    const int64_t* __restrict__ count, int64_t first, int64_t cap, int D4,
                                                    float scale, float4* __restrict__ buf) {
    const int64_t total = max((int64_t)0, min(*count - first, cap)) * D4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        float4 v = weight[rows[e / D4] * D4 + (e % D4)];
        // reference: weight[unique] / world_size  (:277) -- a true division, not a multiply
        if (scale != 1.0f) { v.x /= scale; v.y /= scale; v.z /= scale; v.w /= scale; }
        buf[e] = v;
    }
}

__global__ void __launch_bounds__(256) k_agg_scatter(float4* __restrict__ weight, const int64_t* __restrict__ rows,
                                                     const int64_t* __restrict__ count, int64_t first, int64_t cap, int D4,
                                                     const float4* __restrict__ buf) {
    const int64_t total = max((int64_t)0, min(*count - first, cap)) * D4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
        weight[rows[e / D4] * D4 + (e % D4)] = buf[e];
}

extern "C" int cdlrm_agg_compact(cdlrm_ctx* ctx, uint8_t* touched, int64_t total_rows, int64_t* rows_out, int64_t cap,
                                 int64_t* count_out, void* stream) {
    CDLRM_REQUIRE(ctx && touched && rows_out && count_out, "null argument");
    // own scratch: this runs on the caller's (main) stream while the next window's plan may be compacting on the
    // plan stream with ctx->d_scan
    int rc = cdlrm_scan_reserve_agg(ctx, cdiv(total_rows, 4096) + 1);
    if (rc) return rc;
    return cdlrm_compact_flags(ctx, touched, nullptr, total_rows, nullptr, rows_out, cap, count_out, 1,
                               (hipStream_t)stream, ctx->d_scan_agg, ctx->scan_agg_cap);
}

extern "C" int cdlrm_agg_gather(cdlrm_ctx* ctx, const int64_t* rows, const int64_t* count, float scale, float* buf,
                                int64_t cap, int64_t first, void* stream) {
    CDLRM_REQUIRE(ctx && rows && count && buf && ctx->weight, "null argument");
    if (cap == 0) return 0;
    int64_t gx = cdiv(cap * (ctx->D / 4), 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_agg_gather, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(ctx->weight), rows, count, first, cap, ctx->D / 4, scale,
                       reinterpret_cast<float4*>(buf));
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_agg_scatter(cdlrm_ctx* ctx, const int64_t* rows, const int64_t* count, const float* buf, int64_t cap,
                                 int64_t first, void* stream) {
    CDLRM_REQUIRE(ctx && rows && count && buf && ctx->weight, "null argument");
    if (cap == 0) return 0;
    int64_t gx = cdiv(cap * (ctx->D / 4), 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_agg_scatter, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(ctx->weight), rows, count, first, cap, ctx->D / 4,
                       reinterpret_cast<const float4*>(buf));
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Deadlines for the touched-row merge (round 4).  The merge of step j has to be applied to a row before the first step after
// j that USES the row on any rank -- not before step j + 1 as a whole.  The look-ahead window says which rows the next batches
// use (their slot ids are resolved already: cdlrm_window_resolve), so the rows of a merge are classed by the first batch that
// needs them and exchanged in that order, most of them in the background of the following steps (engine.MergePump).
//   cdlrm_agg_mark_tier : tier[row] = value for every cache slot a [T, n] view of resolved slot ids names (aux slots skipped);
//                         the engine calls it class by class from the LATEST deadline to the earliest, so the earliest wins
//   cdlrm_agg_split     : stable counting sort of the merge's sorted row list by tier byte: rows grouped by class, ascending
//                         inside a class -- the same list on every rank -- and the class offsets
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mark_tier(const TableDesc* __restrict__ tab, int ways, const int32_t* __restrict__ slots,
                                                   int64_t n, int64_t ld, uint8_t value, uint8_t* __restrict__ tier) {
    const int t = blockIdx.y;
    const int64_t rb = tab[t].row_base, first_aux = tab[t].P * ways;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = slots[(int64_t)t * ld + i];
        if (s >= 0 && s < first_aux) tier[rb + s] = value;
    }
}

extern "C" int cdlrm_agg_mark_tier(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, int64_t ld, int32_t value, uint8_t* tier,
                                   void* stream) {
    CDLRM_REQUIRE(ctx && slots && tier && ld >= n && value >= 0 && value <= 255, "bad argument");
    if (n == 0) return 0;
    int64_t gx = cdiv(n, 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_mark_tier, dim3((unsigned)gx, (unsigned)ctx->T), dim3(256), 0, (hipStream_t)stream, ctx->d_tab, ctx->ways,
                       slots, n, ld, (uint8_t)value, tier);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

#define SPLIT_PER_BLOCK 1024      // elements per block: 256 threads x 4 consecutive elements
#define SPLIT_MAX_CLASSES 8
// per block and class: how many of the block's elements fall into the class; layout class-major [C][nblocks], so that ONE
// exclusive scan over the whole array yields every (class, block) group's place in the stable counting sort
__global__ void __launch_bounds__(256) k_split_hist(const int64_t* __restrict__ rows, int64_t n, const uint8_t* __restrict__ tier,
                                                    int C, int64_t nblocks, int64_t* __restrict__ hist) {
    __shared__ int cnt[SPLIT_MAX_CLASSES];
    if (threadIdx.x < SPLIT_MAX_CLASSES) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SPLIT_PER_BLOCK + threadIdx.x * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (base + i < n) {
            int c = tier[rows[base + i]];
            c = c < C ? c : C - 1;
            atomicAdd(&cnt[c], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < C) hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = cnt[threadIdx.x];
}

__global__ void __launch_bounds__(256) k_split_scatter(const int64_t* __restrict__ rows, int64_t n, const uint8_t* __restrict__ tier,
                                                       int C, int64_t nblocks, const int64_t* __restrict__ scanned,
                                                       int64_t* __restrict__ rows_out) {
    __shared__ int smem[32];
    const int64_t base = (int64_t)blockIdx.x * SPLIT_PER_BLOCK + threadIdx.x * 4;
    int cls[4];
    int64_t r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cls[i] = -1;
        r[i] = 0;
        if (base + i < n) {
            r[i] = rows[base + i];
            const int c = tier[r[i]];
            cls[i] = c < C ? c : C - 1;
        }
    }
    for (int c = 0; c < C; ++c) {           // (block-uniform trip count: the scans synchronise)
        int mine = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) mine += cls[i] == c;
        int total;
        int pos = block_excl_scan(mine, smem, &total);
        const int64_t at = scanned[(int64_t)c * nblocks + blockIdx.x];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (cls[i] == c) rows_out[at + pos++] = r[i];
    }
}

__global__ void k_split_offsets(const int64_t* __restrict__ scanned, const int64_t* __restrict__ total, int C, int64_t nblocks,
                                int64_t* __restrict__ class_off) {
    const int c = threadIdx.x;
    if (c < C) class_off[c] = scanned[(int64_t)c * nblocks];
    if (c == C) class_off[C] = *total;
}

extern "C" int cdlrm_agg_split(cdlrm_ctx* ctx, const int64_t* rows, int64_t count, const uint8_t* tier, int32_t n_classes,
                               int64_t* rows_out, int64_t* class_off, void* stream) {
    CDLRM_REQUIRE(ctx && rows && tier && rows_out && class_off && count >= 0, "null argument");
    CDLRM_REQUIRE(n_classes >= 1 && n_classes <= SPLIT_MAX_CLASSES, "1 .. 8 classes");
    hipStream_t s = (hipStream_t)stream;
    if (count == 0) {
        CDLRM_HIP_CHECK(hipMemsetAsync(class_off, 0, sizeof(int64_t) * (n_classes + 1), s));
        return 0;
    }
    const int64_t nblocks = cdiv(count, SPLIT_PER_BLOCK);
    // the merge path's own scan scratch (the plan stream may be compacting with ctx->d_scan): [C][nblocks] counts + the total
    int rc = cdlrm_scan_reserve_agg(ctx, (int64_t)n_classes * nblocks + 2);
    if (rc) return rc;
    int64_t* hist = ctx->d_scan_agg;
    int64_t* total = ctx->d_scan_agg + (int64_t)n_classes * nblocks;
    hipLaunchKernelGGL(k_split_hist, dim3((unsigned)nblocks), dim3(256), 0, s, rows, count, tier, (int)n_classes, nblocks, hist);
    hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(1024), 0, s, hist, (int64_t)n_classes * nblocks, total);
    hipLaunchKernelGGL(k_split_scatter, dim3((unsigned)nblocks), dim3(256), 0, s, rows, count, tier, (int)n_classes, nblocks,
                       (const int64_t*)hist, rows_out);
    hipLaunchKernelGGL(k_split_offsets, dim3(1), dim3(64), 0, s, (const int64_t*)hist, (const int64_t*)total, (int)n_classes,
                       nblocks, class_off);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// small generic helpers of the drop-in surface
// ---------------------------------------------------------------------------------------------
// dst[index[i], :] = rows[i, :]  or  (dst + rows) / 2   (Prefetcher.eviction_manager, cache_manager.py:57-62)
__global__ void __launch_bounds__(256) k_scatter_rows(float4* __restrict__ dst, const int64_t* __restrict__ index,
                                                      const float4* __restrict__ rows, int64_t count, int D4, int average) {
    const int64_t total = count * D4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / D4;
        const int c = (int)(e % D4);
        float4 v = rows[e];
        float4* d = dst + index[r] * D4 + c;
        if (average) {
            const float4 h = *d;
            v.x = (h.x + v.x) / 2; v.y = (h.y + v.y) / 2; v.z = (h.z + v.z) / 2; v.w = (h.w + v.w) / 2;
        }
        *d = v;
    }
}

extern "C" int cdlrm_scatter_rows(float* dst, const int64_t* index, const float* rows, int64_t count, int32_t dim,
                                  int average, void* stream) {
    CDLRM_REQUIRE(dst && index && rows && dim % 4 == 0, "bad argument");
    CDLRM_REQUIRE((((uintptr_t)dst | (uintptr_t)rows) & 15) == 0, "16-byte aligned rows");
    if (count == 0) return 0;
    int64_t gx = cdiv(count * (dim / 4), 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(dst), index, reinterpret_cast<const float4*>(rows), count, dim / 4,
                       average);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// out[i, :] = (dst[index[i], :] + rows[i, :]) / 2: the gather half of the averaging write-back for index lists that
// may repeat an index (every entry is computed from the OLD destination row, cache_manager.py:62)
__global__ void __launch_bounds__(256) k_blend_rows(const float4* __restrict__ dst, const int64_t* __restrict__ index,
                                                    const float4* __restrict__ rows, int64_t count, int D4,
                                                    float4* __restrict__ out) {
    const int64_t total = count * D4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / D4;
        const int c = (int)(e % D4);
        const float4 v = rows[e];
        const float4 h = dst[index[r] * D4 + c];
        out[e] = make_float4((h.x + v.x) / 2, (h.y + v.y) / 2, (h.z + v.z) / 2, (h.w + v.w) / 2);
    }
}

extern "C" int cdlrm_blend_rows(const float* dst, const int64_t* index, const float* rows, int64_t count, int32_t dim,
                                float* out, void* stream) {
    CDLRM_REQUIRE(dst && index && rows && out && dim % 4 == 0, "bad argument");
    CDLRM_REQUIRE((((uintptr_t)dst | (uintptr_t)rows | (uintptr_t)out) & 15) == 0, "16-byte aligned rows");
    if (count == 0) return 0;
    int64_t gx = cdiv(count * (dim / 4), 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_blend_rows, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(dst), index, reinterpret_cast<const float4*>(rows), count, dim / 4,
                       reinterpret_cast<float4*>(out));
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// touched[row_base_t + slots[t, i]] = 1  (cache_group_idxs -> touched-row flags, main_no_ddp.py:417-423)
__global__ void __launch_bounds__(256) k_mark_rows(const TableDesc* __restrict__ tab, const int32_t* __restrict__ slots,
                                                   int64_t n, uint8_t* __restrict__ touched) {
    const int t = blockIdx.y;
    const int64_t rb = tab[t].row_base, rows = tab[t].rows;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = slots[(int64_t)t * n + i];
        if (s >= 0 && s < rows) touched[rb + s] = 1;
    }
}

extern "C" int cdlrm_mark_rows(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, uint8_t* touched, void* stream) {
    CDLRM_REQUIRE(ctx && slots && touched, "null argument");
    if (n == 0) return 0;
    int64_t gx = cdiv(n, 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_mark_rows, dim3((unsigned)gx, (unsigned)ctx->T), dim3(256), 0, (hipStream_t)stream, ctx->d_tab,
                       slots, n, touched);
    CDLRM_LAUNCH_CHECK();
    return 0;
}
