// Per-iteration embedding path on gfx950: tag probe (K6), fused multi-table sum-pool gather (K7, the
// HBM-roofline kernel) and atomics-free backward + sparse SGD (K8).
//
// Data layout in HBM: one flat fp32 weight buffer [sum_k rows_k, D] (row = 4*D bytes,
// 16-byte lanes), one flat int64 tag buffer, tags of a set contiguous (ways*8 bytes = one 128-B line
// at 16 ways).  Every kernel covers all T tables in one launch: blockIdx.y = table.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// K6a: tag probe.  LPL lanes cooperate on one lookup: lane g reads ways g, g+LPL, ... (8-byte tags,
// consecutive lanes -> consecutive tags -> one coalesced segment per set).
// model_no_ddp.py:166-174
// ---------------------------------------------------------------------------------------------
template <int LPL>
__global__ void __launch_bounds__(256) k_probe(const TableDesc* __restrict__ tab, int ways,
                                               const int64_t* __restrict__ tags,
                                               const int64_t* __restrict__ idx, int64_t n, int64_t ld_idx,
                                               int32_t* __restrict__ slots, int* err) {
    const int t = blockIdx.y;
    const TableDesc d = tab[t];
    const int g = threadIdx.x % LPL;
    const int64_t grp0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPL;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / LPL;
    const int64_t n_round = cdiv_dev(n, ngrp) * ngrp;
    for (int64_t i = grp0; i < n_round; i += ngrp) {
        const bool valid = i < n;
        int64_t v = valid ? idx[(int64_t)t * ld_idx + i] : 0;
        bool bad = valid && (v < 0 || v >= d.n_rows);
        if (bad) v = 0;
        const int64_t set = mod_sets(v, d.P);
        const int64_t* tg = tags + d.tag_base + set * ways;
        int found = 0x7fffffff;
        for (int w = g; w < ways; w += LPL)
            if (tg[w] == v) found = w;
#pragma unroll
        for (int m = LPL >> 1; m >= 1; m >>= 1) found = min(found, __shfl_xor(found, m, LPL));
        if (valid && g == 0) {
            slots[(int64_t)t * n + i] = (found == 0x7fffffff) ? -1 : (int32_t)(d.P * found + set);
            if (bad) atomicOr(err, 1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K6b: ordered miss resolution, one workgroup per table: the i-th miss (in position order) gets aux
// slot P*ways + i (model_no_ddp.py:176-177).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_resolve(const TableDesc* __restrict__ tab, int ways, int aux,
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
                row[i] = (int32_t)(d.P * ways + r);
                mp[r] = (int32_t)i;
            } else {
                row[i] = (int32_t)(d.P * ways);   // keep addresses legal; the call reports the error
                atomicOr(err, 2);
            }
        }
        running += total;
    }
    if (threadIdx.x == 0) miss_count[t] = min(running, aux);
}

// ---------------------------------------------------------------------------------------------
// K6c: aux-row fill: cache.weight[aux_i] = W_host[idx_miss_i]  (model_no_ddp.py:179), zero-copy reads
// of the pinned host table over PCIe, 16 bytes per lane.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fill_aux(const TableDesc* __restrict__ tab, int ways, int D4,
                                                  float4* __restrict__ weight, float* const* __restrict__ host_rows,
                                                  const int64_t* __restrict__ idx, int64_t n, int64_t ld_idx,
                                                  const int32_t* __restrict__ miss_pos,
                                                  const int32_t* __restrict__ miss_count) {
    const int t = blockIdx.y;
    const int m = miss_count[t];
    if (m == 0) return;
    const TableDesc d = tab[t];
    const float4* src = reinterpret_cast<const float4*>(host_rows[t]);
    const int64_t total = (int64_t)m * D4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / D4), c = (int)(e % D4);
        const int64_t v = idx[(int64_t)t * ld_idx + miss_pos[(int64_t)t * n + r]];
        weight[(d.row_base + d.P * ways + r) * D4 + c] = src[v * D4 + c];
    }
}

// ---------------------------------------------------------------------------------------------
// K7: fused multi-table EmbeddingBag(sum) forward.  LPR lanes x 16 B cover one row chunk; a wave
// holds 64/LPR bags per step and UNROLL steps in flight (independent slot -> row -> store chains).
// Algorithmic bytes per lookup (Criteo layout): 4D row read + 4D output write + 8 index + 8 offset.
// ---------------------------------------------------------------------------------------------
#define FWD_UNROLL 4
template <int LPR>
__global__ void __launch_bounds__(256) k_embbag_fwd_arange(const TableDesc* __restrict__ tab, int D4,
                                                           const float4* __restrict__ weight,
                                                           const int32_t* __restrict__ slots, int64_t n,
                                                           float* __restrict__ out, int64_t ld_bag, int64_t ld_table) {
    const int t = blockIdx.y;
    const int64_t row_base = tab[t].row_base;
    const int c = threadIdx.x % LPR;
    const int gpb = blockDim.x / LPR;
    const int gid = threadIdx.x / LPR;
    const int32_t* sl = slots + (int64_t)t * n;
    float* o = out + (int64_t)t * ld_table;
    for (int64_t b0 = ((int64_t)blockIdx.x * gpb + gid) * FWD_UNROLL; b0 < n;
         b0 += (int64_t)gridDim.x * gpb * FWD_UNROLL) {
        int32_t s[FWD_UNROLL];
#pragma unroll
        for (int u = 0; u < FWD_UNROLL; ++u) s[u] = (b0 + u < n) ? sl[b0 + u] : -1;
        for (int cc = c; cc < D4; cc += LPR) {
            float4 v[FWD_UNROLL];
#pragma unroll
            for (int u = 0; u < FWD_UNROLL; ++u)
                if (s[u] >= 0) v[u] = weight[(row_base + s[u]) * D4 + cc];
#pragma unroll
            for (int u = 0; u < FWD_UNROLL; ++u)
                if (s[u] >= 0) *reinterpret_cast<float4*>(o + (b0 + u) * ld_bag + cc * 4) = v[u];
        }
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
// K8: backward + sparse SGD without atomics.
//   1. sort (slot, position) keys per table (LDS bitonic chunks + rank-merge passes)
//   2. one LPR-lane group per sorted position; a chunk head sums <= SEG_CH gradient rows in position
//      order; single-chunk segments update the row at once, longer ones leave partial sums
//   3. segment heads of long segments add their partials in chunk order and update the row.
// Repeated slots therefore accumulate in a fixed order: results are bitwise reproducible.
// ---------------------------------------------------------------------------------------------
#define SORT_CHUNK 8192          // keys per LDS bitonic sort (64 KiB of LDS)
#define SORT_THREADS 1024
#define SEG_CH 32

__global__ void __launch_bounds__(SORT_THREADS) k_sort_chunks(const int32_t* __restrict__ slots, int64_t n,
                                                              uint64_t* __restrict__ keys, int npow2) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
    const int t = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * SORT_CHUNK;
    const int64_t cnt = min((int64_t)SORT_CHUNK, n - base);
    for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        uint64_t k = ~0ull;
        if (i < cnt) {
            const int64_t p = base + i;
            k = ((uint64_t)(uint32_t)slots[(int64_t)t * n + p] << 32) | (uint64_t)p;
        }
        sk[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow2 / 2; i += blockDim.x) {
                // i-th compare-exchange pair of this stage
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const uint64_t a = sk[lo], b = sk[hi];
                if ((a > b) == up) { sk[lo] = b; sk[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) keys[(int64_t)t * n + base + i] = sk[i];
}

// merge sorted runs of length `run` pairwise by ranking (keys are unique: position is part of the key)
__global__ void __launch_bounds__(256) k_merge_pass(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                    int64_t n, int64_t run) {
    const int t = blockIdx.y;
    const uint64_t* a = in + (int64_t)t * n;
    uint64_t* o = out + (int64_t)t * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pair = i / (2 * run);
        const int64_t s0 = pair * 2 * run;
        const int64_t s1 = min(s0 + run, n), s2 = min(s0 + 2 * run, n);
        const uint64_t key = a[i];
        int64_t pos;
        if (i < s1) pos = i + (lower_bound_u64(a, s1, s2, key) - s1);
        else pos = (i - s1) + lower_bound_u64(a, s0, s1, key);
        o[pos] = key;
    }
}

template <int LPR, bool ARANGE>
__global__ void __launch_bounds__(256) k_bwd_chunks(const TableDesc* __restrict__ tab, int D4,
                                                    float4* __restrict__ weight, const uint64_t* __restrict__ keys,
                                                    const int64_t* __restrict__ offsets, int64_t n, int64_t n_bags,
                                                    int64_t ld_off, const float* __restrict__ grad, int64_t ld_bag,
                                                    int64_t ld_table, float lr, float4* __restrict__ partials,
                                                    int64_t pstride, int64_t* __restrict__ longlist,
                                                    int32_t* __restrict__ longcount, uint8_t* __restrict__ touched) {
    const int t = blockIdx.y;
    const int64_t row_base = tab[t].row_base;
    const int c = threadIdx.x % LPR;
    const int gpb = blockDim.x / LPR;
    const int gid = threadIdx.x / LPR;
    const uint64_t* kt = keys + (int64_t)t * n;
    const float* g = grad + (int64_t)t * ld_table;
    const int64_t* off = ARANGE ? nullptr : offsets + (int64_t)t * ld_off;
    for (int64_t p = (int64_t)blockIdx.x * gpb + gid; p < n; p += (int64_t)gridDim.x * gpb) {
        const uint64_t key = kt[p];
        const uint32_t slot = (uint32_t)(key >> 32);
        // cheap rejection: most positions are singletons or chunk interiors
        const bool head = (p == 0) || ((uint32_t)(kt[p - 1] >> 32) != slot);
        int64_t start = p;
        if (!head) {
            start = lower_bound_u64(kt, 0, p, (uint64_t)slot << 32);
            if (((p - start) % SEG_CH) != 0) continue;
        }
        const int64_t lim = min(n, p + SEG_CH + 1);
        int64_t e = p + 1;                       // end of this chunk / probe of the next key
        while (e < lim && (uint32_t)(kt[e] >> 32) == slot) ++e;
        const bool more = (e == p + SEG_CH + 1);  // segment continues past this chunk
        const int64_t cend = more ? p + SEG_CH : e;
        const bool single = head && !more;
        for (int cc = c; cc < D4; cc += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t q = p; q < cend; ++q) {
                const int64_t pos = (int64_t)(kt[q] & 0xffffffffull);
                const int64_t bag = ARANGE ? pos : bag_of(off, n_bags, pos);
                const float4 v = *reinterpret_cast<const float4*>(g + bag * ld_bag + cc * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            if (single) {
                float4 w = weight[(row_base + slot) * D4 + cc];
                w.x = fmaf(-lr, acc.x, w.x); w.y = fmaf(-lr, acc.y, w.y);
                w.z = fmaf(-lr, acc.z, w.z); w.w = fmaf(-lr, acc.w, w.w);
                weight[(row_base + slot) * D4 + cc] = w;
            } else {
                // at most two chunk heads of long segments share one SEG_CH-aligned bucket
                const int64_t pi = 2 * (p / SEG_CH) + (head ? 1 : 0);
                partials[((int64_t)t * pstride + pi) * D4 + cc] = acc;
            }
        }
        if (c == 0) {
            if (single) {
                if (touched) touched[row_base + slot] = 1;
            } else if (head) {
                const int li = atomicAdd(longcount, 1);
                longlist[li] = ((int64_t)t << 40) | p;
            }
        }
    }
}

template <int LPR>
__global__ void __launch_bounds__(256) k_bwd_long(const TableDesc* __restrict__ tab, int D4,
                                                  float4* __restrict__ weight, const uint64_t* __restrict__ keys,
                                                  int64_t n, float lr, const float4* __restrict__ partials,
                                                  int64_t pstride, const int64_t* __restrict__ longlist,
                                                  const int32_t* __restrict__ longcount, uint8_t* __restrict__ touched) {
    const int c = threadIdx.x % LPR;
    const int gpb = blockDim.x / LPR;
    const int gid = threadIdx.x / LPR;
    const int cnt = *longcount;
    for (int li = blockIdx.x * gpb + gid; li < cnt; li += gridDim.x * gpb) {
        const int64_t e = longlist[li];
        const int t = (int)(e >> 40);
        const int64_t p0 = e & (((int64_t)1 << 40) - 1);
        const uint64_t* kt = keys + (int64_t)t * n;
        const uint32_t slot = (uint32_t)(kt[p0] >> 32);
        const int64_t end = lower_bound_u64(kt, p0, n, ((uint64_t)slot + 1) << 32);
        const int64_t row = tab[t].row_base + slot;
        for (int cc = c; cc < D4; cc += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t p = p0; p < end; p += SEG_CH) {
                const int64_t pi = 2 * (p / SEG_CH) + (p == p0 ? 1 : 0);
                const float4 v = partials[((int64_t)t * pstride + pi) * D4 + cc];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            float4 w = weight[row * D4 + cc];
            w.x = fmaf(-lr, acc.x, w.x); w.y = fmaf(-lr, acc.y, w.y);
            w.z = fmaf(-lr, acc.z, w.z); w.w = fmaf(-lr, acc.w, w.w);
            weight[row * D4 + cc] = w;
        }
        if (c == 0 && touched) touched[row] = 1;
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
                                  int32_t* slots_out, int32_t* miss_pos, int32_t* miss_count, void* stream) {
    CDLRM_REQUIRE(ctx && idx && slots_out && miss_pos && miss_count, "null argument");
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
#define PROBE_CALL(L) hipLaunchKernelGGL(k_probe<L>, grid, dim3(256), 0, s, ctx->d_tab, ctx->ways, ctx->tags, idx, n, ld_idx, slots_out, ctx->d_err)
    switch (lpl) {
        case 1: PROBE_CALL(1); break;
        case 2: PROBE_CALL(2); break;
        case 4: PROBE_CALL(4); break;
        case 8: PROBE_CALL(8); break;
        default: PROBE_CALL(16); break;
    }
#undef PROBE_CALL
    hipLaunchKernelGGL(k_resolve, dim3(ctx->T), dim3(1024), 0, s, ctx->d_tab, ctx->ways, ctx->aux, slots_out, n,
                       miss_pos, miss_count, ctx->d_err);
    if (ctx->aux > 0) {
        CDLRM_REQUIRE(ctx->h_host_rows[0] != nullptr, "cdlrm_ctx_bind_host_tables first");
        const int D4 = ctx->D / 4;
        int64_t fx = cdiv((int64_t)(n < ctx->aux ? n : ctx->aux) * D4, 256);
        if (fx > 256) fx = 256;
        hipLaunchKernelGGL(k_fill_aux, dim3((unsigned)fx, (unsigned)ctx->T), dim3(256), 0, s, ctx->d_tab, ctx->ways, D4,
                           reinterpret_cast<float4*>(ctx->weight), ctx->d_host_rows, idx, n, ld_idx, miss_pos,
                           miss_count);
    }
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_embbag_fwd(cdlrm_ctx* ctx, const int32_t* slots, const int64_t* offsets, int64_t n,
                                int64_t n_bags, int64_t ld_off, float* out, int64_t ld_bag, int64_t ld_table,
                                void* stream) {
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
    if (!offsets) {
        int64_t gx = cdiv(n, (int64_t)gpb * FWD_UNROLL);
        if (gx > 65535) gx = 65535;
        dim3 grid((unsigned)gx, (unsigned)ctx->T);
#define FWD_CALL(L) hipLaunchKernelGGL(k_embbag_fwd_arange<L>, grid, dim3(256), 0, s, ctx->d_tab, D4, w, slots, n, out, ld_bag, ld_table)
        DISPATCH_LPR(lpr, FWD_CALL)
#undef FWD_CALL
    } else {
        int64_t gx = cdiv(n_bags, gpb);
        if (gx > 65535) gx = 65535;
        dim3 grid((unsigned)gx, (unsigned)ctx->T);
#define FWD_CALL(L) hipLaunchKernelGGL(k_embbag_fwd_bags<L>, grid, dim3(256), 0, s, ctx->d_tab, D4, w, slots, offsets, n, n_bags, ld_off, out, ld_bag, ld_table)
        DISPATCH_LPR(lpr, FWD_CALL)
#undef FWD_CALL
    }
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// work layout: keys A [T*n] u64 | keys B [T*n] u64 | partials [T * pstride * D] f32 | longlist [T*n/SEG_CH+T] i64 | count
static int64_t bwd_pstride(int64_t n) { return 2 * (cdiv(n, SEG_CH) + 1); }
static uint64_t align256(uint64_t v) { return (v + 255) & ~(uint64_t)255; }

extern "C" uint64_t cdlrm_embbag_bwd_work_bytes(int32_t T, int64_t n, int32_t dim) {
    uint64_t keys = align256((uint64_t)T * n * 8);
    uint64_t part = align256((uint64_t)T * bwd_pstride(n) * dim * 4);
    uint64_t ll = align256((uint64_t)T * (n / SEG_CH + 2) * 8);
    return 2 * keys + part + ll + 256;
}

extern "C" int cdlrm_embbag_bwd_sgd(cdlrm_ctx* ctx, const int32_t* slots, const int64_t* offsets, int64_t n,
                                    int64_t n_bags, int64_t ld_off, const float* grad, int64_t ld_bag,
                                    int64_t ld_table, float lr, void* work, uint8_t* touched, void* stream) {
    CDLRM_REQUIRE(ctx && slots && grad && work, "null argument");
    CDLRM_REQUIRE(ctx->weight, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(((uintptr_t)grad & 15) == 0 && ld_bag % 4 == 0 && ld_table % 4 == 0 && ((uintptr_t)work & 255) == 0,
                  "aligned grad rows / work");
    CDLRM_REQUIRE(offsets != nullptr || n_bags == n, "Criteo layout needs n_bags == n");
    CDLRM_REQUIRE(n < ((int64_t)1 << 31) && ctx->T < (1 << 20), "n < 2^31");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return 0;
    const int T = ctx->T, D4 = ctx->D / 4;
    const int lpr = lanes_per_row(D4);
    const int gpb = 256 / lpr;
    char* wp = (char*)work;
    uint64_t* keysA = (uint64_t*)wp; wp += align256((uint64_t)T * n * 8);
    uint64_t* keysB = (uint64_t*)wp; wp += align256((uint64_t)T * n * 8);
    const int64_t pstride = bwd_pstride(n);
    float4* partials = (float4*)wp; wp += align256((uint64_t)T * pstride * ctx->D * 4);
    int64_t* longlist = (int64_t*)wp; wp += align256((uint64_t)T * (n / SEG_CH + 2) * 8);
    int32_t* longcount = (int32_t*)wp;
    CDLRM_HIP_CHECK(hipMemsetAsync(longcount, 0, sizeof(int32_t), s));
    // 1. sort
    const int64_t nchunks = cdiv(n, SORT_CHUNK);
    int npow2 = pow2ceil((int)(n < SORT_CHUNK ? n : SORT_CHUNK));
    if (npow2 < 2) npow2 = 2;
    static bool attr_set = false;
    if (!attr_set) {
        CDLRM_HIP_CHECK(hipFuncSetAttribute((const void*)k_sort_chunks, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            SORT_CHUNK * 8));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_sort_chunks, dim3((unsigned)nchunks, (unsigned)T), dim3(SORT_THREADS), (size_t)npow2 * 8, s,
                       slots, n, keysA, npow2);
    uint64_t* cur = keysA;
    uint64_t* alt = keysB;
    for (int64_t run = SORT_CHUNK; run < n; run *= 2) {
        int64_t gx = cdiv(n, 256);
        if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(k_merge_pass, dim3((unsigned)gx, (unsigned)T), dim3(256), 0, s, cur, alt, n, run);
        uint64_t* tmp = cur; cur = alt; alt = tmp;
    }
    // 2. chunk sums / single-chunk updates
    {
        int64_t gx = cdiv(n, gpb);
        if (gx > 65535) gx = 65535;
        dim3 grid((unsigned)gx, (unsigned)T);
        float4* w = reinterpret_cast<float4*>(ctx->weight);
#define BWD_CALL(L)                                                                                             \
    if (offsets)                                                                                                \
        hipLaunchKernelGGL((k_bwd_chunks<L, false>), grid, dim3(256), 0, s, ctx->d_tab, D4, w, cur, offsets, n, \
                           n_bags, ld_off, grad, ld_bag, ld_table, lr, partials, pstride, longlist, longcount,  \
                           touched);                                                                            \
    else                                                                                                        \
        hipLaunchKernelGGL((k_bwd_chunks<L, true>), grid, dim3(256), 0, s, ctx->d_tab, D4, w, cur, offsets, n,  \
                           n_bags, ld_off, grad, ld_bag, ld_table, lr, partials, pstride, longlist, longcount,  \
                           touched)
        DISPATCH_LPR(lpr, BWD_CALL)
#undef BWD_CALL
        // 3. long segments
        int64_t lx = cdiv((int64_t)T * (n / SEG_CH + 1), gpb);
        if (lx > 1024) lx = 1024;
#define LONG_CALL(L) hipLaunchKernelGGL(k_bwd_long<L>, dim3((unsigned)lx), dim3(256), 0, s, ctx->d_tab, D4, w, cur, n, lr, partials, pstride, longlist, longcount, touched)
        DISPATCH_LPR(lpr, LONG_CALL)
#undef LONG_CALL
    }
    CDLRM_LAUNCH_CHECK();
    return 0;
}
