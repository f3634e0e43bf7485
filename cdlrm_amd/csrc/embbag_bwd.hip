// K8: EmbeddingBag backward + sparse SGD on the cache rows, fused and atomics-free (gfx950).
// Reference: nn.EmbeddingBag(sparse=True) backward + optim.SGD step, main_no_ddp.py:376, 409, 413.
//
//   prepare (depends only on the slot ids, so it runs right after the probe, under the MLPs):
//     1. sort (slot, position) keys per table: LDS bitonic chunks (+ rank-merge passes when a table has
//        more than SORT_CHUNK lookups)
//     2. meta[p] = distance of sorted position p from the start of its run of equal slots
//   apply:
//     3. one LPR-lane group (16 B per lane) per sorted position; only chunk heads (meta % SEG_CH == 0) work:
//        they add <= SEG_CH gradient rows in position order; a run that fits one chunk updates its row at
//        once (W[slot] += -lr * sum), longer runs leave per-chunk partial sums
//     4. heads of long runs add their partials in chunk order and update the row.
// Repeated slots therefore accumulate in a fixed order: bitwise reproducible, no float atomics
// (cdna_hip_programming.md Guideline 12 / Appendix B "store pass + per-destination sum pass").
// HBM-bound: algorithmic bytes per lookup 4D (grad) + 2*4D (row read-modify-write) + 8.
#include "common.h"

#define SORT_CHUNK 8192          // keys per LDS bitonic sort (64 KiB of LDS)
#define SORT_THREADS 1024
#define SEG_CH 32

__global__ void __launch_bounds__(SORT_THREADS) k_sort_chunks(const int32_t* __restrict__ slots, int64_t n,
                                                              uint64_t* __restrict__ keys, int32_t* __restrict__ meta,
                                                              int npow2, int write_meta) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
    __shared__ int wmax[16];
    const int t = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * SORT_CHUNK;
    const int cnt = (int)min((int64_t)SORT_CHUNK, n - base);
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
    if (!write_meta) return;
    // run starts by an inclusive max-scan of head positions: thread owns E consecutive sorted keys
    const int E = (npow2 + SORT_THREADS - 1) / SORT_THREADS;
    const int i0 = threadIdx.x * E;
    int local = -1;     // last head position inside my range
    for (int e = 0; e < E; ++e) {
        const int i = i0 + e;
        if (i < cnt && (i == 0 || (uint32_t)(sk[i] >> 32) != (uint32_t)(sk[i - 1] >> 32))) local = i;
    }
    int inc = local;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc = max(inc, o);
    }
    if (lane == 63) wmax[wid] = inc;
    __syncthreads();
    int pre = -1;       // max over all previous threads
    for (int w = 0; w < wid; ++w) pre = max(pre, wmax[w]);
    const int up = __shfl_up(inc, 1, 64);
    if (lane > 0) pre = max(pre, up);
    int run = pre;
    for (int e = 0; e < E; ++e) {
        const int i = i0 + e;
        if (i >= cnt) break;
        if (i == 0 || (uint32_t)(sk[i] >> 32) != (uint32_t)(sk[i - 1] >> 32)) run = i;
        meta[(int64_t)t * n + base + i] = i - run;
    }
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

// meta for tables sorted in several chunks: distance to the run start by binary search
__global__ void __launch_bounds__(256) k_seg_meta(const uint64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ meta) {
    const int t = blockIdx.y;
    const uint64_t* kt = keys + (int64_t)t * n;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t s = kt[p] >> 32;
        int32_t r = 0;
        if (p > 0 && (kt[p - 1] >> 32) == s) r = (int32_t)(p - lower_bound_u64(kt, 0, p, s << 32));
        meta[(int64_t)t * n + p] = r;
    }
}

template <int LPR, bool ARANGE>
__global__ void __launch_bounds__(256) k_bwd_chunks(const TableDesc* __restrict__ tab, int D4,
                                                    float4* __restrict__ weight, const uint64_t* __restrict__ keys,
                                                    const int32_t* __restrict__ meta,
                                                    const int64_t* __restrict__ offsets, int64_t n, int64_t n_bags,
                                                    int64_t ld_off, const float* __restrict__ grad, int64_t ld_bag,
                                                    int64_t ld_table, float lr, float4* __restrict__ partials,
                                                    int64_t pstride, int64_t* __restrict__ longlist,
                                                    int32_t* __restrict__ longcount, uint8_t* __restrict__ touched,
                                                    int64_t aux_total) {
    constexpr int KM = (SEG_CH + LPR - 1) / LPR;
    const int t = blockIdx.y;
    const int64_t row_base = tab[t].row_base;
    // aux rows (transient copies of host rows, rewritten by every forward) are never flagged: the cross-rank row merge
    // leaves them alone, so it cannot collide with the NEXT batch's aux fill running ahead in the other aux region
    const uint32_t first_aux = (uint32_t)(tab[t].rows - aux_total);
    const int c = threadIdx.x % LPR;
    const int gpb = blockDim.x / LPR;
    const int gid = threadIdx.x / LPR;
    const int gshift = ((threadIdx.x & 63) / LPR) * LPR;
    const uint64_t* kt = keys + (int64_t)t * n;
    const int32_t* mt = meta + (int64_t)t * n;
    const float* g = grad + (int64_t)t * ld_table;
    const int64_t* off = ARANGE ? nullptr : offsets + (int64_t)t * ld_off;
    for (int64_t p = (int64_t)blockIdx.x * gpb + gid; p < n; p += (int64_t)gridDim.x * gpb) {
        const int r0 = mt[p];
        if (r0 % SEG_CH) continue;               // chunk interior: some other group owns this position
        const bool head = r0 == 0;
        const uint32_t slot = (uint32_t)(kt[p] >> 32);
        // first position after p (within the chunk window) that starts another run, found by the group at once
        int jstop = SEG_CH;                      // SEG_CH: the run continues past this chunk
#pragma unroll
        for (int m = KM - 1; m >= 0; --m) {
            const int j = c + m * LPR;
            const int64_t q = p + 1 + j;
            const bool stop = (j < SEG_CH) && (q >= n || mt[q] == 0);
            const unsigned long long b = __ballot(stop);
            const unsigned long long bits = LPR == 64 ? b : ((b >> gshift) & ((1ull << LPR) - 1));
            if (bits) jstop = m * LPR + (__ffsll((long long)bits) - 1);
        }
        const bool more = jstop == SEG_CH;
        const int len = more ? SEG_CH : jstop + 1;
        const bool single = head && !more;
        for (int cc = c; cc < D4; cc += LPR) {
            float4 w;
            if (single) w = weight[(row_base + slot) * D4 + cc];     // issued early, consumed after the sums
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int q = 0;
            for (; q + 4 <= len; q += 4) {
                int64_t bag[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t pos = (int64_t)(kt[p + q + u] & 0xffffffffull);
                    bag[u] = ARANGE ? pos : bag_of(off, n_bags, pos);
                }
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(g + bag[u] * ld_bag + cc * 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; q < len; ++q) {
                const int64_t pos = (int64_t)(kt[p + q] & 0xffffffffull);
                const int64_t bag = ARANGE ? pos : bag_of(off, n_bags, pos);
                const float4 v = *reinterpret_cast<const float4*>(g + bag * ld_bag + cc * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            if (single) {
                w.x = fmaf(-lr, acc.x, w.x); w.y = fmaf(-lr, acc.y, w.y);
                w.z = fmaf(-lr, acc.z, w.z); w.w = fmaf(-lr, acc.w, w.w);
                weight[(row_base + slot) * D4 + cc] = w;
            } else {
                // at most two chunk heads of long runs share one SEG_CH-aligned bucket of positions
                const int64_t pi = 2 * (p / SEG_CH) + (head ? 1 : 0);
                partials[((int64_t)t * pstride + pi) * D4 + cc] = acc;
            }
        }
        if (c == 0) {
            if (single) {
                if (touched && slot < first_aux) touched[row_base + slot] = 1;
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
                                                  const int32_t* __restrict__ longcount, uint8_t* __restrict__ touched,
                                                  int64_t aux_total) {
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
            float4 w = weight[row * D4 + cc];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t p = p0; p < end; p += SEG_CH) {
                const int64_t pi = 2 * (p / SEG_CH) + (p == p0 ? 1 : 0);
                const float4 v = partials[((int64_t)t * pstride + pi) * D4 + cc];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            w.x = fmaf(-lr, acc.x, w.x); w.y = fmaf(-lr, acc.y, w.y);
            w.z = fmaf(-lr, acc.z, w.z); w.w = fmaf(-lr, acc.w, w.w);
            weight[row * D4 + cc] = w;
        }
        if (c == 0 && touched && (int64_t)slot < tab[t].rows - aux_total) touched[row] = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int lanes_per_row_b(int D4) { int l = pow2ceil(D4); return l > 64 ? 64 : (l < 4 ? 4 : l); }

#define DISPATCH_LPR_B(lpr, CALL)               \
    switch (lpr) {                              \
        case 4: { CALL(4); break; }             \
        case 8: { CALL(8); break; }             \
        case 16: { CALL(16); break; }           \
        case 32: { CALL(32); break; }           \
        default: { CALL(64); break; }           \
    }

// work layout: keys A [T*n] u64 | keys B [T*n] u64 | meta [T*n] i32 | partials [T*pstride*D] f32 |
//              longlist [T*(n/SEG_CH+2)] i64 | sorted-buffer selector + long-run counter
static int64_t bwd_pstride(int64_t n) { return 2 * (cdiv(n, SEG_CH) + 1); }
static uint64_t align256(uint64_t v) { return (v + 255) & ~(uint64_t)255; }

struct BwdWork {
    uint64_t *keysA, *keysB;
    int32_t* meta;
    float4* partials;
    int64_t* longlist;
    int32_t* longcount;
    int64_t pstride;
};

static BwdWork carve(void* work, int T, int64_t n, int D) {
    BwdWork w;
    char* wp = (char*)work;
    w.keysA = (uint64_t*)wp; wp += align256((uint64_t)T * n * 8);
    w.keysB = (uint64_t*)wp; wp += align256((uint64_t)T * n * 8);
    w.meta = (int32_t*)wp; wp += align256((uint64_t)T * n * 4);
    w.pstride = bwd_pstride(n);
    w.partials = (float4*)wp; wp += align256((uint64_t)T * w.pstride * D * 4);
    w.longlist = (int64_t*)wp; wp += align256((uint64_t)T * (n / SEG_CH + 2) * 8);
    w.longcount = (int32_t*)wp;
    return w;
}

extern "C" uint64_t cdlrm_embbag_bwd_work_bytes(int32_t T, int64_t n, int32_t dim) {
    return 2 * align256((uint64_t)T * n * 8) + align256((uint64_t)T * n * 4) +
           align256((uint64_t)T * bwd_pstride(n) * dim * 4) + align256((uint64_t)T * (n / SEG_CH + 2) * 8) + 256;
}

// number of rank-merge passes decides which key buffer ends up sorted
static bool sorted_in_B(int64_t n) {
    int passes = 0;
    for (int64_t run = SORT_CHUNK; run < n; run *= 2) ++passes;
    return passes & 1;
}

extern "C" int cdlrm_embbag_bwd_prepare(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, void* work, void* stream) {
    CDLRM_REQUIRE(ctx && slots && work, "null argument");
    CDLRM_REQUIRE(((uintptr_t)work & 255) == 0, "work must be 256-byte aligned");
    CDLRM_REQUIRE(n < ((int64_t)1 << 31) && ctx->T < (1 << 20), "n < 2^31");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return 0;
    const int T = ctx->T;
    BwdWork w = carve(work, T, n, ctx->D);
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
                       slots, n, w.keysA, w.meta, npow2, nchunks == 1 ? 1 : 0);
    uint64_t* cur = w.keysA;
    uint64_t* alt = w.keysB;
    for (int64_t run = SORT_CHUNK; run < n; run *= 2) {
        int64_t gx = cdiv(n, 256);
        if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(k_merge_pass, dim3((unsigned)gx, (unsigned)T), dim3(256), 0, s, cur, alt, n, run);
        uint64_t* tmp = cur; cur = alt; alt = tmp;
    }
    if (nchunks > 1) {
        int64_t gx = cdiv(n, 256);
        if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(k_seg_meta, dim3((unsigned)gx, (unsigned)T), dim3(256), 0, s, cur, n, w.meta);
    }
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_embbag_bwd_apply(cdlrm_ctx* ctx, const int64_t* offsets, int64_t n, int64_t n_bags, int64_t ld_off,
                                      const float* grad, int64_t ld_bag, int64_t ld_table, float lr, void* work,
                                      uint8_t* touched, void* stream) {
    CDLRM_REQUIRE(ctx && grad && work, "null argument");
    CDLRM_REQUIRE(ctx->weight, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(((uintptr_t)grad & 15) == 0 && ld_bag % 4 == 0 && ld_table % 4 == 0 && ((uintptr_t)work & 255) == 0,
                  "aligned grad rows / work");
    CDLRM_REQUIRE(offsets != nullptr || n_bags == n, "Criteo layout needs n_bags == n");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return 0;
    const int T = ctx->T, D4 = ctx->D / 4;
    const int lpr = lanes_per_row_b(D4);
    const int gpb = 256 / lpr;
    BwdWork w = carve(work, T, n, ctx->D);
    const uint64_t* cur = sorted_in_B(n) ? w.keysB : w.keysA;
    CDLRM_HIP_CHECK(hipMemsetAsync(w.longcount, 0, sizeof(int32_t), s));
    int64_t gx = cdiv(n, gpb);
    if (gx > 65535) gx = 65535;
    dim3 grid((unsigned)gx, (unsigned)T);
    float4* wt = reinterpret_cast<float4*>(ctx->weight);
    const int64_t aux_total = (int64_t)ctx->aux * ctx->aux_phases;
#define BWD_CALL(L)                                                                                                \
    if (offsets)                                                                                                   \
        hipLaunchKernelGGL((k_bwd_chunks<L, false>), grid, dim3(256), 0, s, ctx->d_tab, D4, wt, cur, w.meta, offsets, n, \
                           n_bags, ld_off, grad, ld_bag, ld_table, lr, w.partials, w.pstride, w.longlist, w.longcount,  \
                           touched, aux_total);                                                                    \
    else                                                                                                           \
        hipLaunchKernelGGL((k_bwd_chunks<L, true>), grid, dim3(256), 0, s, ctx->d_tab, D4, wt, cur, w.meta, offsets, n,  \
                           n_bags, ld_off, grad, ld_bag, ld_table, lr, w.partials, w.pstride, w.longlist, w.longcount,  \
                           touched, aux_total)
    DISPATCH_LPR_B(lpr, BWD_CALL)
#undef BWD_CALL
    int64_t lx = cdiv((int64_t)T * (n / SEG_CH + 1), gpb);
    if (lx > 1024) lx = 1024;
#define LONG_CALL(L) hipLaunchKernelGGL(k_bwd_long<L>, dim3((unsigned)lx), dim3(256), 0, s, ctx->d_tab, D4, wt, cur, n, lr, w.partials, w.pstride, w.longlist, w.longcount, touched, aux_total)
    DISPATCH_LPR_B(lpr, LONG_CALL)
#undef LONG_CALL
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_embbag_bwd_sgd(cdlrm_ctx* ctx, const int32_t* slots, const int64_t* offsets, int64_t n,
                                    int64_t n_bags, int64_t ld_off, const float* grad, int64_t ld_bag,
                                    int64_t ld_table, float lr, void* work, uint8_t* touched, void* stream) {
    int rc = cdlrm_embbag_bwd_prepare(ctx, slots, n, work, stream);
    if (rc) return rc;
    return cdlrm_embbag_bwd_apply(ctx, offsets, n, n_bags, ld_off, grad, ld_bag, ld_table, lr, work, touched, stream);
}
