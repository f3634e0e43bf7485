// Reduce-then-scan stream compaction for gfx950.  Off the per-iteration critical path (window plan,
// table-agg), so it favours simplicity: 16 flags per lane per 16-byte load, wave-scan via shuffles.
#include "scan.h"

#define CF_THREADS 256
#define CF_PER_THREAD 16
#define CF_PER_BLOCK (CF_THREADS * CF_PER_THREAD)

int cdlrm_scan_reserve(cdlrm_ctx* ctx, int64_t nblocks) {
    if (nblocks <= ctx->scan_cap) return 0;
    if (ctx->d_scan) (void)hipFree(ctx->d_scan);
    ctx->d_scan = nullptr;
    ctx->scan_cap = 0;
    CDLRM_HIP_CHECK(hipMalloc(&ctx->d_scan, sizeof(int64_t) * (nblocks + 1)));
    ctx->scan_cap = nblocks;
    return 0;
}

int cdlrm_scan_reserve_agg(cdlrm_ctx* ctx, int64_t nblocks) {
    if (nblocks <= ctx->scan_agg_cap) return 0;
    if (ctx->d_scan_agg) (void)hipFree(ctx->d_scan_agg);      // hipFree waits for the kernels still reading it
    ctx->d_scan_agg = nullptr;
    ctx->scan_agg_cap = 0;
    CDLRM_HIP_CHECK(hipMalloc(&ctx->d_scan_agg, sizeof(int64_t) * (nblocks + 1)));
    ctx->scan_agg_cap = nblocks;
    return 0;
}

__device__ __forceinline__ int load_flags16(const uint8_t* flags, int64_t base, int64_t n, uint8_t f[CF_PER_THREAD]) {
    int cnt = 0;
    if (base + CF_PER_THREAD <= n) {
        uint4 v = *reinterpret_cast<const uint4*>(flags + base);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f[i] = (w[i >> 2] >> ((i & 3) * 8)) & 0xff;
            cnt += f[i] != 0;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f[i] = (base + i < n) ? flags[base + i] : 0;
            cnt += f[i] != 0;
        }
    }
    return cnt;
}

__global__ void __launch_bounds__(CF_THREADS) k_cf_count(const uint8_t* flags, const int64_t* d_n, int64_t n_max,
                                                         int64_t* sums) {
    __shared__ int smem[32];
    const int64_t n = d_n ? min(d_n[0], n_max) : n_max;
    const int64_t base = ((int64_t)blockIdx.x * CF_THREADS + threadIdx.x) * CF_PER_THREAD;
    uint8_t f[CF_PER_THREAD];
    int cnt = (base < n) ? load_flags16(flags, base, n, f) : 0;
    int total;
    block_excl_scan(cnt, smem, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) k_scan_tops(int64_t* sums, int64_t n, int64_t* d_total) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int64_t base = 0; base < n; base += 1024) {
        int64_t i = base + threadIdx.x;
        int64_t v = i < n ? sums[i] : 0;
        int64_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int64_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        int64_t woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        int64_t carry = carry_s;
        if (i < n) sums[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0 && d_total) *d_total = carry_s;
}

__global__ void __launch_bounds__(CF_THREADS) k_cf_emit(uint8_t* flags, const int64_t* d_n, int64_t n_max,
                                                        const int64_t* sums, int32_t* out32, int64_t* out64,
                                                        int64_t cap, int clear, int* err) {
    __shared__ int smem[32];
    const int64_t n = d_n ? min(d_n[0], n_max) : n_max;
    const int64_t base = ((int64_t)blockIdx.x * CF_THREADS + threadIdx.x) * CF_PER_THREAD;
    if ((int64_t)blockIdx.x * CF_PER_BLOCK >= n) return;
    uint8_t f[CF_PER_THREAD];
    int cnt = (base < n) ? load_flags16(flags, base, n, f) : 0;
    int total;
    int ex = block_excl_scan(cnt, smem, &total);
    int64_t pos = sums[blockIdx.x] + ex;
    if (cnt) {
#pragma unroll
        for (int i = 0; i < CF_PER_THREAD; ++i) {
            if (f[i]) {
                if (pos < cap) {
                    if (out32) out32[pos] = (int32_t)(base + i);
                    if (out64) out64[pos] = base + i;
                } else if (!(clear & 2)) {
                    atomicOr(err, 4);
                }
                ++pos;
            }
        }
        if (clear & 1) {
            if (base + CF_PER_THREAD <= n) {
                *reinterpret_cast<uint4*>(flags + base) = make_uint4(0, 0, 0, 0);
            } else {
                for (int i = 0; i < CF_PER_THREAD; ++i)
                    if (base + i < n) flags[base + i] = 0;
            }
        }
    }
}

int cdlrm_compact_flags(cdlrm_ctx* ctx, uint8_t* flags, const int64_t* d_n, int64_t n_max, int32_t* out32,
                        int64_t* out64, int64_t cap, int64_t* d_count, int clear, hipStream_t s, int64_t* scratch,
                        int64_t scratch_cap) {
    if (n_max <= 0) {
        if (d_count) CDLRM_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(int64_t), s));
        return 0;
    }
    CDLRM_REQUIRE(((uintptr_t)flags & 15) == 0, "flags must be 16-byte aligned");
    const int64_t nblocks = cdiv(n_max, CF_PER_BLOCK);
    if (!scratch) { scratch = ctx->d_scan; scratch_cap = ctx->scan_cap; }
    CDLRM_REQUIRE(nblocks <= scratch_cap, "scan scratch too small (cdlrm_scan_reserve)");
    hipLaunchKernelGGL(k_cf_count, dim3((unsigned)nblocks), dim3(CF_THREADS), 0, s, flags, d_n, n_max, scratch);
    hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(1024), 0, s, scratch, nblocks, d_count);
    hipLaunchKernelGGL(k_cf_emit, dim3((unsigned)nblocks), dim3(CF_THREADS), 0, s, flags, d_n, n_max, scratch,
                       out32, out64, cap, clear, ctx->d_err);
    CDLRM_LAUNCH_CHECK();
    return 0;
}
