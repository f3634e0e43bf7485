// FP32-MFMA tiled GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, fp32 vector peak).
//
//   C[m,n] = sum_k A(m,k) * B(k,n)      block tile (64*TM) x (64*TN) x 32, 4 waves as 2x2, each wave
//   TM x TN MFMA tiles of 32x32; global -> register prefetch -> LDS staging, one barrier pair per K tile.
//   A_KC: A(m,k) = A[m*lda + k]  (contraction contiguous)   else A[k*lda + m]
//   B_KC: B(k,n) = B[n*ldb + k]                             else B[k*ldb + n]
// LDS images: contraction-contiguous operands as [rows][32+4] (pitch = odd multiple of 16 B: ds_write_b128
// staging and ds_read_b128 fragment fetches are both conflict-free), the others as [32][rows].
// The tile shape is picked per call so that the grid has >= 2 workgroups per CU: with one 4-wave
// workgroup per CU every SIMD holds a single wave and nothing hides its barrier / staging stalls.
// Measured (tools/gemm_ablate.hip, 8192x512x512): 80 TF with staging, 107 TF with the staging ablated; the vendor
// sgemm reaches 95-113 TF on the forward/dgrad shapes and 40-77 TF on the wgrad shapes, i.e. the MLP as a whole
// runs at the library's speed.  Tried without gain: LDS double-buffering, 2-stage register prefetch, BK=64,
// step-major MFMA order, two accumulators per wave in the 64x64 tile (two MFMA dependency chains: 90.5 vs 93.7 TF).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef GBK
#define GBK 32
#endif
#define LDS_KC (GBK + 4)   // pitch = odd multiple of 16 B: aligned rows, conflict-free ds_read_b128 / ds_write_b128

struct GemmArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc; int64_t slab;     // slab: elements between split-K outputs
    int64_t M; int N; int64_t K; int64_t kchunk;
    const float* bias; int act;               // epilogue: + bias[n], 1 ReLU, 2 sigmoid
    int vecA, vecB;                           // 16-byte loads legal
    // fused backward pieces
    const float* mask; int64_t ldmask; int mask_act;   // epilogue: C *= act'(mask[m,n]) (1 ReLU output, 2 sigmoid output)
    float* colsum;                            // !A_KC only: colsum[z*M + m] = sum over this split's k of A(m,k)
    int vecC;                                 // 16-byte stores legal (k_gemm_direct)
    int fastep;                               // epilogue operands fetched ahead of their use: k_gemm2's loads-first epilogue of full
                                              // tiles, the short-batch kernels' prefetch at kernel start (direct_prefetch)
    int alone;                                // the caller's hint CDLRM_GEMM_ALONE: no other GEMM runs beside this launch (the wide
                                              // kernel of gemm_wide.h wants a whole CU: 96 KB of LDS, one wave per SIMD)
};

static inline GemmArgs gemm_args() {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.fastep = (g_cdlrm_debug[7] & 1) ? 0 : 1;       // (development switch: the epilogues before round 5)
    return g;
}

// Operand staging.  Steady-state tiles (FULLK) are loaded with NO predicate: rows past the matrix edge are
// clamped to the last valid row (their products land in output rows/columns that are never stored), so the
// loads stay in flight across the MFMA loop.  Only the last, partial K tile zero-fills (at LDS-store time).
template <bool KC, int ROWS, bool VEC>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int64_t ld, int64_t r0, int64_t rmax,
                                          int64_t k0, int64_t kmax, float4 (&v)[ROWS * GBK / 1024]) {
    // STRAIGHT-LINE code on purpose: no branch, wave-uniform or not, may surround these loads.  hipcc places an
    // s_waitcnt vmcnt(0) at the join of any branch that contains loads, which would drain the prefetch before the
    // MFMA loop instead of letting it fly underneath (seen in the ISA: every K tile then pays a full load latency).
    // Out-of-range rows / contraction indices are clamped to valid addresses; the partial last tile is zero-filled
    // when it is written to LDS.
#pragma unroll
    for (int i = 0; i < ROWS * GBK / 1024; ++i) {
        const int f = threadIdx.x + i * 256;
        if (KC) {       // ROWS rows of the non-contraction index, GBK contraction elements per row
            const int64_t r = f / (GBK / 4), c = (f % (GBK / 4)) * 4;
            const int64_t rr = min(r0 + r, rmax - 1);
            if (VEC) {
                v[i] = *reinterpret_cast<const float4*>(P + rr * ld + min(k0 + c, kmax - 4));
            } else {
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) e[u] = P[rr * ld + min(k0 + c + u, kmax - 1)];
                v[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        } else {        // GBK rows of the contraction index, ROWS non-contraction elements per row
            const int64_t c = f / (ROWS / 4), r = (f % (ROWS / 4)) * 4;
            const int64_t kk = min(k0 + c, kmax - 1);
            if (VEC) {
                v[i] = *reinterpret_cast<const float4*>(P + kk * ld + min(r0 + r, rmax - 4));
            } else {
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) e[u] = P[kk * ld + min(r0 + r + u, rmax - 1)];
                v[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    }
}

template <bool KC, int ROWS>
__device__ __forceinline__ void tile_store(float* __restrict__ S, const float4 (&v)[ROWS * GBK / 1024], int64_t k0,
                                           int64_t kmax) {
#pragma unroll
    for (int i = 0; i < ROWS * GBK / 1024; ++i) {
        const int f = threadIdx.x + i * 256;
        float4 x = v[i];
        if (KC) {
            const int r = f / (GBK / 4), c = (f % (GBK / 4)) * 4;
            x.x = (k0 + c + 0 < kmax) ? x.x : 0.f;      // selects, not branches
            x.y = (k0 + c + 1 < kmax) ? x.y : 0.f;
            x.z = (k0 + c + 2 < kmax) ? x.z : 0.f;
            x.w = (k0 + c + 3 < kmax) ? x.w : 0.f;
            *reinterpret_cast<float4*>(S + r * LDS_KC + c) = x;
        } else {
            const int c = f / (ROWS / 4), r = (f % (ROWS / 4)) * 4;
            const bool ok = k0 + c < kmax;
            x.x = ok ? x.x : 0.f; x.y = ok ? x.y : 0.f; x.z = ok ? x.z : 0.f; x.w = ok ? x.w : 0.f;
            *reinterpret_cast<float4*>(S + c * ROWS + r) = x;
        }
    }
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an L2), so give
// each XCD a contiguous run of logical tiles: the tiles of one row panel then re-read A from their own L2
// instead of fetching it once per XCD.  Bijective for any grid size; placement only affects speed.
__device__ __forceinline__ unsigned xcd_remap(unsigned orig, unsigned nwg) {
    const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = orig & 7;
    return (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (orig >> 3);
}

// Several independent GEMMs of one operand layout in ONE launch (the weight gradients of all layers of the MLPs: at
// small batches every launch costs ~10 us of fixed latency against a few us of MFMA work, and one problem alone
// cannot fill the chip).  Workgroup -> problem by the running workgroup count.
#define GEMM_GROUP_MAX 8
struct GemmGroup {
    int n;
    unsigned first[GEMM_GROUP_MAX + 1];      // first[p] .. first[p+1]: workgroups of problem p (x fastest, then y, then z)
    GemmArgs g[GEMM_GROUP_MAX];
};

template <bool A_KC, bool B_KC, int TM, int TN, bool VA, bool VB>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs& g, unsigned bx, unsigned by, unsigned bz,
                                               float* __restrict__ As, float* __restrict__ Bs) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)by * BM;
    const int64_t n0 = (int64_t)bx * BN;
    const int64_t kbeg = (int64_t)bz * g.kchunk;
    const int64_t kend = min(g.K, kbeg + g.kchunk);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[BM * GBK / 1024], rb[BN * GBK / 1024];
    // bias gradient fused into the weight-gradient GEMM: the first column panel also sums its A tiles (= dZ^T) over
    // the contraction (VALU only, under the MFMAs)
    const bool do_colsum = !A_KC && g.colsum != nullptr && bx == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    tile_load<A_KC, BM, VA>(g.A, g.lda, m0, g.M, kbeg, kend, ra);
    tile_load<B_KC, BN, VB>(g.B, g.ldb, n0, g.N, kbeg, kend, rb);
#ifndef GEMM_ABLATE
#define GEMM_ABLATE 0      // tools/gemm_ablate.hip only: 1 no global loads, 2 no LDS staging, 3 neither
#endif
    // (LDS double-buffering with one barrier per tile was tried and measured SLOWER on these shapes -- 73.6 vs
    //  80 TF at 8192x512x512, 20 vs 26 TF at M=1024: the doubled LDS footprint costs more occupancy than the saved
    //  barrier buys.)
    for (int64_t k0 = kbeg; k0 < kend; k0 += GBK) {
        if (!(GEMM_ABLATE & 2) || k0 == kbeg) {
            __syncthreads();
            tile_store<A_KC, BM>(As, ra, k0, kend);
            tile_store<B_KC, BN>(Bs, rb, k0, kend);
            __syncthreads();
        }
        if (!A_KC && do_colsum) {       // wave-uniform; no loads inside
#pragma unroll
            for (int i = 0; i < BM * GBK / 1024; ++i) {
                const int c = (threadIdx.x + i * 256) / (BM / 4);
                const bool ok = k0 + c < kend;
                csum.x += ok ? ra[i].x : 0.f; csum.y += ok ? ra[i].y : 0.f;
                csum.z += ok ? ra[i].z : 0.f; csum.w += ok ? ra[i].w : 0.f;
            }
        }
        if (!(GEMM_ABLATE & 1)) {   // unconditional: past the last tile the clamped addresses just re-read valid data
            tile_load<A_KC, BM, VA>(g.A, g.lda, m0, g.M, k0 + GBK, kend, ra);
            tile_load<B_KC, BN, VB>(g.B, g.ldb, n0, g.N, k0 + GBK, kend, rb);
        }
        // One MFMA consumes 2 contraction indices (lanes 0-31 the first, lanes 32-63 the second).  Any pairing works
        // as long as A and B agree, so an 8-index group is consumed in 4 steps with lane-half lk holding indices
        // 4*lk .. 4*lk+3: contraction-contiguous operands then need ONE ds_read_b128 per 4 MFMA steps.
        const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
        for (int kg = 0; kg < GBK / 8; ++kg) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * (32 * TM) + i * 32 + lr;
                if (A_KC) {
                    a[i] = *reinterpret_cast<const float4*>(As + r * LDS_KC + kg * 8 + 4 * lk);
                } else {
                    const float* q = As + (kg * 8 + 4 * lk) * BM + r;
                    a[i] = make_float4(q[0], q[BM], q[2 * BM], q[3 * BM]);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int c = wn * (32 * TN) + j * 32 + lr;
                if (B_KC) {
                    b[j] = *reinterpret_cast<const float4*>(Bs + c * LDS_KC + kg * 8 + 4 * lk);
                } else {
                    const float* q = Bs + (kg * 8 + 4 * lk) * BN + c;
                    b[j] = make_float4(q[0], q[BN], q[2 * BN], q[3 * BN]);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
    if (!A_KC && do_colsum) {
        // the BM/4 threads with equal (tid % (BM/4)) hold the same 4 rows for different contraction indices:
        // combine them through LDS in a fixed order
        __syncthreads();                        // every wave is done reading As
        const int q = threadIdx.x / (BM / 4), r4 = (threadIdx.x % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(As + q * BM + r4) = csum;
        __syncthreads();
        if (threadIdx.x < BM && m0 + threadIdx.x < g.M) {
            float s = 0.f;
#pragma unroll
            for (int qq = 0; qq < 1024 / BM; ++qq) s += As[qq * BM + threadIdx.x];
            g.colsum[(int64_t)bz * g.M + m0 + threadIdx.x] = s;
        }
    }
    float* C = g.C + (int64_t)bz * g.slab;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t col = n0 + wn * (32 * TN) + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * (32 * TM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= g.M) continue;
                float v = acc[i][j][r] + bv;
                if (g.act == 1) v = v > 0.f ? v : 0.f;
                else if (g.act == 2) v = 1.0f / (1.0f + expf(-v));
                if (g.mask_act) {               // activation backward of the layer below, fused into the dgrad
                    const float x = g.mask[row * g.ldmask + col];
                    v = g.mask_act == 1 ? (x > 0.f ? v : 0.f) : v * ((1.0f - x) * x);
                }
                C[row * g.ldc + col] = v;
            }
        }
}

template <bool A_KC, bool B_KC, int TM, int TN, bool VA, bool VB>
__global__ void __launch_bounds__(256) k_gemm(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[64 * TM * LDS_KC];
    __shared__ __attribute__((aligned(16))) float Bs[64 * TN * LDS_KC];
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned wgid = xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, nwg);
    gemm_tile_body<A_KC, B_KC, TM, TN, VA, VB>(g, wgid % gridDim.x, (wgid / gridDim.x) % gridDim.y,
                                               wgid / (gridDim.x * gridDim.y), As, Bs);
}

// grouped launch of 64x64-tile problems, each with its own contraction split (slabs + a grouped reduction)
template <bool A_KC, bool B_KC, bool VA, bool VB>
__global__ void __launch_bounds__(256) k_gemm_group(GemmGroup grp) {
    __shared__ __attribute__((aligned(16))) float As[64 * LDS_KC];
    __shared__ __attribute__((aligned(16))) float Bs[64 * LDS_KC];
    const unsigned wgid = xcd_remap(blockIdx.x, gridDim.x);
    int p = 0;
#pragma unroll
    for (int q = 1; q < GEMM_GROUP_MAX; ++q)
        if (q < grp.n && wgid >= grp.first[q]) p = q;
    const GemmArgs& g = grp.g[p];
    const unsigned local = wgid - grp.first[p];
    const unsigned gx = (unsigned)((g.N + 63) / 64), gy = (unsigned)((g.M + 63) / 64);
    gemm_tile_body<A_KC, B_KC, 1, 1, VA, VB>(g, local % gx, (local / gx) % gy, local / (gx * gy), As, Bs);
}

// ---- small problems: LDS-free GEMM -------------------------------------------------------------------
// When the tiled kernel's grid cannot fill the chip (M <= ~2048 with these layer widths: 128 workgroups, one wave per
// SIMD, every K tile paying a full load latency -- 20 us for 0.5 GFLOP) the work is re-cut finer: one workgroup per
// 32x32 output tile, its 4 waves splitting the CONTRACTION four ways, operands loaded from global/L2 straight into
// the MFMA operand layout (lane = row, 4 consecutive contraction indices per lane half), 64 contraction indices per
// batch of loads, two batches in flight.  The 4 partial tiles meet in LDS and are summed in wave order (fixed
// order: reproducible), then bias / activation / mask and a row-contiguous store.
// Contraction-index pairing of the LDS-free kernel: a batch is 64 indices = two groups of 32; inside a group lane-half
// lk owns the 16 CONSECUTIVE indices 16*lk .. 16*lk+15, fetched by four back-to-back 16-byte loads (j = 0..3).  A row of
// a contraction-contiguous operand is then read 64 contiguous bytes per lane, the two lane halves completing the 128-B
// line, and the four instructions that touch a line are adjacent in issue order (the earlier pairing 8*s + 4*lk spread
// them over the whole batch: with 8 waves per CU the lines were evicted from the 32-KB vector L1 in between and came
// from L2 four times -- forward layout 15.3 -> see DESIGN.md).  Any pairing is valid as long as A and B agree.
__device__ __forceinline__ int64_t direct_k(int64_t k, int s, int lk) { return k + 32 * (s >> 2) + 16 * lk + 4 * (s & 3); }

template <bool KC, bool VEC>
__device__ __forceinline__ void direct_load(const float* __restrict__ p, int64_t ld, int64_t k, int64_t kmax, int lk,
                                            float4 (&v)[8]) {
    // straight-line, clamped addresses (see tile_load)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int64_t kk = direct_k(k, s, lk);
        if (KC) {
            if (VEC) {
                v[s] = *reinterpret_cast<const float4*>(p + min(kk, kmax - 4));
            } else {
                v[s] = make_float4(p[min(kk, kmax - 1)], p[min(kk + 1, kmax - 1)], p[min(kk + 2, kmax - 1)],
                                   p[min(kk + 3, kmax - 1)]);
            }
        } else {
            v[s] = make_float4(p[min(kk, kmax - 1) * ld], p[min(kk + 1, kmax - 1) * ld], p[min(kk + 2, kmax - 1) * ld],
                               p[min(kk + 3, kmax - 1) * ld]);
        }
    }
}

// Aligned fast path (contraction length and split boundaries multiples of 32, waves starting on multiples of 32): the
// contraction part of every address is wave-uniform -- a scalar base per load, computed on the scalar ALU -- and the
// lane part (row / column and the lane half's 16*lk) is ONE 32-bit byte offset per operand.  The generic loader keeps a
// 64-bit address per load in flight: 280-320 VGPRs for the contraction-strided layouts (dgrad, wgrad), one wave per
// SIMD, and hipcc then parks loaded values in AGPRs behind `s_waitcnt vmcnt(0)` -- a dozen serialised round trips at
// the head of the kernel (seen in the ISA).  A 32-group that starts inside [0, K) lies inside it entirely (K % 32 == 0),
// so the uniform clamp below only moves addresses of fully masked groups.
template <bool KC>
__device__ __forceinline__ void direct_load_al(const float* __restrict__ base, int64_t ld, unsigned voff, int64_t k,
                                               int64_t K, float4 (&v)[8]) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int64_t ku = k + 32 * (s >> 2) + 4 * (s & 3);
        if (KC) {
            const char* sb = reinterpret_cast<const char*>(base + min(ku, K - 20));
            v[s] = *reinterpret_cast<const float4*>(sb + voff);
        } else {
            float e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const char* sb = reinterpret_cast<const char*>(base + min(ku + u, K - 17) * ld);
                e[u] = *reinterpret_cast<const float*>(sb + voff);
            }
            v[s] = make_float4(e[0], e[1], e[2], e[3]);
        }
    }
}

#ifndef DIRECT_ABLATE
#define DIRECT_ABLATE 0     // tools/gemm_direct_ablate.hip only: 1 no loads inside the loop, 2 no MFMAs
#endif
__device__ __forceinline__ void direct_mma(const float4 (&a)[8], const float4 (&b)[8], int64_t k, int64_t kw1, int lk,
                                           f32x16& acc, float& cs) {
#if DIRECT_ABLATE & 2
#pragma unroll
    for (int s = 0; s < 8; ++s) cs += (a[s].x + a[s].y) + (a[s].z + a[s].w) + (b[s].x + b[s].y) + (b[s].z + b[s].w);
    acc[0] = cs;
    return;
#endif
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int64_t kk = direct_k(k, s, lk);
        // zeroing ONE operand past the end of this wave's contraction range is enough
        const float ax = kk + 0 < kw1 ? a[s].x : 0.f, ay = kk + 1 < kw1 ? a[s].y : 0.f;
        const float az = kk + 2 < kw1 ? a[s].z : 0.f, aw = kk + 3 < kw1 ? a[s].w : 0.f;
        cs += (ax + ay) + (az + aw);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, b[s].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, b[s].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(az, b[s].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, b[s].w, acc, 0, 0, 0);
    }
}

// The epilogue's operand words -- this thread's four bias values and four words of the activation mask -- fetched at the START of
// the kernel (round 5): behind the contraction loop and the LDS exchange they were one more load latency in front of the store,
// in every one of the 13 GEMM launches of a short-batch step, whose kernels take 7-13 us each.  Same words, same arithmetic.
// ok: the thread stores a whole float4 inside the matrix and both operands are 16-byte loadable; otherwise direct_finish loads
// them itself, element by element, as before.  (cdlrm_debug_set(7, 1) -> GemmArgs.fastep = 0: never.)
struct DirectPre {
    float4 b, m;
    bool ok;
};
__device__ __forceinline__ DirectPre direct_prefetch(const GemmArgs& g, int64_t m0, int64_t n0) {
    DirectPre p;
    p.b = make_float4(0.f, 0.f, 0.f, 0.f);
    p.m = p.b;
    const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const int64_t gm = m0 + row, gn = n0 + c4;
    p.ok = g.fastep && g.vecC && gm < g.M && gn + 3 < (int64_t)g.N &&
           (g.bias == nullptr || (((uintptr_t)g.bias) & 15) == 0) &&
           (g.mask_act == 0 || ((((uintptr_t)g.mask) & 15) == 0 && (g.ldmask & 3) == 0));
    if (p.ok) {
        if (g.bias) p.b = *reinterpret_cast<const float4*>(g.bias + gn);
        if (g.mask_act) p.m = *reinterpret_cast<const float4*>(g.mask + gm * g.ldmask + gn);
    }
    return p;
}

// The 4 partial tiles of a workgroup meet in LDS and are summed in wave order (fixed order: reproducible), then bias /
// activation / mask and a row-contiguous store; the bias-gradient column sums likewise.
template <bool A_KC>
__device__ __forceinline__ void direct_finish(const GemmArgs& g, unsigned bx, unsigned bz, int64_t m0, int64_t n0, int wave,
                                              int lr, int lk, const f32x16& acc, float cs, float (*red)[32][33],
                                              float (*csr)[32], const DirectPre& pre) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lk][lr] = acc[r];
    if (!A_KC && g.colsum != nullptr && bx == 0) {
        cs += __shfl_xor(cs, 32);
        if (lk == 0) csr[wave][lr] = cs;
    }
    __syncthreads();
    if (!A_KC && g.colsum != nullptr && bx == 0 && threadIdx.x < 32 && m0 + threadIdx.x < g.M)
        g.colsum[(int64_t)bz * g.M + m0 + threadIdx.x] =
            ((csr[0][threadIdx.x] + csr[1][threadIdx.x]) + csr[2][threadIdx.x]) + csr[3][threadIdx.x];
    const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const int64_t gm = m0 + row;
    if (gm >= g.M) return;
    float* C = g.C + (int64_t)bz * g.slab + gm * g.ldc;
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t gn = n0 + c4 + u;
        float x = ((red[0][row][c4 + u] + red[1][row][c4 + u]) + red[2][row][c4 + u]) + red[3][row][c4 + u];
        if (gn < g.N) {
            if (g.bias) x += pre.ok ? (u == 0 ? pre.b.x : u == 1 ? pre.b.y : u == 2 ? pre.b.z : pre.b.w) : g.bias[gn];
            if (g.act == 1) x = x > 0.f ? x : 0.f;
            else if (g.act == 2) x = 1.0f / (1.0f + expf(-x));
            if (g.mask_act) {
                const float y = pre.ok ? (u == 0 ? pre.m.x : u == 1 ? pre.m.y : u == 2 ? pre.m.z : pre.m.w)
                                       : g.mask[gm * g.ldmask + gn];
                x = g.mask_act == 1 ? (y > 0.f ? x : 0.f) : x * ((1.0f - y) * y);
            }
        }
        v[u] = x;
    }
    if (g.vecC && n0 + c4 + 3 < g.N) {
        *reinterpret_cast<float4*>(C + n0 + c4) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (n0 + c4 + u < g.N) C[n0 + c4 + u] = v[u];
    }
}

// MODE 0: each wave's share of the contraction is <= 64 (one batch of loads, no loop); 1: <= 128 (two batches, no
// loop); 2: longer (two batches in flight around a loop, the last <= 128 indices peeled so that no batch is fetched
// past the end: at K = 512 that trailing prefetch was a third of all the loads)
template <bool A_KC, bool B_KC, bool VA, bool VB, int MODE, bool AL>
__device__ __forceinline__ void direct_body(const GemmArgs& g, unsigned bx, unsigned by, unsigned bz,
                                            float (*red)[32][33], float (*csr)[32]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: the contraction range is uniform
    const int lr = lane & 31, lk = lane >> 5;
    const int64_t m0 = (int64_t)by * 32, n0 = (int64_t)bx * 32;
    const DirectPre pre = direct_prefetch(g, m0, n0);
    __builtin_amdgcn_sched_barrier(0);          // (the loads stay HERE: hipcc would sink them to their use behind the loop)
    const int64_t kbeg = (int64_t)bz * g.kchunk;
    const int64_t kend = min(g.K, kbeg + g.kchunk);
    // a quarter of the range: rounded up to 8, or (aligned path) to 32 so that every wave starts on a 32-group
    const int64_t kq = AL ? ((kend - kbeg + 127) / 128) * 32 : ((kend - kbeg + 31) / 32) * 8;
    const int64_t kw0 = kbeg + wave * kq, kw1 = min(kend, kw0 + kq);
    const int64_t am = min(m0 + lr, g.M - 1), bn = min(n0 + lr, (int64_t)g.N - 1);
    const float* pa = A_KC ? g.A + am * g.lda : g.A + am;
    const float* pb = B_KC ? g.B + bn * g.ldb : g.B + bn;
    // aligned path: this lane's byte offset into each operand (its row / column + its half's 16 contraction indices)
    const unsigned va_off = (unsigned)((A_KC ? am * g.lda + 16 * lk : 16 * lk * g.lda + am) * 4);
    const unsigned vb_off = (unsigned)((B_KC ? bn * g.ldb + 16 * lk : 16 * lk * g.ldb + bn) * 4);
#define DIRECT_LOAD_A(k_, dst_)                                                   \
    do {                                                                          \
        if (AL) direct_load_al<A_KC>(g.A, g.lda, va_off, (k_), g.K, dst_);        \
        else direct_load<A_KC, VA>(pa, g.lda, (k_), g.K, lk, dst_);               \
    } while (0)
#define DIRECT_LOAD_B(k_, dst_)                                                   \
    do {                                                                          \
        if (AL) direct_load_al<B_KC>(g.B, g.ldb, vb_off, (k_), g.K, dst_);        \
        else direct_load<B_KC, VB>(pb, g.ldb, (k_), g.K, lk, dst_);               \
    } while (0)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float cs = 0.f;
    float4 a0[8], b0[8];
    DIRECT_LOAD_A(kw0, a0);
    DIRECT_LOAD_B(kw0, b0);
    if (MODE >= 1) {
        // Both MFMA batches of an iteration are UNCONDITIONAL (indices past kw1 are zeroed in direct_mma): with a
        // condition on the second batch the compiler sinks its loads into the conditional block, right in front
        // of their use, and the double buffering is gone (seen in the ISA).
        float4 a1[8], b1[8];
        int64_t k = kw0;
        // sched_barrier: the machine scheduler otherwise drags each load down to just before its use (fewer
        // live registers, 2-4 loads in flight, one exposed latency per step -- seen in the ISA)
        if (MODE == 2) {
            for (; k + 128 < kw1; k += 128) {
                if (!(DIRECT_ABLATE & 1) || k == kw0) {
                    DIRECT_LOAD_A(k + 64, a1);
                    DIRECT_LOAD_B(k + 64, b1);
                }
                __builtin_amdgcn_sched_barrier(0);
                direct_mma(a0, b0, k, kw1, lk, acc, cs);
                __builtin_amdgcn_sched_barrier(0);
                if (!(DIRECT_ABLATE & 1)) {
                    DIRECT_LOAD_A(k + 128, a0);
                    DIRECT_LOAD_B(k + 128, b0);
                }
                __builtin_amdgcn_sched_barrier(0);
                direct_mma(a1, b1, k + 64, kw1, lk, acc, cs);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        DIRECT_LOAD_A(k + 64, a1);
        DIRECT_LOAD_B(k + 64, b1);
        __builtin_amdgcn_sched_barrier(0);
        direct_mma(a0, b0, k, kw1, lk, acc, cs);
        __builtin_amdgcn_sched_barrier(0);
        direct_mma(a1, b1, k + 64, kw1, lk, acc, cs);
    } else {
        __builtin_amdgcn_sched_barrier(0);
        direct_mma(a0, b0, kw0, kw1, lk, acc, cs);
    }
#undef DIRECT_LOAD_A
#undef DIRECT_LOAD_B
    direct_finish<A_KC>(g, bx, bz, m0, n0, wave, lr, lk, acc, cs, red, csr, pre);
}

template <bool A_KC, bool B_KC, bool VA, bool VB, int MODE, bool AL>
__global__ void __launch_bounds__(256) k_gemm_direct(GemmArgs g) {
    __shared__ float red[4][32][33];
    __shared__ float csr[4][32];
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned wgid = xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, nwg);
    direct_body<A_KC, B_KC, VA, VB, MODE, AL>(g, wgid % gridDim.x, (wgid / gridDim.x) % gridDim.y,
                                          wgid / (gridDim.x * gridDim.y), red, csr);
}

template <bool A_KC, bool B_KC, bool VA, bool VB, int MODE, bool AL>
__global__ void __launch_bounds__(256) k_gemm_direct_group(GemmGroup grp) {
    __shared__ float red[4][32][33];
    __shared__ float csr[4][32];
    const unsigned wgid = xcd_remap(blockIdx.x, gridDim.x);
    int p = 0;
#pragma unroll
    for (int q = 1; q < GEMM_GROUP_MAX; ++q)
        if (q < grp.n && wgid >= grp.first[q]) p = q;
    const GemmArgs& g = grp.g[p];
    const unsigned local = wgid - grp.first[p];
    const unsigned gx = (unsigned)((g.N + 31) / 32), gy = (unsigned)((g.M + 31) / 32);
    direct_body<A_KC, B_KC, VA, VB, MODE, AL>(g, local % gx, (local / gx) % gy, local / (gx * gy), red, csr);
}

// ---- small problems, staged: coalesced loads + a wave-private LDS transpose -------------------------------------
// The LDS-free kernel reads a contraction-contiguous operand with lane = row: every 16-byte load instruction touches
// 32 different 128-B lines, and the vector memory pipeline (address coalescer / L1 tag look-ups) -- not L2, not the
// MFMAs -- bounds it (1024x512x512: 12.9 us with 3.4 us of MFMA work).  Here each wave fetches its 32 x 64 (rows x
// contraction) sub-panels with fully coalesced 16-byte loads (a 256-B run of a row per 16 lanes), parks them in a
// wave-PRIVATE LDS region and reads them back in the MFMA operand layout.  No workgroup barrier: the LDS pipeline
// executes a wave's accesses in order, and the regions of the four waves are disjoint.  Same cut as the LDS-free
// kernel otherwise (32x32 tile per workgroup, 4 waves split the contraction, partial tiles summed in wave order).
// Requires the aligned conditions (K % 32 == 0, split boundaries on multiples of 32), K >= 64, 16-byte-loadable
// operands, and non-contraction extents % 4 == 0 for contraction-strided operands.
#define ST_KC_PITCH 68      // floats; 17 * 16 B: ds_write_b128 / ds_read_b128 conflict-free
#define ST_KS_PITCH 36      // contraction-strided operand kept as [k][32 + 4]
#define ST_OP_FLOATS 2304   // per operand per wave: max(32 * 68, 64 * 36)
#define ST_LDS_BYTES (4 * 2 * ST_OP_FLOATS * 4)

// global -> registers: 8 float4 per lane cover a 32 x 64 (KC) or 64 x 32 (contraction-strided) sub-panel
template <bool KC>
__device__ __forceinline__ void st_load(const float* __restrict__ base, int64_t ld, const unsigned (&voff)[8], int64_t ks,
                                        f32x4 (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (KC) {       // lane: row (lane >> 4) + 4 i, contraction offset (lane & 15) * 4
            const char* sb = reinterpret_cast<const char*>(base + ks);
            v[i] = *reinterpret_cast<const f32x4*>(sb + voff[i]);
        } else {        // lane: contraction row (lane >> 3) + 8 i, 4 consecutive non-contraction elements
            const char* sb = reinterpret_cast<const char*>(base + (ks + 8 * i) * ld);
            v[i] = *reinterpret_cast<const f32x4*>(sb + voff[0]);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void st_park(float* __restrict__ S, int lane, const f32x4 (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (KC) *reinterpret_cast<f32x4*>(S + ((lane >> 4) + 4 * i) * ST_KC_PITCH + (lane & 15) * 4) = v[i];
        else *reinterpret_cast<f32x4*>(S + ((lane >> 3) + 8 * i) * ST_KS_PITCH + (lane & 7) * 4) = v[i];
    }
}

// fragment of contraction group gq (8 indices; lane half lk owns 4 consecutive ones); shift: the panel in LDS starts
// `shift` indices before this batch (a batch that would straddle the end of K is fetched from K - 64)
template <bool KC>
__device__ __forceinline__ f32x4 st_frag(const float* __restrict__ S, int lr, int lk, int gq, int shift) {
    const int j0 = (8 * gq + 4 * lk + shift) & 63;
    if (KC) return *reinterpret_cast<const f32x4*>(S + lr * ST_KC_PITCH + j0);
    const float* q = S + j0 * ST_KS_PITCH + lr;
    f32x4 r;
    r.x = q[0]; r.y = q[ST_KS_PITCH]; r.z = q[2 * ST_KS_PITCH]; r.w = q[3 * ST_KS_PITCH];
    return r;
}

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void st_batch(const float* __restrict__ Sa, const float* __restrict__ Sb, int lr, int lk,
                                         int shift, int64_t k, int64_t kw1, f32x16& acc, float& cs) {
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {
        const f32x4 a = st_frag<A_KC>(Sa, lr, lk, gq, shift);
        const f32x4 b = st_frag<B_KC>(Sb, lr, lk, gq, shift);
        const int64_t kk = k + 8 * gq + 4 * lk;
        // zeroing ONE operand past the end of this wave's contraction range is enough
        const float ax = kk + 0 < kw1 ? a.x : 0.f, ay = kk + 1 < kw1 ? a.y : 0.f;
        const float az = kk + 2 < kw1 ? a.z : 0.f, aw = kk + 3 < kw1 ? a.w : 0.f;
        cs += (ax + ay) + (az + aw);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(az, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, b.w, acc, 0, 0, 0);
    }
}

template <bool A_KC, bool B_KC, int MODE>
__device__ __forceinline__ void staged_body(const GemmArgs& g, unsigned bx, unsigned by, unsigned bz, float* __restrict__ lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int64_t m0 = (int64_t)by * 32, n0 = (int64_t)bx * 32;
    const DirectPre pre = direct_prefetch(g, m0, n0);
    __builtin_amdgcn_sched_barrier(0);
    const int64_t kbeg = (int64_t)bz * g.kchunk;
    const int64_t kend = min(g.K, kbeg + g.kchunk);
    const int64_t kq = ((kend - kbeg + 127) / 128) * 32;
    const int64_t kw0 = kbeg + wave * kq, kw1 = min(kend, kw0 + kq);
    float* Sa = lds + wave * (2 * ST_OP_FLOATS);
    float* Sb = Sa + ST_OP_FLOATS;
    // per-lane byte offsets of the coalesced loads (rows / columns past the edge are clamped to valid ones: their
    // products land in output rows / columns that are never stored)
    unsigned va[8], vb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (A_KC) va[i] = (unsigned)((min(m0 + (lane >> 4) + 4 * i, g.M - 1) * g.lda + (lane & 15) * 4) * 4);
        else va[i] = (unsigned)(((lane >> 3) * g.lda + min(m0 + (lane & 7) * 4, g.M - 4)) * 4);
        if (B_KC) vb[i] = (unsigned)((min(n0 + (lane >> 4) + 4 * i, (int64_t)g.N - 1) * g.ldb + (lane & 15) * 4) * 4);
        else vb[i] = (unsigned)(((lane >> 3) * g.ldb + min(n0 + (lane & 7) * 4, (int64_t)g.N - 4)) * 4);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float cs = 0.f;
    f32x4 ra0[8], rb0[8], ra1[8], rb1[8];
    // a batch is fetched from ks = min(k, K - 64): `shift` = k - ks (0 or 32 inside the range) relocates the fragments
    const int64_t klast = g.K - 64;
#define ST_KS(k_) min((int64_t)(k_), klast)
#define ST_SHIFT(k_) ((int)min((int64_t)(k_) - ST_KS(k_), (int64_t)32))
    st_load<A_KC>(g.A, g.lda, va, ST_KS(kw0), ra0);
    st_load<B_KC>(g.B, g.ldb, vb, ST_KS(kw0), rb0);
    if (MODE >= 1) {
        st_load<A_KC>(g.A, g.lda, va, ST_KS(kw0 + 64), ra1);
        st_load<B_KC>(g.B, g.ldb, vb, ST_KS(kw0 + 64), rb1);
    }
    __builtin_amdgcn_sched_barrier(0);
    int64_t k = kw0;
    if (MODE == 2) {
        // two batches in flight around the loop; the last <= 128 indices are peeled so that nothing is fetched past the end
        for (; k + 128 < kw1; k += 128) {
            st_park<A_KC>(Sa, lane, ra0);
            st_park<B_KC>(Sb, lane, rb0);
            __builtin_amdgcn_sched_barrier(0);
            st_load<A_KC>(g.A, g.lda, va, ST_KS(k + 128), ra0);      // into the registers just parked
            st_load<B_KC>(g.B, g.ldb, vb, ST_KS(k + 128), rb0);
            __builtin_amdgcn_sched_barrier(0);
            st_batch<A_KC, B_KC>(Sa, Sb, lr, lk, ST_SHIFT(k), k, kw1, acc, cs);
            __builtin_amdgcn_sched_barrier(0);
            st_park<A_KC>(Sa, lane, ra1);
            st_park<B_KC>(Sb, lane, rb1);
            __builtin_amdgcn_sched_barrier(0);
            st_load<A_KC>(g.A, g.lda, va, ST_KS(k + 192), ra1);
            st_load<B_KC>(g.B, g.ldb, vb, ST_KS(k + 192), rb1);
            __builtin_amdgcn_sched_barrier(0);
            st_batch<A_KC, B_KC>(Sa, Sb, lr, lk, ST_SHIFT(k + 64), k + 64, kw1, acc, cs);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    st_park<A_KC>(Sa, lane, ra0);
    st_park<B_KC>(Sb, lane, rb0);
    __builtin_amdgcn_sched_barrier(0);
    st_batch<A_KC, B_KC>(Sa, Sb, lr, lk, ST_SHIFT(k), k, kw1, acc, cs);
    if (MODE >= 1) {
        __builtin_amdgcn_sched_barrier(0);
        st_park<A_KC>(Sa, lane, ra1);
        st_park<B_KC>(Sb, lane, rb1);
        __builtin_amdgcn_sched_barrier(0);
        st_batch<A_KC, B_KC>(Sa, Sb, lr, lk, ST_SHIFT(k + 64), k + 64, kw1, acc, cs);
    }
#undef ST_KS
#undef ST_SHIFT
    __syncthreads();        // every wave is done with its staging region: the reduction buffers alias it
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(lds);
    float (*csr)[32] = reinterpret_cast<float (*)[32]>(lds + 4 * 32 * 33);
    direct_finish<A_KC>(g, bx, bz, m0, n0, wave, lr, lk, acc, cs, red, csr, pre);
}

template <bool A_KC, bool B_KC, int MODE>
__global__ void __launch_bounds__(256) k_gemm_staged(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float st_lds[];
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned wgid = xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, nwg);
    staged_body<A_KC, B_KC, MODE>(g, wgid % gridDim.x, (wgid / gridDim.x) % gridDim.y, wgid / (gridDim.x * gridDim.y), st_lds);
}

template <bool A_KC, bool B_KC, int MODE>
__global__ void __launch_bounds__(256) k_gemm_staged_group(GemmGroup grp) {
    extern __shared__ __attribute__((aligned(16))) float st_lds[];
    const unsigned wgid = xcd_remap(blockIdx.x, gridDim.x);
    int p = 0;
#pragma unroll
    for (int q = 1; q < GEMM_GROUP_MAX; ++q)
        if (q < grp.n && wgid >= grp.first[q]) p = q;
    const GemmArgs& g = grp.g[p];
    const unsigned local = wgid - grp.first[p];
    const unsigned gx = (unsigned)((g.N + 31) / 32), gy = (unsigned)((g.M + 31) / 32);
    staged_body<A_KC, B_KC, MODE>(g, local % gx, (local / gx) % gy, local / (gx * gy), st_lds);
}

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// tile choice: largest tile whose grid still has >= GEMM_MIN_BLOCKS workgroups
#define GEMM_MIN_BLOCKS 512
static inline void gemm_pick_tile(int64_t M, int64_t N, int64_t splits, int* tm, int* tn) {
    // measured on MI355X at M=8192, N=K=512 (tools/gemm_ablate.hip, branch-free staging):
    //   64x64 91.5 TF (1024 workgroups) > 64x128 86.5 = 128x64 86.3 > 128x128 74.9 (256 workgroups)
    // -> take the largest tile that still leaves >= 4 workgroups per CU
    const int cand[4][2] = {{2, 2}, {1, 2}, {2, 1}, {1, 1}};
    for (int c = 0; c < 4; ++c) {
        const int64_t blocks = cdiv(M, 64 * cand[c][0]) * cdiv(N, 64 * cand[c][1]) * splits;
        if (blocks >= 2 * GEMM_MIN_BLOCKS || c == 3) {
            *tm = cand[c][0]; *tn = cand[c][1];
            return;
        }
    }
}

template <bool A_KC, bool B_KC, int TM, int TN>
static void launch_gemm_v(const GemmArgs& g, dim3 grid, hipStream_t s) {
    if (g.vecA && g.vecB) hipLaunchKernelGGL((k_gemm<A_KC, B_KC, TM, TN, true, true>), grid, dim3(256), 0, s, g);
    else if (g.vecA) hipLaunchKernelGGL((k_gemm<A_KC, B_KC, TM, TN, true, false>), grid, dim3(256), 0, s, g);
    else if (g.vecB) hipLaunchKernelGGL((k_gemm<A_KC, B_KC, TM, TN, false, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_gemm<A_KC, B_KC, TM, TN, false, false>), grid, dim3(256), 0, s, g);
}

// the tiled kernel needs this many 64x64 workgroups to be the better choice (2 per CU); below it the
// LDS-free kernel's 4x finer cut wins.  CDLRM_GEMM_DIRECT=0 / 1 forces one or the other (experiments).
// Fewer than 256 tiles of 64x64 (one per CU): the 32x32-tile kernels (staged / LDS-free), whose workgroups split the contraction.
// The bound was 512 until round 4 -- set when the tiled kernel was the register-staged one; with the LDS-DMA kernel a 512-wide
// layer at M = 2048 (256 tiles) is better off tiled: same box, same process, five rounds each, 0.2710 -> 0.2418 ms per step at a
// per-rank batch of 2048 (-10.8 %), 0.1997 -> 0.1832 at c2 (B = 2048, D = 32), ties at 4096 and 8192; lower bounds LOSE at 1024
// (128 tiles: 0.1869 at 512 / 256 against 0.1986 / 0.2046 / 0.2074 at 128 / 64 / 32).
static inline bool gemm_use_direct(int64_t M, int64_t N, int64_t splits) {
    return cdiv(M, 64) * cdiv(N, 64) * splits < 256;
}

// per-wave share of the contraction (a quarter, rounded up to 8 -- to 32 on the aligned path): one batch of 64, two
// batches, or a loop
static inline int direct_mode(int64_t klen, bool al) {
    const int64_t kq = al ? cdiv(klen, 128) * 32 : cdiv(klen, 32) * 8;
    return kq <= 64 ? 0 : kq <= 128 ? 1 : 2;
}

// the aligned loader applies: contraction and split boundaries on multiples of 32, 16-byte loads legal on the
// contraction-contiguous operands, lane offsets within 32 bits
template <bool A_KC, bool B_KC>
static inline bool direct_aligned(const GemmArgs& g, int64_t klen) {
    if (g.K < 32 || g.K % 32 != 0 || (klen < g.K && klen % 32 != 0)) return false;
    if ((A_KC && !g.vecA) || (B_KC && !g.vecB)) return false;
    const int64_t lim = (int64_t)1 << 30;      // elements: byte offsets stay below 2^32
    if ((A_KC ? g.M * g.lda : 16 * g.lda + g.M) >= lim || (B_KC ? (int64_t)g.N * g.ldb : 16 * g.ldb + g.N) >= lim) return false;
    return true;
}

// the staged kernel applies: the aligned conditions, K >= 64, and 16-byte loads along the contiguous direction of
// contraction-strided operands (extent and pitch multiples of 4)
template <bool A_KC, bool B_KC>
static inline bool direct_staged(const GemmArgs& g, int64_t klen) {
    if (!direct_aligned<A_KC, B_KC>(g, klen) || g.K < 64) return false;
    if (!A_KC && (g.M % 4 != 0 || g.lda % 4 != 0 || !aligned16(g.A))) return false;
    if (!B_KC && (g.N % 4 != 0 || g.ldb % 4 != 0 || !aligned16(g.B))) return false;
    const int64_t lim = (int64_t)1 << 30;
    if ((A_KC ? g.M * g.lda : 8 * g.lda + g.M) >= lim || (B_KC ? (int64_t)g.N * g.ldb : 8 * g.ldb + g.N) >= lim) return false;
    return true;
}

template <typename KERN>
static inline void staged_lds_attr(KERN kernel) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS_BYTES);
}

template <bool A_KC, bool B_KC>
static void launch_gemm_staged(const GemmArgs& g, dim3 grid, int mode, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        staged_lds_attr(k_gemm_staged<A_KC, B_KC, 0>);
        staged_lds_attr(k_gemm_staged<A_KC, B_KC, 1>);
        staged_lds_attr(k_gemm_staged<A_KC, B_KC, 2>);
        attr = true;
    }
    if (mode == 2) CDLRM_LAUNCH_EV((k_gemm_staged<A_KC, B_KC, 2>), grid, dim3(256), ST_LDS_BYTES, s, g);
    else if (mode == 1) CDLRM_LAUNCH_EV((k_gemm_staged<A_KC, B_KC, 1>), grid, dim3(256), ST_LDS_BYTES, s, g);
    else CDLRM_LAUNCH_EV((k_gemm_staged<A_KC, B_KC, 0>), grid, dim3(256), ST_LDS_BYTES, s, g);
}

template <bool A_KC, bool B_KC>
static void launch_gemm_direct(GemmArgs g, int splits, hipStream_t s) {
    g.vecC = aligned16(g.C) && g.ldc % 4 == 0 && g.slab % 4 == 0;
    const bool va = A_KC && g.vecA, vb = B_KC && g.vecB;     // only contraction-contiguous operands use 16-B loads
    dim3 grid((unsigned)cdiv(g.N, 32), (unsigned)cdiv(g.M, 32), (unsigned)splits);
    const int64_t klen = g.kchunk < g.K ? g.kchunk : g.K;
    const bool al = direct_aligned<A_KC, B_KC>(g, klen);
    const int mode = direct_mode(klen, al);
    if (direct_staged<A_KC, B_KC>(g, klen)) {
        launch_gemm_staged<A_KC, B_KC>(g, grid, mode, s);
        return;
    }
    if (al) {
        if (mode == 2) hipLaunchKernelGGL((k_gemm_direct<A_KC, B_KC, A_KC, B_KC, 2, true>), grid, dim3(256), 0, s, g);
        else if (mode == 1) hipLaunchKernelGGL((k_gemm_direct<A_KC, B_KC, A_KC, B_KC, 1, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((k_gemm_direct<A_KC, B_KC, A_KC, B_KC, 0, true>), grid, dim3(256), 0, s, g);
        return;
    }
#define CDLRM_DIRECT(VA_, VB_)                                                                              \
    do {                                                                                                    \
        if (mode == 2) hipLaunchKernelGGL((k_gemm_direct<A_KC, B_KC, VA_, VB_, 2, false>), grid, dim3(256), 0, s, g);      \
        else if (mode == 1) hipLaunchKernelGGL((k_gemm_direct<A_KC, B_KC, VA_, VB_, 1, false>), grid, dim3(256), 0, s, g); \
        else hipLaunchKernelGGL((k_gemm_direct<A_KC, B_KC, VA_, VB_, 0, false>), grid, dim3(256), 0, s, g);                \
    } while (0)
    if (va && vb) CDLRM_DIRECT(A_KC, B_KC);
    else if (va) CDLRM_DIRECT(A_KC, false);
    else if (vb) CDLRM_DIRECT(false, B_KC);
    else CDLRM_DIRECT(false, false);
#undef CDLRM_DIRECT
}

// The two grouped weight-gradient launches of a short-batch step as ONE: the layers the LDS-free kernel takes (a 13-wide
// input, a 1-wide output: a launch of ~8 us for next to no work, all of it launch floor) ride as extra workgroups in front of
// the LDS-tiled layers' group.  One kernel, two bodies, picked by workgroup index; the LDS of the two is one raw buffer.
template <int MODE, bool AL>
__global__ void __launch_bounds__(256) k_wgrad_mixed(GemmGroup dgrp, GemmGroup tgrp, unsigned n_direct) {
    constexpr int RED_FLOATS = 4 * 32 * 33 + 4 * 32, TILE_FLOATS = 2 * 64 * LDS_KC;
    __shared__ __attribute__((aligned(16))) float lds[RED_FLOATS > TILE_FLOATS ? RED_FLOATS : TILE_FLOATS];
    const unsigned wgid = xcd_remap(blockIdx.x, gridDim.x);
    if (wgid < n_direct) {
        int p = 0;
#pragma unroll
        for (int q = 1; q < GEMM_GROUP_MAX; ++q)
            if (q < dgrp.n && wgid >= dgrp.first[q]) p = q;
        const GemmArgs& g = dgrp.g[p];
        const unsigned local = wgid - dgrp.first[p];
        const unsigned gx = (unsigned)((g.N + 31) / 32), gy = (unsigned)((g.M + 31) / 32);
        float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(lds);
        float (*csr)[32] = reinterpret_cast<float (*)[32]>(lds + 4 * 32 * 33);
        direct_body<false, false, false, false, MODE, AL>(g, local % gx, (local / gx) % gy, local / (gx * gy), red, csr);
    } else {
        const unsigned w = wgid - n_direct;
        int p = 0;
#pragma unroll
        for (int q = 1; q < GEMM_GROUP_MAX; ++q)
            if (q < tgrp.n && w >= tgrp.first[q]) p = q;
        const GemmArgs& g = tgrp.g[p];
        const unsigned local = w - tgrp.first[p];
        const unsigned gx = (unsigned)((g.N + 63) / 64), gy = (unsigned)((g.M + 63) / 64);
        gemm_tile_body<false, false, 1, 1, true, true>(g, local % gx, (local / gx) % gy, local / (gx * gy), lds, lds + 64 * LDS_KC);
    }
}

// returns 1 when it launched, 0 when the combination is not one it takes (the caller then launches the groups separately)
static inline int launch_wgrad_mixed(const GemmArgs* direct, int nd, const GemmGroup& tgrp, unsigned tblocks, hipStream_t s) {
    if (nd < 1 || nd > GEMM_GROUP_MAX || tgrp.n < 1) return 0;
    GemmGroup dgrp;
    memset(&dgrp, 0, sizeof(dgrp));
    unsigned blocks = 0;
    int mode = -1;
    bool al = false;
    for (int i = 0; i < nd; ++i) {
        const int64_t kc = direct[i].kchunk > 0 && direct[i].kchunk < direct[i].K ? direct[i].kchunk : direct[i].K;
        GemmArgs pg = direct[i];
        pg.kchunk = kc;
        const bool a = direct_aligned<false, false>(pg, kc);
        if (direct_staged<false, false>(pg, kc)) return 0;
        const int m = direct_mode(kc, a);
        if (i > 0 && (m != mode || a != al)) return 0;         // one variant of the LDS-free body per launch
        mode = m; al = a;
        dgrp.first[i] = blocks;
        dgrp.g[i] = pg;
        dgrp.g[i].vecC = aligned16(pg.C) && pg.ldc % 4 == 0 && pg.slab % 4 == 0;
        blocks += (unsigned)(cdiv(pg.M, 32) * cdiv(pg.N, 32) * cdiv(pg.K, kc));
    }
    dgrp.n = nd;
    dgrp.first[nd] = blocks;
    const dim3 grid(blocks + tblocks);
#define CDLRM_WMIX(MODE_, AL_) hipLaunchKernelGGL((k_wgrad_mixed<MODE_, AL_>), grid, dim3(256), 0, s, dgrp, tgrp, blocks)
    if (al) {
        if (mode == 2) CDLRM_WMIX(2, true); else if (mode == 1) CDLRM_WMIX(1, true); else CDLRM_WMIX(0, true);
    } else {
        if (mode == 2) CDLRM_WMIX(2, false); else if (mode == 1) CDLRM_WMIX(1, false); else CDLRM_WMIX(0, false);
    }
#undef CDLRM_WMIX
    return 1;
}

// up to GEMM_GROUP_MAX un-split problems of the weight-gradient layout (both operands contraction-strided) per launch
static inline int launch_wgrad_group(const GemmArgs* probs, int n, hipStream_t s) {
    static bool st_attr = false;
    if (!st_attr) {
        staged_lds_attr(k_gemm_staged_group<false, false, 0>);
        staged_lds_attr(k_gemm_staged_group<false, false, 1>);
        staged_lds_attr(k_gemm_staged_group<false, false, 2>);
        st_attr = true;
    }
    for (int want = 0; want < 9; ++want) {
        const int want_mode = want % 3;
        const bool want_al = want >= 3;
        const bool want_st = want >= 6;
        GemmGroup grp;
        memset(&grp, 0, sizeof(grp));
        unsigned blocks = 0;
        auto flush = [&]() {
            if (grp.n == 0) return;
            grp.first[grp.n] = blocks;
#define CDLRM_DGROUP(MODE_, AL_) \
    hipLaunchKernelGGL((k_gemm_direct_group<false, false, false, false, MODE_, AL_>), dim3(blocks), dim3(256), 0, s, grp)
            if (want_st) {
                if (want_mode == 2)
                    hipLaunchKernelGGL((k_gemm_staged_group<false, false, 2>), dim3(blocks), dim3(256), ST_LDS_BYTES, s, grp);
                else if (want_mode == 1)
                    hipLaunchKernelGGL((k_gemm_staged_group<false, false, 1>), dim3(blocks), dim3(256), ST_LDS_BYTES, s, grp);
                else
                    hipLaunchKernelGGL((k_gemm_staged_group<false, false, 0>), dim3(blocks), dim3(256), ST_LDS_BYTES, s, grp);
            } else if (want_al) {
                if (want_mode == 2) CDLRM_DGROUP(2, true);
                else if (want_mode == 1) CDLRM_DGROUP(1, true);
                else CDLRM_DGROUP(0, true);
            } else {
                if (want_mode == 2) CDLRM_DGROUP(2, false);
                else if (want_mode == 1) CDLRM_DGROUP(1, false);
                else CDLRM_DGROUP(0, false);
            }
#undef CDLRM_DGROUP
            grp.n = 0;
            blocks = 0;
        };
        for (int i = 0; i < n; ++i) {
            // kchunk < K: the contraction is cut into slabs (C + z*slab, colsum + z*M), summed by the caller
            const int64_t kc = probs[i].kchunk > 0 && probs[i].kchunk < probs[i].K ? probs[i].kchunk : probs[i].K;
            GemmArgs pg = probs[i];
            pg.kchunk = kc;
            const bool al = direct_aligned<false, false>(pg, kc);
            const bool st = direct_staged<false, false>(pg, kc);
            if (st != want_st || (!st && al != (want_al && !want_st)) || direct_mode(kc, al) != want_mode) continue;
            if (grp.n == GEMM_GROUP_MAX) flush();
            grp.first[grp.n] = blocks;
            grp.g[grp.n] = probs[i];
            grp.g[grp.n].kchunk = kc;
            grp.g[grp.n].vecC = aligned16(probs[i].C) && probs[i].ldc % 4 == 0 && probs[i].slab % 4 == 0;
            blocks += (unsigned)(cdiv(probs[i].M, 32) * cdiv(probs[i].N, 32) * cdiv(probs[i].K, kc));
            grp.n++;
        }
        flush();
    }
    CDLRM_LAUNCH_CHECK();
    return 0;
}

