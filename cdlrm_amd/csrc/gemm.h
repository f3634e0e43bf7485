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
// step-major MFMA order.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
};

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

template <bool A_KC, bool B_KC, int TM, int TN, bool VA, bool VB>
__global__ void __launch_bounds__(256) k_gemm(GemmArgs g) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_KC];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_KC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an L2), so give
    // each XCD a contiguous run of logical tiles: the tiles of one row panel then re-read A from their own L2
    // instead of fetching it once per XCD.  Bijective for any grid size; placement only affects speed.
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned orig = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = orig & 7;
    const unsigned wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (orig >> 3);
    const unsigned bx = wgid % gridDim.x, by = (wgid / gridDim.x) % gridDim.y, bz = wgid / (gridDim.x * gridDim.y);
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
                C[row * g.ldc + col] = v;
            }
        }
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

template <bool A_KC, bool B_KC>
static int launch_gemm(GemmArgs g, int splits, hipStream_t s) {
    int tm, tn;
    gemm_pick_tile(g.M, g.N, splits, &tm, &tn);
    if (g.N <= 32) tn = 1;
    if (g.M <= 32) tm = 1;
    if (tm == 2 && tn == 1) { tm = 1; tn = g.N <= 64 ? 1 : 2; }     // 128x64 is never the best shape here
    // vector loads also need extents >= 4 in the vectorised direction (clamped addresses must stay inside)
    if (g.K < 4) g.vecA = g.vecB = 0;
    if (!A_KC && g.M < 4) g.vecA = 0;
    if (!B_KC && g.N < 4) g.vecB = 0;
    dim3 grid((unsigned)cdiv(g.N, 64 * tn), (unsigned)cdiv(g.M, 64 * tm), (unsigned)splits);
    if (tm == 2 && tn == 2) launch_gemm_v<A_KC, B_KC, 2, 2>(g, grid, s);
    else if (tm == 1 && tn == 2) launch_gemm_v<A_KC, B_KC, 1, 2>(g, grid, s);
    else launch_gemm_v<A_KC, B_KC, 1, 1>(g, grid, s);
    CDLRM_LAUNCH_CHECK();
    return 0;
}
