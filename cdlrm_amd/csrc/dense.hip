// Dense part of the training step on gfx950: FP32-MFMA Linear layers (K10), the pairwise-dot feature
// interaction (K9), BCE loss (K11) and dense SGD (K12).  Reference: model_no_ddp.py:244-316,
// main_no_ddp.py:212-221, 375, 415.
//
// fp32 everywhere (1e-5 loss-trajectory parity): v_mfma_f32_32x32x2_f32 is an exact k-ordered fp32
// fma chain at the fp32 vector peak (MI355X_MICROARCH.md "Matrix cores").
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// =================================================================================================
// Tiled GEMM  C[m,n] = sum_k A(m,k) * B(k,n)   128x128x32 tile, 4 waves x (2x2 MFMA 32x32 tiles),
// register-prefetched global->LDS staging.
//   A_KC: A(m,k) = A[m*lda + k]  (contraction contiguous)   else A[k*lda + m]
//   B_KC: B(k,n) = B[n*ldb + k]                             else B[k*ldb + n]
// =================================================================================================
#define GBM 128
#define GBN 128
#define GBK 32
#define LDS_KC (GBK + 1)

struct GemmArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc; int64_t slab;     // slab: elements between split-K outputs
    int64_t M; int N; int64_t K; int64_t kchunk;
    const float* bias; int act;               // epilogue: + bias[n], 1 ReLU, 2 sigmoid
    int vecA, vecB;                           // 16-byte loads legal
};

template <bool KC>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int64_t ld, int64_t r0, int64_t rmax,
                                          int64_t k0, int64_t kmax, bool vec, float4 v[4]) {
    // KC: rows = non-contraction index (128), 32 contraction elements per row
    // !KC: rows = contraction index (32), 128 non-contraction elements per row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        int64_t r, c;   // r: non-contraction index offset, c: contraction offset (first of 4 for KC)
        const float* p;
        bool rok;
        int64_t navail;
        if (KC) {
            r = f >> 3; c = (f & 7) * 4;
            rok = r0 + r < rmax;
            navail = kmax - (k0 + c);
            p = P + (r0 + r) * ld + k0 + c;
        } else {
            c = f >> 5; r = (f & 31) * 4;
            rok = k0 + c < kmax;
            navail = rmax - (r0 + r);
            p = P + (k0 + c) * ld + r0 + r;
        }
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rok && navail > 0) {
            if (vec && navail >= 4) {
                x = *reinterpret_cast<const float4*>(p);
            } else {
                x.x = p[0];
                if (navail > 1) x.y = p[1];
                if (navail > 2) x.z = p[2];
                if (navail > 3) x.w = p[3];
            }
        }
        v[i] = x;
    }
}

template <bool KC>
__device__ __forceinline__ void tile_store(float* __restrict__ S, const float4 v[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        if (KC) {
            const int r = f >> 3, c = (f & 7) * 4;
            float* d = S + r * LDS_KC + c;
            d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        } else {
            const int c = f >> 5, r = (f & 31) * 4;
            *reinterpret_cast<float4*>(S + c * GBM + r) = v[i];
        }
    }
}

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256) k_gemm(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[GBM * LDS_KC];
    __shared__ __attribute__((aligned(16))) float Bs[GBN * LDS_KC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * GBM;
    const int64_t n0 = (int64_t)blockIdx.x * GBN;
    const int64_t kbeg = (int64_t)blockIdx.z * g.kchunk;
    const int64_t kend = min(g.K, kbeg + g.kchunk);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[4], rb[4];
    tile_load<A_KC>(g.A, g.lda, m0, g.M, kbeg, kend, g.vecA, ra);
    tile_load<B_KC>(g.B, g.ldb, n0, g.N, kbeg, kend, g.vecB, rb);
    for (int64_t k0 = kbeg; k0 < kend; k0 += GBK) {
        __syncthreads();
        tile_store<A_KC>(As, ra);
        tile_store<B_KC>(Bs, rb);
        __syncthreads();
        if (k0 + GBK < kend) {
            tile_load<A_KC>(g.A, g.lda, m0, g.M, k0 + GBK, kend, g.vecA, ra);
            tile_load<B_KC>(g.B, g.ldb, n0, g.N, k0 + GBK, kend, g.vecB, rb);
        }
        const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 64 + i * 32 + lr;
                a[i] = A_KC ? As[r * LDS_KC + kk + lk] : As[(kk + lk) * GBM + r];
                const int c = wn * 64 + i * 32 + lr;
                b[i] = B_KC ? Bs[c * LDS_KC + kk + lk] : Bs[(kk + lk) * GBN + c];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float* C = g.C + (int64_t)blockIdx.z * g.slab;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= g.M) continue;
                float v = acc[i][j][r] + bv;
                if (g.act == 1) v = v > 0.f ? v : 0.f;
                else if (g.act == 2) v = 1.0f / (1.0f + expf(-v));
                C[row * g.ldc + col] = v;
            }
        }
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <bool A_KC, bool B_KC>
static int launch_gemm(GemmArgs g, int splits, hipStream_t s) {
    dim3 grid((unsigned)cdiv(g.N, GBN), (unsigned)cdiv(g.M, GBM), (unsigned)splits);
    hipLaunchKernelGGL((k_gemm<A_KC, B_KC>), grid, dim3(256), 0, s, g);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_linear_fwd(const float* X, int64_t ld_x, const float* W, const float* bias, float* Y, int64_t ld_y,
                                int64_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    CDLRM_REQUIRE(X && W && Y && M >= 0 && N >= 1 && K >= 1 && ld_x >= K && ld_y >= N, "bad argument");
    if (M == 0) return 0;
    GemmArgs g;
    g.A = X; g.lda = ld_x; g.B = W; g.ldb = K; g.C = Y; g.ldc = ld_y; g.slab = 0;
    g.M = M; g.N = N; g.K = K; g.kchunk = K; g.bias = bias; g.act = act;
    g.vecA = aligned16(X) && ld_x % 4 == 0;
    g.vecB = aligned16(W) && K % 4 == 0;
    return launch_gemm<true, true>(g, 1, (hipStream_t)stream);
}

// ---- backward helpers -----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_act_grad(const float* __restrict__ Y, int64_t ld_y, float* __restrict__ dY,
                                                  int64_t ld_dy, int64_t M, int N, int act) {
    const int64_t total = M * N;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / N;
        const int n = (int)(e % N);
        const float y = Y[m * ld_y + n];
        float d = dY[m * ld_dy + n];
        if (act == 1) d = y > 0.f ? d : 0.f;              // threshold_backward
        else if (act == 2) d = d * ((1.0f - y) * y);      // sigmoid_backward
        dY[m * ld_dy + n] = d;
    }
}

#define CS_ROWS 256
__global__ void __launch_bounds__(256) k_colsum_partial(const float* __restrict__ Z, int64_t ld, int64_t M, int N,
                                                        float* __restrict__ part) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
    float s = 0.f;
    for (int64_t m = r0; m < r1; ++m) s += Z[m * ld + n];
    part[(int64_t)blockIdx.y * N + n] = s;
}

__global__ void __launch_bounds__(256) k_reduce_slabs(const float* __restrict__ part, int64_t count, int splits,
                                                      float* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += part[(int64_t)z * count + e];
        out[e] = s;
    }
}

static int wgrad_splits(int64_t M, int N, int K) {
    const int64_t tiles = cdiv(N, GBM) * cdiv(K, GBN);
    int64_t s = cdiv(512, tiles);                     // aim at ~2 workgroups per CU
    const int64_t smax = cdiv(M, 4 * GBK);
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    return (int)s;
}

extern "C" uint64_t cdlrm_linear_bwd_work_bytes(int64_t M, int32_t N, int32_t K) {
    const uint64_t slabs = (uint64_t)wgrad_splits(M, N, K) * N * K * 4;
    const uint64_t cs = (uint64_t)cdiv(M, CS_ROWS) * N * 4;
    return ((slabs + 255) & ~(uint64_t)255) + ((cs + 255) & ~(uint64_t)255) + 256;
}

extern "C" int cdlrm_linear_bwd(const float* X, int64_t ld_x, const float* W, const float* Y, int64_t ld_y, float* dY,
                                int64_t ld_dy, float* dX, int64_t ld_dx, float* dW, float* db, int64_t M, int32_t N,
                                int32_t K, int32_t act, void* work, void* stream) {
    CDLRM_REQUIRE(X && W && dY && dW && work && M >= 1 && N >= 1 && K >= 1, "bad argument");
    CDLRM_REQUIRE(act == 0 || Y, "activation backward needs Y");
    CDLRM_REQUIRE(((uintptr_t)work & 255) == 0, "work must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (act != 0) {
        int64_t gx = cdiv(M * N, 256);
        if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(k_act_grad, dim3((unsigned)gx), dim3(256), 0, s, Y, ld_y, dY, ld_dy, M, N, act);
    }
    if (dX) {   // dX[M,K] = dZ[M,N] W[N,K]
        GemmArgs g;
        g.A = dY; g.lda = ld_dy; g.B = W; g.ldb = K; g.C = dX; g.ldc = ld_dx; g.slab = 0;
        g.M = M; g.N = K; g.K = N; g.kchunk = N; g.bias = nullptr; g.act = 0;
        g.vecA = aligned16(dY) && ld_dy % 4 == 0;
        g.vecB = aligned16(W) && K % 4 == 0;
        int rc = launch_gemm<true, false>(g, 1, s);
        if (rc) return rc;
    }
    // dW[N,K] = dZ[M,N]^T X[M,K], split over M into slabs summed in slab order
    const int splits = wgrad_splits(M, N, K);
    float* slabs = (float*)work;
    float* cs = (float*)((char*)work + ((((uint64_t)splits * N * K * 4) + 255) & ~(uint64_t)255));
    {
        GemmArgs g;
        g.A = dY; g.lda = ld_dy; g.B = X; g.ldb = ld_x; g.C = splits > 1 ? slabs : dW; g.ldc = K;
        g.slab = (int64_t)N * K;
        g.M = N; g.N = K; g.K = M; g.kchunk = cdiv(cdiv(M, splits), GBK) * GBK; g.bias = nullptr; g.act = 0;
        g.vecA = aligned16(dY) && ld_dy % 4 == 0;
        g.vecB = aligned16(X) && ld_x % 4 == 0;
        const int zs = (int)cdiv(M, g.kchunk);
        int rc = launch_gemm<false, false>(g, zs, s);
        if (rc) return rc;
        if (splits > 1) {
            int64_t gx = cdiv((int64_t)N * K, 256);
            if (gx > 2048) gx = 2048;
            hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)gx), dim3(256), 0, s, slabs, (int64_t)N * K, zs, dW);
        }
    }
    if (db) {
        const int ny = (int)cdiv(M, CS_ROWS);
        hipLaunchKernelGGL(k_colsum_partial, dim3((unsigned)cdiv(N, 256), (unsigned)ny), dim3(256), 0, s, dY, ld_dy, M, N,
                           cs);
        hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, cs, (int64_t)N, ny, db);
    }
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// K9: pairwise-dot interaction, one wave per sample, Z = T T^T on the 32x32x2 fp32 MFMA with the
// sample's [F, D] features staged in LDS ([32][D+1], conflict-free fragment reads).
// =================================================================================================
__device__ __forceinline__ int pair_base(int i, int itself) { return itself ? i * (i + 1) / 2 : i * (i - 1) / 2; }

__global__ void __launch_bounds__(256) k_interact_fwd(const float* __restrict__ feat, int64_t B, int F, int D, int itself,
                                                      float* __restrict__ R, int64_t ld_r) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ldt = D + 1;
    float* Ts = smem + wave * (32 * ldt + 32);
    for (int e = lane; e < 32 * ldt + 32; e += 64) Ts[e] = 0.f;
    const int D4 = D >> 2;
    const int64_t nb = cdiv_dev(B, 4);
    for (int64_t blk = blockIdx.x; blk < nb; blk += gridDim.x) {
        const int64_t b = blk * 4 + wave;
        const bool valid = b < B;
        __syncthreads();
        if (valid) {
            const float4* src = reinterpret_cast<const float4*>(feat + b * F * D);
            for (int e = lane; e < F * D4; e += 64) {
                const int row = e / D4, c = (e % D4) * 4;
                const float4 v = src[e];
                float* d = Ts + row * ldt + c;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
        __syncthreads();
        if (!valid) continue;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* tp = Ts + (lane & 31) * ldt + (lane >> 5);
        for (int k0 = 0; k0 < D; k0 += 2) {
            const float v = tp[k0];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v, v, acc, 0, 0, 0);
        }
        float* out = R + b * ld_r;
        for (int c = lane; c < D; c += 64) out[c] = Ts[c];
        const int j = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (i < F && j < i + (itself ? 1 : 0)) out[D + pair_base(i, itself) + j] = acc[r];
        }
    }
}

__global__ void __launch_bounds__(256) k_interact_bwd(const float* __restrict__ feat, const float* __restrict__ dR,
                                                      int64_t ld_r, int64_t B, int F, int D, int itself,
                                                      float* __restrict__ dfeat) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ldt = D + 1;
    const int npairs = pair_base(F, itself);
    const int per_wave = 32 * ldt + 32 + 32 * 33 + ((npairs + 3) & ~3);
    float* Ts = smem + wave * per_wave;
    float* Ss = Ts + 32 * ldt + 32;
    float* Gs = Ss + 32 * 33;
    for (int e = lane; e < 32 * ldt + 32; e += 64) Ts[e] = 0.f;
    const int D4 = D >> 2;
    const int off = itself ? 1 : 0;
    const int64_t nb = cdiv_dev(B, 4);
    for (int64_t blk = blockIdx.x; blk < nb; blk += gridDim.x) {
        const int64_t b = blk * 4 + wave;
        const bool valid = b < B;
        __syncthreads();
        if (valid) {
            const float4* src = reinterpret_cast<const float4*>(feat + b * F * D);
            for (int e = lane; e < F * D4; e += 64) {
                const int row = e / D4, c = (e % D4) * 4;
                const float4 v = src[e];
                float* d = Ts + row * ldt + c;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            const float* g = dR + b * ld_r + D;
            for (int e = lane; e < npairs; e += 64) Gs[e] = g[e];
        }
        __syncthreads();
        if (valid) {
            for (int e = lane; e < 1024; e += 64) {
                const int i = e >> 5, j = e & 31;
                float v = 0.f;
                if (i < F && j < F) {
                    if (j < i + off) v += Gs[pair_base(i, itself) + j];
                    if (i < j + off) v += Gs[pair_base(j, itself) + i];
                }
                Ss[i * 33 + j] = v;
            }
        }
        __syncthreads();
        if (!valid) continue;
        const float* sp = Ss + (lane & 31) * 33 + (lane >> 5);
        float* out = dfeat + b * F * D;
        const float* gx = dR + b * ld_r;
        for (int n0 = 0; n0 < D; n0 += 32) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* tp = Ts + (lane >> 5) * ldt + n0 + (lane & 31);
#pragma unroll
            for (int k0 = 0; k0 < 32; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sp[k0], tp[k0 * ldt], acc, 0, 0, 0);
            const int col = n0 + (lane & 31);
            if (col < D) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (i < F) out[i * D + col] = acc[r] + (i == 0 ? gx[col] : 0.f);
                }
            }
        }
    }
}

extern "C" int cdlrm_interact_fwd(const float* feat, int64_t B, int32_t F, int32_t D, int32_t itself, float* R,
                                  int64_t ld_r, void* stream) {
    CDLRM_REQUIRE(feat && R && F >= 1 && F <= 32 && D >= 4 && D % 4 == 0 && D <= 512, "unsupported shape (F<=32, D%4==0)");
    CDLRM_REQUIRE(aligned16(feat) && ld_r >= D + (itself ? F * (F + 1) / 2 : F * (F - 1) / 2), "alignment / ld_r");
    if (B == 0) return 0;
    const size_t lds = (size_t)4 * (32 * (D + 1) + 32) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        CDLRM_HIP_CHECK(hipFuncSetAttribute((const void*)k_interact_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    int64_t gx = cdiv(B, 4);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(k_interact_fwd, dim3((unsigned)gx), dim3(256), lds, (hipStream_t)stream, feat, B, F, D, itself, R, ld_r);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_interact_bwd(const float* feat, const float* dR, int64_t ld_r, int64_t B, int32_t F, int32_t D,
                                  int32_t itself, float* dfeat, void* stream) {
    CDLRM_REQUIRE(feat && dR && dfeat && F >= 1 && F <= 32 && D >= 4 && D % 4 == 0 && D <= 512, "unsupported shape");
    CDLRM_REQUIRE(aligned16(feat), "alignment");
    if (B == 0) return 0;
    const int npairs = itself ? F * (F + 1) / 2 : F * (F - 1) / 2;
    const size_t lds = (size_t)4 * (32 * (D + 1) + 32 + 32 * 33 + ((npairs + 3) & ~3)) * sizeof(float);
    CDLRM_REQUIRE(lds <= 160 * 1024, "LDS budget");
    static size_t attr = 0;
    if (lds > attr) {
        CDLRM_HIP_CHECK(hipFuncSetAttribute((const void*)k_interact_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    int64_t gx = cdiv(B, 4);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(k_interact_bwd, dim3((unsigned)gx), dim3(256), lds, (hipStream_t)stream, feat, dR, ld_r, B, F, D,
                       itself, dfeat);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// K11 BCELoss(mean) on the sigmoid output + its gradient; K12 dense SGD
// =================================================================================================
__global__ void __launch_bounds__(256) k_bce_partial(const float* __restrict__ Z, const float* __restrict__ T, int64_t n,
                                                     float* __restrict__ part, float* __restrict__ dZ) {
    __shared__ float red[4];
    float s = 0.f;
    const float inv_n = 1.0f / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float z = Z[i], t = T[i];
        // torch binary_cross_entropy: log terms clamped at -100
        const float l1 = fmaxf(logf(z), -100.f), l0 = fmaxf(log1pf(-z), -100.f);
        s += (t - 1.0f) * l0 - t * l1;
        // binary_cross_entropy_backward: (z - t) / max((1 - z) z, 1e-12) * grad, grad = 1/n
        if (dZ) dZ[i] = (z - t) / fmaxf((1.0f - z) * z, 1e-12f) * inv_n;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(64) k_bce_final(const float* __restrict__ part, int nparts, int64_t n,
                                                  float* __restrict__ loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if (threadIdx.x == 0) loss[0] = s / (float)n;
}

#define BCE_PARTS 64
extern "C" int cdlrm_bce_fwd_bwd(const float* Z, const float* target, int64_t n, float* loss_out, float* dZ, void* stream) {
    CDLRM_REQUIRE(Z && target && loss_out && n >= 1, "bad argument");
    // partial sums live behind the loss word: loss_out must have room for 1 + 64 floats
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bce_partial, dim3(BCE_PARTS), dim3(256), 0, s, Z, target, n, loss_out + 1, dZ);
    hipLaunchKernelGGL(k_bce_final, dim3(1), dim3(64), 0, s, loss_out + 1, BCE_PARTS, n, loss_out);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

__global__ void __launch_bounds__(256) k_sgd(float* __restrict__ p, const float* __restrict__ g, int64_t n, float lr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = fmaf(-lr, g[i], p[i]);
}

extern "C" int cdlrm_sgd_step(float* param, const float* grad, int64_t n, float lr, void* stream) {
    CDLRM_REQUIRE(param && grad && n >= 0, "bad argument");
    if (n == 0) return 0;
    int64_t gx = cdiv(n, 256);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(k_sgd, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, param, grad, n, lr);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// x /= divisor  (aggregate_gradients: layer.weight.grad /= world_size, main_no_ddp.py:239) -- a true
// division so the result matches the reference for every world size, not only powers of two
__global__ void __launch_bounds__(256) k_scale_div(float* __restrict__ x, int64_t n, float divisor) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = x[i] / divisor;
}

extern "C" int cdlrm_scale_div(float* x, int64_t n, float divisor, void* stream) {
    CDLRM_REQUIRE(x && n >= 0 && divisor != 0.f, "bad argument");
    if (n == 0) return 0;
    int64_t gx = cdiv(n, 256);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(k_scale_div, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, x, n, divisor);
    CDLRM_LAUNCH_CHECK();
    return 0;
}
