// Dense part of the training step on gfx950: FP32-MFMA Linear layers (K10), the pairwise-dot feature
// interaction (K9), BCE loss (K11) and dense SGD (K12).  Reference: model_no_ddp.py:244-316,
// main_no_ddp.py:212-221, 375, 415.
//
// fp32 everywhere (1e-5 loss-trajectory parity): v_mfma_f32_32x32x2_f32 is an exact k-ordered fp32
// fma chain at the fp32 vector peak (MI355X_MICROARCH.md "Matrix cores").
#include "gemm_wide.h"


// Linear forward with a short contraction (K <= 32: the first bottom layer reads the 13 dense features).  An MFMA
// tile would be mostly padding -- the small-batch GEMM spends 13 us at M = 1024 fetching clamped indices for 13.6 MFLOP --
// so this one runs on the vector ALU: a workgroup owns 32 rows x 128 columns, X tile and transposed W tile in LDS,
// each thread 4 x 4 outputs (x values broadcast, w values one conflict-free 16-byte read), 16-byte row-contiguous stores.
#define SK_KMAX 32
__global__ void __launch_bounds__(256) k_linear_smallk(const float* __restrict__ X, int64_t ld_x, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ Y, int64_t ld_y,
                                                       int64_t M, int N, int K, int act) {
    __shared__ float Xs[32][SK_KMAX + 1];
    __shared__ __attribute__((aligned(16))) float Wt[SK_KMAX][128 + 4];
    const int64_t m0 = (int64_t)blockIdx.y * 32;
    const int n0 = blockIdx.x * 128;
    for (int e = threadIdx.x; e < 32 * K; e += 256) {
        const int r = e / K, k = e % K;
        Xs[r][k] = X[min(m0 + r, M - 1) * ld_x + k];
    }
    for (int e = threadIdx.x; e < 128 * K; e += 256) {
        const int n = e / K, k = e % K;
        Wt[k][n] = W[(int64_t)min(n0 + n, N - 1) * K + k];
    }
    __syncthreads();
    const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k = 0; k < K; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(&Wt[k][4 * cg]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = Xs[rg * 4 + i][k];
            acc[i][0] = fmaf(x, w.x, acc[i][0]); acc[i][1] = fmaf(x, w.y, acc[i][1]);
            acc[i][2] = fmaf(x, w.z, acc[i][2]); acc[i][3] = fmaf(x, w.w, acc[i][3]);
        }
    }
    const int n = n0 + 4 * cg;
    if (n >= N) return;                 // N % 4 == 0: a column group is inside or outside as a whole
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = make_float4(bias[n], bias[n + 1], bias[n + 2], bias[n + 3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + rg * 4 + i;
        if (m >= M) break;
        float v[4] = {acc[i][0] + b.x, acc[i][1] + b.y, acc[i][2] + b.z, acc[i][3] + b.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (act == 1) v[j] = v[j] > 0.f ? v[j] : 0.f;
            else if (act == 2) v[j] = 1.0f / (1.0f + expf(-v[j]));
        }
        *reinterpret_cast<float4*>(Y + m * ld_y + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// The same layer with the WEIGHTS in registers (round 4): a lane owns 4 output columns and keeps their 4 x K weights, a wave
// walks rows -- the row's K inputs are wave-uniform (scalar loads), so a row costs 4 K fused multiply-adds per lane and one
// 1 KB store per wave, and nothing is staged through LDS.  Same arithmetic per output as k_linear_smallk (k ascending from
// 0, bias added last): bit-identical.  The LDS-tiled version above spent its 13 us at M = 8192 on tile set-up (a 32-row tile
// is 53 k multiply-adds behind two index-division loops and a barrier); this one is bound by the 16.8 MB it writes.
template <int K_>
__global__ void __launch_bounds__(256) k_linear_smallk_rows(const float* __restrict__ X, int64_t ld_x,
                                                            const float* __restrict__ W, const float* __restrict__ bias,
                                                            float* __restrict__ Y, int64_t ld_y, int64_t M, int N, int rows_per_wg,
                                                            int act) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ncg = N >> 8;                                   // column groups of 256 (N % 256 == 0)
    const int cgp = (int)(blockIdx.x % (unsigned)ncg);
    const int64_t r0 = (int64_t)(blockIdx.x / (unsigned)ncg) * rows_per_wg;
    const int n = cgp * 256 + lane * 4;
    float w[4][K_];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < K_; ++k) w[j][k] = W[(int64_t)(n + j) * K_ + k];
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = *reinterpret_cast<const float4*>(bias + n);
    int64_t r1 = r0 + rows_per_wg;
    if (r1 > M) r1 = M;
    float xn[K_];                                              // the next row's inputs, fetched one row ahead
    {
        const int64_t rf = r0 + wave < M ? r0 + wave : M - 1;
        const float* __restrict__ xr = X + rf * ld_x;          // wave-uniform address: scalar loads
#pragma unroll
        for (int k = 0; k < K_; ++k) xn[k] = xr[k];
    }
    for (int64_t r = r0 + wave; r < r1; r += 4) {
        float x[K_];
#pragma unroll
        for (int k = 0; k < K_; ++k) x[k] = xn[k];
        {
            const int64_t rn = r + 4 < M ? r + 4 : M - 1;
            const float* __restrict__ xr = X + rn * ld_x;
#pragma unroll
            for (int k = 0; k < K_; ++k) xn[k] = xr[k];
        }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < K_; ++k) a = fmaf(x[k], w[j][k], a);
            v[j] = a;
        }
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (act == 1) v[j] = v[j] > 0.f ? v[j] : 0.f;
            else if (act == 2) v[j] = 1.0f / (1.0f + expf(-v[j]));
        }
        *reinterpret_cast<float4*>(Y + r * ld_y + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

extern "C" int cdlrm_linear_fwd(const float* X, int64_t ld_x, const float* W, const float* bias, float* Y, int64_t ld_y,
                                int64_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    CDLRM_REQUIRE(X && W && Y && M >= 0 && N >= 1 && K >= 1 && ld_x >= K && ld_y >= N, "bad argument");
    const int alone = (act & CDLRM_GEMM_ALONE) != 0;       // scheduling hint riding on the activation code
    act &= ~CDLRM_GEMM_ALONE;
    CDLRM_REQUIRE(act >= 0 && act <= 2, "bad activation code");
    if (M == 0) return 0;
    CDLRM_CLEAR_STALE();
    // (Round 6, measured and removed: this layer on the matrix cores -- a wave owning 16 rows x 256 columns, the weights as
    //  16x16x4 fragments in registers, ascending k, bit-identical -- 10.9 us alone against 8.6 for the register kernel below at
    //  M = 8192 (64 scattered 4-byte weight loads per lane for 64 MFMAs), 0.5542 against 0.5519 ms per c3 step.)
    if (K == 13 && N % 256 == 0 && ld_y % 4 == 0 && aligned16(Y) && (!bias || aligned16(bias)) && M >= 256 &&
        !g_cdlrm_debug[0]) {
        // Rows per workgroup: ~512-1024 workgroups at the c3 batch; short batches get 8 rows per workgroup -- a wave's 52
        // weight loads are then amortised over two rows only, but 32 rows left 64 workgroups for 256 CUs at M = 1024 and the
        // kernel took 10.3 us in the per-rank step, against 6.6 for the LDS-tiled one.
        const int rpw = M >= 32768 ? 128 : M >= 4096 ? 32 : 8;
        const int64_t blocks = cdiv(M, rpw) * (N >> 8);
        if (blocks <= 0x7fffffff) {
            hipLaunchKernelGGL((k_linear_smallk_rows<13>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, ld_x, W,
                               bias, Y, ld_y, M, (int)N, rpw, (int)act);
            CDLRM_LAUNCH_CHECK();
            return 0;
        }
    }
    if (K <= SK_KMAX && N % 4 == 0 && ld_y % 4 == 0 && aligned16(Y) && cdiv(M, 32) <= 65535) {
        dim3 grid((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 32));
        hipLaunchKernelGGL(k_linear_smallk, grid, dim3(256), 0, (hipStream_t)stream, X, ld_x, W, bias, Y, ld_y, M, (int)N,
                           (int)K, (int)act);
        CDLRM_LAUNCH_CHECK();
        return 0;
    }
    GemmArgs g = gemm_args();
    g.A = X; g.lda = ld_x; g.B = W; g.ldb = K; g.C = Y; g.ldc = ld_y; g.slab = 0;
    g.M = M; g.N = N; g.K = K; g.kchunk = K; g.bias = bias; g.act = act;
    g.vecA = aligned16(X) && ld_x % 4 == 0 && K % 4 == 0;
    g.vecB = aligned16(W) && K % 4 == 0;
    g.alone = alone;
    return launch_gemm<true, true>(g, 1, (hipStream_t)stream);
}

// ---- backward helpers -----------------------------------------------------------------------------
// Stand-alone activation backward (dZ = dY * act'(Y) in place).  The training step does not need it: there the
// activation backward of layer l-1 rides in the epilogue of layer l's dgrad GEMM (x_act) and the loss kernel
// applies the final sigmoid's; this kernel serves callers that hand over post-activation gradients.
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

// Two reductions in one job (dW slabs and the bias-gradient partials of the same layer): logical blocks [0, gxa) work
// on part A, the rest on part B.  Fixed summation order (slab 0, 1, 2, ...): reproducible; 4 slabs in flight.
// (pA / pB: the parameters the gradients belong to, updated in the same pass -- p -= lr * g, the dense SGD step -- when the
//  caller has no exchange between the two: cdlrm_mlp_wgrad_sgd; NULL: gradients only)
__device__ __forceinline__ void reduce_slabs_body(int bid, const float* __restrict__ partA, int64_t countA, int splitsA,
                                                  float* __restrict__ outA, int gxa, const float* __restrict__ partB,
                                                  int64_t countB, int splitsB, float* __restrict__ outB,
                                                  float (*red)[64], float* __restrict__ pA = nullptr,
                                                  float* __restrict__ pB = nullptr, float lr = 0.f, int novec = 0) {
    if (bid >= gxa) {
        // part B (bias gradient): few elements, many partials -> 4 lanes per element, each summing every 4th
        // partial, combined in a fixed order through LDS
        const int c = threadIdx.x & 63, sub = threadIdx.x >> 6;
        const int64_t e = (int64_t)(bid - gxa) * 64 + c;
        float s = 0.f;
        if (e < countB)
            for (int z = sub; z < splitsB; z += 4) s += partB[(int64_t)z * countB + e];
        red[sub][c] = s;
        __syncthreads();
        if (sub == 0 && e < countB) {
            const float g = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
            outB[e] = g;
            if (pB) pB[e] = fmaf(-lr, g, pB[e]);
        }
        return;
    }
    if (!novec && (countA & 3) == 0 && ((((uintptr_t)partA | (uintptr_t)outA | (uintptr_t)pA) & 15) == 0)) {
        // 16-byte version (round 4): a thread owns four consecutive elements, eight slabs in flight; per element the same
        // additions in the same order as the scalar loop below (slab 0, 1, 2, ...): bit-identical
        const int64_t n4 = countA >> 2;
        const float4* __restrict__ p4 = reinterpret_cast<const float4*>(partA);
        for (int64_t e = (int64_t)bid * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gxa * blockDim.x) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            // (the parameter word of the fused SGD step travels WITH the first slabs, not behind the last: one load latency less
            //  at the end of a launch that sits on the training queue of every step)
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pA) q = reinterpret_cast<const float4*>(pA)[e];
            __builtin_amdgcn_sched_barrier(0);
            int z = 0;
            for (; z + 8 <= splitsA; z += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p4[(int64_t)(z + u) * n4 + e];
#pragma unroll
                for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
            for (; z < splitsA; ++z) {
                const float4 v = p4[(int64_t)z * n4 + e];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            reinterpret_cast<float4*>(outA)[e] = s;
            if (pA) {
                q.x = fmaf(-lr, s.x, q.x); q.y = fmaf(-lr, s.y, q.y); q.z = fmaf(-lr, s.z, q.z); q.w = fmaf(-lr, s.w, q.w);
                reinterpret_cast<float4*>(pA)[e] = q;
            }
        }
        return;
    }
    for (int64_t e = (int64_t)bid * blockDim.x + threadIdx.x; e < countA; e += (int64_t)gxa * blockDim.x) {
        float s = 0.f;
        int z = 0;
        for (; z + 4 <= splitsA; z += 4) {
            const float v0 = partA[(int64_t)z * countA + e], v1 = partA[(int64_t)(z + 1) * countA + e];
            const float v2 = partA[(int64_t)(z + 2) * countA + e], v3 = partA[(int64_t)(z + 3) * countA + e];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; z < splitsA; ++z) s += partA[(int64_t)z * countA + e];
        outA[e] = s;
        if (pA) pA[e] = fmaf(-lr, s, pA[e]);
    }
}

__global__ void __launch_bounds__(256) k_reduce_slabs(const float* __restrict__ partA, int64_t countA, int splitsA,
                                                      float* __restrict__ outA, int gxa,
                                                      const float* __restrict__ partB, int64_t countB, int splitsB,
                                                      float* __restrict__ outB) {
    __shared__ float red[4][64];
    reduce_slabs_body((int)blockIdx.x, partA, countA, splitsA, outA, gxa, partB, countB, splitsB, outB, red);
}

// the reductions of several layers in one launch (grouped weight gradients)
struct ReduceJob {
    const float* partA; int64_t countA; float* outA; int gxa;
    const float* partB; int64_t countB; float* outB;
    int splits;
    float* pA; float* pB; float lr;         // fused SGD step (NULL: none)
    int novec;                              // development: the scalar loop (g_cdlrm_debug[2])
};
struct ReduceGroup {
    int n;
    unsigned first[GEMM_GROUP_MAX + 1];
    ReduceJob j[GEMM_GROUP_MAX];
};
__global__ void __launch_bounds__(256) k_reduce_group(ReduceGroup grp) {
    __shared__ float red[4][64];
    int p = 0;
#pragma unroll
    for (int q = 1; q < GEMM_GROUP_MAX; ++q)
        if (q < grp.n && blockIdx.x >= grp.first[q]) p = q;
    const ReduceJob& r = grp.j[p];
    reduce_slabs_body((int)(blockIdx.x - grp.first[p]), r.partA, r.countA, r.splits, r.outA, r.gxa, r.partB, r.countB,
                      r.splits, r.outB, red, r.pA, r.pB, r.lr, r.novec);
}

// Weight-gradient contraction (over the batch) is cut into slabs when the batch is long: small batches go to the
// LDS-free kernel un-split (no reduction launch at all), long ones to the tiled kernel with ~1024 workgroups.
#define WGRAD_DIRECT_MAX_M 2048
static int wgrad_splits(int64_t M, int N, int K) {
    if (M <= WGRAD_DIRECT_MAX_M) return 1;
    const int64_t tiles = cdiv(N, 64) * cdiv(K, 64);    // the 64x64 tile launch_gemm picks for these shapes
    const int64_t target = 1024;        // measured at c3: 512 -> 0.687 ms/step, 1024 -> 0.668, 2048 the same
    int64_t s = cdiv(target, tiles);                  // aim at ~4 workgroups of 64x64 per CU (2 of 128x64)
    const int64_t smax = cdiv(M, 8 * GBK);
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    return (int)s;
}

// (Round 5, measured and removed: a STREAMING kernel for the weight gradient of the 13-wide layer at long batches -- a thread owns
//  four columns of dZ, 16-byte loads, eight rows in flight, the slab's X rows in LDS, 128-row slabs into the grouped reduction;
//  correct against fp64 at M = 16384 / 20011 -- because the LDS-free MFMA kernel takes 251 us for it inside a c5 step (134 MB of
//  dZ, 17 us of HBM time).  In the step: c5 3.7080 against 3.7059 ms, a tie; forced at c3 0.5680 against 0.5599, at 4096 0.3394
//  against 0.3288.  That launch lies in the half of the step that is bound by the SUM of its work, not by any kernel's length.)
extern "C" uint64_t cdlrm_linear_bwd_work_bytes(int64_t M, int32_t N, int32_t K) {
    const uint64_t splits = (uint64_t)wgrad_splits(M, N, K);
    const uint64_t slabs = splits * N * K * 4;
    const uint64_t cs = splits * N * 4;
    return ((slabs + 255) & ~(uint64_t)255) + ((cs + 255) & ~(uint64_t)255) + 256;
}

// scratch for cdlrm_mlp_wgrad: every layer's slabs at once on the grouped small-batch path, the largest layer's
// otherwise
extern "C" uint64_t cdlrm_mlp_wgrad_work_bytes(int32_t n_layers, int64_t M, const int32_t* N, const int32_t* K) {
    uint64_t total = 256, largest = 0;
    if (!N || !K) return 0;
    if (M <= WGRAD_DIRECT_MAX_M) {
        const uint64_t zs = (uint64_t)cdiv(M, 4 * GBK) + 1;         // upper bound of the split count
        for (int i = 0; i < n_layers; ++i)
            total += ((zs * N[i] * K[i] * 4 + 255) & ~(uint64_t)255) + ((zs * N[i] * 4 + 255) & ~(uint64_t)255);
        return total;
    }
    (void)largest;
    for (int i = 0; i < n_layers; ++i) total += cdlrm_linear_bwd_work_bytes(M, N[i], K[i]);      // every layer its own slabs
    return total;
}

extern "C" int cdlrm_linear_bwd(const float* X, int64_t ld_x, const float* W, const float* Y, int64_t ld_y, float* dY,
                                int64_t ld_dy, float* dX, int64_t ld_dx, float* dW, float* db, int64_t M, int32_t N,
                                int32_t K, int32_t act, int32_t x_act, void* work, void* stream) {
    CdlrmStopScope stop_scope;          // (first: every exit below flushes an attached completion event)
    const int alone = (act & CDLRM_GEMM_ALONE) != 0;       // scheduling hint for the dgrad GEMM, riding on the activation code
    act &= ~CDLRM_GEMM_ALONE;
    CDLRM_REQUIRE(X && W && dY && work && M >= 1 && N >= 1 && K >= 1, "bad argument");
    CDLRM_REQUIRE(dW || !db, "db without dW (the bias gradient is a by-product of the weight-gradient GEMM)");
    CDLRM_REQUIRE(act == 0 || Y, "activation backward needs Y");
    CDLRM_REQUIRE(act >= 0 && act <= 2 && x_act >= 0 && x_act <= 2, "bad activation code");
    CDLRM_REQUIRE(((uintptr_t)work & 255) == 0, "work must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // a completion event waiting for this call (cdlrm_event_attach_next) rides on the dgrad GEMM when that is the call's only
    // launch (the training step's use); with several launches it is recorded behind the last one
    if (act != 0 || dW || !dX) stop_scope.hold(s);
    const int splits = wgrad_splits(M, N, K);
    float* slabs = (float*)work;
    float* cs = (float*)((char*)work + ((((uint64_t)splits * N * K * 4) + 255) & ~(uint64_t)255));
    if (act != 0) {     // dZ = dY * act'(Y) in place
        const int64_t nb = cdiv(M * N, 256), blocks = nb < 2048 ? nb : 2048;
        hipLaunchKernelGGL(k_act_grad, dim3((unsigned)blocks), dim3(256), 0, s, Y, ld_y, dY, ld_dy, M, N, act);
    }
    if (dX) {   // dX[M,K] = dZ[M,N] W[N,K]  (* act'(X) when X is the activation output of the layer below)
        GemmArgs g = gemm_args();
        g.A = dY; g.lda = ld_dy; g.B = W; g.ldb = K; g.C = dX; g.ldc = ld_dx; g.slab = 0;
        g.M = M; g.N = K; g.K = N; g.kchunk = N; g.bias = nullptr; g.act = 0;
        g.vecA = aligned16(dY) && ld_dy % 4 == 0 && N % 4 == 0;
        g.vecB = aligned16(W) && K % 4 == 0;
        g.mask = X; g.ldmask = ld_x; g.mask_act = x_act;
        g.alone = alone;
        int rc = launch_gemm<true, false>(g, 1, s);
        if (rc) return rc;
    }
    // dW[N,K] = dZ[M,N]^T X[M,K], split over M into slabs summed in slab order; the first column panel of the
    // same GEMM sums dZ over the batch (bias gradient)
    if (dW) {
        GemmArgs g = gemm_args();
        g.A = dY; g.lda = ld_dy; g.B = X; g.ldb = ld_x; g.ldc = K;
        g.slab = (int64_t)N * K;
        g.M = N; g.N = K; g.K = M; g.kchunk = cdiv(cdiv(M, splits), GBK) * GBK; g.bias = nullptr; g.act = 0;
        g.vecA = aligned16(dY) && ld_dy % 4 == 0 && N % 4 == 0;
        g.vecB = aligned16(X) && ld_x % 4 == 0 && K % 4 == 0;
        const int zs = (int)cdiv(M, g.kchunk);      // <= splits
        g.C = zs > 1 ? slabs : dW;
        g.colsum = db ? (zs > 1 ? cs : db) : nullptr;
        int rc = launch_gemm<false, false>(g, zs, s);
        if (rc) return rc;
        if (zs > 1) {   // one launch sums the dW slabs and the bias-gradient partials
            int64_t gxa = cdiv((int64_t)N * K, 256);
            if (gxa > 2048) gxa = 2048;
            const int64_t gxb = db ? cdiv(N, 64) : 0;
            hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)(gxa + gxb)), dim3(256), 0, s, slabs, (int64_t)N * K, zs, dW,
                               (int)gxa, cs, (int64_t)N, zs, db);
        }
    }
    CDLRM_LAUNCH_CHECK();
    return 0;                           // (stop_scope records an event no launch carried)
}

// (Round 4 built the two layers with a thin side -- the 13-wide first layer and the 1-wide output layer -- as vector-ALU
//  reductions over the batch: lanes <-> columns of the wide operand, the thin operand's rows staged in LDS and read back as
//  broadcasts, sixteen rows in flight per wave, the four waves of a workgroup summed through LDS in a fixed order, into the same
//  split-M slabs.  Parity-green, 12.5 / 11.4 us stand-alone with their reductions at M = 8192 -- and a TIE in the step against
//  the LDS-free MFMA kernel they were to replace: 0.6125 against 0.6113 ms (13-wide), 0.6148 against 0.6146 (1-wide), same box,
//  same process, tools/ab_step.py.  Where these launches sit -- the end of the backward, three queues deep -- a kernel's
//  duration is set by what runs beside it (30 us in the trace for either version), not by its own instruction mix.  Removed.)
// Weight (and bias) gradients of SEVERAL layers at once, from the pre-activation gradients dZ[i] that the dgrad chain
// left behind (cdlrm_linear_bwd with dW = NULL): dW[i] = dZ[i]^T X[i], db[i] = column sums of dZ[i].
// Small batches: all layers in one grouped launch of the LDS-free kernel (no slabs, no reduction); long batches: the
// tiled split-M path, layer after layer.
__global__ void __launch_bounds__(256) k_sgd(float* __restrict__ p, const float* __restrict__ g, int64_t n, float lr);

// P_w / P_b (both or neither): the layers' parameters, stepped by -lr * gradient in the reduction pass (layers whose
// gradient needs no reduction: one elementwise launch behind it)
static int mlp_wgrad_impl(int32_t n_layers, const float* const* X, const int64_t* ld_x, const float* const* dZ,
                          const int64_t* ld_dz, float* const* dW, float* const* db, int64_t M, const int32_t* N,
                          const int32_t* K, void* work, void* stream, float* const* P_w, float* const* P_b, float lr) {
    CDLRM_REQUIRE(n_layers >= 0 && (n_layers == 0 || (X && ld_x && dZ && ld_dz && dW && db && N && K)) && M >= 1,
                  "bad argument");
    hipStream_t s = (hipStream_t)stream;
    std::vector<char> stepped((size_t)(n_layers > 0 ? n_layers : 0), 0);
    auto step_rest = [&]() -> int {             // layers the reduction did not cover
        if (!P_w) return 0;
        for (int i = 0; i < n_layers; ++i) {
            if (stepped[i]) continue;
            const int64_t cnt = (int64_t)N[i] * K[i];
            int64_t gx = cdiv(cnt, 256);
            if (gx > 2048) gx = 2048;
            hipLaunchKernelGGL(k_sgd, dim3((unsigned)gx), dim3(256), 0, s, P_w[i], (const float*)dW[i], cnt, lr);
            if (db[i] && P_b && P_b[i])
                hipLaunchKernelGGL(k_sgd, dim3((unsigned)cdiv(N[i], 256)), dim3(256), 0, s, P_b[i], (const float*)db[i], (int64_t)N[i], lr);
        }
        CDLRM_LAUNCH_CHECK();
        return 0;
    };
    if (M <= WGRAD_DIRECT_MAX_M) {
        // Small batches, all layers at once.  Layers whose operands are 16-byte loadable go through the LDS-tiled
        // kernel as ONE grouped launch, the contraction (the batch) cut into slabs so that the group has ~1000
        // workgroups; the rest (13-wide input, 1-wide output) through the grouped LDS-free kernel with the same slabs;
        // then ONE grouped reduction of all layers (fixed slab order).  
        const int use_tiled = 1;
        std::vector<GemmArgs> direct, tiled;
        std::vector<int> direct_layer, tiled_layer;
        int64_t tiles = 0;
        for (int i = 0; i < n_layers; ++i) {
            CDLRM_REQUIRE(X[i] && dZ[i] && dW[i] && N[i] >= 1 && K[i] >= 1 && ld_x[i] >= K[i] && ld_dz[i] >= N[i],
                          "bad layer argument");
            GemmArgs g = gemm_args();
            g.A = dZ[i]; g.lda = ld_dz[i]; g.B = X[i]; g.ldb = ld_x[i]; g.C = dW[i]; g.ldc = K[i];
            g.M = N[i]; g.N = K[i]; g.K = M; g.kchunk = M; g.colsum = db[i];
            g.vecA = aligned16(dZ[i]) && ld_dz[i] % 4 == 0 && N[i] % 4 == 0;
            g.vecB = aligned16(X[i]) && ld_x[i] % 4 == 0 && K[i] % 4 == 0;
            if (use_tiled && g.vecA && g.vecB && M >= 256) {
                tiled.push_back(g);
                tiled_layer.push_back(i);
                tiles += cdiv(N[i], 64) * cdiv(K[i], 64);
            } else {
                direct.push_back(g);
                direct_layer.push_back(i);
            }
        }
        // one split count for the whole group (slabs of the batch); 1 = no slabs, no reduction
        int64_t kchunk = M;
        int zs = 1;
        if (!tiled.empty()) {
            // workgroups the grouped launch aims at (slabs = target / tiles).  1024 until round 4 (7 slabs for the top MLP's 160
            // tiles); measured in the step, five rounds each: per-rank batch 1024 -- 384 / 640 / 768 / 896 / 1024 / 1536 ->
            // 0.1802 / 0.1814 / 0.1797 / 0.1813 / 0.1856 / 0.1851 ms; 2048 -- 256 / 384 / 512 / 640 / 768 / 1024 / 2048 ->
            // 0.2495 / 0.2489 / 0.2372 / 0.2367 / 0.2404 / 0.2416 / 0.2405: four to five slabs, not seven (fewer partial
            // slabs to write and reduce; three is too few workgroups at 2048)
            const int64_t target_wgs = M <= 1024 ? 768 : 640;
            int64_t splits = cdiv(target_wgs, tiles);
            const int64_t smax = cdiv(M, 4 * GBK);
            if (splits > smax) splits = smax;
            if (splits < 1) splits = 1;
            kchunk = cdiv(cdiv(M, splits), GBK) * GBK;
            zs = (int)cdiv(M, kchunk);
        }
        CDLRM_REQUIRE(zs == 1 || (work && ((uintptr_t)work & 255) == 0), "work must be 256-byte aligned");
        char* wp = (char*)work;
        std::vector<ReduceJob> jobs;
        auto slabbed = [&](GemmArgs& g, int li) {          // redirect one problem's outputs to its slabs
            const int64_t cnt = (int64_t)N[li] * K[li];
            g.kchunk = kchunk;
            g.slab = cnt;
            if (zs == 1) return;
            float* slabs = (float*)wp;
            wp += (((uint64_t)zs * cnt * 4) + 255) & ~(uint64_t)255;
            float* cs = (float*)wp;
            wp += (((uint64_t)zs * N[li] * 4) + 255) & ~(uint64_t)255;
            g.C = slabs;
            g.colsum = db[li] ? cs : nullptr;
            ReduceJob r;
            r.partA = slabs; r.countA = cnt; r.outA = dW[li];
            int64_t gxa = cdiv(cnt, 1024);
            if (gxa > 256) gxa = 256;
            r.gxa = (int)gxa;
            r.partB = cs; r.countB = db[li] ? N[li] : 0; r.outB = db[li];
            r.splits = zs;
            r.novec = g_cdlrm_debug[2];
            r.pA = P_w ? P_w[li] : nullptr; r.pB = (P_w && P_b && db[li]) ? P_b[li] : nullptr; r.lr = lr;
            if (P_w) stepped[li] = 1;
            jobs.push_back(r);
        };
        for (size_t q = 0; q < direct.size(); ++q) slabbed(direct[q], direct_layer[q]);
        for (size_t q = 0; q < tiled.size(); ++q) slabbed(tiled[q], tiled_layer[q]);
        bool direct_done = direct.empty();
        for (size_t q0 = 0; q0 < tiled.size(); q0 += GEMM_GROUP_MAX) {
            GemmGroup grp;
            memset(&grp, 0, sizeof(grp));
            unsigned blocks = 0;
            for (size_t q = q0; q < tiled.size() && q < q0 + GEMM_GROUP_MAX; ++q) {
                grp.first[grp.n] = blocks;
                grp.g[grp.n] = tiled[q];
                blocks += (unsigned)(cdiv(tiled[q].M, 64) * cdiv(tiled[q].N, 64) * zs);
                grp.n++;
            }
            grp.first[grp.n] = blocks;
            // the LDS-free layers ride in front of the first tiled group's launch where the combination allows (one launch less
            // per sub-network on a launch-bound step: 0.2008 -> 0.1963 ms at a local batch of 1024, 0.2955 -> 0.2883 at 2048)
            if (!direct_done && launch_wgrad_mixed(direct.data(), (int)direct.size(), grp, blocks, s)) {
                direct_done = true;
                continue;
            }
            hipLaunchKernelGGL((k_gemm_group<false, false, true, true>), dim3(blocks), dim3(256), 0, s, grp);
        }
        if (!direct_done) {
            int rc = launch_wgrad_group(direct.data(), (int)direct.size(), s);
            if (rc) return rc;
        }
        for (size_t q0 = 0; q0 < jobs.size(); q0 += GEMM_GROUP_MAX) {
            ReduceGroup red;
            memset(&red, 0, sizeof(red));
            unsigned rblocks = 0;
            for (size_t q = q0; q < jobs.size() && q < q0 + GEMM_GROUP_MAX; ++q) {
                red.first[red.n] = rblocks;
                red.j[red.n] = jobs[q];
                rblocks += (unsigned)(jobs[q].gxa + cdiv(jobs[q].countB, 64));
                red.n++;
            }
            red.first[red.n] = rblocks;
            hipLaunchKernelGGL(k_reduce_group, dim3(rblocks), dim3(256), 0, s, red);
        }
        CDLRM_LAUNCH_CHECK();
        return step_rest();
    }
    // Long batches: per layer one split-M GEMM of the tiled (or, for degenerate shapes, the LDS-free) kernel into the
    // layer's own slabs, then ONE grouped reduction of all layers' slabs and bias partials.
    CDLRM_REQUIRE(work && ((uintptr_t)work & 255) == 0, "work must be 256-byte aligned");
    char* wp = (char*)work;
    std::vector<ReduceJob> jobs;
    for (int i = 0; i < n_layers; ++i) {
        CDLRM_REQUIRE(X[i] && dZ[i] && dW[i] && N[i] >= 1 && K[i] >= 1 && ld_x[i] >= K[i] && ld_dz[i] >= N[i],
                      "bad layer argument");
        const int splits = wgrad_splits(M, N[i], K[i]);
        const int64_t cnt = (int64_t)N[i] * K[i];
        GemmArgs g = gemm_args();
        g.A = dZ[i]; g.lda = ld_dz[i]; g.B = X[i]; g.ldb = ld_x[i]; g.ldc = K[i];
        g.slab = cnt;
        g.M = N[i]; g.N = K[i]; g.K = M; g.kchunk = cdiv(cdiv(M, splits), GBK) * GBK;
        g.vecA = aligned16(dZ[i]) && ld_dz[i] % 4 == 0 && N[i] % 4 == 0;
        g.vecB = aligned16(X[i]) && ld_x[i] % 4 == 0 && K[i] % 4 == 0;
        const int zs = (int)cdiv(M, g.kchunk);
        g.C = dW[i];
        g.colsum = db[i];
        if (zs > 1) {
            float* slabs = (float*)wp;
            wp += (((uint64_t)zs * cnt * 4) + 255) & ~(uint64_t)255;
            float* cs = (float*)wp;
            wp += (((uint64_t)zs * N[i] * 4) + 255) & ~(uint64_t)255;
            g.C = slabs;
            g.colsum = db[i] ? cs : nullptr;
            ReduceJob r;
            r.partA = slabs; r.countA = cnt; r.outA = dW[i];
            int64_t gxa = cdiv(cnt, 256);
            if (gxa > 1024) gxa = 1024;
            r.gxa = (int)gxa;
            r.partB = cs; r.countB = db[i] ? N[i] : 0; r.outB = db[i];
            r.splits = zs;
            r.novec = g_cdlrm_debug[2];
            r.pA = P_w ? P_w[i] : nullptr; r.pB = (P_w && P_b && db[i]) ? P_b[i] : nullptr; r.lr = lr;
            if (P_w) stepped[i] = 1;
            jobs.push_back(r);
        }
        int rc = launch_gemm<false, false>(g, zs, s);
        if (rc) return rc;
    }
    for (size_t q0 = 0; q0 < jobs.size(); q0 += GEMM_GROUP_MAX) {
        ReduceGroup red;
        memset(&red, 0, sizeof(red));
        unsigned rblocks = 0;
        for (size_t q = q0; q < jobs.size() && q < q0 + GEMM_GROUP_MAX; ++q) {
            red.first[red.n] = rblocks;
            red.j[red.n] = jobs[q];
            rblocks += (unsigned)(jobs[q].gxa + cdiv(jobs[q].countB, 64));
            red.n++;
        }
        red.first[red.n] = rblocks;
        hipLaunchKernelGGL(k_reduce_group, dim3(rblocks), dim3(256), 0, s, red);
    }
    CDLRM_LAUNCH_CHECK();
    return step_rest();
}

extern "C" int cdlrm_mlp_wgrad(int32_t n_layers, const float* const* X, const int64_t* ld_x, const float* const* dZ,
                               const int64_t* ld_dz, float* const* dW, float* const* db, int64_t M, const int32_t* N,
                               const int32_t* K, void* work, void* stream) {
    return mlp_wgrad_impl(n_layers, X, ld_x, dZ, ld_dz, dW, db, M, N, K, work, stream, nullptr, nullptr, 0.f);
}

// cdlrm_mlp_wgrad followed by the dense SGD step of the same layers (W[i] -= lr * dW[i], b[i] -= lr * db[i]:
// optimizer_mlps.step(), main_no_ddp.py:415) in the same launches -- for callers with nothing between the two (one rank: no
// gradient exchange).  Same arithmetic as cdlrm_mlp_wgrad + cdlrm_sgd_step.
extern "C" int cdlrm_mlp_wgrad_sgd(int32_t n_layers, const float* const* X, const int64_t* ld_x, const float* const* dZ,
                                   const int64_t* ld_dz, float* const* dW, float* const* db, float* const* W,
                                   float* const* b, float lr, int64_t M, const int32_t* N, const int32_t* K, void* work,
                                   void* stream) {
    CDLRM_REQUIRE(n_layers == 0 || (W && b), "parameters missing");
    return mlp_wgrad_impl(n_layers, X, ld_x, dZ, ld_dz, dW, db, M, N, K, work, stream, W, b, lr);
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
                                                      int64_t ld_r, int64_t B, int F, int D, int itself, int x_act,
                                                      float* __restrict__ dfeat) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ldt = D + 1;
    const int npairs = pair_base(F, itself);
    const int per_wave = 32 * ldt + 32 + ((npairs + 3) & ~3);
    float* Ts = smem + wave * per_wave;
    float* Gs = Ts + 32 * ldt + 32;
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
        if (!valid) continue;
        // A fragments of S = G + G^T straight from the staged gradient: lane holds S[i][lk + 2m], m = 0..15
        float sreg[16];
        {
            const int i = lane & 31, lk = lane >> 5;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int j = lk + 2 * m;
                float v = 0.f;
                if (i < F && j < F) {
                    if (j < i + off) v += Gs[pair_base(i, itself) + j];
                    if (i < j + off) v += Gs[pair_base(j, itself) + i];
                }
                sreg[m] = v;
            }
        }
        float* out = dfeat + b * F * D;
        const float* gx = dR + b * ld_r;
        for (int n0 = 0; n0 < D; n0 += 32) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* tp = Ts + (lane >> 5) * ldt + n0 + (lane & 31);
#pragma unroll
            for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sreg[m], tp[2 * m * ldt], acc, 0, 0, 0);
            const int col = n0 + (lane & 31);
            if (col < D) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (i < F) {
                        float v = acc[r];
                        if (i == 0) {       // the dense feature: + direct path, * act'(bottom-MLP output) if asked
                            v += gx[col];
                            const float y = Ts[col];
                            if (x_act == 1) v = y > 0.f ? v : 0.f;
                            else if (x_act == 2) v = v * ((1.0f - y) * y);
                        }
                        out[i * D + col] = v;
                    }
                }
            }
        }
    }
}

// ---- software-pipelined variants (D = 32 / 64 / 128 / 256) ---------------------------------------------------
// One wave per sample as above, but a wave walks its samples with the NEXT sample's rows already in flight: global ->
// registers (prefetch) -> LDS (ds_write_b128, row pitch D+4 floats = odd multiple of 16 B: conflict-free b128 writes
// and fragment reads) -> MFMA.  The feature matrix is both MFMA operands of T T^T, so one ds_read_b128 feeds four
// MFMA steps (lane half lk holds contraction indices 8g + 4 lk .. +3, as in gemm.h).  No workgroup barrier: every
// wave owns its LDS slice.  HBM-bound by design: 15.7 KB in + 1.9 KB out per sample against 64 MFMAs.
typedef float v4f __attribute__((ext_vector_type(4)));
#ifdef IA_STAMP     // tools/interact_ablate.hip: s_memtime stamps of wave 0 of workgroup 0 (4 per sample)
__device__ long long g_ia_stamp[64];
#define IA_STAMP_DECL int ia_n = 0;
#define IA_STAMP_AT(k) { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && threadIdx.x == 0 && ia_n < 60) g_ia_stamp[ia_n++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define IA_STAMP_DECL
#define IA_STAMP_AT(k)
#endif
#ifndef IA_ABL
#define IA_ABL 0      // tools/interact_ablate.hip: 1 no MFMA, 2 no global loads in the loop, 3 no output stores
#endif

template <int D4>
__device__ __forceinline__ void interact_prefetch(const float* __restrict__ feat, int64_t b, int FD4, int lane,
                                                  v4f (&v)[D4 / 2]) {
    const v4f* src = reinterpret_cast<const v4f*>(feat) + b * FD4;
#pragma unroll
    for (int i = 0; i < D4 / 2; ++i) v[i] = src[min(lane + 64 * i, FD4 - 1)];       // straight-line, clamped
}

template <int D4>
__device__ __forceinline__ void interact_stage(float* __restrict__ Ts, int FD4, int lane, const v4f (&v)[D4 / 2]) {
    constexpr int PITCH = 4 * D4 + 4;
    // unconditional: slots past row F-1 receive (finite) copies of the last element; rows >= F never reach an output
    // (forward: only pairs i, j < F are stored; backward: their S coefficients are zero).  A condition here makes
    // the compiler keep the prefetch registers in scratch memory.
#pragma unroll
    for (int i = 0; i < D4 / 2; ++i) {
        const int e = lane + 64 * i;
        *reinterpret_cast<v4f*>(Ts + (e / D4) * PITCH + (e % D4) * 4) = v[i];
    }
}

template <int D4>
__global__ void __launch_bounds__(256) k_interact_fwd_p(const float* __restrict__ feat, int64_t B, int F, int itself,
                                                        float* __restrict__ R, int64_t ld_r) {
    constexpr int D = 4 * D4, PITCH = D + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* Ts = smem + wave * (32 * PITCH);
    for (int e = lane; e < 32 * PITCH; e += 64) Ts[e] = 0.f;        // rows F..31 stay zero
    const int FD4 = F * D4;
    const int64_t nw = (int64_t)gridDim.x * 4;
    int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= B) return;
    v4f nxt[D4 / 2];
    interact_prefetch<D4>(feat, b, FD4, lane, nxt);
    IA_STAMP_DECL
    auto one = [&](int64_t b) __attribute__((always_inline)) {
        IA_STAMP_AT(0)
        interact_stage<D4>(Ts, FD4, lane, nxt);
        IA_STAMP_AT(1)
#if IA_ABL != 2
        interact_prefetch<D4>(feat, min(b + nw, B - 1), FD4, lane, nxt);
#endif
        __builtin_amdgcn_sched_barrier(0);
        // Z = T T^T is symmetric and only its lower triangle is stored: three 16x16 tiles -- (0,0), (1,0), (1,1) -- on
        // the 16x16x4 fp32 MFMA (same FLOP rate as 32x32x2) instead of one 32x32 tile: 96 MFMAs x 32 cycles against
        // 64 x 64, and three independent accumulator chains.  Lane l = (row l % 16 of its tile's 16 rows, contraction
        // group l / 16): one ds_read_b128 per row block holds contraction indices 16g + 4 (l/16) .. +3, four MFMA steps.
        v4f acc00 = {0.f, 0.f, 0.f, 0.f}, acc10 = acc00, acc11 = acc00;
        const int l16 = lane & 15, g4 = lane >> 4;
        const float* tp = Ts + l16 * PITCH + 4 * g4;
#if IA_ABL == 1
        acc00[0] = tp[0];
#else
#pragma unroll 2
#endif
        for (int g = 0; g < (IA_ABL == 1 ? 0 : D / 16); ++g) {
            const float4 a0 = *reinterpret_cast<const float4*>(tp + 16 * g);                  // rows 0..15
            const float4 a1 = *reinterpret_cast<const float4*>(tp + 16 * PITCH + 16 * g);     // rows 16..31
            acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, a0.x, acc00, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, a0.x, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, a1.x, acc11, 0, 0, 0);
            acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, a0.y, acc00, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, a0.y, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, a1.y, acc11, 0, 0, 0);
            acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, a0.z, acc00, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, a0.z, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, a1.z, acc11, 0, 0, 0);
            acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, a0.w, acc00, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, a0.w, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, a1.w, acc11, 0, 0, 0);
        }
        IA_STAMP_AT(2)
        float* out = R + b * ld_r;
        const int off = itself ? 1 : 0;
#if IA_ABL == 3
        if (acc00[0] + acc10[1] + acc11[2] == 1.2345f) out[0] = 1.f;
#else
        for (int c = lane; c < D; c += 64) out[c] = Ts[c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // accumulator register r of lane l: tile row 4 (l / 16) + r, tile column l % 16
            const int i0 = 4 * g4 + r, i1 = 16 + 4 * g4 + r;
            if (i0 < F && l16 < i0 + off) out[D + pair_base(i0, itself) + l16] = acc00[r];
            if (i1 < F) out[D + pair_base(i1, itself) + l16] = acc10[r];
            if (i1 < F && 16 + l16 < i1 + off) out[D + pair_base(i1, itself) + 16 + l16] = acc11[r];
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        IA_STAMP_AT(3)
    };
    for (; b < B; b += nw) one(b);
}

// Column-slab forward (the kernel the step runs): the sample's [F, D] block is staged NS slabs of D / NS columns at a time into a
// slice of 32 x (D / NS + 4) floats; the three accumulator chains run on across the slabs -- the contraction order of
// k_interact_fwd_p, bit-identical results --; each slab's registers are refilled with the NEXT sample's slab right after they are
// staged, so a wave's loads, LDS traffic and MFMAs interleave at slab granularity instead of sample granularity.  That, not
// occupancy, is what pays (tools/interact_ablate.hip, B = 8192, F = 27, D = 128): 22.6 us on ONE workgroup per CU (a wave streams
// 8 samples) against 29.3 us for whole-row staging on two; with more waves per CU (the slice is 4.6 KB instead of 16.9 KB: up to
// 20 fit) the kernel gets SLOWER, 24.2 us at 8, 26.6-27.5 us at 12-20.  Rows F .. 31 of a slab are copies of row F - 1 (clamped
// row index in the load: finite values whose products land in rows / columns >= F of Z, never stored).
// The output row [dense D | pairs | pad] is put together in a staging row of its own and leaves as whole float4 words: a fixed
// number of unconditional, fully coalesced stores.  (Storing the accumulators straight from their lanes is a dozen 4-byte stores
// under lane-dependent conditions; behind them the compiler can no longer count the outstanding memory operations and makes the
// next sample's staging wait for vmcnt(0) -- for these stores' acknowledgements -- in every iteration.)  The first sample is
// peeled off the loop so that the loop is only ever entered with the same memory operations in flight as its own back edge
// leaves -- the next sample's loads, then this one's output stores: the wait in front of the staging then counts the loads alone.
//
// G (the fused gather, cdlrm_gather_interact_fwd): rows 1 .. F-1 of a sample are not read from a feature block but straight from
// the cache rows the batch's slot ids name -- the sum-pool of a one-index bag IS its row (Criteo layout), so the [B, T, D] block
// the stand-alone gather writes (and this kernel reads back) never exists: 109 MB written + 113 MB read per c3 step that are
// not moved at all.  `feat` is then the dense feature alone (row 0, pitch ga.ldx4).  Per sample a lane needs the slot ids of
// the NR rows its registers cover; they are loaded TWO samples ahead (slot -> row address -> row is a dependent chain: the
// rows of the next sample are in flight while this one is computed, the slot ids of the one after are in flight behind them).
// Same operands in the same lanes, same MFMA order: bit-identical to gather + interaction.
struct IaGather {
    const TableDesc* tab;       // row_base of every table's cache rows
    const v4f* weight;          // cache rows (and the auxiliary rows behind them), D4 float4 words each
    const int32_t* slots;       // [T, n] slot ids of the batch (cdlrm_embbag_probe / cdlrm_embbag_take)
    int64_t n;
    int64_t ldx4;               // pitch of the dense-feature rows, float4 words
    const uint8_t* once;        // backward, ONCE: lookups whose slot occurs once in the batch (embbag_bwd.hip: the slot sort),
    int64_t ld_once;            //                 table t's flags at once + t * ld_once
    float lr;                   //                 learning rate of the embedding rows
};

// DB (round 6): the slab slice is DOUBLE-BUFFERED and the fragment reads run one 16-deep group ahead of the MFMAs.  With ONE
// slice a wave -- one per SIMD, in-order -- stages slab s, waits out the LDS round trip, reads the first fragments, waits again and
// only then issues MFMAs: ~250 cycles per slab in which its SIMD's matrix pipe idles, and with every row already on the die the
// kernel still took 22.7 us for 10.7 us of MFMA time (tools/gather_warm_cold.py).  Here slab s+1 (the NEXT sample's slab 0 behind
// the last one) is staged into the other slice in front of slab s's MFMAs and its first fragments are read behind slab s's last
// group: no LDS latency is left in the MFMA stream.  Same operands in the same lanes, same MFMA order: bit-identical.
template <int D4, int NS, bool G, bool DB = false>
__global__ void __launch_bounds__(256) k_interact_fwd_s(const float* __restrict__ feat, IaGather ga, int64_t B, int F,
                                                        int itself, float* __restrict__ R, int64_t ld_r) {
    constexpr int D = 4 * D4, CS = D4 / NS, DS = 4 * CS, PITCH = DS + 4, NR = CS / 2;
    constexpr int OSW = D + 532;                    // output staging row: D + up to 528 pairs + the pad word
    constexpr int TSW = (DB ? 2 : 1) * 32 * PITCH;  // slab slice(s) of a wave
    static_assert(!DB || (NS % 2 == 0 && DS / 16 >= 1), "double buffering alternates two slices over an even number of slabs");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* Ts = smem + wave * (TSW + OSW);
    float* Os = Ts + TSW;
    const int64_t nw = (int64_t)gridDim.x * 4;
    int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int l16 = lane & 15, g4 = lane >> 4;
    const int off = itself ? 1 : 0;
    // per register i of a slab: (row, chunk) of this lane's 16 bytes; the source row is clamped to F - 1
    int soff[NR], doff[NR];
    int srow[G ? NR : 1];               // G: where the row's slot ids start in `slots` (T * n < 2^31, checked by the host) ...
    const v4f* wbase[G ? NR : 1];       //    ... and this lane's 16 bytes of slot 0 of the row's table
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int e = lane + 64 * i, row = e / CS, c = e % CS;
        soff[i] = G ? c : min(row, F - 1) * D4 + c;
        doff[i] = row * PITCH + 4 * c;
        if constexpr (G) {
            const int t = max(min(row, F - 1) - 1, 0);
            srow[i] = t * (int)ga.n;
            wbase[i] = ga.weight + (ga.tab[t].row_base * D4 + c);
        }
    }
    v4f nxt[NS][NR];
    const v4f* pn[G ? NR : 1];          // G: this lane's NR sources of the sample whose rows are fetched next
    int32_t sl[G ? NR : 1];             //    and the slot ids of the sample after it
    auto load_slots = [&](int64_t bb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NR; ++i) sl[i] = ga.slots[srow[i] + bb];
    };
    auto make_ptrs = [&](int64_t bb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const v4f* p = wbase[i] + (int64_t)sl[i] * D4;
            if (i == 0 && lane < CS) p = reinterpret_cast<const v4f*>(feat) + bb * ga.ldx4 + soff[i];   // row 0: the dense feature
            pn[i] = p;
        }
    };
    auto prefetch = [&](int s, int64_t bb) __attribute__((always_inline)) {
        if constexpr (G) {
#pragma unroll
            for (int i = 0; i < NR; ++i) nxt[s][i] = pn[i][s * CS];
        } else {
            const v4f* src = reinterpret_cast<const v4f*>(feat) + bb * (int64_t)F * D4 + s * CS;
#pragma unroll
            for (int i = 0; i < NR; ++i) nxt[s][i] = src[soff[i]];
        }
    };
    if constexpr (G) {
        load_slots(b);
        make_ptrs(b);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) prefetch(s, b);
    if constexpr (G) load_slots(min(b + nw, B - 1));
    if constexpr (DB) {
        // ---- the double-buffered form (see above).  Slices Ts (even slabs) and Ts + 32 * PITCH (odd slabs).
        float* const T0 = Ts;
        float* const T1 = Ts + 32 * PITCH;
        constexpr int NG = DS / 16;                 // 16-deep groups per slab
#pragma unroll
        for (int i = 0; i < NR; ++i) *reinterpret_cast<v4f*>(T0 + doff[i]) = nxt[0][i];      // the first sample's slab 0
        const int toff = l16 * PITCH + 4 * g4;
        float4 fa0 = *reinterpret_cast<const float4*>(T0 + toff);                              // fragments of (slab 0, group 0)
        float4 fa1 = *reinterpret_cast<const float4*>(T0 + toff + 16 * PITCH);
        auto one_db = [&](int64_t b) __attribute__((always_inline)) {
            const int64_t bn = min(b + nw, B - 1);
            if constexpr (G) {
                make_ptrs(bn);
                load_slots(min(bn + nw, B - 1));
            }
            prefetch(0, bn);                            // nxt[0] is free: this sample's slab 0 is staged
            if (lane < CS) *reinterpret_cast<v4f*>(Os + 4 * lane) = *reinterpret_cast<const v4f*>(T0 + 4 * lane);    // row 0 of slab 0
            v4f acc00 = {0.f, 0.f, 0.f, 0.f}, acc10 = acc00, acc11 = acc00;
            // the j-th MFMA of a 16-deep group (the order of k_interact_fwd_p / the single-slice form)
            auto mf = [&](int j, const float4& a0, const float4& a1) __attribute__((always_inline)) {
                const int c = j / 3, w = j % 3;
                const float x0 = c == 0 ? a0.x : c == 1 ? a0.y : c == 2 ? a0.z : a0.w;
                const float x1 = c == 0 ? a1.x : c == 1 ? a1.y : c == 2 ? a1.z : a1.w;
                if (w == 0) acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x0, acc00, 0, 0, 0);
                else if (w == 1) acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, x0, acc10, 0, 0, 0);
                else acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, x1, acc11, 0, 0, 0);
            };
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float* cur = (s & 1) ? T1 : T0;
                float* oth = (s & 1) ? T0 : T1;
                // Slab s: NG groups of 12 MFMAs.  In their shadows, ONE instruction per MFMA slot (the wave is alone on its SIMD and
                // in order: whatever is not placed between two MFMAs is paid in full): the next group's two fragment reads, then --
                // first group only -- the next slab's staging into the other slice (its last readers, slab s-1's MFMAs, are done),
                // the dense part of the output row and the register refill with the next sample's slab.
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const float4 a0 = fa0, a1 = fa1;
                    const float* np_ = (g + 1 < NG) ? cur + toff + 16 * (g + 1) : oth + toff;
#pragma unroll
                    for (int j = 0; j < 12; ++j) {
                        mf(j, a0, a1);
                        const int k = g * 12 + j;           // slot inside the slab
                        if (k >= 0 && k < NR) {
                            const int sn = s + 1 < NS ? s + 1 : 0;       // the slab being staged (the NEXT sample's slab 0 behind the last)
                            *reinterpret_cast<v4f*>(oth + doff[k]) = nxt[sn][k];
                        }
                        if (k == NR && s + 1 < NS && lane < CS) *reinterpret_cast<v4f*>(Os + (s + 1) * DS + 4 * lane) = nxt[s + 1][0];
                        if (k == NR + 1 && s + 1 < NS) prefetch(s + 1, bn);
                        // (with one group per slab the reads follow the staging; else they go first in the LAST group)
                        if (j == (NG == 1 ? NR + 2 : 0) && g == NG - 1) {
                            fa0 = *reinterpret_cast<const float4*>(np_);
                            fa1 = *reinterpret_cast<const float4*>(np_ + 16 * PITCH);
                        }
                        if (j == 0 && g + 1 < NG) {
                            fa0 = *reinterpret_cast<const float4*>(np_);
                            fa1 = *reinterpret_cast<const float4*>(np_ + 16 * PITCH);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // the output row: accumulators -> staging row -> whole 16-byte words.  (Deferred into the next sample's first slab --
            // a second staging row, the accumulators parked in 12 more registers, its LDS writes / reads / stores one per MFMA
            // slot -- it measured SLOWER: 21.1 / 24.2 us warm / cold against 20.3 / 23.5, tools/gather_warm_cold.py; removed.)
            float* out = R + b * ld_r;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i0 = 4 * g4 + r, i1 = 16 + 4 * g4 + r;
                if (i0 < F && l16 < i0 + off) Os[D + pair_base(i0, itself) + l16] = acc00[r];
                if (i1 < F) Os[D + pair_base(i1, itself) + l16] = acc10[r];
                if (i1 < F && 16 + l16 < i1 + off) Os[D + pair_base(i1, itself) + 16 + l16] = acc11[r];
            }
            const int width = D + pair_base(F, itself);
            if (lane < 4) Os[width + lane] = 0.f;
            const int W4 = (width + 3) >> 2;
            constexpr int NST = (D4 + 132 + 63) / 64;
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int e = min(64 * k + lane, W4 - 1);
                *reinterpret_cast<v4f*>(out + 4 * e) = *reinterpret_cast<const v4f*>(Os + 4 * e);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        one_db(b);
        for (b += nw; b < B; b += nw) one_db(b);
        return;
    }
    auto one = [&](int64_t b) __attribute__((always_inline)) {
        const int64_t bn = min(b + nw, B - 1);
        if constexpr (G) {
            make_ptrs(bn);                          // (its slot ids were requested a whole sample ago)
            load_slots(min(bn + nw, B - 1));
        }
        v4f acc00 = {0.f, 0.f, 0.f, 0.f}, acc10 = acc00, acc11 = acc00;
        const float* tp = Ts + l16 * PITCH + 4 * g4;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < NR; ++i) *reinterpret_cast<v4f*>(Ts + doff[i]) = nxt[s][i];
            if (lane < CS) *reinterpret_cast<v4f*>(Os + s * DS + 4 * lane) = nxt[s][0];      // row 0: the dense part of the output
            prefetch(s, bn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 2
            for (int g = 0; g < DS / 16; ++g) {
                const float4 a0 = *reinterpret_cast<const float4*>(tp + 16 * g);                  // rows 0..15
                const float4 a1 = *reinterpret_cast<const float4*>(tp + 16 * PITCH + 16 * g);     // rows 16..31
                acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, a0.x, acc00, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, a0.x, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, a1.x, acc11, 0, 0, 0);
                acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, a0.y, acc00, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, a0.y, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, a1.y, acc11, 0, 0, 0);
                acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, a0.z, acc00, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, a0.z, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, a1.z, acc11, 0, 0, 0);
                acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, a0.w, acc00, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, a0.w, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, a1.w, acc11, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float* out = R + b * ld_r;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i0 = 4 * g4 + r, i1 = 16 + 4 * g4 + r;
            if (i0 < F && l16 < i0 + off) Os[D + pair_base(i0, itself) + l16] = acc00[r];
            if (i1 < F) Os[D + pair_base(i1, itself) + l16] = acc10[r];
            if (i1 < F && 16 + l16 < i1 + off) Os[D + pair_base(i1, itself) + 16 + l16] = acc11[r];
        }
        const int width = D + pair_base(F, itself);
        if (lane < 4) Os[width + lane] = 0.f;       // the row pitch's pad columns (ld_r >= width rounded up to 4)
        const int W4 = (width + 3) >> 2;
        constexpr int NST = (D4 + 132 + 63) / 64;   // covers F = 32 with `itself`; a compile-time count
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = min(64 * k + lane, W4 - 1);
            *reinterpret_cast<v4f*>(out + 4 * e) = *reinterpret_cast<const v4f*>(Os + 4 * e);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    one(b);                 // first sample peeled (above)
    for (b += nw; b < B; b += nw) one(b);
}

// backward, whole-row staging (F <= 16 or an unaligned dfeat): dR row staged as float4 (needs ld_r % 4 == 0 and a 16-byte
// aligned dR), the accumulators stored straight from their lanes
template <int D4>
__global__ void __launch_bounds__(256) k_interact_bwd_p(const float* __restrict__ feat, const float* __restrict__ dR,
                                                        int64_t ld_r, int64_t B, int F, int itself, int x_act,
                                                        float* __restrict__ dfeat) {
    constexpr int D = 4 * D4, PITCH = D + 4;
    constexpr int GMAX = D + 528;               // dense part + up to 32*33/2 pair gradients
    constexpr int NG = (GMAX / 4 + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lk = lane >> 5;
    float* Ts = smem + wave * (32 * PITCH + GMAX);
    float* Gs = Ts + 32 * PITCH;
    for (int e = lane; e < 32 * PITCH; e += 64) Ts[e] = 0.f;
    const int FD4 = F * D4;
    const int off = itself ? 1 : 0;
    const int npairs = pair_base(F, itself);
    const int G4 = (D + npairs + 3) / 4;        // float4 words of one dR row (the pad word lies inside the pitch)
    const int64_t nw = (int64_t)gridDim.x * 4;
    int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= B) return;
    v4f nxt[D4 / 2], gn[NG];
    interact_prefetch<D4>(feat, b, FD4, lane, nxt);
#pragma unroll
    for (int i = 0; i < NG; ++i) gn[i] = reinterpret_cast<const v4f*>(dR + b * ld_r)[min(lane + 64 * i, G4 - 1)];
    auto one = [&](int64_t b) __attribute__((always_inline)) {
        interact_stage<D4>(Ts, FD4, lane, nxt);
#pragma unroll
        for (int i = 0; i < NG; ++i) *reinterpret_cast<v4f*>(Gs + 4 * min(lane + 64 * i, GMAX / 4 - 1)) = gn[i];
        const int64_t bn = min(b + nw, B - 1);
        interact_prefetch<D4>(feat, bn, FD4, lane, nxt);
#pragma unroll
        for (int i = 0; i < NG; ++i) gn[i] = reinterpret_cast<const v4f*>(dR + bn * ld_r)[min(lane + 64 * i, G4 - 1)];
        __builtin_amdgcn_sched_barrier(0);
        // A fragments of S = G + G^T: lane holds S[lr][lk + 2m], m = 0..15
        float sreg[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int j = lk + 2 * m;
            float v = 0.f;
            if (lr < F && j < F) {
                if (j < lr + off) v += Gs[D + pair_base(lr, itself) + j];
                if (lr < j + off) v += Gs[D + pair_base(j, itself) + lr];
            }
            sreg[m] = v;
        }
        float* out = dfeat + b * F * D;
        for (int n0 = 0; n0 < D; n0 += 32) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* tp = Ts + lk * PITCH + n0 + lr;
#pragma unroll
            for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sreg[m], tp[2 * m * PITCH], acc, 0, 0, 0);
            const int col = n0 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (i < F) {
                    float v = acc[r];
                    if (i == 0) {       // the dense feature: + direct path, * act'(bottom-MLP output) if asked
                        v += Gs[col];
                        const float y = Ts[col];
                        if (x_act == 1) v = y > 0.f ? v : 0.f;
                        else if (x_act == 2) v = v * ((1.0f - y) * y);
                    }
                    out[i * D + col] = v;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (; b < B; b += nw) one(b);
}

// Column-slab backward (the kernel the step runs at F > 16): dT[:, slab] = (G + G^T) T[:, slab] needs only the slab's columns of
// T, so the sample's features are staged NS slabs of D / NS columns at a time (slice 32 x (D / NS + 4) floats instead of
// 32 x (D + 4)), each slab's registers refilled with the next sample's slab as soon as they are staged; the dR row and the S
// fragments are per sample.  The 32x32 accumulator tile of a 32-column block leaves through LDS -- 16 rows at a time into the
// part of the dR staging row that is dead once the S fragments are built -- as whole float4 words of valid rows only: 4
// unconditional, fully coalesced stores per block (see the forward kernel for why).  Same products in the same order as
// k_interact_bwd_p: bit-identical.  49.1 -> 41.2 us at B = 8192, F = 27, D = 128 (two workgroups per CU; one: 45.6).
// Needs F > 16, a 16-byte aligned dfeat / dR and ld_r % 4 == 0.
// G (cdlrm_gather_interact_bwd): as in the forward kernel -- the rows are read again from the cache (they are still the
// forward's: the embedding update of this batch runs behind this kernel), `feat` is the dense feature alone.
// ONCE (cdlrm_gather_interact_bwd_sgd, round 6): a lookup whose slot no other lookup of the batch shares (flagged by the
// backward's slot sort) gets its row's SGD update HERE: the lane that holds 16 B of the row's gradient holds the same 16 B of the
// row in the slab slice (the operand it was staged as), so W[slot] = fma(-lr, 0 + g, W[slot]) -- exactly what the sorted path
// computes for a run of one: a single addend has no order -- leaves as the one store that used to carry the gradient.  The
// gradient row is NOT written (cdlrm_embbag_bwd_apply_rest never reads it): per such lookup 4D bytes written instead of 4D written
// + 4D + 4D read + 4D written, and no work for it in the sorted path.  Nobody else reads that row in this batch (that is what
// "once" means), the forward has read it, the next reader is ordered behind the embedding update as before.
template <int D4, int NS, bool G, bool ONCE = false>
__global__ void __launch_bounds__(256, (G && D4 <= 32) ? 2 : 1) k_interact_bwd_s(const float* __restrict__ feat, IaGather ga,
                                                        const float* __restrict__ dR, int64_t ld_r, int64_t B, int F,
                                                        int itself, int x_act, float* __restrict__ dfeat) {
    constexpr int D = 4 * D4, CS = D4 / NS, DS = 4 * CS, PITCH = DS + 4, NR = CS / 2;
    constexpr int GMAX = D + 528;               // dense part + up to 32*33/2 pair gradients
    constexpr int NG = (GMAX / 4 + 63) / 64;
    static_assert(DS % 32 == 0, "a slab is a whole number of 32-column MFMA blocks");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lk = lane >> 5;
    float* Ts = smem + wave * (32 * PITCH + GMAX);
    float* Gs = Ts + 32 * PITCH;
    float* Os = Gs + D;                         // [16][32] output staging: the pair part of the dR row, dead once S is built
    const int off = itself ? 1 : 0;
    const int npairs = pair_base(F, itself);
    const int G4 = (D + npairs + 3) / 4;        // float4 words of one dR row (the pad word lies inside the pitch)
    const int64_t nw = (int64_t)gridDim.x * 4;
    int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= B) return;
    int soff[NR], doff[NR];
    int srow[G ? NR : 1];
    int orow[ONCE ? NR : 1];
    const v4f* wbase[G ? NR : 1];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int e = lane + 64 * i, row = e / CS, c = e % CS;
        soff[i] = G ? c : min(row, F - 1) * D4 + c;
        doff[i] = row * PITCH + 4 * c;
        if constexpr (G) {
            const int t = max(min(row, F - 1) - 1, 0);
            srow[i] = t * (int)ga.n;
            if constexpr (ONCE) orow[i] = t * (int)ga.ld_once;
            wbase[i] = ga.weight + (ga.tab[t].row_base * D4 + c);
        }
    }
    static_assert(!ONCE || (G && CS == 8 && DS == 32), "the output store's lane -> (row, word) map is the staging map");
    v4f nxt[NS][NR], gn[NG];
    const v4f* pn[G ? NR : 1];
    int32_t sl[G ? NR : 1];
    // ONCE: the row pointers and flags of the sample being worked (pn / fn are the next sample's by then; fl: two ahead)
    const v4f* pc[ONCE ? NR : 1];
    uint8_t fl[ONCE ? NR : 1], fn[ONCE ? NR : 1], fc[ONCE ? NR : 1];
    auto load_slots = [&](int64_t bb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            sl[i] = ga.slots[srow[i] + bb];
            if constexpr (ONCE) fl[i] = ga.once[orow[i] + bb];
        }
    };
    auto make_ptrs = [&](int64_t bb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const v4f* p = wbase[i] + (int64_t)sl[i] * D4;
            if constexpr (ONCE) fn[i] = fl[i];
            if (i == 0 && lane < CS) {
                p = reinterpret_cast<const v4f*>(feat) + bb * ga.ldx4 + soff[i];
                if constexpr (ONCE) fn[i] = 0;      // row 0 is the dense feature
            }
            pn[i] = p;
        }
    };
    auto prefetch = [&](int s, int64_t bb) __attribute__((always_inline)) {
        if constexpr (G) {
#pragma unroll
            for (int i = 0; i < NR; ++i) nxt[s][i] = pn[i][s * CS];
        } else {
            const v4f* src = reinterpret_cast<const v4f*>(feat) + bb * (int64_t)F * D4 + s * CS;
#pragma unroll
            for (int i = 0; i < NR; ++i) nxt[s][i] = src[soff[i]];
        }
    };
    if constexpr (G) {
        load_slots(b);
        make_ptrs(b);
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) gn[i] = reinterpret_cast<const v4f*>(dR + b * ld_r)[min(lane + 64 * i, G4 - 1)];
#pragma unroll
    for (int s = 0; s < NS; ++s) prefetch(s, b);
    if constexpr (G) load_slots(min(b + nw, B - 1));
    auto one = [&](int64_t b) __attribute__((always_inline)) {
        const int64_t bn = min(b + nw, B - 1);
        if constexpr (G) {
            if constexpr (ONCE) {
#pragma unroll
                for (int i = 0; i < NR; ++i) { pc[i] = pn[i]; fc[i] = fn[i]; }
            }
            make_ptrs(bn);
            load_slots(min(bn + nw, B - 1));
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) *reinterpret_cast<v4f*>(Gs + 4 * min(lane + 64 * i, GMAX / 4 - 1)) = gn[i];
#pragma unroll
        for (int i = 0; i < NG; ++i) gn[i] = reinterpret_cast<const v4f*>(dR + bn * ld_r)[min(lane + 64 * i, G4 - 1)];
        __builtin_amdgcn_sched_barrier(0);
        // A fragments of S = G + G^T: lane holds S[lr][lk + 2m], m = 0..15
        float sreg[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int j = lk + 2 * m;
            float v = 0.f;
            if (lr < F && j < F) {
                if (j < lr + off) v += Gs[D + pair_base(lr, itself) + j];
                if (lr < j + off) v += Gs[D + pair_base(j, itself) + lr];
            }
            sreg[m] = v;
        }
        float* out = dfeat + b * F * D;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < NR; ++i) *reinterpret_cast<v4f*>(Ts + doff[i]) = nxt[s][i];
            prefetch(s, bn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nl = 0; nl < DS; nl += 32) {
                const int n0 = s * DS + nl;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float* tp = Ts + lk * PITCH + nl + lr;
#pragma unroll
                for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sreg[m], tp[2 * m * PITCH], acc, 0, 0, 0);
                {   // row 0 (register 0 of the lanes lk = 0): the dense feature: + direct path, * act'(bottom-MLP output)
                    float v = acc[0] + Gs[n0 + lr];
                    const float y = Ts[nl + lr];
                    if (x_act == 1) v = y > 0.f ? v : 0.f;
                    else if (x_act == 2) v = v * ((1.0f - y) * y);
                    acc[0] = lk == 0 ? v : acc[0];
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr)      // accumulator register 8h + rr: row 16h + (rr&3) + 8 (rr>>2) + 4 lk
                        Os[((rr & 3) + 8 * (rr >> 2) + 4 * lk) * 32 + lr] = acc[8 * h + rr];
                    const int nvalid = min(F - 16 * h, 16) * 8;     // float4 words of valid rows (F > 16: >= 8)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int idx = min(64 * k + lane, nvalid - 1);     // clamped: spare lanes repeat the last word
                        const int row = idx >> 3, c4 = idx & 7;
                        const v4f gv = *reinterpret_cast<const v4f*>(Os + row * 32 + 4 * c4);
                        if constexpr (ONCE) {
                            // (row, c4) of an unclamped lane is the (row, word) it staged as piece 2h + k: pc / fc are this row's
                            if (64 * k + lane < nvalid && fc[2 * h + k]) {
                                v4f w = *reinterpret_cast<const v4f*>(Ts + (16 * h + row) * PITCH + nl + 4 * c4);
                                v4f a = {0.f, 0.f, 0.f, 0.f};
                                a += gv;                    // the sorted path's sum of one addend (0 + g: -0 becomes +0 there too)
                                w.x = fmaf(-ga.lr, a.x, w.x); w.y = fmaf(-ga.lr, a.y, w.y);
                                w.z = fmaf(-ga.lr, a.z, w.z); w.w = fmaf(-ga.lr, a.w, w.w);
                                *const_cast<v4f*>(pc[2 * h + k] + s * CS) = w;
                                continue;
                            }
                        }
                        *reinterpret_cast<v4f*>(out + (16 * h + row) * D + n0 + 4 * c4) = gv;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    one(b);                 // first sample peeled: see k_interact_fwd_s
    for (b += nw; b < B; b += nw) one(b);
}

template <typename K>
static int interact_set_lds(K kernel, size_t lds, size_t* cached) {
    if (lds > *cached) {
        CDLRM_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        *cached = lds;
    }
    return 0;
}

extern "C" int cdlrm_interact_fwd(const float* feat, int64_t B, int32_t F, int32_t D, int32_t itself, float* R,
                                  int64_t ld_r, void* stream) {
    CDLRM_REQUIRE(feat && R && F >= 1 && F <= 32 && D >= 4 && D % 4 == 0 && D <= 512, "unsupported shape (F<=32, D%4==0)");
    CDLRM_REQUIRE(aligned16(feat) && ld_r >= D + (itself ? F * (F + 1) / 2 : F * (F - 1) / 2), "alignment / ld_r");
    if (B == 0) return 0;
    if (D == 32 || D == 64 || D == 128 || D == 256) {
        const int width_p = D + (itself ? F * (F + 1) / 2 : F * (F - 1) / 2);
        // whole-float4 output rows: the row pitch holds the last (padded) word
        const bool vec_out_p = ld_r % 4 == 0 && aligned16(R) && ld_r >= ((width_p + 3) & ~3);
        int64_t gp = cdiv(B, 4);
        if (vec_out_p) {
            // column-slab kernel, slabs of 32 columns, ONE workgroup per CU (each wave streams 8 samples at B = 8192).  Measured at
            // B = 8192, F = 27, D = 128 (tools/interact_ablate.hip): 22.6 us against 29.3 for the whole-row staging kernel at two
            // workgroups per CU (26.9 at one); 2 / 4 / 8 slabs at one workgroup per CU 23.3 / 22.6 / 23.3, at two 24.7 / 24.2 / 24.3,
            // at three to five 26.6-27.5 -- the finer-grained pipeline, not occupancy, is what pays
            const size_t lds_s = (size_t)4 * (32 * 36 + D + 532) * sizeof(float);
            if (gp > 256) gp = 256;
            static size_t s32 = 0, s64 = 0, s128 = 0, s256 = 0;
#define IFWD_S(D4_, A_)                                                                                               \
    do {                                                                                                              \
        int rc = interact_set_lds(k_interact_fwd_s<D4_, D4_ / 8, false>, lds_s, &A_);                                 \
        if (rc) return rc;                                                                                            \
        hipLaunchKernelGGL((k_interact_fwd_s<D4_, D4_ / 8, false>), dim3((unsigned)gp), dim3(256), lds_s,             \
                           (hipStream_t)stream, feat, IaGather{}, B, F, itself, R, ld_r);                             \
    } while (0)
            if (D == 32) IFWD_S(8, s32);
            else if (D == 64) IFWD_S(16, s64);
            else if (D == 128) IFWD_S(32, s128);
            else IFWD_S(64, s256);
#undef IFWD_S
            CDLRM_LAUNCH_CHECK();
            return 0;
        }
        // output rows that cannot leave as float4 words (pitch or alignment): whole-row staging, scalar stores
        const size_t ldsp = (size_t)4 * 32 * (D + 4) * sizeof(float);
        if (gp > 512) gp = 512;             // 2 workgroups per CU (LDS), each wave streams ~4 samples
        static size_t a32 = 0, a64 = 0, a128 = 0, a256 = 0;
#define IFWD(D4_, A_)                                                                                          \
    do {                                                                                                       \
        int rc = interact_set_lds(k_interact_fwd_p<D4_>, ldsp, &A_);                                           \
        if (rc) return rc;                                                                                     \
        hipLaunchKernelGGL((k_interact_fwd_p<D4_>), dim3((unsigned)gp), dim3(256), ldsp, (hipStream_t)stream,  \
                           feat, B, F, itself, R, ld_r);                                                       \
    } while (0)
        if (D == 32) IFWD(8, a32);
        else if (D == 64) IFWD(16, a64);
        else if (D == 128) IFWD(32, a128);
        else IFWD(64, a256);
#undef IFWD
        CDLRM_LAUNCH_CHECK();
        return 0;
    }
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
                                  int32_t itself, int32_t x_act, float* dfeat, void* stream) {
    CdlrmStopScope stop_scope;          // (first: every exit below flushes an attached completion event)
    CDLRM_REQUIRE(feat && dR && dfeat && F >= 1 && F <= 32 && D >= 4 && D % 4 == 0 && D <= 512, "unsupported shape");
    CDLRM_REQUIRE(aligned16(feat), "alignment");
    if (B == 0) return 0;
    const int npairs = itself ? F * (F + 1) / 2 : F * (F - 1) / 2;
    const bool vec_in = ld_r % 4 == 0 && aligned16(dR) && ld_r >= ((D + npairs + 3) & ~3);
    if ((D == 32 || D == 64 || D == 128 || D == 256) && vec_in && F > 16 && aligned16(dfeat)) {
        // column-slab kernel, slabs of 32 columns, two workgroups per CU.  Measured at B = 8192, F = 27, D = 128
        // (tools/interact_ablate.hip): 41.2 us against 49.1 for the whole-row staging kernel; 4 slabs on 256 / 512 / 768 workgroups
        // 45.6 / 41.2 / 44.2, 2 slabs 46.7 / 51.8 / 54.2
        const size_t lds_s = (size_t)4 * (32 * 36 + D + 528) * sizeof(float);
        int64_t gp = cdiv(B, 4);
        // (D = 256: 336 registers per lane, one wave per SIMD -- one workgroup per CU is all that fits: 96.7 us against 102.2)
        const int64_t gmax = D == 256 ? 256 : 512;
        if (gp > gmax) gp = gmax;
        static size_t s32 = 0, s64 = 0, s128 = 0, s256 = 0;
#define IBWD_S(D4_, A_)                                                                                               \
    do {                                                                                                              \
        int rc = interact_set_lds(k_interact_bwd_s<D4_, D4_ / 8, false>, lds_s, &A_);                                 \
        if (rc) return rc;                                                                                            \
        CDLRM_LAUNCH_EV((k_interact_bwd_s<D4_, D4_ / 8, false>), dim3((unsigned)gp), dim3(256), lds_s,                \
                        (hipStream_t)stream, feat, IaGather{}, dR, ld_r, B, F, itself, x_act, dfeat);                 \
    } while (0)
        if (D == 32) IBWD_S(8, s32);
        else if (D == 64) IBWD_S(16, s64);
        else if (D == 128) IBWD_S(32, s128);
        else IBWD_S(64, s256);
#undef IBWD_S
        CDLRM_LAUNCH_CHECK();
        return 0;
    }
    if ((D == 32 || D == 64 || D == 128) && vec_in) {
        // F <= 16 (or an unaligned dfeat): whole-row staging, the accumulators stored straight from their lanes
        const size_t ldsp = (size_t)4 * (32 * (D + 4) + D + 528) * sizeof(float);
        int64_t gp = cdiv(B, 4);
        if (gp > 512) gp = 512;
        static size_t b32 = 0, b64 = 0, b128 = 0;
#define IBWD(D4_, A_)                                                                                          \
    do {                                                                                                       \
        int rc = interact_set_lds(k_interact_bwd_p<D4_>, ldsp, &A_);                                           \
        if (rc) return rc;                                                                                     \
        CDLRM_LAUNCH_EV((k_interact_bwd_p<D4_>), dim3((unsigned)gp), dim3(256), ldsp, (hipStream_t)stream,     \
                        feat, dR, ld_r, B, F, itself, x_act, dfeat);                                           \
    } while (0)
        if (D == 32) IBWD(8, b32);
        else if (D == 64) IBWD(16, b64);
        else IBWD(32, b128);
#undef IBWD
        CDLRM_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = (size_t)4 * (32 * (D + 1) + 32 + ((npairs + 3) & ~3)) * sizeof(float);
    CDLRM_REQUIRE(lds <= 160 * 1024, "LDS budget");
    static size_t attr = 0;
    if (lds > attr) {
        CDLRM_HIP_CHECK(hipFuncSetAttribute((const void*)k_interact_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    int64_t gx = cdiv(B, 4);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(k_interact_bwd, dim3((unsigned)gx), dim3(256), lds, (hipStream_t)stream, feat, dR, ld_r, B, F, D,
                       itself, x_act, dfeat);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// ---- fused gather + interaction (Criteo layout: one index per bag, so a bag's sum-pool is its cache row) ------------------
// cached EmbeddingBag forward (model_no_ddp.py:200-203) + interact_features "dot" (:272-293) in one launch, and the
// interaction backward reading the same rows again: the [B, T, D] block between the two operators is never written or read.
// Shapes the column-slab kernels take: D in {32, 64, 128, 256}, 16 < F = T + 1 <= 32, float4 output / gradient rows.
static bool gather_interact_shape_ok(const cdlrm_ctx* ctx) {
    const int D = ctx->D, F = ctx->T + 1;
    return (D == 32 || D == 64 || D == 128 || D == 256) && F > 16 && F <= 32;
}

extern "C" int cdlrm_gather_interact_supported(cdlrm_ctx* ctx) {
    return ctx && gather_interact_shape_ok(ctx) ? 1 : 0;
}

extern "C" int cdlrm_gather_interact_fwd(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, const float* x, int64_t ld_x,
                                         int64_t B, int32_t itself, float* R, int64_t ld_r, void* stream) {
    CdlrmStopScope stop_scope;          // (first: every exit below flushes an attached completion event)
    CDLRM_CLEAR_STALE();
    CDLRM_REQUIRE(ctx && slots && x && R, "null argument");
    CDLRM_REQUIRE(ctx->weight, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(gather_interact_shape_ok(ctx), "unsupported shape (D in 32/64/128/256, 16 < T + 1 <= 32): gather + interaction as two calls");
    const int D = ctx->D, F = ctx->T + 1;
    const int width = D + (itself ? F * (F + 1) / 2 : F * (F - 1) / 2);
    CDLRM_REQUIRE(n >= B && (int64_t)ctx->T * n < INT32_MAX && aligned16(x) && ld_x % 4 == 0 && ld_x >= D,
                  "slot pitch / dense-feature rows");
    CDLRM_REQUIRE(ld_r % 4 == 0 && aligned16(R) && ld_r >= ((width + 3) & ~3), "whole-float4 output rows");
    if (B == 0) return 0;       // (an armed event pair stays armed for the next launch that can carry it)
    hipEvent_t ev0 = (hipEvent_t)ctx->ev_start, ev1 = (hipEvent_t)ctx->ev_stop;     // cdlrm_ctx_time_next_gather
    ctx->ev_start = ctx->ev_stop = nullptr;
    // a completion event attached to this launch (cdlrm_event_attach_next: the trainer's side queue starts the batch's slot sort
    // and the look-ahead resolve behind the interaction forward): it IS the launch's stop event -- unless the launch is being
    // timed (the roofline samples of bench.py carry their own pair), then it is recorded behind the launch
    hipEvent_t se = cdlrm_take_stop_event((hipStream_t)stream);
    if (!ev1 && se) { ev1 = se; se = nullptr; }
    IaGather ga{ctx->d_tab, reinterpret_cast<const v4f*>(ctx->weight), slots, n, ld_x / 4};
    const size_t lds_s = (size_t)4 * (32 * 36 + D + 532) * sizeof(float);
    int64_t gp = cdiv(B, 4);
    // ONE workgroup per CU, a wave walks 8 samples of a c3 batch: in the step 23.6 us against 25.2 on two and 26.5 on three
    // (bench.py --debug 4=<n>; stand-alone 27.4 / 27.2 / 27.2).  Two samples' rows in flight per wave (a second register set,
    // the loop unrolled by two) measured SLOWER in the step, twice: 25.1 us as hipcc compiled it (its wait counts at the joins
    // of the unrolled loop wait for part of the younger sample's rows at every slab), and 25.7 against 24.0 us on one box
    // with every vector-memory instruction issued from inline asm and ONE counted wait per sample (exact: vmcnt(19) in front
    // of the address arithmetic, nothing else; bit-identical once the 16-byte stores had wait states behind them -- the
    // compiler reuses a store's data registers at once when it does not know the statement is a store).  More rows in flight
    // do not help: a c3 launch is 1024 waves x 8 samples, 2.9 us per sample against 2.3 us at c5 (64 samples per wave, 0.89
    // of 8 TB/s) -- the difference is ramp, and neither variant shortens it.  Both removed
    const int per_cu = g_cdlrm_debug[4] > 0 ? g_cdlrm_debug[4] : 1;
    if (gp > 256 * per_cu) gp = 256 * per_cu;
    static size_t s32 = 0, s64 = 0, s128 = 0, s256 = 0;
#define GIFWD(D4_, A_)                                                                                                \
    do {                                                                                                              \
        int rc = interact_set_lds(k_interact_fwd_s<D4_, D4_ / 8, true>, lds_s, &A_);                                  \
        if (rc) return rc;                                                                                            \
        hipExtLaunchKernelGGL((k_interact_fwd_s<D4_, D4_ / 8, true>), dim3((unsigned)gp), dim3(256), lds_s,           \
                              (hipStream_t)stream, ev0, ev1, 0, x, ga, B, F, itself, R, ld_r);                        \
    } while (0)
    // (slabs of 256 or 512 B instead of 128 -- NS = 2, 1 at D = 128 -- measured the same in the step and 24.6-25.8 / 26.6-28.3 against
    //  27.2 us stand-alone; the variants were removed)
    // the double-buffered form (round 6; cdlrm_debug_set(7, 2): the single-slice form, for A/Bs -- bit-identical)
    static size_t d64 = 0, d128 = 0, d256 = 0;
    const size_t lds_d = (size_t)4 * (2 * 32 * 36 + D + 532) * sizeof(float);
#define GIFWD_DB(D4_, A_)                                                                                             \
    do {                                                                                                              \
        int rc = interact_set_lds(k_interact_fwd_s<D4_, D4_ / 8, true, true>, lds_d, &A_);                            \
        if (rc) return rc;                                                                                            \
        hipExtLaunchKernelGGL((k_interact_fwd_s<D4_, D4_ / 8, true, true>), dim3((unsigned)gp), dim3(256), lds_d,     \
                              (hipStream_t)stream, ev0, ev1, 0, x, ga, B, F, itself, R, ld_r);                        \
    } while (0)
    const bool db = !(g_cdlrm_debug[7] & 2);
    if (D == 32) GIFWD(8, s32);                    // (one slab per sample: nothing to alternate)
    else if (D == 64) { if (db) GIFWD_DB(16, d64); else GIFWD(16, s64); }
    else if (D == 128) { if (db) GIFWD_DB(32, d128); else GIFWD(32, s128); }
    else { if (db) GIFWD_DB(64, d256); else GIFWD(64, s256); }
#undef GIFWD_DB
#undef GIFWD
    if (se) CDLRM_HIP_CHECK(hipEventRecord(se, (hipStream_t)stream));
    CDLRM_LAUNCH_CHECK();
    return 0;
}

static int cdlrm_gather_interact_bwd_core(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, const float* x, int64_t ld_x,
                               const float* dR, int64_t ld_r, int64_t B, int32_t itself, int32_t x_act,
                               float* dfeat, const uint8_t* once, int64_t ld_once, float lr, void* stream) {
    CdlrmStopScope stop_scope;          // (first: every exit below flushes an attached completion event)
    CDLRM_REQUIRE(ctx && slots && x && dR && dfeat, "null argument");
    CDLRM_REQUIRE(ctx->weight, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(gather_interact_shape_ok(ctx), "unsupported shape (D in 32/64/128/256, 16 < T + 1 <= 32): cdlrm_interact_bwd on a gathered block");
    const int D = ctx->D, F = ctx->T + 1;
    const int npairs = itself ? F * (F + 1) / 2 : F * (F - 1) / 2;
    CDLRM_REQUIRE(n >= B && (int64_t)ctx->T * n < INT32_MAX && aligned16(x) && ld_x % 4 == 0 && ld_x >= D,
                  "slot pitch / dense-feature rows");
    CDLRM_REQUIRE(ld_r % 4 == 0 && aligned16(dR) && ld_r >= ((D + npairs + 3) & ~3) && aligned16(dfeat), "whole-float4 gradient rows");
    if (B == 0) return 0;
    IaGather ga{ctx->d_tab, reinterpret_cast<const v4f*>(ctx->weight), slots, n, ld_x / 4, once, ld_once, lr};
    const size_t lds_s = (size_t)4 * (32 * 36 + D + 528) * sizeof(float);
    int64_t gp = cdiv(B, 4);
    const int per_cu = g_cdlrm_debug[5] > 0 ? g_cdlrm_debug[5] : (D == 256 ? 1 : 2);
    if (gp > 256 * per_cu) gp = 256 * per_cu;
    static size_t s32 = 0, s64 = 0, s128 = 0, s256 = 0, o32 = 0, o64 = 0, o128 = 0, o256 = 0;
#define GIBWD(D4_, A_, ONCE_)                                                                                         \
    do {                                                                                                              \
        int rc = interact_set_lds(k_interact_bwd_s<D4_, D4_ / 8, true, ONCE_>, lds_s, &A_);                           \
        if (rc) return rc;                                                                                            \
        CDLRM_LAUNCH_EV((k_interact_bwd_s<D4_, D4_ / 8, true, ONCE_>), dim3((unsigned)gp), dim3(256), lds_s,          \
                        (hipStream_t)stream, x, ga, dR, ld_r, B, F, itself, x_act, dfeat);                            \
    } while (0)
    if (once) {
        if (D == 32) GIBWD(8, o32, true);
        else if (D == 64) GIBWD(16, o64, true);
        else if (D == 128) GIBWD(32, o128, true);
        else GIBWD(64, o256, true);
    } else {
        if (D == 32) GIBWD(8, s32, false);
        else if (D == 64) GIBWD(16, s64, false);
        else if (D == 128) GIBWD(32, s128, false);
        else GIBWD(64, s256, false);
    }
#undef GIBWD
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_gather_interact_bwd(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, const float* x, int64_t ld_x,
                                         const float* dR, int64_t ld_r, int64_t B, int32_t itself, int32_t x_act,
                                         float* dfeat, void* stream) {
    return cdlrm_gather_interact_bwd_core(ctx, slots, n, x, ld_x, dR, ld_r, B, itself, x_act, dfeat, nullptr, 0, 0.f, stream);
}

extern "C" int cdlrm_gather_interact_bwd_sgd(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, const float* x, int64_t ld_x,
                                             const float* dR, int64_t ld_r, int64_t B, int32_t itself, int32_t x_act,
                                             float* dfeat, const uint8_t* once, int64_t ld_once, float lr, void* stream) {
    if (!(ctx && once && ld_once >= B && (int64_t)ctx->T * ld_once < INT32_MAX)) {
        CdlrmStopScope stop_scope;      // (an attached completion event must not outlive a refused call)
        CDLRM_REQUIRE(ctx && once, "null argument");
        CDLRM_REQUIRE(ld_once >= B && (int64_t)ctx->T * ld_once < INT32_MAX, "flag pitch");
    }
    return cdlrm_gather_interact_bwd_core(ctx, slots, n, x, ld_x, dR, ld_r, B, itself, x_act, dfeat, once, ld_once, lr, stream);
}

// =================================================================================================
// K11 BCELoss(mean) on the sigmoid output + its gradient; K12 dense SGD
// =================================================================================================
__global__ void __launch_bounds__(256) k_bce_partial(const float* __restrict__ Z, const float* __restrict__ T, int64_t n,
                                                     float* __restrict__ part, float* __restrict__ dZ,
                                                     int sigmoid_bwd) {
    __shared__ float red[4];
    float s = 0.f;
    const float inv_n = 1.0f / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float z = Z[i], t = T[i];
        // torch binary_cross_entropy: log terms clamped at -100
        const float l1 = fmaxf(logf(z), -100.f), l0 = fmaxf(log1pf(-z), -100.f);
        s += (t - 1.0f) * l0 - t * l1;
        // binary_cross_entropy_backward: (z - t) / max((1 - z) z, 1e-12) * grad, grad = 1/n
        if (dZ) {
            float d = (z - t) / fmaxf((1.0f - z) * z, 1e-12f) * inv_n;
            if (sigmoid_bwd) d = d * ((1.0f - z) * z);      // sigmoid_backward of the layer that produced z
            dZ[i] = d;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// the whole loss in ONE workgroup (n up to a few 10k: a training batch) -- one launch instead of two
__global__ void __launch_bounds__(1024) k_bce_one(const float* __restrict__ Z, const float* __restrict__ T, int64_t n,
                                                  float* __restrict__ loss, float* __restrict__ dZ, int sigmoid_bwd) {
    __shared__ float red[16];
    float s = 0.f;
    const float inv_n = 1.0f / (float)n;
    // 8 elements per thread per round, all 16 loads of a round issued before any is used (a dependent load per
    // element made one workgroup over a batch of 8192 take 14 us; clamped indices keep the loads branch-free)
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += 8 * 1024) {
        float z[8], t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = min(i0 + u * 1024, n - 1);
            z[u] = Z[i];
            t[u] = T[i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = i0 + u * 1024;
            if (i < n) {
                const float l1 = fmaxf(logf(z[u]), -100.f), l0 = fmaxf(log1pf(-z[u]), -100.f);
                s += (t[u] - 1.0f) * l0 - t[u] * l1;
                if (dZ) {
                    float d = (z[u] - t[u]) / fmaxf((1.0f - z[u]) * z[u], 1e-12f) * inv_n;
                    if (sigmoid_bwd) d = d * ((1.0f - z[u]) * z[u]);
                    dZ[i] = d;
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += red[w];
        loss[0] = tot / (float)n;
    }
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
extern "C" int cdlrm_bce_fwd_bwd(const float* Z, const float* target, int64_t n, float* loss_out, float* dZ,
                                 int32_t sigmoid_bwd, void* stream) {
    CDLRM_REQUIRE(Z && target && loss_out && n >= 1, "bad argument");
    // partial sums live behind the loss word: loss_out must have room for 1 + 64 floats
    hipStream_t s = (hipStream_t)stream;
    if (n <= 32768) {
        hipLaunchKernelGGL(k_bce_one, dim3(1), dim3(1024), 0, s, Z, target, n, loss_out, dZ, (int)sigmoid_bwd);
        CDLRM_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_bce_partial, dim3(BCE_PARTS), dim3(256), 0, s, Z, target, n, loss_out + 1, dZ, (int)sigmoid_bwd);
    hipLaunchKernelGGL(k_bce_final, dim3(1), dim3(64), 0, s, loss_out + 1, BCE_PARTS, n, loss_out);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// -------------------------------------------------------------------------------------------------
// The other arms of the loss (main_no_ddp.py:212-221, 364-372) and the --loss-threshold clamp of the prediction
// (model_no_ddp.py:311-314), element-wise: kind 0 BCELoss, 1 MSELoss, 2 weighted BCE (loss_ws[T.long()] * BCE(none),
// mean).  Returns the element's loss term (before the division by n); *d = dL/d(pre-clamp prediction).
// -------------------------------------------------------------------------------------------------
struct LossCfg {
    int kind;
    float w0, w1;        // wbce: weight of target class 0 / 1
    float thr;           // 0 < thr < 1: z = clamp(p, thr, 1 - thr); gradient only where thr <= p <= 1 - thr
};

__device__ __forceinline__ float loss_elem(float p, float t, const LossCfg& c, int64_t n, float* zc, float* d) {
    float z = p;
    bool pass = true;
    if (c.thr > 0.f && c.thr < 1.f) {
        const float hi = 1.0f - c.thr;
        z = fminf(fmaxf(p, c.thr), hi);
        pass = (p >= c.thr) && (p <= hi);
    }
    *zc = z;
    float l, g;
    if (c.kind == 1) {                      // mse_loss: (z - t)^2, backward 2 (z - t) / n
        const float df = z - t;
        l = df * df;
        g = 2.0f * df * (1.0f / (float)n);
    } else {
        // torch binary_cross_entropy: log terms clamped at -100; backward (z - t) / max((1 - z) z, 1e-12) * grad
        const float l1 = fmaxf(logf(z), -100.f), l0 = fmaxf(log1pf(-z), -100.f);
        l = (t - 1.0f) * l0 - t * l1;
        float go = 1.0f / (float)n;
        if (c.kind == 2) {                  // the weights are float64 in the reference: w / n is rounded once
            const float w = ((long)t == 0) ? c.w0 : c.w1;
            l *= w;
            go = (float)((double)w / (double)n);
        }
        g = (z - t) / fmaxf((1.0f - z) * z, 1e-12f) * go;
    }
    *d = pass ? g : 0.f;
    return l;
}

__global__ void __launch_bounds__(1024) k_loss_one(const float* __restrict__ Z, const float* __restrict__ T, int64_t n,
                                                   LossCfg c, float* __restrict__ loss, float* __restrict__ dZ,
                                                   float* __restrict__ Zc, int sigmoid_bwd) {
    __shared__ float red[16];
    __shared__ int redc[16];
    float s = 0.f;
    int ok = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float p = Z[i];
        const float t = T[i];
        float zc, d;
        s += loss_elem(p, t, c, n, &zc, &d);
        ok += rintf(zc) == t;                        // main_no_ddp.py:431: (np.round(S, 0) == T), half to even
        if (sigmoid_bwd) d = d * ((1.0f - p) * p);
        if (dZ) dZ[i] = d;
        if (Zc) Zc[i] = zc;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s += __shfl_down(s, d, 64); ok += __shfl_down(ok, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; redc[threadIdx.x >> 6] = ok; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        int cnt = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { tot += red[w]; cnt += redc[w]; }
        loss[0] = tot / (float)n;
        loss[1] = (float)cnt;                        // correct predictions of this batch
        loss[2] = (tot / (float)n) * (float)n;       // L * mbs as the reference accumulates it (:433), in fp32
    }
}

extern "C" int cdlrm_loss_fwd_bwd(const float* Z, const float* target, int64_t n, int32_t kind, float w0, float w1,
                                  float threshold, float* loss_out, float* dZ, float* Zc, int32_t sigmoid_bwd,
                                  void* stream) {
    CDLRM_REQUIRE(Z && target && loss_out && n >= 1 && kind >= 0 && kind <= 2, "bad argument");
    LossCfg c;
    c.kind = kind; c.w0 = w0; c.w1 = w1; c.thr = threshold;
    hipLaunchKernelGGL(k_loss_one, dim3(1), dim3(1024), 0, (hipStream_t)stream, Z, target, n, c, loss_out, dZ, Zc,
                       (int)sigmoid_bwd);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// -------------------------------------------------------------------------------------------------
// Output head in ONE launch: the last top-MLP layer (out_features = 1, sigmoid; main_no_ddp.py:358), the loss and
// that layer's input gradient.  At a local batch of 1024 the three stand-alone kernels (256 -> 1 GEMM, loss, 1 -> 256
// outer product) cost 8.7 + 7.4 + 7.5 us of fixed latency for 2 MB of traffic.  One wave per row: dot over K by
// lanes, sigmoid, loss term, dz (pre-activation gradient of the last layer), then dY[row, :] = dz * w, times the
// derivative of the activation that produced Y.  Loss: per-workgroup partials in `scratch`, summed in index order
// (reproducible) by cdlrm_head_finish -- on the same stream, or (finish = 0) wherever the caller launches it behind
// this kernel: nothing on the training step's critical path reads the loss value.
// -------------------------------------------------------------------------------------------------

// The loss: per-workgroup partial sums, added up in index order by k_head_finish, a second one-workgroup launch.  (One
// launch with a "last workgroup to arrive sums" tail needs agent-scope release/acquire fences, which on the 8-XCD part
// write back / invalidate a whole L2 -- freshly filled with this kernel's 8 MB of dY: the tail cost 16 of the kernel's
// 24 us at B = 8192, whatever the arrangement of the arrival counters.)  A wave takes 4 rows per pass (independent load
// chains).
#ifndef HEAD_ABL
#define HEAD_ABL 0        // tools/head_ablate.hip: 1 no partial sums, 2 no dY stores
#endif
#define HEAD_MAX_BLOCKS 1024
#define HEAD_TAIL_BLOCKS 128

template <bool VEC, bool TAIL>
__global__ void __launch_bounds__(256) k_head(const float* __restrict__ Y, int64_t ldy, const float* __restrict__ w,
                                              const float* __restrict__ bias, const float* __restrict__ T, int64_t B,
                                              int K, LossCfg c, int x_act, float* __restrict__ Zout,
                                              float* __restrict__ Zc, float* __restrict__ dZ, float* __restrict__ dY,
                                              int64_t lddy, float* __restrict__ loss, float* __restrict__ scratch) {
    __shared__ float red[4];
    __shared__ int redc[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float b0 = bias ? bias[0] : 0.f;
    float lsum = 0.f;
    int nok = 0;
    constexpr int R = 4;                            // rows a wave works on at once: R independent load chains in flight
    const bool one = VEC && K <= 256;               // a row is one float4 per lane: kept in registers for the dY pass
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
    float4 w1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (one && lane * 4 < K) w1 = *reinterpret_cast<const float4*>(w + lane * 4);
    for (int64_t r0 = wid * R; r0 < B; r0 += nw * R) {
        float4 y1[R];
        float acc[R], tt[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {               // all loads of the R rows first (clamped rows: results discarded)
            const int64_t row = min(r0 + i, B - 1);
            const float* y = Y + row * ldy;
            tt[i] = T[row];
            acc[i] = 0.f;
            y1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (one) {
                if (lane * 4 < K) y1[i] = *reinterpret_cast<const float4*>(y + lane * 4);
            } else if (VEC) {
                for (int k = lane * 4; k < K; k += 256) {
                    const float4 a = *reinterpret_cast<const float4*>(y + k);
                    const float4 ww = *reinterpret_cast<const float4*>(w + k);
                    acc[i] = fmaf(a.x, ww.x, acc[i]); acc[i] = fmaf(a.y, ww.y, acc[i]);
                    acc[i] = fmaf(a.z, ww.z, acc[i]); acc[i] = fmaf(a.w, ww.w, acc[i]);
                }
            } else {
                for (int k = lane; k < K; k += 64) acc[i] = fmaf(y[k], w[k], acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int64_t row = r0 + i;
            if (one) {
                acc[i] = fmaf(y1[i].x, w1.x, acc[i]); acc[i] = fmaf(y1[i].y, w1.y, acc[i]);
                acc[i] = fmaf(y1[i].z, w1.z, acc[i]); acc[i] = fmaf(y1[i].w, w1.w, acc[i]);
            }
            float a = acc[i];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
            if (row >= B) continue;                 // wave-uniform
            const float p = 1.0f / (1.0f + expf(-(a + b0)));
            float zc, d;
            const float l = loss_elem(p, tt[i], c, B, &zc, &d);
            d = d * ((1.0f - p) * p);                   // sigmoid backward: gradient w.r.t. the pre-activation
            if (lane == 0) {
                lsum += l;
                nok += rintf(zc) == tt[i];              // main_no_ddp.py:431: (np.round(S, 0) == T)
                Zout[row] = p;
                if (Zc) Zc[row] = zc;
                dZ[row] = d;
            }
            if (!dY || HEAD_ABL == 2) continue;
            const float* y = Y + row * ldy;
            float* dy = dY + row * lddy;
            if (one) {
                if (lane * 4 < K) {
                    float4 o = make_float4(d * w1.x, d * w1.y, d * w1.z, d * w1.w);
                    if (x_act == 1) {
                        o.x = y1[i].x > 0.f ? o.x : 0.f; o.y = y1[i].y > 0.f ? o.y : 0.f;
                        o.z = y1[i].z > 0.f ? o.z : 0.f; o.w = y1[i].w > 0.f ? o.w : 0.f;
                    } else if (x_act == 2) {
                        o.x *= (1.0f - y1[i].x) * y1[i].x; o.y *= (1.0f - y1[i].y) * y1[i].y;
                        o.z *= (1.0f - y1[i].z) * y1[i].z; o.w *= (1.0f - y1[i].w) * y1[i].w;
                    }
                    *reinterpret_cast<float4*>(dy + lane * 4) = o;
                }
            } else if (VEC) {
                for (int k = lane * 4; k < K; k += 256) {
                    const float4 av = *reinterpret_cast<const float4*>(y + k);
                    const float4 ww = *reinterpret_cast<const float4*>(w + k);
                    float4 o = make_float4(d * ww.x, d * ww.y, d * ww.z, d * ww.w);
                    if (x_act == 1) {
                        o.x = av.x > 0.f ? o.x : 0.f; o.y = av.y > 0.f ? o.y : 0.f;
                        o.z = av.z > 0.f ? o.z : 0.f; o.w = av.w > 0.f ? o.w : 0.f;
                    } else if (x_act == 2) {
                        o.x *= (1.0f - av.x) * av.x; o.y *= (1.0f - av.y) * av.y;
                        o.z *= (1.0f - av.z) * av.z; o.w *= (1.0f - av.w) * av.w;
                    }
                    *reinterpret_cast<float4*>(dy + k) = o;
                }
            } else {
                for (int k = lane; k < K; k += 64) {
                    const float av = y[k];
                    float o = d * w[k];
                    if (x_act == 1) o = av > 0.f ? o : 0.f;
                    else if (x_act == 2) o *= (1.0f - av) * av;
                    dy[k] = o;
                }
            }
        }
    }
#if HEAD_ABL == 1
    if (lsum == 1.2345f) scratch[0] = lsum;
    return;
#endif
    if (lane == 0) { red[wave] = lsum; redc[wave] = nok; }
    __syncthreads();
    if (threadIdx.x == 0) {
        scratch[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        scratch[HEAD_MAX_BLOCKS + blockIdx.x] = (float)(redc[0] + redc[1] + redc[2] + redc[3]);
    }
    if (!TAIL) return;
    // Short batches (<= HEAD_TAIL_BLOCKS workgroups, little dirty data in the L2s): the workgroup that arrives last sums the
    // partials itself -- a second launch costs its 5 us floor, more than the fences do here.  Same summation order as
    // k_head_finish.  scratch[2 * HEAD_MAX_BLOCKS] is the arrival counter: zero before the first call, left zero.
    __shared__ unsigned last;
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(reinterpret_cast<unsigned*>(scratch + 2 * HEAD_MAX_BLOCKS), 1u);
    }
    __syncthreads();
    if (last == gridDim.x - 1 && wave == 0) {
        __threadfence();
        float s = 0.f, cnt = 0.f;
        for (unsigned i = lane; i < gridDim.x; i += 64) {
            s += __builtin_nontemporal_load(scratch + i);
            cnt += __builtin_nontemporal_load(scratch + HEAD_MAX_BLOCKS + i);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { s += __shfl_xor(s, d, 64); cnt += __shfl_xor(cnt, d, 64); }
        if (lane == 0) {
            loss[0] = s / (float)B;
            loss[1] = cnt;
            loss[2] = (s / (float)B) * (float)B;
            *reinterpret_cast<unsigned*>(scratch + 2 * HEAD_MAX_BLOCKS) = 0u;
        }
    }
}

// fixed order for a given number of partials: lane i sums partials i, i + 64, ..., then a butterfly over the lanes
__global__ void __launch_bounds__(64) k_head_finish(const float* __restrict__ scratch, int nparts, int64_t B,
                                                    float* __restrict__ loss, double* __restrict__ acc) {
    const int lane = threadIdx.x;
    float s = 0.f, cnt = 0.f;
    for (int i = lane; i < nparts; i += 64) {
        s += scratch[i];
        cnt += scratch[HEAD_MAX_BLOCKS + i];         // integers < 2^24: exact
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s += __shfl_xor(s, d, 64); cnt += __shfl_xor(cnt, d, 64); }
    if (lane == 0) {
        loss[0] = s / (float)B;
        loss[1] = cnt;                              // correct predictions of this batch
        const float lm = (s / (float)B) * (float)B; // L * mbs as the reference accumulates it (:433), in fp32
        loss[2] = lm;
        if (acc) {                                  // running print statistics (main_no_ddp.py:427-433): float64 sums over steps
            acc[0] += (double)cnt;
            acc[1] += (double)lm;
        }
    }
}

extern "C" int64_t cdlrm_head_scratch_floats(void) { return 2 * HEAD_MAX_BLOCKS + 1; }

static int64_t head_grid(int64_t B) {
    int64_t gx = cdiv(B, 16);                       // 4 waves x 4 rows per pass
    return gx > HEAD_MAX_BLOCKS ? HEAD_MAX_BLOCKS : gx;     // B > 16384: more passes
}

extern "C" int cdlrm_head_fwd_bwd(const float* Y, int64_t ldy, const float* w, const float* bias, const float* target,
                                  int64_t B, int32_t K, int32_t kind, float w0, float w1, float threshold,
                                  int32_t x_act, float* Z, float* Zc, float* dZ, float* dY, int64_t lddy,
                                  float* loss_out, float* scratch, int32_t finish, void* stream) {
    CDLRM_REQUIRE(Y && w && target && Z && dZ && loss_out && scratch && B >= 1 && K >= 1, "bad argument");
    CDLRM_REQUIRE(kind >= 0 && kind <= 2 && x_act >= 0 && x_act <= 2, "bad loss kind / activation");
    LossCfg c;
    c.kind = kind; c.w0 = w0; c.w1 = w1; c.thr = threshold;
    const int64_t gx = head_grid(B);
    const bool vec = K % 4 == 0 && ldy % 4 == 0 && (!dY || lddy % 4 == 0) && (((uintptr_t)Y | (uintptr_t)w | (uintptr_t)dY) & 15) == 0;
    const bool tail = finish && gx <= HEAD_TAIL_BLOCKS;
#define HEAD_CALL(V_, T_)                                                                                      \
    hipLaunchKernelGGL((k_head<V_, T_>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, Y, ldy, w, bias, target, B, \
                       (int)K, c, (int)x_act, Z, Zc, dZ, dY, lddy, loss_out, scratch)
    if (vec && tail) HEAD_CALL(true, true);
    else if (vec) HEAD_CALL(true, false);
    else if (tail) HEAD_CALL(false, true);
    else HEAD_CALL(false, false);
#undef HEAD_CALL
    CDLRM_LAUNCH_CHECK();
    if (tail) return 0;
    if (finish) return cdlrm_head_finish(scratch, B, loss_out, nullptr, stream);
    return 0;
}

extern "C" int cdlrm_head_finish(const float* scratch, int64_t B, float* loss_out, double* acc, void* stream) {
    CDLRM_REQUIRE(scratch && loss_out && B >= 1, "bad argument");
    hipLaunchKernelGGL(k_head_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, (int)head_grid(B), B, loss_out, acc);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// dX *= act'(X) over an [M, N] block with row pitches (the "cat" interaction has no interaction backward whose
// epilogue could apply the bottom MLP's last activation: model_no_ddp.py:297-299)
__global__ void __launch_bounds__(256) k_act_bwd(float* __restrict__ dX, int64_t lddx, const float* __restrict__ X,
                                                 int64_t ldx, int64_t M, int N, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * N; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / N, c = i % N;
        const float x = X[r * ldx + c];
        float d = dX[r * lddx + c];
        d = act == 1 ? (x > 0.f ? d : 0.f) : act == 2 ? d * ((1.0f - x) * x) : d;
        dX[r * lddx + c] = d;
    }
}

extern "C" int cdlrm_act_bwd(float* dX, int64_t lddx, const float* X, int64_t ldx, int64_t M, int32_t N, int32_t act,
                             void* stream) {
    CDLRM_REQUIRE(dX && X && M >= 0 && N >= 1 && act >= 0 && act <= 2, "bad argument");
    if (M == 0 || act == 0) return 0;
    int64_t gx = cdiv(M * N, 256);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, dX, lddx, X, ldx, M, (int)N, (int)act);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// two ranges of one flat buffer in one launch (a sub-network's weights and, behind all weights, its biases)
__global__ void __launch_bounds__(256) k_sgd2(float* __restrict__ p, const float* __restrict__ g, int64_t off0, int64_t n0,
                                              int64_t off1, int64_t n1, float lr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = i < n0 ? off0 + i : off1 + (i - n0);
        p[j] = fmaf(-lr, g[j], p[j]);
    }
}

extern "C" int cdlrm_sgd_step2(float* param, const float* grad, int64_t off0, int64_t n0, int64_t off1, int64_t n1,
                               float lr, void* stream) {
    CDLRM_REQUIRE(param && grad && off0 >= 0 && n0 >= 0 && off1 >= 0 && n1 >= 0, "bad argument");
    if (n0 + n1 == 0) return 0;
    int64_t gx = cdiv(n0 + n1, 256);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(k_sgd2, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, param, grad, off0, n0, off1, n1, lr);
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
