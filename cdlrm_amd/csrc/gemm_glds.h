// FP32-MFMA tiled GEMM, second generation: operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR
// round trip, no ds_write pass), two LDS stages, ONE barrier per 32-deep K tile.
//
//   iteration t:  barrier (tile t has landed for every wave; every wave is done reading stage (t+1)&1)
//                 -> issue the DMA of tile t+1 into stage (t+1)&1   (lands under this tile's MFMAs)
//                 -> fragments of tile t from stage t&1 -> MFMA 32x32x2 (exact fp32 products)
//
// The first-generation kernel (gemm.h: k_gemm) parks every tile in registers, waits, writes it to ONE LDS buffer
// between two barriers: per K tile a barrier -> 6-8 ds_write_b128 -> barrier -> address arithmetic -> global loads
// -> ds_read chain during which the wave issues no MFMA (profiles/r01_mfma_pmc.json: 42-53 % MFMA-busy).
//
// LDS images (an LDS-DMA instruction writes 1 KiB lane-linear: the image cannot be padded, so the conflict-free layout
// comes from permuting the per-lane SOURCE address and applying the same permutation to the read):
//   contraction-contiguous operand: [rows][32] floats, 128-B rows of eight 16-B chunks; chunk c of row r is stored at
//     position c ^ ((r >> 1) & 7).  A ds_read_b128 serves lanes in groups of 16 whose rows are {0-3, 12-15, 20-27} or
//     {4-11, 16-19, 28-31} (+32 k): with this permutation the 16 (row parity, position) pairs of a group are distinct,
//     i.e. 16 different 16-B bank slots of the 256-B bank row -> conflict-free.
//   contraction-strided operand: [32][rows] floats as it lies in memory; fragments are ds_read_b32 along the rows
//     (32 consecutive floats per lane half: conflict-free).
// Requirements (else the caller falls back to k_gemm): contraction range of every split a multiple of 32, 16-byte
// loadable rows (the vecA / vecB flags), extents >= 4 along vectorised non-contraction directions.
#pragma once
#include "gemm.h"

#define G2_BK 32

typedef __attribute__((address_space(3))) void* g2_lds_ptr;

// One LDS-DMA: this lane's 16 bytes at `src` land at LDS byte address lds_wave_base + 16 * lane (wave-uniform base in M0).
// Inline asm on purpose: behind `__builtin_amdgcn_global_load_lds` hipcc (ROCm 7.2) waits `vmcnt(0)` in front of the NEXT
// ds_read -- an LDS load "may alias" the DMA's LDS store and carries no alias scope --, which drains the prefetch at once
// (seen in the ISA) and leaves a serial load -> wait -> compute loop.  The compiler does not count asm memory operations,
// so the kernel waits for them itself (g2_dma_wait) before the barrier that publishes a stage.  M0 is compiler-reserved:
// saved, set and restored inside the one statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void g2_dma16(const float* src, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_wave_base)
                 : "memory");
}
__device__ __forceinline__ void g2_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned g2_lds_addr(const float* p) {
    return (unsigned)(uintptr_t)(g2_lds_ptr)(p);
}

// per-thread source pointers of one operand's tile (ROWS/32 DMA rounds), advanced by one K tile per iteration
template <bool KC, int ROWS>
struct G2Stage {
    const float* src[ROWS / 32];
    int64_t step;
    __device__ __forceinline__ void init(const float* P, int64_t ld, int64_t r0, int64_t rmax, int64_t kbeg) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            if (KC) {
                const int row = i * 32 + (tid >> 3);
                const int c = (tid & 7) ^ ((row >> 1) & 7);
                src[i] = P + min(r0 + row, rmax - 1) * ld + kbeg + 4 * c;
            } else {
                constexpr int CPR = ROWS / 4;                     // 16-byte chunks per contraction row
                const int q = i * 256 + tid;
                const int kk = q / CPR, rc = (q % CPR) * 4;
                src[i] = P + (kbeg + kk) * ld + min(r0 + rc, rmax - 4);
            }
        }
        step = KC ? G2_BK : G2_BK * ld;
    }
    // issue the DMA of the current tile into `stage` (the operand's [ROWS * 32] float image) and move on one tile
    __device__ __forceinline__ void issue(const float* stage) {
        const unsigned base = __builtin_amdgcn_readfirstlane(g2_lds_addr(stage) + (threadIdx.x >> 6) * 1024u);
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            g2_dma16(src[i], base + i * 4096u);
            src[i] += step;
        }
    }
};

template <bool A_KC, bool B_KC, int TM, int TN>
__device__ __forceinline__ void gemm2_tile_body(const GemmArgs& g, unsigned bx, unsigned by, unsigned bz, float* lds) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int A_ST = BM * G2_BK, B_ST = BN * G2_BK;        // floats per stage
    float* As = lds;                                            // [2][A_ST]
    float* Bs = lds + 2 * A_ST;                                 // [2][B_ST]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    const int64_t m0 = (int64_t)by * BM;
    const int64_t n0 = (int64_t)bx * BN;
    const int64_t kbeg = (int64_t)bz * g.kchunk;
    const int64_t kend = min(g.K, kbeg + g.kchunk);
    const int nt = (int)((kend - kbeg) / G2_BK);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // bias gradient fused into the weight-gradient GEMM (A = dZ^T): the waves of the first column panel add up the A
    // fragments they read anyway (VALU under the MFMAs); lane (lr, lk) ends up with its row's sum over its half of the
    // contraction indices
    const bool do_colsum = !A_KC && g.colsum != nullptr && bx == 0 && wn == 0;
    float csum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) csum[i] = 0.f;

    G2Stage<A_KC, BM> sa;
    G2Stage<B_KC, BN> sb;
    sa.init(g.A, g.lda, m0, g.M, kbeg);
    sb.init(g.B, g.ldb, n0, g.N, kbeg);
    sa.issue(As);
    sb.issue(Bs);
    const int sw = (lr >> 1) & 7;                               // read-side chunk permutation of contraction-contiguous images
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        g2_dma_wait();              // this wave's share of tile t has landed ...
        __syncthreads();            // ... and everybody's; every wave is done reading stage cur^1 (tile t-1)
        if (t + 1 < nt) {           // wave-uniform; never a DMA in flight when the workgroup ends (its LDS is re-assigned)
            sa.issue(As + (cur ^ 1) * A_ST);
            sb.issue(Bs + (cur ^ 1) * B_ST);
        }
        const float* Ab = As + cur * A_ST;
        const float* Bb = Bs + cur * B_ST;
#pragma unroll
        for (int kg = 0; kg < G2_BK / 8; ++kg) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * (32 * TM) + i * 32 + lr;
                if (A_KC) {
                    a[i] = *reinterpret_cast<const float4*>(Ab + r * G2_BK + 4 * ((kg * 2 + lk) ^ sw));
                } else {
                    const float* q = Ab + (kg * 8 + 4 * lk) * BM + r;
                    a[i] = make_float4(q[0], q[BM], q[2 * BM], q[3 * BM]);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int c = wn * (32 * TN) + j * 32 + lr;
                if (B_KC) {
                    b[j] = *reinterpret_cast<const float4*>(Bb + c * G2_BK + 4 * ((kg * 2 + lk) ^ sw));
                } else {
                    const float* q = Bb + (kg * 8 + 4 * lk) * BN + c;
                    b[j] = make_float4(q[0], q[BN], q[2 * BN], q[3 * BN]);
                }
            }
            if (!A_KC && do_colsum) {
#pragma unroll
                for (int i = 0; i < TM; ++i) csum[i] += (a[i].x + a[i].y) + (a[i].z + a[i].w);
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
    if (!A_KC && g.colsum != nullptr && bx == 0) {
        // the two lane halves of a wave hold the two halves of the contraction range: add them in a fixed order
        if (wn == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float other = __shfl_xor(csum[i], 32, 64);
                const float s = lk == 0 ? csum[i] + other : other + csum[i];
                const int64_t row = m0 + wm * (32 * TM) + i * 32 + lr;
                if (lk == 0 && row < g.M) g.colsum[(int64_t)bz * g.M + row] = s;
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
                if (g.mask_act) {               // activation backward of the layer below, fused into the dgrad
                    const float x = g.mask[row * g.ldmask + col];
                    v = g.mask_act == 1 ? (x > 0.f ? v : 0.f) : v * ((1.0f - x) * x);
                }
                C[row * g.ldc + col] = v;
            }
        }
}

template <bool A_KC, bool B_KC, int TM, int TN>
__global__ void __launch_bounds__(256) k_gemm2(GemmArgs g) {
    // ONE LDS object (a second one beside an LDS-DMA staging array makes hipcc wait vmcnt(0) before every fragment read)
    __shared__ __attribute__((aligned(1024))) float lds[2 * 64 * (TM + TN) * G2_BK];
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned wgid = xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, nwg);
    gemm2_tile_body<A_KC, B_KC, TM, TN>(g, wgid % gridDim.x, (wgid / gridDim.x) % gridDim.y,
                                        wgid / (gridDim.x * gridDim.y), lds);
}

// the DMA kernel applies: every split's contraction range a multiple of 32, 16-byte loadable rows, >= 4 elements along
// the vectorised non-contraction directions
template <bool A_KC, bool B_KC>
static inline bool gemm2_applies(const GemmArgs& g) {
    static int off = -1;
    if (off < 0) {
        const char* e = getenv("CDLRM_GEMM_GLDS");
        off = (e && atoi(e) == 0) ? 1 : 0;
    }
    if (off || !g.vecA || !g.vecB) return false;
    const int64_t kc = g.kchunk < g.K ? g.kchunk : g.K;
    if (g.K < G2_BK || g.K % G2_BK != 0 || kc % G2_BK != 0) return false;
    if (!A_KC && g.M < 4) return false;
    if (!B_KC && g.N < 4) return false;
    return true;
}

template <bool A_KC, bool B_KC>
static void launch_gemm2(const GemmArgs& g, int tm, int tn, int splits, hipStream_t s) {
    dim3 grid((unsigned)cdiv(g.N, 64 * tn), (unsigned)cdiv(g.M, 64 * tm), (unsigned)splits);
    if (tm == 2 && tn == 2) hipLaunchKernelGGL((k_gemm2<A_KC, B_KC, 2, 2>), grid, dim3(256), 0, s, g);
    else if (tm == 2 && tn == 1) hipLaunchKernelGGL((k_gemm2<A_KC, B_KC, 2, 1>), grid, dim3(256), 0, s, g);
    else if (tm == 1 && tn == 2) hipLaunchKernelGGL((k_gemm2<A_KC, B_KC, 1, 2>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_gemm2<A_KC, B_KC, 1, 1>), grid, dim3(256), 0, s, g);
}
