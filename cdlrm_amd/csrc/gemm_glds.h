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
#ifndef G2_PRIO
#define G2_PRIO 0       // experiment (tools/gemm2_bench.hip): wave priority of the first-dispatched half of the grid
#endif
#ifndef G2_ABL
#define G2_ABL 0        // diagnostic builds: 1 no DMA in the loop, 2 no barrier in the loop, 4 pieces behind every 2nd MFMA
#endif

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

// per-thread source pointers of one operand's tile (ROWS/32 DMA pieces per thread), advanced by one K tile per issue
template <bool KC, int ROWS>
struct G2Stage {
    static constexpr int NP = ROWS / 32;
    const float* src[NP];
    int64_t step;
    __device__ __forceinline__ void init(const float* P, int64_t ld, int64_t r0, int64_t rmax, int64_t kbeg) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
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
    // piece i of the current tile -> stage image at LDS byte address `base` (this wave's first piece); moves on one tile
    __device__ __forceinline__ void piece(int i, unsigned base) {
        g2_dma16(src[i], base + i * 4096u);
        src[i] += step;
    }
};

// the MFMA fragments of one 8-deep contraction group: lane half lk holds indices 4*lk .. 4*lk+3 of the group
template <bool A_KC, bool B_KC, int TM, int TN>
__device__ __forceinline__ void g2_frags(const float* Ab, const float* Bb, int kg, int ra, int rb, int lk, int sw,
                                         float4 (&a)[TM], float4 (&b)[TN]) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = ra + i * 32;
        if (A_KC) {
            a[i] = *reinterpret_cast<const float4*>(Ab + r * G2_BK + 4 * ((kg * 2 + lk) ^ sw));
        } else {
            const float* q = Ab + (kg * 8 + 4 * lk) * BM + r;
            a[i] = make_float4(q[0], q[BM], q[2 * BM], q[3 * BM]);
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = rb + j * 32;
        if (B_KC) {
            b[j] = *reinterpret_cast<const float4*>(Bb + c * G2_BK + 4 * ((kg * 2 + lk) ^ sw));
        } else {
            const float* q = Bb + (kg * 8 + 4 * lk) * BN + c;
            b[j] = make_float4(q[0], q[BN], q[2 * BN], q[3 * BN]);
        }
    }
}

template <int TM, int TN>
__device__ __forceinline__ void g2_epilogue(const GemmArgs& g, f32x16 (&acc)[TM][TN], int64_t m0, int64_t n0, unsigned bz,
                                            int wm, int wn, int lane) {
    // Epilogue.  The MFMAs were issued with the operand roles swapped (first operand = the B fragment), so an accumulator
    // tile is C^T: lane l holds output ROW (l & 31) and, per group of four registers, four CONSECUTIVE columns
    // 8*(r>>2) + 4*(l>>5) + (0..3) -- one 16-byte store per lane and register group (4 per 32x32 tile instead of the 16
    // scalar stores of the row-per-register layout: the store tail of the first version took 20 k cycles per workgroup,
    // a fifth of the kernel).  Same products in the same k order: bit-identical results.
    float* C = g.C + (int64_t)bz * g.slab;
    const bool vbias = g.bias != nullptr && (((uintptr_t)g.bias) & 15) == 0;
    const bool vmask = g.mask_act != 0 && (((uintptr_t)g.mask) & 15) == 0 && (g.ldmask & 3) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int64_t row = m0 + wm * (32 * TM) + i * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t col = n0 + wn * (32 * TN) + j * 32 + 8 * q + 4 * (lane >> 5);
                if (row >= g.M || col >= g.N) continue;          // N % 4 == 0: a group of four columns is in or out
                float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                if (g.bias) {
                    float4 bv;
                    if (vbias) bv = *reinterpret_cast<const float4*>(g.bias + col);
                    else bv = make_float4(g.bias[col], g.bias[col + 1], g.bias[col + 2], g.bias[col + 3]);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                if (g.act == 1) {
                    v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                    v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                } else if (g.act == 2) {
                    v.x = 1.0f / (1.0f + expf(-v.x)); v.y = 1.0f / (1.0f + expf(-v.y));
                    v.z = 1.0f / (1.0f + expf(-v.z)); v.w = 1.0f / (1.0f + expf(-v.w));
                }
                if (g.mask_act) {               // activation backward of the layer below, fused into the dgrad
                    float4 x;
                    const float* mp = g.mask + row * g.ldmask + col;
                    if (vmask) x = *reinterpret_cast<const float4*>(mp);
                    else x = make_float4(mp[0], mp[1], mp[2], mp[3]);
                    if (g.mask_act == 1) {
                        v.x = x.x > 0.f ? v.x : 0.f; v.y = x.y > 0.f ? v.y : 0.f;
                        v.z = x.z > 0.f ? v.z : 0.f; v.w = x.w > 0.f ? v.w : 0.f;
                    } else {
                        v.x *= (1.0f - x.x) * x.x; v.y *= (1.0f - x.y) * x.y;
                        v.z *= (1.0f - x.z) * x.z; v.w *= (1.0f - x.w) * x.w;
                    }
                }
                *reinterpret_cast<float4*>(C + row * g.ldc + col) = v;
            }
        }
    }
}

// Epilogue of a tile that lies INSIDE the matrix, forward / dgrad layouts (round 5): every operand word it needs (bias: forward;
// the activation mask: dgrad) is loaded FIRST, then the 4 * TM * TN stores leave back to back -- the generic epilogue above
// interleaves a load, its wait and a store per register group, and behind a store that wait (vmcnt(0)) is the store's round
// trip: the "store tail" of 10.5 k cycles per workgroup was eight of those in a row.  Same values, same stores: bit-identical.
template <int TM, int TN, bool B_KC, int ACT>
__device__ __forceinline__ void g2_epilogue_full(const GemmArgs& g, f32x16 (&acc)[TM][TN], int64_t m0, int64_t n0, int wm, int wn,
                                                 int lane) {
    float4 bv[TN][4];
    float4 mv[B_KC ? 1 : TM][B_KC ? 1 : TN][4];
    const int64_t row0 = m0 + wm * (32 * TM) + (lane & 31);
    const int64_t col0 = n0 + wn * (32 * TN) + 4 * (lane >> 5);
    if (B_KC) {
        if (g.bias) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) bv[j][q] = *reinterpret_cast<const float4*>(g.bias + col0 + j * 32 + 8 * q);
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) bv[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else if (ACT != 0) {      // (mask_act == 0: no mask -- g.mask may be null, nothing is loaded)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    mv[i][j][q] = *reinterpret_cast<const float4*>(g.mask + (row0 + i * 32) * g.ldmask + col0 + j * 32 + 8 * q);
    }
    const bool has_bias = g.bias != nullptr;        // (no bias: no add -- x + 0.f would turn -0.0 into +0.0, the generic epilogue's values differ)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                if (B_KC) {
                    const float4 b = bv[j][q];
                    if (has_bias) { v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                    if (ACT == 1) {
                        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                    } else if (ACT == 2) {
                        v.x = 1.0f / (1.0f + expf(-v.x)); v.y = 1.0f / (1.0f + expf(-v.y));
                        v.z = 1.0f / (1.0f + expf(-v.z)); v.w = 1.0f / (1.0f + expf(-v.w));
                    }
                } else if (ACT != 0) {
                    const float4 x = mv[i][j][q];
                    if (ACT == 1) {
                        v.x = x.x > 0.f ? v.x : 0.f; v.y = x.y > 0.f ? v.y : 0.f;
                        v.z = x.z > 0.f ? v.z : 0.f; v.w = x.w > 0.f ? v.w : 0.f;
                    } else if (ACT == 2) {
                        v.x *= (1.0f - x.x) * x.x; v.y *= (1.0f - x.y) * x.y;
                        v.z *= (1.0f - x.z) * x.z; v.w *= (1.0f - x.w) * x.w;
                    }
                }
                *reinterpret_cast<float4*>(g.C + (row0 + i * 32) * g.ldc + col0 + j * 32 + 8 * q) = v;
            }
}

// what g2_epilogue_full can take: forward = bias (absent or 16-byte loadable) + activation, no mask; dgrad = an activation mask
// with 16-byte loadable rows (or none), no bias / activation.  `fastep` is set by the host (launch_gemm) from the debug switch.
template <bool B_KC>
__device__ __forceinline__ bool g2_full_ok(const GemmArgs& g) {
    if (!g.fastep) return false;
    if (B_KC) return g.mask_act == 0 && (g.bias == nullptr || (((uintptr_t)g.bias) & 15) == 0);
    return g.bias == nullptr && g.act == 0 && (g.mask_act == 0 || ((((uintptr_t)g.mask) & 15) == 0 && (g.ldmask & 3) == 0));
}

#ifdef GEMM2_STAMP      // diagnostic builds only (tools/gemm2_bench.hip): where a workgroup's time goes, and at what clock
__device__ unsigned long long g2_stamps[8 * 4096];
#define G2_STAMP(slot)                                                                            \
    if (threadIdx.x == 0 && g2_wg < 4096) {                                                         \
        g2_stamps[g2_wg * 8 + (slot)] = __builtin_amdgcn_s_memtime();                                \
        if ((slot) == 0 || (slot) == 3) g2_stamps[g2_wg * 8 + 4 + ((slot) ? 1 : 0)] = __builtin_amdgcn_s_memrealtime(); \
        if ((slot) == 0) g2_stamps[g2_wg * 8 + 6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | \
                                                    (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);               \
    }
#else
#define G2_STAMP(slot)
#endif

template <bool A_KC, bool B_KC, int TM, int TN>
__device__ __forceinline__ void gemm2_tile_body(const GemmArgs& g, unsigned bx, unsigned by, unsigned bz, float* lds) {
#ifdef GEMM2_STAMP
    const unsigned g2_wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#endif
    G2_STAMP(0)
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
    constexpr int NPA = G2Stage<A_KC, BM>::NP, NPB = G2Stage<B_KC, BN>::NP, NPIECE = NPA + NPB;
    sa.init(g.A, g.lda, m0, g.M, kbeg);
    sb.init(g.B, g.ldb, n0, g.N, kbeg);
    // LDS byte address of this wave's first DMA piece in stage 0 of each operand (wave-uniform: scalar registers)
    const unsigned a_dst = __builtin_amdgcn_readfirstlane(g2_lds_addr(As) + wave * 1024u);
    const unsigned b_dst = __builtin_amdgcn_readfirstlane(g2_lds_addr(Bs) + wave * 1024u);
    const int ra = wm * (32 * TM) + lr, rb = wn * (32 * TN) + lr;
    const int sw = (lr >> 1) & 7;                               // read-side chunk permutation of contraction-contiguous images

    // One wave's instruction stream is in-order: whatever sits between two MFMAs of a wave delays the second one unless
    // it fits into the 64 cycles the first one executes.  So the loop is software-pipelined by hand:
    //   * fragments are double-buffered in registers: the reads of group kg+1 are issued in front of the MFMAs of group kg;
    //   * the barrier that publishes tile t+1 sits in front of the LAST group of tile t, with that group's MFMAs (fragments
    //     already in registers) queued behind it: they cover the DMA issue of tile t+2 -- one piece after each MFMA --
    //     and the LDS latency of tile t+1's first fragments.
    // (The un-pipelined form of this loop -- barrier, all DMA pieces, fragment reads, MFMAs -- measured 89-96 TF/s at
    //  8192 x 512 x 512: two workgroups per CU run in lockstep, their stalls coincide instead of covering each other.)
    float4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < NPA; ++i) sa.piece(i, a_dst);
#pragma unroll
    for (int i = 0; i < NPB; ++i) sb.piece(i, b_dst);
    g2_dma_wait();
    __syncthreads();
    if (nt > 1) {
#pragma unroll
        for (int i = 0; i < NPA; ++i) sa.piece(i, a_dst + A_ST * 4u);
#pragma unroll
        for (int i = 0; i < NPB; ++i) sb.piece(i, b_dst + B_ST * 4u);
    }
    g2_frags<A_KC, B_KC, TM, TN>(As, Bs, 0, ra, rb, lk, sw, fa[0], fb[0]);
    G2_STAMP(1)

    auto mfma_group = [&](const float4 (&a)[TM], const float4 (&b)[TN]) {
        if (!A_KC && do_colsum) {
#pragma unroll
            for (int i = 0; i < TM; ++i) csum[i] += (a[i].x + a[i].y) + (a[i].z + a[i].w);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].x, a[i].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].y, a[i].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].z, a[i].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].w, a[i].w, acc[i][j], 0, 0, 0);
            }
    };

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const float* Ab = As + cur * A_ST;
        const float* Bb = Bs + cur * B_ST;
        g2_frags<A_KC, B_KC, TM, TN>(Ab, Bb, 1, ra, rb, lk, sw, fa[1], fb[1]);
        mfma_group(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        g2_frags<A_KC, B_KC, TM, TN>(Ab, Bb, 2, ra, rb, lk, sw, fa[0], fb[0]);
        mfma_group(fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        g2_frags<A_KC, B_KC, TM, TN>(Ab, Bb, 3, ra, rb, lk, sw, fa[1], fb[1]);
        mfma_group(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        const bool more = t + 1 < nt, more2 = t + 2 < nt;       // wave-uniform
        if (more) {
            g2_dma_wait();          // this wave's pieces of tile t+1 have landed (issued a whole tile ago) ...
            if (!(G2_ABL & 2))
            __syncthreads();        // ... and everybody's; every wave has read its last fragments of tile t (stage cur)
            g2_frags<A_KC, B_KC, TM, TN>(As + (cur ^ 1) * A_ST, Bs + (cur ^ 1) * B_ST, 0, ra, rb, lk, sw, fa[0], fb[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // last group of tile t: its MFMAs interleaved with the DMA pieces of tile t+2 (into stage cur, free since the barrier)
        if (!A_KC && do_colsum) {
#pragma unroll
            for (int i = 0; i < TM; ++i) csum[i] += (fa[1][i].x + fa[1][i].y) + (fa[1][i].z + fa[1][i].w);
        }
        const unsigned a_st = a_dst + cur * (A_ST * 4u), b_st = b_dst + cur * (B_ST * 4u);
#pragma unroll
        for (int m = 0; m < 4 * TM * TN; ++m) {                 // one DMA piece behind each of the first MFMAs
            const int i = (m >> 2) / TN, j = (m >> 2) % TN, comp = m & 3;
            const float av = comp == 0 ? fa[1][i].x : comp == 1 ? fa[1][i].y : comp == 2 ? fa[1][i].z : fa[1][i].w;
            const float bv = comp == 0 ? fb[1][j].x : comp == 1 ? fb[1][j].y : comp == 2 ? fb[1][j].z : fb[1][j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[i][j], 0, 0, 0);
            constexpr int PSTEP = (G2_ABL & 4) ? 2 : 1;
            if (!(G2_ABL & 1) && (m % PSTEP) == 0 && m / PSTEP < NPIECE && more2) {
                if (m / PSTEP < NPA) sa.piece(m / PSTEP, a_st);
                else sb.piece(m / PSTEP - NPA, b_st);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    G2_STAMP(2)
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
    // a tile inside the matrix of an un-split forward / dgrad launch takes the loads-first epilogue (cdlrm_debug_set(7, 1): never)
    bool fast = A_KC && gridDim.z == 1 && m0 + BM <= g.M && n0 + BN <= (int64_t)g.N && g2_full_ok<B_KC>(g);
    if (fast) {
        const int actk = B_KC ? g.act : g.mask_act;
        if (actk == 1) g2_epilogue_full<TM, TN, B_KC, 1>(g, acc, m0, n0, wm, wn, lane);
        else if (actk == 2) g2_epilogue_full<TM, TN, B_KC, 2>(g, acc, m0, n0, wm, wn, lane);
        else g2_epilogue_full<TM, TN, B_KC, 0>(g, acc, m0, n0, wm, wn, lane);
    } else {
        g2_epilogue<TM, TN>(g, acc, m0, n0, bz, wm, wn, lane);
    }
    G2_STAMP(3)
}

template <bool A_KC, bool B_KC, int TM, int TN>
__global__ void __launch_bounds__(256) k_gemm2(GemmArgs g) {
    // ONE LDS object (a second one beside an LDS-DMA staging array makes hipcc wait vmcnt(0) before every fragment read)
    __shared__ __attribute__((aligned(1024))) float lds[2 * 64 * (TM + TN) * G2_BK];
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned orig = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned wgid = xcd_remap(orig, nwg);
#if G2_PRIO
    // experiment: the first half of the grid (dispatched first: one workgroup per CU) outranks the second half, its CU-mates
    if (orig < (nwg >> 1)) __builtin_amdgcn_s_setprio(G2_PRIO);
#endif
    gemm2_tile_body<A_KC, B_KC, TM, TN>(g, wgid % gridDim.x, (wgid / gridDim.x) % gridDim.y,
                                        wgid / (gridDim.x * gridDim.y), lds);
}

// Tried on top of this kernel and dropped (tools/gemm2_bench.hip, M = 8192, 512-wide layers, in-kernel stamps):
//   * a fifth, DMA-only wave per workgroup (three LDS stages, two tiles in flight, counted vmcnt): 42.7-43.9 us against
//     42.0-42.3 -- the compute waves do not get faster when they issue no DMA (149 cycles per MFMA and wave against 144;
//     132 with the DMA ablated altogether, 129 with the barrier ablated too, ideal 128): what costs them ~8 % is the
//     tile traffic into the CU's LDS itself, not the issue slots;
//   * starting half of the workgroups 2.5-8 k cycles late so that the store tails of the co-resident pairs do not
//     coincide: 40.6 against 42.0 us on the forward layout, nothing on the weight-gradient layout -- and in the step
//     (round 3, tools/ab_step.py) 0.641 / 0.652 / 0.677 ms at 3 / 6 / 12 k cycles against 0.630: the delay is paid by every
//     GEMM of the chain, the tails it separates were hidden under the side queues' kernels anyway;
//   * 128x128 tiles (one workgroup per CU): 42.3-43.4 us, no better than 128x64 with two;
//   * (round 3) the output tile taken through LDS behind the loop and stored row-major -- a wave instruction = 4 whole rows of
//     256 bytes instead of 32 rows x 32 bytes, the dgrad's mask read the same way: bit-identical, and no faster in the step
//     (tools/ab_step.py, c3: 0.6481 against 0.6463 ms; c5: 4.139 against 4.132): the store tail is not bound by the number
//     of write requests;
//   * (round 3) a whole MLP direction as ONE launch, a workgroup taking 32 rows through every layer (no launch boundaries, no
//     lock-step tail; weights streamed per workgroup, the activations re-read from L2): correct, and LDS-bound -- a 32-row
//     block moves 2.3x the LDS bytes per MFMA of the 128x64 tile (136 KB of stages, 62 % of the LDS bandwidth at the MFMA
//     rate) -- top MLP forward 100 us against 108 layer by layer, its input-gradient chain 124 against 106, bottom MLP 63
//     against 51; in the step 0.647 (top forward only) / 0.672 / 0.717 ms (all four chains) against 0.646.  Removed;
//   * (round 3) the split-M weight gradients of a sub-network's layers as ONE grouped launch of this kernel (1536 workgroups
//     for the top MLP: one ramp, the problems' store tails under each other's loops): 0.667 against 0.641 ms per step --
//     the side queue's kernel then holds every workgroup slot for 250 us and the training queue's GEMMs wait for slots.
//   * (round 3, second session) write-through C stores (buffer stores with sc1 / sc0 sc1 / nt cache bits, so that nothing
//     stays dirty in the XCDs' L2s for the end-of-kernel write-back): 41.9 / 41.9 / 42.5 us against 41.9 plain on the forward
//     layout, the epilogue itself 15.6 k cycles instead of 11.6 k -- back-to-back launches of this kernel are 1.4 us apart
//     (first workgroup start -> last end 40.5 us of 41.9), the boundary is not where the time is;
//   * (same session) bias words / the dgrad's activation-mask words fetched into registers BEFORE the K loop (144 VGPRs):
//     the epilogue stays at 10.8 k / 9.6 k cycles (11.6 k / 11.0 k) and the kernel at 41.9 / 42.6 us -- the store tail is the
//     write burst itself (16 MB leaving 512 lock-stepped workgroups in ~5 us = 3.3 TB/s; the weight-gradient layout, whose
//     workgroups drift apart in the loop, shows 5.0 k cycles for the same bytes), not the latency of its operands.
//   * (round 5) the yardstick: the vendor library's fp32 GEMM on these shapes (torch.mm / addmm, tools/bench_kernels.py --only
//     vendor, profiles/r05_gemm_vs_vendor.json): 8192 x 512 x 512 forward 38.9 us against 41.8-45.1 here, 65536 x 512 x 512
//     251 us (137 TF/s) against 298-320 -- its kernel (rocprofv3 name: MT256x256x32_MI16x16x1 ... MIWT8_8 ... SK3) is a 256x256
//     macro tile, 128x128 per wave, ONE workgroup per CU, stream-K over the contraction; the weight gradients go the other way
//     (48.6 against 55.3 us, 294 against 406 at M = 65536), and with the epilogues the library leaves to other kernels (ReLU, act',
//     bias gradient) the MLP's whole GEMM set costs 389 us here against 535 at c3, 2454 against 3287 at c5.  What was built on
//     that evidence, measured and left in tools/ (tools/gemm2p_experiment.h, tools/gemm_big_tile.hip; bit-identical outputs):
//     - this kernel on 128x128 / 256x128 / 128x256 / 256x256 tiles (per wave up to 128x128, one wave per SIMD): 300.7 / 317.6 /
//       318.4 / 312.3 us against 317.7 at M = 65536, 44.2 / 76.4 / 76.9 / 145.2 against 44.2 at M = 8192 -- this loop needs its
//       second wave per SIMD; the library's schedule at one wave per SIMD is hand-placed assembly;
//     - a PERSISTENT form (workgroup slots walk their tiles; the K-tile DMA pipeline runs on across output tiles, so the next
//       tile's first fragments are in LDS when the epilogue ends; the epilogue loads all bias / mask words first and its stores
//       leave back to back and drain under the next tile's MFMAs, counted vmcnt): stand-alone 42.7 against 44.4 us at M = 8192,
//       83.2 against 87.0 at 16384, 316 against 320 at 65536 -- and in the STEP, same box, six rounds (tools/ab_step.py): c3
//       0.5658 against 0.5635 ms, c5 3.7355 against 3.7219, per-rank 4096 a tie: 0.4 % SLOWER (102 + 62 registers against
//       52 + 32; the tails it removes lie under the side queues' kernels in the step).  Not taken.
// Where a workgroup's 83 k cycles go at 8192 x 512 x 512 (128x64 tile, two workgroups per CU, 2.3-2.4 GHz): prologue 2.5 k,
// loop 73.7 k (ideal 65.5 k), epilogue 10.5 k -- the 16 MB of output leave all 512 workgroups at the same moment.

// the DMA kernel applies: every split's contraction range a multiple of 32, 16-byte loadable rows, >= 4 elements along
// the vectorised non-contraction directions
template <bool A_KC, bool B_KC>
static inline bool gemm2_applies(const GemmArgs& g) {
    if (!g.vecA || !g.vecB) return false;
    const int64_t kc = g.kchunk < g.K ? g.kchunk : g.K;
    if (g.K < G2_BK || g.K % G2_BK != 0 || kc % G2_BK != 0) return false;
    if (!A_KC && g.M < 4) return false;
    if (!B_KC && g.N < 4) return false;
    // the epilogue stores four consecutive columns per lane
    if (g.N % 4 != 0 || g.ldc % 4 != 0 || g.slab % 4 != 0 || (((uintptr_t)g.C) & 15) != 0) return false;
    return true;
}

#ifndef G2_EXTRA_LDS
#define G2_EXTRA_LDS(tm, tn) 0
#endif
template <bool A_KC, bool B_KC>
static void launch_gemm2(const GemmArgs& g, int tm, int tn, int splits, hipStream_t s) {
    dim3 grid((unsigned)cdiv(g.N, 64 * tn), (unsigned)cdiv(g.M, 64 * tm), (unsigned)splits);
    const size_t dyn = (size_t)G2_EXTRA_LDS(tm, tn);
    if (tm == 2 && tn == 2) CDLRM_LAUNCH_EV((k_gemm2<A_KC, B_KC, 2, 2>), grid, dim3(256), dyn, s, g);
    else if (tm == 2 && tn == 1) CDLRM_LAUNCH_EV((k_gemm2<A_KC, B_KC, 2, 1>), grid, dim3(256), dyn, s, g);
    else if (tm == 1 && tn == 2) CDLRM_LAUNCH_EV((k_gemm2<A_KC, B_KC, 1, 2>), grid, dim3(256), dyn, s, g);
    else CDLRM_LAUNCH_EV((k_gemm2<A_KC, B_KC, 1, 1>), grid, dim3(256), dyn, s, g);
}

// the wide kernel (gemm_wide.h: k_gemm3, one workgroup per CU on 128x128 tiles) takes the un-split forward / dgrad GEMMs whose
// tiles fill the chip; defined in gemm_wide.h, which a translation unit that calls launch_gemm includes instead of this file
template <bool A_KC, bool B_KC>
static bool gemm3_try(const GemmArgs& g, int splits, hipStream_t s);

template <bool A_KC, bool B_KC>
static int launch_gemm(GemmArgs g, int splits, hipStream_t s) {
    if (g.K < 4) g.vecA = g.vecB = 0;
    if constexpr (A_KC || !B_KC) {
        if (gemm3_try<A_KC, B_KC>(g, splits, s)) {
            CDLRM_LAUNCH_CHECK();
            return 0;
        }
    }
    // (long batches with a narrow output -- the 256 -> 128 layer at M = 8192, forward: 256 tiles of 64x64, and its weight
    //  gradient: 8 tiles x 32 slabs of the batch -- are better off on the LDS-DMA kernel's 64x64 tile than on the LDS-free
    //  one: 10.3 against 13.1 us forward)
    const bool long_narrow = (g.M >= 4096 || g.K >= 4096) && cdiv(g.M, 64) * cdiv(g.N, 64) * splits >= 256 &&
                             gemm2_applies<A_KC, B_KC>(g);
    if (gemm_use_direct(g.M, g.N, splits) && !long_narrow) {
        launch_gemm_direct<A_KC, B_KC>(g, splits, s);
        CDLRM_LAUNCH_CHECK();
        return 0;
    }
    if (gemm2_applies<A_KC, B_KC>(g)) {
        // LDS-DMA kernel.  Measured on the c3 layer shapes (tools/gemm2_bench.hip, M = 8192, all three layouts): 128x64
        // tiles win wherever they leave MORE than one workgroup per CU (512-wide layers 40-42 us against 43-47 for 64x64), the
        // 64x64 tile from there down (128-wide output: 9.9 against 14.3 us).  At exactly one per CU -- the 256-wide layers at
        // M = 8192: top forward 512 -> 256, bottom forward 512 -> 256, bottom dgrad 256 <- 128 -- 512 workgroups of 64x64 beat
        // 256 of 128x64 in the step: 0.5740 against 0.5767 ms, six rounds of 110 steps each, every round (round 4; bit-identical:
        // a tile's k order does not depend on its shape)
        int tm2 = 2, tn2 = 1;
        if (g.M <= 64 || cdiv(g.M, 128) * cdiv(g.N, 64) * splits <= 256) tm2 = 1;
        // ... and so do the short contractions (K <= 256 un-split: the dgrads 512 <- 256 of both sub-networks, eight K tiles per
        // workgroup, where prologue and store tail weigh most): 1024 workgroups of 64x64 instead of 512 of 128x64, 0.5681 against
        // 0.5739 ms per c3 step, six rounds, every round.  (EVERY forward / dgrad on 64x64: 0.5776 against 0.5747; the weight
        // gradients too: 0.5819 / 0.5863 -- the 512-wide layers keep 128x64.)
        if (splits == 1 && g.K <= 256) tm2 = 1;
        // ... and where a CU gets >= 4 tiles of 128x128 (un-split forward / dgrad at M = 65536: c5) that shape, one workgroup per
        // CU, a third less LDS fill per MFMA: stand-alone 297.5 against 313.7 us (512 x 512 forward), 283.1 / 297.7 (512 <- 480),
        // 158.7 / 165.9 (256 <- 512), dgrad 318.0 / 322.5 (profiles/r05_gemm_big_tiles.txt); in the c5 step 3.6927 against
        // 3.7159 ms, ten rounds (-0.6 %; cdlrm_debug_set(6, 16): 128x64 as before).  Bit-identical (a tile's k order does not
        // depend on its shape).  At M = 8192 the same shape is one tile per CU and loses (round 2, and again in round 5).
        if (!(g_cdlrm_debug[6] & 16) && A_KC && splits == 1 && tm2 == 2 && cdiv(g.M, 128) * cdiv(g.N, 128) >= 1024) tn2 = 2;
        // (the same shape for the split-M weight gradients of a long batch: a tie, c5 3.7017 against 3.6995 ms, ten rounds -- not taken)
        launch_gemm2<A_KC, B_KC>(g, tm2, tn2, splits, s);
        CDLRM_LAUNCH_CHECK();
        return 0;
    }
    if (g.N <= 32 || g.M <= 32) {
        // a 13-wide (or 1-wide) side that the DMA kernel cannot load: the LDS-free kernel's 32x32 tiles waste less of the
        // MFMA than the 64x64 staged tile, whatever the number of slabs (the 512 x 13 weight gradient at M = 65536, 128
        // slabs: 1100 us on the tiled kernel, c5's longest launch)
        launch_gemm_direct<A_KC, B_KC>(g, splits, s);
        CDLRM_LAUNCH_CHECK();
        return 0;
    }
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
