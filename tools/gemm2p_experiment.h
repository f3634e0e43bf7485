// EXPERIMENT (round 5), tools only -- never part of libcdlrm_hip.so: a persistent form of the LDS-DMA GEMM (csrc/gemm_glds.h:
// k_gemm2) with the loads-first epilogue.  Measured against k_gemm2 by tools/gemm_big_tile.hip; the numbers and why it was not
// taken are in gemm_glds.h ("Tried on top of this kernel and dropped").
#pragma once
#include "gemm_glds.h"

// ---- persistent variant for long M (round 5) -------------------------------------------------------------------------
// k_gemm2's workgroups run in lock-step pairs per CU: both are in their prologue (2.5 k cycles: two K tiles' DMA latency) at
// the same time and both in their store tail (10.5 k cycles) at the same time, 13 k of a workgroup's 84-87 k cycles with the
// MFMA pipe idle -- at M = 65536, where a CU works through 32 tiles, as much as at M = 8192.  The vendor library reaches
// 137 TF/s on 65536 x 512 x 512 (tools/bench_kernels.py --only vendor, profiles/r05_gemm_vs_vendor.json) where k_gemm2 reaches
// 104-115.  Here a workgroup slot WALKS its tiles (grid = workgroup slots, tile += grid) and the K-tile DMA pipeline runs on
// ACROSS output tiles: during the last two K iterations of tile i the pieces that go out are the first two K tiles of tile
// i + 1, so the next tile's first fragments are in LDS when the epilogue ends, and the epilogue's stores drain under the
// next tile's MFMAs.  Per tile what is left outside the loop is the epilogue's own issue time.
// Wait counts: vmcnt is ONE in-order counter for loads, LDS-DMA and stores on this part.  At the first K iteration of a new
// tile the queue holds [DMA pieces of K tile 1 (issued before the epilogue)] [the epilogue's memory operations: at least
// 4 * TM * TN of them -- its stores -- when the tile was full]; `vmcnt(4 * TM * TN)` therefore covers the DMA without waiting
// for the stores.  A tile with clipped stores waits vmcnt(0).  The barrier is a bare s_barrier behind explicit wait counts
// (__syncthreads()'s fence would wait vmcnt(0)); LDS needs no fence inside a workgroup.
// (g2_epilogue_full, the loads-first epilogue of full tiles, has since moved into gemm_glds.h: the production kernel takes it)
template <int KEEP>
__device__ __forceinline__ void g2_wait_keep() {
    if (KEEP == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if (KEEP == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (KEEP == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void g2_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <bool A_KC, bool B_KC, int TM, int TN>
__global__ void __launch_bounds__(256) k_gemm2p(GemmArgs g, unsigned ntx, unsigned ntiles) {
    static_assert(A_KC, "forward / dgrad layouts (un-split contraction)");
    __shared__ __attribute__((aligned(1024))) float lds[2 * 64 * (TM + TN) * G2_BK];
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int A_ST = BM * G2_BK, B_ST = BN * G2_BK;
    constexpr int NST = 4 * TM * TN;                            // stores per wave of a full tile's epilogue
    float* As = lds;
    float* Bs = lds + 2 * A_ST;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    const unsigned G = gridDim.x;
    unsigned tile = xcd_remap(blockIdx.x, G);
    if (tile >= ntiles) return;
    const int nt = (int)(g.K / G2_BK);                          // >= 2 (launch_gemm2p)
    f32x16 acc[TM][TN];
    G2Stage<A_KC, BM> sa;
    G2Stage<B_KC, BN> sb;
    constexpr int NPA = G2Stage<A_KC, BM>::NP, NPB = G2Stage<B_KC, BN>::NP, NPIECE = NPA + NPB;
    const unsigned a_dst = __builtin_amdgcn_readfirstlane(g2_lds_addr(As) + wave * 1024u);
    const unsigned b_dst = __builtin_amdgcn_readfirstlane(g2_lds_addr(Bs) + wave * 1024u);
    const int ra = wm * (32 * TM) + lr, rb = wn * (32 * TN) + lr;
    const int sw = (lr >> 1) & 7;
    float4 fa[2][TM], fb[2][TN];

    sa.init(g.A, g.lda, (int64_t)(tile / ntx) * BM, g.M, 0);
    sb.init(g.B, g.ldb, (int64_t)(tile % ntx) * BN, g.N, 0);
#pragma unroll
    for (int i = 0; i < NPA; ++i) sa.piece(i, a_dst);
#pragma unroll
    for (int i = 0; i < NPB; ++i) sb.piece(i, b_dst);
    g2_wait_keep<0>();
    g2_barrier();
#pragma unroll
    for (int i = 0; i < NPA; ++i) sa.piece(i, a_dst + A_ST * 4u);
#pragma unroll
    for (int i = 0; i < NPB; ++i) sb.piece(i, b_dst + B_ST * 4u);
    g2_frags<A_KC, B_KC, TM, TN>(As, Bs, 0, ra, rb, lk, sw, fa[0], fb[0]);

    auto mfma_group = [&](const float4 (&a)[TM], const float4 (&b)[TN]) {
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

    unsigned cur = 0;               // LDS stage of the K tile being multiplied (runs on across output tiles)
    bool keep_stores = false;       // the youngest NST memory operations in flight are the previous tile's stores
    for (;;) {
        const unsigned next = tile + G;
        const bool has_next = next < ntiles;                    // wave-uniform
        const unsigned bx = tile % ntx, by = tile / ntx;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int t = 0; t < nt; ++t) {
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
            // the K tiles one and two ahead in the workgroup's SEQUENCE: this output tile's, or the next one's first two
            const bool more = (t + 1 < nt) | has_next, more2 = (t + 2 < nt) | has_next;
            if (t == nt - 2 && has_next) {                      // the pieces issued below are the next tile's K tile 0
                sa.init(g.A, g.lda, (int64_t)(next / ntx) * BM, g.M, 0);
                sb.init(g.B, g.ldb, (int64_t)(next % ntx) * BN, g.N, 0);
            }
            if (more) {
                // this wave's pieces of the next K tile have landed (issued a whole K tile ago; behind them at most the previous
                // output tile's stores), everybody's after the barrier; every wave has read its last fragments of stage cur
                if (keep_stores) g2_wait_keep<NST>();
                else g2_wait_keep<0>();
                keep_stores = false;
                g2_barrier();
                g2_frags<A_KC, B_KC, TM, TN>(As + (cur ^ 1) * A_ST, Bs + (cur ^ 1) * B_ST, 0, ra, rb, lk, sw, fa[0], fb[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const unsigned a_st = a_dst + cur * (A_ST * 4u), b_st = b_dst + cur * (B_ST * 4u);
#pragma unroll
            for (int m = 0; m < 4 * TM * TN; ++m) {             // one DMA piece behind each of the first MFMAs
                const int i = (m >> 2) / TN, j = (m >> 2) % TN, comp = m & 3;
                const float av = comp == 0 ? fa[1][i].x : comp == 1 ? fa[1][i].y : comp == 2 ? fa[1][i].z : fa[1][i].w;
                const float bv = comp == 0 ? fb[1][j].x : comp == 1 ? fb[1][j].y : comp == 2 ? fb[1][j].z : fb[1][j].w;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[i][j], 0, 0, 0);
                if (m < NPIECE && more2) {
                    if (m < NPA) sa.piece(m, a_st);
                    else sb.piece(m - NPA, b_st);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            cur ^= 1;
        }
        // epilogue: its stores drain under the next tile's loop
        const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;
        const bool full = m0 + BM <= g.M && n0 + BN <= (int64_t)g.N;
        const int actk = B_KC ? g.act : g.mask_act;             // (launch_gemm2p: forward carries no mask, dgrad no bias / act)
        if (full) {
            if (actk == 1) g2_epilogue_full<TM, TN, B_KC, 1>(g, acc, m0, n0, wm, wn, lane);
            else if (actk == 2) g2_epilogue_full<TM, TN, B_KC, 2>(g, acc, m0, n0, wm, wn, lane);
            else g2_epilogue_full<TM, TN, B_KC, 0>(g, acc, m0, n0, wm, wn, lane);
        } else {
            g2_epilogue<TM, TN>(g, acc, m0, n0, 0, wm, wn, lane);
        }
        keep_stores = full;
        if (!has_next) break;
        tile = next;
    }
}

// Forward / dgrad launches with an un-split contraction take k_gemm2p whenever its full-tile epilogue applies.  Measured
// (tools/gemm_big_tile.hip, round 5; bit-identical outputs): at M = 8192, one tile per slot, 42.7 against 44.4 us (512 x 512,
// forward) and 41.3-41.6 against 43.7-44.4 on 64x64 tiles -- the epilogue alone --; at M = 16384 83.2 against 87.0 us; at
// M = 65536 316 against 320: there the two co-resident workgroups of a CU drift apart by themselves and the tails already lie
// under each other's loops -- what is left to the vendor's 251 us is its 256x256 macro tile (a third of the LDS fill per MFMA),
// which this loop structure does not carry at one wave per SIMD (same file: 256x256 / 256x128 tiles 312-318 us).
template <bool A_KC, bool B_KC>
static inline bool gemm2p_applies(const GemmArgs& g, int splits) {
    if (!A_KC || splits != 1 || g.colsum != nullptr || g.K < 2 * G2_BK) return false;
    // the full-tile epilogue: forward = bias (16-byte loadable) + activation, dgrad = activation mask (16-byte loadable rows)
    if (B_KC && (g.mask_act != 0 || (g.bias != nullptr && (((uintptr_t)g.bias) & 15) != 0))) return false;
    if (!B_KC && (g.bias != nullptr || g.act != 0)) return false;
    if (!B_KC && g.mask_act != 0 && ((((uintptr_t)g.mask) & 15) != 0 || (g.ldmask & 3) != 0)) return false;
    return true;
}

template <bool A_KC, bool B_KC>
static void launch_gemm2p(const GemmArgs& g, int tm, int tn, hipStream_t s) {
    if constexpr (A_KC) {
        const unsigned ntx = (unsigned)cdiv(g.N, 64 * tn), nty = (unsigned)cdiv(g.M, 64 * tm);
        const unsigned ntiles = ntx * nty;
        // workgroup slots: what a CU holds of this tile shape (LDS: 48 KB per 128x64 workgroup, 32 KB per 64x64 one), the rest
        // of the tiles are walked.  cdlrm_debug_set(7, n): n slots
        unsigned slots = tm == 2 ? 512u : 1024u;
        if (g_cdlrm_debug[7] > 0) slots = (unsigned)g_cdlrm_debug[7];
        const dim3 grid(ntiles < slots ? ntiles : slots);
        if (tm == 2 && tn == 1) CDLRM_LAUNCH_EV((k_gemm2p<true, B_KC, 2, 1>), grid, dim3(256), 0, s, g, ntx, ntiles);
        else if (tm == 2 && tn == 2) CDLRM_LAUNCH_EV((k_gemm2p<true, B_KC, 2, 2>), grid, dim3(256), 0, s, g, ntx, ntiles);
        else if (tm == 1 && tn == 2) CDLRM_LAUNCH_EV((k_gemm2p<true, B_KC, 1, 2>), grid, dim3(256), 0, s, g, ntx, ntiles);
        else CDLRM_LAUNCH_EV((k_gemm2p<true, B_KC, 1, 1>), grid, dim3(256), 0, s, g, ntx, ntiles);
    }
}

