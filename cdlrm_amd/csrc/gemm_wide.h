// FP32-MFMA GEMM, third generation ("wide"): ONE workgroup per CU, ONE wave per SIMD, each wave a (16*IM) x (16*JN) accumulator
// tile of v_mfma_f32_16x16x4_f32 blocks, operands global -> LDS by LDS-DMA into a ring of NS stages with COUNTED vmcnt waits
// (the DMA of tile t+NS-1 is issued while tile t is multiplied; nothing in the loop ever waits for a load that was issued
// less than NS-2 whole K tiles ago), one barrier per 32-deep K tile placed in front of the tile's LAST MFMAs so that these
// cover the first fragment reads of the next tile.
//
// Why a third kernel (profiles/r05_gemm_vs_vendor.json): k_gemm2 (gemm_glds.h) runs two 4-wave workgroups
// per CU on 128x64 tiles, its MFMA stream is 12 % longer than the MFMA time (144 cycles per 128), and 512 lock-stepped
// workgroups pay prologue + store tail together; the vendor library's kernel for these shapes keeps one workgroup per CU on
// tiles with 2-4x the MFMAs per barrier.  Here a wave issues 128 MFMAs (4096 cycles) per barrier at IM = JN = 4, every LDS
// read and every DMA piece sits in the shadow of an MFMA (one per MFMA slot), and the ring keeps two K tiles in flight.
//
//   C[m,n] = sum_k A(m,k) * B(k,n)
//   A_KC: A(m,k) = A[m*lda + k] (forward and dgrad)         else A[k*lda + m] (weight gradient: A = dZ^T, split over the batch)
//   B_KC: B(k,n) = B[n*ldb + k] (forward: weights [N, K])   else B[k*ldb + n] (dgrad: weights [K, N]; weight gradient: X)
//
// Contraction order inside a 16-deep group differs from k_gemm2's (lane quarter kq of a 16x16x4 MFMA holds k = 4*kq + c of
// MFMA c: 0,4,8,12, 1,5,9,13, ...), so results differ from the 32x32x2 kernels in the last bits (both are exact fp32 fma
// chains; tests compare against fp64 / torch fp32 with 1e-5 relative).  A tile's k order does not depend on M, N or the grid.
//
// LDS images per stage (an LDS-DMA writes 1 KiB lane-linear, so the conflict-free layout is a permutation of the per-lane
// SOURCE address, undone by the reads):
//   contraction-contiguous operand [rows][32] floats: chunk c (16 B) of row r at position c ^ ((r >> 1) & 7); a
//     ds_read_b128 of fragment (rows r0 .. r0+15, chunk 4g + kq) is conflict-free (tools/lds_bank_check.py);
//   contraction-strided operand [32][cols] floats: row k rotated by 16 floats when (k >> 2) & 1 (the two lane quarters of a
//     ds_read_b32 half-wave hit different bank halves).
#pragma once
#include <type_traits>
#include "gemm_glds.h"

#define G3_BK 32
#ifndef G3_ABL
#define G3_ABL 0       // diagnostic builds (tools/gemm3_bench.hip): 1 no DMA in the steady loop, 2 no barrier, 4 no vmcnt wait
#endif

// One LDS-DMA with a scalar base + 32-bit per-lane byte offset: no per-piece 64-bit pointer arithmetic in the loop.
__device__ __forceinline__ void g3_dma16(unsigned voff, const float* sbase, unsigned lds_wave_base) {
    // M0 is compiler-reserved: saved, set and restored inside the one statement (cdna_hip_programming.md 5.7); the scalar moves
    // cost nothing measurable (the loop without any DMA takes the same cycles: tools/gemm3_bench.hip, -DG3_ABL=1)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_wave_base)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void g3_vmwait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The MFMA as an asm statement with the accumulator tied in place ("+v": destination = SrcC, in VGPRs -- gfx950's register
// file is unified).  Behind the builtin hipcc (ROCm 7.2) renames the 16 accumulators of this loop through the AGPR file and
// copies them back at the loop edge: ~100 v_accvgpr_read / _write per K tile in the MFMA stream (seen in the ISA of the first
// version of this kernel and of k_gemm2 on 128x128 wave tiles).  Volatile asm statements keep their program order, so the
// loop's issue order is the source order.  The hazard recognizer does not look into asm: a dependent MFMA on the SAME
// destination needs no wait states, and the epilogue waits out the last MFMAs itself (g3_mfma_drain).
__device__ __forceinline__ void g3_mfma(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// Region edges.  Wherever compiler-placed code may follow or precede the asm MFMAs (a second copy of the tile body, the last
// tile, the epilogue: it moves accumulators between its register assignments there) the wait states travel INSIDE the asm
// statement: the first MFMA of such a region waits out a VALU write of its accumulator, the last one waits until every MFMA
// of the region has written back (12 wait states behind the last, 20 behind the one before).  The loop body has neither
// (tools/mfma_hazard_check.py verifies that nothing touches an accumulator at the loop's edge).
__device__ __forceinline__ void g3_mfma_first(f32x4& acc, float a, float b) {
    asm volatile("s_nop 3\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void g3_mfma_last(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 3" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void g3_mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory"); }

template <bool KC, int ROWS>
struct G3Stage {
    static constexpr int NP = ROWS / 32;        // DMA pieces per wave and K tile (tile = ROWS * 128 B, 4 waves, 1 KiB pieces)
    unsigned off[NP];                           // per-lane byte offsets from `base` (fixed for the kernel)
    const float* base;                          // wave-uniform: first element of the current K tile
    int64_t step;
    __device__ __forceinline__ void init(const float* P, int64_t ld, int64_t r0, int64_t rmax, int64_t kbeg) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int gp = i * 4 + wave;        // piece index inside the tile image: lands at image + gp KiB
            if (KC) {
                const int row = gp * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                const int64_t rr = min(r0 + row, rmax - 1) - r0;
                off[i] = (unsigned)((rr * ld + 4 * c) * 4);
            } else {
                constexpr int CPR = ROWS / 4;
                const int q = gp * 64 + lane;
                const int kk = q / CPR, pos = q % CPR;
                const int ch = pos ^ (((kk >> 2) & 1) * 4);
                const int64_t cc = min(r0 + 4 * ch, rmax - 4) - r0;
                off[i] = (unsigned)((kk * ld + cc) * 4);
            }
        }
        base = KC ? P + r0 * ld + kbeg : P + kbeg * ld + r0;
        step = KC ? G3_BK : G3_BK * ld;
    }
    __device__ __forceinline__ void piece(int i, unsigned lds_wave_base) const { g3_dma16(off[i], base, lds_wave_base + i * 4096u); }
    __device__ __forceinline__ void advance() { base += step; }
};

#ifdef G3_STAMP
__device__ unsigned long long g3_stamps[8 * 4096];
#define G3_STAMP_AT(slot)                                                                       \
    if (threadIdx.x == 0 && g3_wg < 4096) {                                                      \
        g3_stamps[g3_wg * 8 + (slot)] = __builtin_amdgcn_s_memtime();                             \
        if ((slot) == 0 || (slot) == 4) g3_stamps[g3_wg * 8 + 5 + ((slot) ? 1 : 0)] = __builtin_amdgcn_s_memrealtime(); \
    }
#else
#define G3_STAMP_AT(slot)
#endif

// (Round 6, measured and not kept: NS = 2 -- a 64 KB ring, TWO workgroups per CU at 256 registers each, one's prologue / store
//  tail under the other's MFMAs -- for batches with many tiles per CU: 281 / 290 us against 286 / 304 at M = 65536, scratch
//  spills in the tail tiles; a persistent tile loop would be the real remedy for the ~3 us between a CU's workgroups there.)
template <bool A_KC, bool B_KC, int IM, int JN, int NS>
__global__ void __launch_bounds__(256) k_gemm3(GemmArgs g) {
    constexpr int BM = 32 * IM, BN = 32 * JN;
    constexpr int A_ST = BM * G3_BK, B_ST = BN * G3_BK, STAGE = A_ST + B_ST;       // floats
    constexpr int NPA = G3Stage<A_KC, BM>::NP, NPB = G3Stage<B_KC, BN>::NP, NPW = NPA + NPB;
    constexpr int MG = 4 * IM * JN;                    // MFMAs per 16-deep group and wave
    constexpr int NRA = A_KC ? IM : 4 * IM;             // LDS read instructions per group and wave: A fragments ...
    constexpr int NRD = NRA + (B_KC ? JN : 4 * JN);     // ... and all
    static_assert(NS >= 3 && NS <= 4 && NS * STAGE * 4 <= 160 * 1024, "ring does not fit");
    // the schedule of a 16-deep group (MG MFMA slots): the NRD reads of the next group behind the first MFMAs, then one DMA piece
    // every PSTEP-th MFMA; in the second group the wait + barrier sit TAILC + NRD MFMAs before its end (the next tile's first
    // reads behind them, the last TAILC MFMAs cover the reads' latency)
    constexpr int PSTEP = (MG - NRD) / NPW >= 4 ? 4 : (MG - NRD) / NPW;
    constexpr int TAILC = MG - NRD >= 16 ? 16 : (MG - NRD) / 2;
    static_assert(PSTEP >= 1 && NRD + PSTEP * (NPW - 1) < MG && TAILC >= 4, "schedule does not fit the group");
    // ONE LDS object (a second one beside an LDS-DMA staging array makes hipcc wait vmcnt(0) before every fragment read)
    __shared__ __attribute__((aligned(1024))) float lds[NS * STAGE];

    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned orig = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned wgid = xcd_remap(orig, nwg);
    const unsigned bx = wgid % gridDim.x, by = (wgid / gridDim.x) % gridDim.y, bz = wgid / (gridDim.x * gridDim.y);
#ifdef G3_STAMP
    const unsigned g3_wg = orig;
#endif
    G3_STAMP_AT(0)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, kq = lane >> 4;
    const int64_t m0 = (int64_t)by * BM;
    const int64_t n0 = (int64_t)bx * BN;
    const int64_t kbeg = (int64_t)bz * g.kchunk;
    const int64_t kend = min(g.K, kbeg + g.kchunk);
    const int nt = (int)((kend - kbeg) / G3_BK);

    f32x4 acc[IM][JN];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    G3Stage<A_KC, BM> sa;
    G3Stage<B_KC, BN> sb;
    sa.init(g.A, g.lda, m0, g.M, kbeg);
    sb.init(g.B, g.ldb, n0, g.N, kbeg);
    const unsigned lds0 = g2_lds_addr(lds);
    const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024u);
    const unsigned b_dst = __builtin_amdgcn_readfirstlane(lds0 + A_ST * 4u + wave * 1024u);

    // per-lane fragment addresses (floats from the start of a stage), see the header comment
    const int sw = lr >> 1;
    int fa0, fa1, fb0, fb1;
    if (A_KC) {
        fa0 = (wm * 16 * IM + lr) * G3_BK + 4 * (kq ^ sw);
        fa1 = (wm * 16 * IM + lr) * G3_BK + 4 * ((4 + kq) ^ sw);
    } else {
        fa0 = 4 * kq * BM + wm * 16 * IM + lr + 16 * (kq & 1);               // even blocks
        fa1 = 4 * kq * BM + wm * 16 * IM + lr - 16 * (kq & 1);               // odd blocks
    }
    if (B_KC) {
        fb0 = A_ST + (wn * 16 * JN + lr) * G3_BK + 4 * (kq ^ sw);
        fb1 = A_ST + (wn * 16 * JN + lr) * G3_BK + 4 * ((4 + kq) ^ sw);
    } else {
        fb0 = A_ST + 4 * kq * BN + wn * 16 * JN + lr + 16 * (kq & 1);        // even blocks
        fb1 = A_ST + 4 * kq * BN + wn * 16 * JN + lr - 16 * (kq & 1);        // odd blocks
    }

    // fragment registers: a contraction-contiguous operand's fragment is one ds_read_b128 (float4: the four MFMAs of a group
    // take its components), a contraction-strided one's four ds_read_b32 into four scalars (as components of a float4 every
    // read would redefine the whole vector for the register allocator: copies and spills in the MFMA stream)
    float4 fa4[2][A_KC ? IM : 1], fb4[2][B_KC ? JN : 1];
    float fas[2][A_KC ? 1 : IM][4], fbs[2][B_KC ? 1 : JN][4];
    // read instruction q (0 .. NRD-1) of group gsel of the stage whose per-lane addresses are pa / pb into buffer `buf`
    auto rd = [&](const float* pa0, const float* pa1, const float* pb0, const float* pb1, int gsel, int buf, int q) {
        if (q < NRA) {
            if constexpr (A_KC) {
                fa4[buf][q] = *reinterpret_cast<const float4*>((gsel ? pa1 : pa0) + q * 16 * G3_BK);
            } else {
                const int i = q >> 2, c = q & 3;
                fas[buf][i][c] = ((i & 1) ? pa1 : pa0)[(16 * gsel + c) * BM + 16 * i];
            }
        } else if constexpr (B_KC) {
            const int j = q - NRA;
            fb4[buf][j] = *reinterpret_cast<const float4*>((gsel ? pb1 : pb0) + j * 16 * G3_BK);
        } else {
            const int j = (q - NRA) >> 2, c = (q - NRA) & 3;
            fbs[buf][j][c] = ((j & 1) ? pb1 : pb0)[(16 * gsel + c) * BN + 16 * j];
        }
    };
    auto comp4 = [](const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; };
    auto afrag = [&](int buf, int i, int c) -> float {
        if constexpr (A_KC) return comp4(fa4[buf][i], c);
        else return fas[buf][i][c];
    };
    auto bfrag = [&](int buf, int j, int c) -> float {
        if constexpr (B_KC) return comp4(fb4[buf][j], c);
        else return fbs[buf][j][c];
    };
    // bias gradient fused into the weight-gradient GEMM (A = dZ^T, contraction-strided): the waves of the first column panel
    // add up the A fragments they read anyway (VALU under the MFMAs); lane (lr, kq) collects its row's sum over the contraction
    // indices of its quarter, the four quarters are added in a fixed order behind the loop
    const bool do_colsum = !A_KC && g.colsum != nullptr && bx == 0 && __builtin_amdgcn_readfirstlane(wn) == 0;     // scalar condition
    float csum[IM];
#pragma unroll
    for (int i = 0; i < IM; ++i) csum[i] = 0.f;
    auto colsum_add = [&](int buf, int i) {
        if constexpr (!A_KC) {
            if (do_colsum) csum[i] += (fas[buf][i][0] + fas[buf][i][1]) + (fas[buf][i][2] + fas[buf][i][3]);
        }
    };
    auto dma_piece = [&](int p, int stage) {
        if (p < NPA) sa.piece(p, a_dst + stage * (STAGE * 4u));
        else sb.piece(p - NPA, b_dst + stage * (STAGE * 4u));
        if (p == NPA - 1) sa.advance();
        if (p == NPW - 1) sb.advance();
    };

    // ---- epilogue operands.  MFMA operands are swapped (first = the B fragment): an accumulator block is C^T, lane (lr, kq)
    //      holds output row lr and the four consecutive columns 4*kq .. 4*kq+3 of the block: one 16-byte store per block.
    float* C = g.C + (int64_t)bz * g.slab;
    const int64_t row0 = m0 + wm * 16 * IM + lr;
    const int64_t col0 = n0 + wn * 16 * JN + 4 * kq;
    const bool vbias = g.bias != nullptr && (((uintptr_t)g.bias) & 15) == 0;
    const bool vmask = g.mask_act != 0 && (((uintptr_t)g.mask) & 15) == 0 && (g.ldmask & 3) == 0;
    // a tile inside the matrix with 16-byte loadable epilogue operands: its blocks are finished and stored one by one under
    // the MFMAs of the last 16-deep group (wave-uniform)
#ifdef G3_NOFAST          // (development: every tile through the generic epilogue)
    const bool fast = false;
#else
    const bool fast = m0 + BM <= g.M && n0 + BN <= (int64_t)g.N && (g.bias == nullptr || vbias) && (g.mask_act == 0 || vmask);
#endif
    float4 bv[JN], mv[IM][JN];
#pragma unroll
    for (int j = 0; j < JN; ++j) bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) mv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fast && g.bias) {
#pragma unroll
        for (int j = 0; j < JN; ++j) bv[j] = *reinterpret_cast<const float4*>(g.bias + col0 + 16 * j);
    }
    // the dgrad's activation words, fetched one K tile ahead of the last one (16 loads per lane, ~2 us of HBM latency)
    auto prefetch_mask = [&]() {
        if (fast && g.mask_act) {
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < JN; ++j)
                    mv[i][j] = *reinterpret_cast<const float4*>(g.mask + (row0 + 16 * i) * g.ldmask + col0 + 16 * j);
        }
    };
    auto epi_block = [&](int i, int j) {
        float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (g.mask_act) {
            const float4 x = mv[i][j];
            if (g.mask_act == 1) {
                v.x = x.x > 0.f ? v.x : 0.f; v.y = x.y > 0.f ? v.y : 0.f;
                v.z = x.z > 0.f ? v.z : 0.f; v.w = x.w > 0.f ? v.w : 0.f;
            } else {
                v.x *= (1.0f - x.x) * x.x; v.y *= (1.0f - x.y) * x.y;
                v.z *= (1.0f - x.z) * x.z; v.w *= (1.0f - x.w) * x.w;
            }
        } else {
            if (g.bias) { v.x += bv[j].x; v.y += bv[j].y; v.z += bv[j].z; v.w += bv[j].w; }
            if (g.act == 1) {
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            } else if (g.act == 2) {
                v.x = 1.0f / (1.0f + expf(-v.x)); v.y = 1.0f / (1.0f + expf(-v.y));
                v.z = 1.0f / (1.0f + expf(-v.z)); v.w = 1.0f / (1.0f + expf(-v.w));
            }
        }
        *reinterpret_cast<float4*>(C + (row0 + 16 * i) * g.ldc + col0 + 16 * j) = v;
    };

    // prologue: tiles 0 .. NS-2 in flight, wait for tile 0
    const int npre = nt < NS - 1 ? nt : NS - 1;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < npre) {
#pragma unroll
            for (int p = 0; p < NPW; ++p) dma_piece(p, s);
        }
    if (npre >= 3) g3_vmwait<2 * NPW>();
    else if (npre == 2) g3_vmwait<NPW>();
    else g3_vmwait<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < NRD; ++q) rd(lds + fa0, lds + fa1, lds + fb0, lds + fb1, 0, 0, q);
    G3_STAMP_AT(1)

    // One K tile other than the last.  `issue` (wave-uniform): the DMA of tile t+NS-1 goes out in this tile's first group.
    // `rem`: whole tiles that may stay in flight behind tile t+1 at the wait.  ONE body for the steady state and the ring's
    // run-out (run-time flags, scalar branches): every further copy of this code gets its own register assignment, and the
    // compiler then moves the accumulators between the copies with VALU instructions it places right behind the last MFMA --
    // it cannot know the latency of an asm MFMA (the first version with one body per case returned wrong sums for the blocks
    // whose last MFMAs were still in flight).  Where a second copy is needed (tile nt-2, the last tile) g3_mfma_drain() sits
    // in front; tools/mfma_hazard_check.py checks the built code object for such reads.
    auto tile = [&](auto edge_c, bool issue, int rem, int cur, int nxt) {
        constexpr bool EDGE = decltype(edge_c)::value;
        const float* st = lds + cur * STAGE;
        const float* st1 = lds + (cur + 1 == NS ? 0 : cur + 1) * STAGE;
        const float *pa0 = st + fa0, *pa1 = st + fa1, *pb0 = st + fb0, *pb1 = st + fb1;
        const float *qa0 = st1 + fa0, *qa1 = st1 + fa1, *qb0 = st1 + fb0, *qb1 = st1 + fb1;
        // ---- group 0: MFMAs on fragment buffer 0; the reads of group 1 and the DMA pieces in their shadows
#pragma unroll
        for (int m = 0; m < MG; ++m) {
            const int c = m / (IM * JN), blk = m % (IM * JN), i = blk / JN, j = blk % JN;
            if (EDGE && m < IM * JN) g3_mfma_first(acc[i][j], bfrag(0, j, c), afrag(0, i, c));
            else g3_mfma(acc[i][j], bfrag(0, j, c), afrag(0, i, c));
            if (m >= MG - IM) colsum_add(0, m - (MG - IM));
            if (m < NRD) rd(pa0, pa1, pb0, pb1, 1, 1, m);
            if (!(G3_ABL & 1) && m >= NRD && (m - NRD) % PSTEP == 0 && (m - NRD) / PSTEP < NPW) {
                if (issue) dma_piece((m - NRD) / PSTEP, nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- group 1: MFMAs on fragment buffer 1; in front of its last NRD + 16 MFMAs the wait + barrier that publish tile
        //      t+1, then the reads of tile t+1's first group (fragment buffer 0 is free)
        constexpr int MB = MG - NRD - TAILC;
#pragma unroll
        for (int m = 0; m < MG; ++m) {
            const int c = m / (IM * JN), blk = m % (IM * JN), i = blk / JN, j = blk % JN;
            if (m == MB) {
                // raw barrier: behind __syncthreads() hipcc waits vmcnt(0) for its own outstanding loads (the mask prefetch)
                if (!(G3_ABL & 4)) {
                    if (NS >= 4 && rem >= 2) g3_vmwait<2 * NPW>();
                    else if (rem >= 1) g3_vmwait<NPW>();
                    else g3_vmwait<0>();
                }
                if (!(G3_ABL & 2)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if (EDGE && m == MG - 1) g3_mfma_last(acc[i][j], bfrag(1, j, c), afrag(1, i, c));
            else g3_mfma(acc[i][j], bfrag(1, j, c), afrag(1, i, c));
            if (m < IM) colsum_add(1, m);
            if (m >= MB && m - MB < NRD) rd(qa0, qa1, qb0, qb1, 0, 0, m - MB);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // The last K tile.  PROG (a tile inside the matrix): the second group runs block pair by block pair and every finished
    // pair is stored under the next pair's MFMAs.
    auto last_tile = [&](auto prog_c, int cur) {
        constexpr bool PROG = decltype(prog_c)::value;
        const float* st = lds + cur * STAGE;
        const float *pa0 = st + fa0, *pa1 = st + fa1, *pb0 = st + fb0, *pb1 = st + fb1;
#pragma unroll
        for (int m = 0; m < MG; ++m) {
            const int c = m / (IM * JN), blk = m % (IM * JN), i = blk / JN, j = blk % JN;
            if (m < IM * JN) g3_mfma_first(acc[i][j], bfrag(0, j, c), afrag(0, i, c));
            else if (PROG && m == MG - 1) g3_mfma_last(acc[i][j], bfrag(0, j, c), afrag(0, i, c));
            else g3_mfma(acc[i][j], bfrag(0, j, c), afrag(0, i, c));
            if (m >= MG - IM) colsum_add(0, m - (MG - IM));
            if (m < NRD) rd(pa0, pa1, pb0, pb1, 1, 1, m);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PROG) {
            // the BUILTIN MFMA from here on: the epilogue's VALU reads the accumulators between these MFMAs, and only for the
            // builtin does the compiler know the latency (behind the asm form it placed accumulator copies 8 wait states
            // after the MFMA that writes them: wrong sums).  The drain covers the change of form.
            constexpr int NB = IM * JN;
#pragma unroll
            for (int p = 0; p < NB / 2; ++p) {
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const int c = s8 >> 1, blk = 2 * p + (s8 & 1), i = blk / JN, j = blk % JN;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfrag(1, j, c), afrag(1, i, c), acc[i][j], 0, 0, 0);
                    if (p == 0 && s8 < IM) colsum_add(1, s8);
                    if (p > 0 && s8 == 2) epi_block((2 * p - 2) / JN, (2 * p - 2) % JN);
                    if (p > 0 && s8 == 5) epi_block((2 * p - 1) / JN, (2 * p - 1) % JN);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            epi_block((NB - 2) / JN, (NB - 2) % JN);
            epi_block((NB - 1) / JN, (NB - 1) % JN);
        } else {
#pragma unroll
            for (int m = 0; m < MG; ++m) {
                const int c = m / (IM * JN), blk = m % (IM * JN), i = blk / JN, j = blk % JN;
                if (m == MG - 1) g3_mfma_last(acc[i][j], bfrag(1, j, c), afrag(1, i, c));
                else g3_mfma(acc[i][j], bfrag(1, j, c), afrag(1, i, c));
                if (m < IM) colsum_add(1, m);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    int cur = 0;                                        // stage of tile t
    int nxt = NS - 1;                                   // stage the DMA of tile t+NS-1 goes to (= stage of tile t-1)
    for (int t = 0; t + 2 < nt; ++t) {                  // tiles 0 .. nt-3
        const int rem = nt - 2 - t;
        tile(std::integral_constant<bool, false>{}, t + NS - 1 < nt, rem, cur, nxt);
        nxt = cur;
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
    g3_mfma_drain();                                    // (the loop's last MFMAs: a copy of the accumulators may follow)
    prefetch_mask();                                    // one K tile ahead of the last one
    if (nt >= 2) {
        tile(std::integral_constant<bool, true>{}, false, 0, cur, nxt);       // tile nt-2
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
    if (fast) last_tile(std::integral_constant<bool, true>{}, cur);
    else last_tile(std::integral_constant<bool, false>{}, cur);
    G3_STAMP_AT(2)
    if (!A_KC && g.colsum != nullptr && bx == 0 && wn == 0) {
        // the four lane quarters of a wave hold four quarters of the contraction range: add them in a fixed order
#pragma unroll
        for (int i = 0; i < IM; ++i) {
            const float q1 = __shfl(csum[i], lr + 16, 64), q2 = __shfl(csum[i], lr + 32, 64), q3 = __shfl(csum[i], lr + 48, 64);
            const float q0 = __shfl(csum[i], lr, 64);
            const float sum = (q0 + q1) + (q2 + q3);
            const int64_t row = m0 + wm * 16 * IM + 16 * i + lr;
            if (kq == 0 && row < g.M) g.colsum[(int64_t)bz * g.M + row] = sum;
        }
    }
    if (!fast) {
        // edge tiles and unaligned operands: the generic form behind the loop (same values)
#pragma unroll
        for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                const int64_t row = row0 + 16 * i, col = col0 + 16 * j;
                if (row >= g.M || col >= g.N) continue;          // N % 4 == 0: four columns are in or out together
                float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                if (g.bias) {
                    v.x += g.bias[col]; v.y += g.bias[col + 1]; v.z += g.bias[col + 2]; v.w += g.bias[col + 3];
                }
                if (g.act == 1) {
                    v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                    v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                } else if (g.act == 2) {
                    v.x = 1.0f / (1.0f + expf(-v.x)); v.y = 1.0f / (1.0f + expf(-v.y));
                    v.z = 1.0f / (1.0f + expf(-v.z)); v.w = 1.0f / (1.0f + expf(-v.w));
                }
                if (g.mask_act) {
                    const float* mp = g.mask + row * g.ldmask + col;
                    const float4 x = make_float4(mp[0], mp[1], mp[2], mp[3]);
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
    G3_STAMP_AT(3)
#ifdef G3_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G3_STAMP_AT(4)
#endif
}

// (Round 6, built, measured and removed -- tools/gemm3_bench.hip p, profiles/r06_gemm3_vs_gemm2.txt: a PERSISTENT form for batches
//  with several tiles per CU (c5: 8 per CU at M = 65536) -- one workgroup per CU walking its tiles, the K-tile ring running on
//  across output tiles, the next tile's first fragments read behind a barrier in front of the progressive epilogue, the
//  accumulators restarted by an init-form MFMA (SrcC = 0) instead of VALU moves; bit-identical, correct on every shape tried
//  (K = 64 ... 512, N = 128 ... 512, ragged workgroup counts) -- 281.0 / 306.5 us against 285.2 / 301.3 for this kernel at
//  65536 x 512 x 512 forward / dgrad: no gain.  A tile takes ~35 us either way (81 k cycles at 2.3 GHz where prologue + loop +
//  tail of ONE workgroup are 77 k at M = 8192): at this size the part is clock-limited under the MFMA stream, and what the
//  vendor library's kernel has is less LDS traffic per MFMA (128x128 accumulators per wave).  A 256x128 tile (128x64 per wave)
//  of THIS kernel does not fit 256 VGPRs beside the asm MFMAs' "+v" accumulators (scratch spills in the MFMA stream).)

// what k_gemm3 can take: A contraction-contiguous, 16-byte loadable rows, every split's contraction range a multiple of 32
template <bool A_KC, bool B_KC>
static inline bool gemm3_applies(const GemmArgs& g) {
    if (!g.vecA || !g.vecB) return false;
    if (!A_KC && g.M < 4) return false;
    const int64_t kc = g.kchunk < g.K ? g.kchunk : g.K;
    if (g.K < G3_BK || g.K % G3_BK != 0 || kc % G3_BK != 0) return false;
    if (g.N < 4 || g.N % 4 != 0 || g.ldc % 4 != 0 || g.slab % 4 != 0 || (((uintptr_t)g.C) & 15) != 0) return false;
    if (A_KC && g.colsum != nullptr) return false;
    return true;
}

template <bool A_KC, bool B_KC, int IM, int JN, int NS>
static void launch_gemm3(const GemmArgs& g, int splits, hipStream_t s) {
    dim3 grid((unsigned)cdiv(g.N, 32 * JN), (unsigned)cdiv(g.M, 32 * IM), (unsigned)splits);
    CDLRM_LAUNCH_EV((k_gemm3<A_KC, B_KC, IM, JN, NS>), grid, dim3(256), 0, s, g);
}

// launch_gemm's hook (gemm_glds.h).  Taken where the 128x128 tiles fill whole rounds of one workgroup per CU (>= 90 % of the
// slots of the last round too: c3's 512-wide layers at M = 8192 are exactly 256 tiles, c5's 2048 and 1024) -- measured against
// k_gemm2 on one box (tools/gemm3_bench.hip, profiles/r06_gemm3_vs_gemm2.txt).
template <bool A_KC, bool B_KC>
static bool gemm3_try(const GemmArgs& g, int splits, hipStream_t s) {
    // Only for launches the caller marks as running ALONE (CDLRM_GEMM_ALONE: the top MLP's forward and its dgrad chain in the
    // training step).  Beside the weight-gradient GEMMs of the side queues a workgroup of this kernel (96 KB of LDS, 340
    // registers per lane) waits for a CU to drain: the bottom MLP's 512 <- 256 dgrad took 115.6 us there against 62.9 on
    // k_gemm2's 1024 small workgroups, the c3 step 0.5790 against 0.5580 ms (profiles/r06_ab_gemm3_in_step.txt).
    // cdlrm_debug_set(6, 32): never; (6, 256): every eligible launch (the stand-alone benches).
    if (g_cdlrm_debug[6] & 32) return false;
    if (!A_KC && (g_cdlrm_debug[6] & 512)) {
        // (A/B: the split-M weight gradients on this kernel)
    } else if (!A_KC && (g_cdlrm_debug[6] & 1024) && (int64_t)g.M * g.N >= 512 * 480) {
        // (A/B: the two 512-wide ones only)
    } else if (!g.alone && !(g_cdlrm_debug[6] & 256)) return false;
    const int64_t kc = g.kchunk < g.K ? g.kchunk : g.K;
    if ((A_KC && splits != 1) || kc < 2 * G3_BK || !gemm3_applies<A_KC, B_KC>(g)) return false;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // 128x128 tiles where they fill whole rounds of one workgroup per CU (>= 90 % of the last round's slots), else 64x128 tiles
    // under the same rule (M = 8192 x 256-wide layers, per-rank batches of 4096 x 512-wide: 256 tiles)
    auto fills = [&](int64_t tiles) { return tiles * 10 >= cdiv(tiles, n_cu) * n_cu * 9; };
    const int64_t t128 = cdiv(g.M, 128) * cdiv(g.N, 128) * splits, t64 = cdiv(g.M, 64) * cdiv(g.N, 128) * splits;
    // (short contractions, K <= 256, on k_gemm2's 64x64 tiles instead: c3 step 0.5594 against 0.5558 ms; on the 64x128 tiles: a tie)
    // (ragged tiles -- N = 480: a quarter of the tiles take the generic epilogue behind the loop -- only where every CU has ONE
    //  tile: at M = 65536 the 480 <- 512 dgrad took 333 us here against 304 on k_gemm2, at M = 8192 44.4 against 47.2)
    if ((g.N % 128 != 0 || g.M % 64 != 0) && t128 > n_cu) return false;
    const bool big = fills(t128);
    if (big) launch_gemm3<A_KC, B_KC, 4, 4, 3>(g, splits, s);
    else if (A_KC && fills(t64)) launch_gemm3<A_KC, B_KC, 2, 4, 3>(g, splits, s);
    else return false;
    return true;
}
