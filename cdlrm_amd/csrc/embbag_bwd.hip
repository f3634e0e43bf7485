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

#define SORT_CHUNK 8192          // most keys one workgroup sorts (64 KiB of LDS)
#define SORT_THREADS 1024
#define SEG_CH 32

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int lane_mask) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, lane_mask, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), lane_mask, 64);
    return ((uint64_t)hi << 32) | lo;
}

// Sort of one chunk of (slot, position) keys by ONE workgroup of 1024 threads, E keys per thread (chunk padded to
// 1024 * E with ~0 keys):
//   1. every wave sorts its 64 * E keys in REGISTERS -- a bitonic network whose exchanges at distance < E stay inside a
//      thread and at distance >= E are lane shuffles: no LDS, no barrier;
//   2. the 16 sorted runs meet in LDS and are merged pairwise by RANK: a key's place in the merged run is its place in
//      its own run plus the number of smaller keys in the sibling run (a binary search in LDS): log2(16) = 4 passes with
//      two barriers each.
// The all-LDS bitonic network this replaces took 91 compare-exchange stages with a 1024-thread barrier after each
// (91-113 us for 26 tables x 8192 lookups, profiles/r01_v7_c3_n1_kernel_stats.csv).
template <int E>
__global__ void __launch_bounds__(SORT_THREADS) k_sort_chunks(const int32_t* __restrict__ slots, int64_t n,
                                                              uint64_t* __restrict__ keys, int32_t* __restrict__ meta,
                                                              int npow2, int write_meta, int32_t* __restrict__ longcount,
                                                              uint8_t* __restrict__ once, int nb, int64_t ld_in,
                                                              int64_t batch_len, int nbt, int j0) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
    __shared__ int wmax[16];
    const int t = blockIdx.y;
    // output list of input list t: a SLICE of a window chunk (nb of its nbt batches, from batch j0 on) lands in the chunk's
    // [T, nbt] lists; a batch's own sort and a whole chunk: ol == t
    const int ol = (t / nb) * nbt + j0 + t % nb;
    // the long-run list of THIS work buffer starts empty: cleared here, by the first kernel of every prepare (an apply always
    // follows a prepare on the same buffer, in stream order), so no clearing launch sits on the queue (a 4-byte hipMemsetAsync
    // is a 5.4 us kernel of its own) and two backward passes in flight on different work buffers share nothing
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { longcount[0] = 0; longcount[2] = 0; }
    const int chunk = SORT_THREADS * E;       // == sort_chunk(n): whole chunks, a shorter last one
    const int64_t base = (int64_t)blockIdx.x * chunk;
    const int cnt = (int)min((int64_t)chunk, n - base);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WK = 64 * E;              // keys per wave
    constexpr int NT = SORT_THREADS * E;    // keys per workgroup (>= npow2)
    uint64_t k[E];
    int pos[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        // coalesced load: element e of lane l is chunk index wave*WK + e*64 + l (which key a thread starts with is irrelevant)
        const int i = wave * WK + e * 64 + lane;
        k[e] = ~0ull;
        if (i < cnt) {
            const int64_t p = base + i;
            // (window form, cdlrm_embbag_bwd_prepare_window: list t = batch t % nb of table t / nb, read out of the resolver's
            //  [T, ld_in] slot ids; one batch: nb = 1, ld_in = n)
            k[e] = ((uint64_t)(uint32_t)slots[(int64_t)(t / nb) * ld_in + (int64_t)(t % nb) * batch_len + p] << 32) | (uint64_t)p;
        }
    }
    // 1. wave-local bitonic sort; the key with wave-local index g = lane * E + e lives in register e of lane `lane`
#pragma unroll
    for (int size = 2; size <= WK; size <<= 1) {
#pragma unroll
        for (int j = size >> 1; j >= 1; j >>= 1) {
            if (j >= E) {
                const int lm = j / E;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const uint64_t o = shfl_xor_u64(k[e], lm);
                    const int g = lane * E + e;
                    const bool up = (g & size) == 0, lower = (g & j) == 0;
                    const uint64_t mn = k[e] < o ? k[e] : o, mx = k[e] < o ? o : k[e];
                    k[e] = (lower == up) ? mn : mx;
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int p = e ^ j;
                    if (p > e) {
                        const int g = lane * E + e;
                        const bool up = (g & size) == 0;
                        const uint64_t a = k[e], b = k[p];
                        const bool sw = (a > b) == up;
                        k[e] = sw ? b : a;
                        k[p] = sw ? a : b;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        pos[e] = wave * WK + lane * E + e;
        sk[pos[e]] = k[e];
    }
    __syncthreads();
    // 2. rank-merge the 16 runs pairwise; ties (only the ~0 pad keys repeat) go left run first.  Branch-free binary
    //    searches with a fixed step count, the E searches of a thread interleaved (E independent LDS reads in flight per
    //    step instead of one dependent read at a time)
    for (int run = WK; run < NT; run <<= 1) {
        int sib[E], cntl[E];
        bool left[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int s0 = pos[e] & ~(2 * run - 1);
            left[e] = (pos[e] & run) == 0;
            sib[e] = left[e] ? s0 + run : s0;       // first key of the sibling run
            cntl[e] = 0;                            // sibling keys that sort in front of k[e], so far
        }
        for (int st = run >> 1; st > 0; st >>= 1) {
            uint64_t v[E];
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = sk[sib[e] + cntl[e] + st - 1];
#pragma unroll
            for (int e = 0; e < E; ++e) cntl[e] += (left[e] ? (v[e] < k[e]) : (v[e] <= k[e])) ? st : 0;
        }
        {
            uint64_t v[E];
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = sk[sib[e] + cntl[e]];        // cntl <= run - 1 here: in range
#pragma unroll
            for (int e = 0; e < E; ++e) cntl[e] += (left[e] ? (v[e] < k[e]) : (v[e] <= k[e])) ? 1 : 0;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int s0 = pos[e] & ~(2 * run - 1);
            pos[e] = s0 + (pos[e] - (left[e] ? s0 : s0 + run)) + cntl[e];
        }
        __syncthreads();                    // every search of this pass has read the old layout
#pragma unroll
        for (int e = 0; e < E; ++e) sk[pos[e]] = k[e];
        __syncthreads();
    }
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) keys[(int64_t)ol * n + base + i] = sk[i];
    if (!write_meta) return;
    // run starts by an inclusive max-scan of head positions: thread owns E consecutive sorted keys
    const int i0 = threadIdx.x * E;
    int local = -1;     // last head position inside my range
    for (int e = 0; e < E; ++e) {
        const int i = i0 + e;
        if (i < cnt && (i == 0 || (uint32_t)(sk[i] >> 32) != (uint32_t)(sk[i - 1] >> 32))) local = i;
    }
    int inc = local;
    const int wid = wave;
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
        meta[(int64_t)ol * n + base + i] = i - run;
        // a slot this batch reads ONCE (a run of one lookup), flagged at the lookup's position in the batch
        const bool last = i + 1 >= cnt || (uint32_t)(sk[i + 1] >> 32) != (uint32_t)(sk[i] >> 32);
        once[(int64_t)ol * n + (uint32_t)sk[i]] = (run == i && last) ? 1 : 0;
    }
}

// merge sorted runs of length `run` pairwise by ranking (keys are unique: position is part of the key)
__global__ void __launch_bounds__(256) k_merge_pass(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                    int64_t n, int64_t run, int nb, int nbt, int j0) {
    const int t = blockIdx.y;
    const int ol = (t / nb) * nbt + j0 + t % nb;
    const uint64_t* a = in + (int64_t)ol * n;
    uint64_t* o = out + (int64_t)ol * n;
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
__global__ void __launch_bounds__(256) k_seg_meta(const uint64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ meta,
                                                  uint8_t* __restrict__ once, int nb, int nbt, int j0) {
    // A tile of 256 consecutive sorted positions per pass: the run start of a position is the LAST run start at or in front of
    // it -- an inclusive max-scan inside the tile (wave shuffles + one LDS hop) -- and only the run that reaches into the tile
    // from the left needs a search over the keys, ONE per tile instead of one per position (round 6: a tiny table's run is
    // thousands of positions, each of which walked 13 dependent loads; 12-31 us beside the step for 2 x 26 lists).
    __shared__ int wlast[4];
    __shared__ int64_t left_start;
    const int t = blockIdx.y;
    const int ol = (t / nb) * nbt + j0 + t % nb;
    const uint64_t* kt = keys + (int64_t)ol * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t p0 = (int64_t)blockIdx.x * 256; p0 < n; p0 += (int64_t)gridDim.x * 256) {
        const int64_t p = p0 + threadIdx.x;
        const bool in = p < n;
        const uint64_t key = in ? kt[p] : 0ull;
        const uint64_t s = key >> 32;
        const bool start = in && (p == 0 || (kt[p - 1] >> 32) != s);
        if (threadIdx.x == 0) left_start = -1;
        // last run start at or in front of this thread, inside the tile (-1: none)
        int v = start ? (int)threadIdx.x : -1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(v, d, 64);
            if (lane >= d) v = max(v, o);
        }
        if (lane == 63) wlast[wave] = v;
        __syncthreads();
        for (int w = 0; w < wave; ++w) v = max(v, wlast[w]);
        if (in && v < 0 && threadIdx.x == 0) left_start = lower_bound_u64(kt, 0, p0, s << 32);      // the run of the tile's first key
        __syncthreads();
        if (in) {
            const int64_t st = v >= 0 ? p0 + v : left_start;
            const int32_t r = (int32_t)(p - st);
            meta[(int64_t)ol * n + p] = r;
            // a slot this batch reads ONCE, flagged at the lookup's position in the batch (see k_sort_chunks)
            const bool last = p + 1 >= n || (kt[p + 1] >> 32) != s;
            once[(int64_t)ol * n + (uint32_t)key] = (r == 0 && last) ? 1 : 0;
        }
        __syncthreads();
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
                                                    int64_t aux_total, int32_t* __restrict__ runend, int64_t kstride,
                                                    int ways, int aux_add) {
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
    // (kstride: elements between two tables' sorted lists -- n, or nb * n inside a window's sorted chunk; aux_add: the sorted
    //  chunk holds phase-0 aux slots, the batch trains on aux region aux_add / aux)
    const uint64_t* kt = keys + (int64_t)t * kstride;
    const int32_t* mt = meta + (int64_t)t * kstride;
    const uint32_t aux0 = (uint32_t)(tab[t].P * ways);
    const float* g = grad + (int64_t)t * ld_table;
    const int64_t* off = ARANGE ? nullptr : offsets + (int64_t)t * ld_off;
    for (int64_t p = (int64_t)blockIdx.x * gpb + gid; p < n; p += (int64_t)gridDim.x * gpb) {
        const int r0 = mt[p];
        if (r0 % SEG_CH) continue;               // chunk interior: some other group owns this position
        const bool head = r0 == 0;
        uint32_t slot = (uint32_t)(kt[p] >> 32);
        if (slot >= aux0) slot += aux_add;
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
            // the LAST chunk of a long run knows where the run ends: leave it at the run's head for k_bwd_long (which
            // would otherwise find it by a 13-step binary search over the keys, one dependent global read per step)
            if (!single && !more) runend[(int64_t)t * n + (p - r0)] = (int32_t)(p + len);
        }
    }
}

// The same sums in the same order (bit-identical to k_bwd_chunks), laid out for memory-level parallelism.  k_bwd_chunks gives a
// lane group ONE sorted position per trip: meta -> look-ahead meta -> key -> gradient / row loads are four dependent global
// reads, and a group makes ~9 such trips at c3 -- the kernel is a chain of latencies (52 us alone for 166 MB of traffic).
// Here a lane group owns a BLOCK of SEG_CH consecutive sorted positions: one coalesced read brings the keys and run distances of
// the block and of the SEG_CH positions behind it (a chunk that starts in the block ends at most there), run starts and chunk
// heads become two bit masks (a ballot), and the heads of the block are worked FOUR AT A TIME with four gradient rows each in
// flight (16 rows when the block holds one head: the inside of a long run).  Two dependent reads per block instead of four per
// position.
template <int LPR, bool ARANGE, int HU = 4, int RA = 16>
__global__ void __launch_bounds__(256) k_bwd_blocks(const TableDesc* __restrict__ tab, int D4,
                                                    float4* __restrict__ weight, const uint64_t* __restrict__ keys,
                                                    const int32_t* __restrict__ meta,
                                                    const int64_t* __restrict__ offsets, int64_t n, int64_t n_bags,
                                                    int64_t ld_off, const float* __restrict__ grad, int64_t ld_bag,
                                                    int64_t ld_table, float lr, float4* __restrict__ partials,
                                                    int64_t pstride, int64_t* __restrict__ longlist,
                                                    int32_t* __restrict__ longcount, uint8_t* __restrict__ touched,
                                                    int64_t aux_total, int32_t* __restrict__ runend, int skip_once,
                                                    int64_t kstride, int ways, int aux_add) {
    constexpr int GPB = 256 / LPR;
    constexpr int SPAN = 2 * SEG_CH;                // positions a block's chunks can reach
    constexpr int NL = SPAN / LPR;                  // span positions per lane
    static_assert(SEG_CH == 32 && SPAN % LPR == 0, "masks below are 32 / 64 bits wide");
    __shared__ uint32_t s_pos[GPB][SPAN];           // low half of the key: the lookup's position in the batch
    __shared__ uint32_t s_slot[GPB][SEG_CH];
    __shared__ int32_t s_r0[GPB][SEG_CH];
    const int t = blockIdx.y;
    const int64_t row_base = tab[t].row_base;
    const uint32_t first_aux = (uint32_t)(tab[t].rows - aux_total);
    const int c = threadIdx.x % LPR;
    const int gid = threadIdx.x / LPR;
    const int gshift = ((threadIdx.x & 63) / LPR) * LPR;
    const uint64_t* kt = keys + (int64_t)t * kstride;
    const int32_t* mt = meta + (int64_t)t * kstride;
    const uint32_t aux0 = (uint32_t)(tab[t].P * ways);
    const float* g = grad + (int64_t)t * ld_table;
    const int64_t* off = ARANGE ? nullptr : offsets + (int64_t)t * ld_off;
    const int64_t nblk = (n + SEG_CH - 1) / SEG_CH;
    for (int64_t blk = (int64_t)blockIdx.x * GPB + gid; blk < nblk; blk += (int64_t)gridDim.x * GPB) {
        const int64_t p0 = blk * SEG_CH;
        uint64_t S = 0;                              // bit j: position p0 + j starts a run (or lies past the end)
        uint32_t H = 0;                              // bit j < SEG_CH: position p0 + j is a chunk head
        __builtin_amdgcn_wave_barrier();             // the previous block's LDS reads are issued before these writes
#pragma unroll
        for (int m = 0; m < NL; ++m) {
            const int j = c + m * LPR;
            const int64_t q = p0 + j;
            const bool in = q < n;
            const uint64_t key = in ? kt[q] : 0ull;
            const int32_t r = in ? mt[q] : 0;
            s_pos[gid][j] = (uint32_t)key;
            if (j < SEG_CH) {
                uint32_t slot = (uint32_t)(key >> 32);
                if (slot >= aux0) slot += aux_add;
                s_slot[gid][j] = slot;
                s_r0[gid][j] = r;
            }
            const unsigned long long b = __ballot(r == 0);
            const unsigned long long bits = LPR == 64 ? b : ((b >> gshift) & ((1ull << LPR) - 1));
            S |= bits << (m * LPR);
            if (m * LPR < SEG_CH) {
                const unsigned long long hb = __ballot(in && j < SEG_CH && (r % SEG_CH) == 0);
                const unsigned long long hbits = LPR == 64 ? hb : ((hb >> gshift) & ((1ull << LPR) - 1));
                H |= (uint32_t)(hbits << (m * LPR));
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (skip_once) {
            // runs of ONE lookup were updated by the interaction backward itself (cdlrm_gather_interact_bwd_sgd): a start
            // followed by a start.  Only their touched flags are left to set.
            const uint32_t one = (uint32_t)(S & (S >> 1)) & H;
            H &= ~one;
            if (touched) {
#pragma unroll
                for (int m = 0; m * LPR < SEG_CH; ++m) {
                    const int j = c + m * LPR;
                    if (j < SEG_CH && ((one >> j) & 1)) {
                        const uint32_t slot = s_slot[gid][j];
                        if (slot < first_aux) touched[row_base + slot] = 1;
                    }
                }
            }
        }
        const int64_t pbucket = (int64_t)t * pstride + 2 * blk;
        while (H) {
            const bool alone = (H & (H - 1)) == 0;
            if (alone) {
                // one head: 16 gradient rows in flight
                const int j = __ffs((int)H) - 1;
                H = 0;
                const uint64_t rest = S >> (j + 1);
                const int nxt = rest ? (__ffsll((long long)rest) - 1) : 64;
                const bool more = nxt >= SEG_CH;
                const int len = more ? SEG_CH : nxt + 1;
                const bool head = (S >> j) & 1;
                const bool single = head && !more;
                const uint32_t slot = s_slot[gid][j];
                for (int cc = c; cc < D4; cc += LPR) {
                    float4 w;
                    if (single) w = weight[(row_base + slot) * D4 + cc];
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int q = 0; q < len; q += RA) {
                        float4 v[RA];
#pragma unroll
                        for (int u = 0; u < RA; ++u)
                            if (q + u < len) {
                                const int64_t pos = s_pos[gid][j + q + u];
                                const int64_t bag = ARANGE ? pos : bag_of(off, n_bags, pos);
                                v[u] = *reinterpret_cast<const float4*>(g + bag * ld_bag + cc * 4);
                            }
#pragma unroll
                        for (int u = 0; u < RA; ++u)
                            if (q + u < len) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                    }
                    if (single) {
                        w.x = fmaf(-lr, acc.x, w.x); w.y = fmaf(-lr, acc.y, w.y);
                        w.z = fmaf(-lr, acc.z, w.z); w.w = fmaf(-lr, acc.w, w.w);
                        weight[(row_base + slot) * D4 + cc] = w;
                    } else {
                        partials[(pbucket + (head ? 1 : 0)) * D4 + cc] = acc;
                    }
                }
                if (c == 0) {
                    const int64_t p = p0 + j;
                    if (single) {
                        if (touched && slot < first_aux) touched[row_base + slot] = 1;
                    } else if (head) {
                        const int li = atomicAdd(longcount, 1);
                        longlist[li] = ((int64_t)t << 40) | p;
                    }
                    if (!single && !more) runend[(int64_t)t * n + (p - s_r0[gid][j])] = (int32_t)(p + len);
                }
                continue;
            }
            // up to four heads, four gradient rows of each in flight
            int hj[HU], hlen[HU];
            bool hhead[HU], hsingle[HU], hmore[HU];
            uint32_t hslot[HU];
            int maxlen = 0;
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                hlen[u] = 0; hj[u] = 0; hhead[u] = false; hsingle[u] = false; hmore[u] = false; hslot[u] = 0;
                if (H) {
                    const int j = __ffs((int)H) - 1;
                    H &= H - 1;
                    const uint64_t rest = S >> (j + 1);
                    const int nxt = rest ? (__ffsll((long long)rest) - 1) : 64;
                    hmore[u] = nxt >= SEG_CH;
                    hlen[u] = hmore[u] ? SEG_CH : nxt + 1;
                    hhead[u] = (S >> j) & 1;
                    hsingle[u] = hhead[u] && !hmore[u];
                    hj[u] = j;
                    hslot[u] = s_slot[gid][j];
                    maxlen = max(maxlen, hlen[u]);
                }
            }
            for (int cc = c; cc < D4; cc += LPR) {
                float4 w[HU], acc[HU];
#pragma unroll
                for (int u = 0; u < HU; ++u) {
                    acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (hsingle[u]) w[u] = weight[(row_base + hslot[u]) * D4 + cc];
                }
                for (int q = 0; q < maxlen; q += 4) {
                    float4 v[HU][4];
#pragma unroll
                    for (int u = 0; u < HU; ++u)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (q + k < hlen[u]) {
                                const int64_t pos = s_pos[gid][hj[u] + q + k];
                                const int64_t bag = ARANGE ? pos : bag_of(off, n_bags, pos);
                                v[u][k] = *reinterpret_cast<const float4*>(g + bag * ld_bag + cc * 4);
                            }
#pragma unroll
                    for (int u = 0; u < HU; ++u)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (q + k < hlen[u]) {
                                acc[u].x += v[u][k].x; acc[u].y += v[u][k].y; acc[u].z += v[u][k].z; acc[u].w += v[u][k].w;
                            }
                }
#pragma unroll
                for (int u = 0; u < HU; ++u) {
                    if (hlen[u] == 0) continue;
                    if (hsingle[u]) {
                        w[u].x = fmaf(-lr, acc[u].x, w[u].x); w[u].y = fmaf(-lr, acc[u].y, w[u].y);
                        w[u].z = fmaf(-lr, acc[u].z, w[u].z); w[u].w = fmaf(-lr, acc[u].w, w[u].w);
                        weight[(row_base + hslot[u]) * D4 + cc] = w[u];
                    } else {
                        partials[(pbucket + (hhead[u] ? 1 : 0)) * D4 + cc] = acc[u];
                    }
                }
            }
            if (c == 0) {
#pragma unroll
                for (int u = 0; u < HU; ++u) {
                    if (hlen[u] == 0) continue;
                    const int64_t p = p0 + hj[u];
                    if (hsingle[u]) {
                        if (touched && hslot[u] < first_aux) touched[row_base + hslot[u]] = 1;
                    } else if (hhead[u]) {
                        const int li = atomicAdd(longcount, 1);
                        longlist[li] = ((int64_t)t << 40) | p;
                    }
                    if (!hsingle[u] && !hmore[u]) runend[(int64_t)t * n + (p - s_r0[gid][hj[u]])] = (int32_t)(p + hlen[u]);
                }
            }
        }
    }
}

template <int LPR>
__global__ void __launch_bounds__(256) k_bwd_long(const TableDesc* __restrict__ tab, int D4,
                                                  float4* __restrict__ weight, const uint64_t* __restrict__ keys,
                                                  int64_t n, float lr, const float4* __restrict__ partials,
                                                  int64_t pstride, const int64_t* __restrict__ longlist,
                                                  int32_t* __restrict__ longcount, uint8_t* __restrict__ touched,
                                                  int64_t aux_total, const int32_t* __restrict__ runend, int64_t kstride,
                                                  int ways, int aux_add) {
    const int c = threadIdx.x % LPR;
    const int gpb = blockDim.x / LPR;
    const int gid = threadIdx.x / LPR;
    const int cnt = *longcount;
    // The list counter lives in the work buffer: cleared by the first kernel of every prepare (k_sort_chunks) AND left zero by
    // the last workgroup of this kernel to finish -- every workgroup has read it by then --, so a second apply on one prepare
    // (a benchmark loop) starts from an empty list too.  longcount[2] counts the workgroups that are through.
    __syncthreads();
    if (threadIdx.x == 0) {
        const int through = atomicAdd(longcount + 2, 1);
        if (through == (int)gridDim.x - 1) {
            atomicExch(longcount, 0);
            atomicExch(longcount + 2, 0);
        }
    }
    for (int li = blockIdx.x * gpb + gid; li < cnt; li += gridDim.x * gpb) {
        const int64_t e = longlist[li];
        const int t = (int)(e >> 40);
        const int64_t p0 = e & (((int64_t)1 << 40) - 1);
        const uint64_t* kt = keys + (int64_t)t * kstride;
        uint32_t slot = (uint32_t)(kt[p0] >> 32);
        if (slot >= (uint32_t)(tab[t].P * ways)) slot += aux_add;
        const int64_t end = runend[(int64_t)t * n + p0];
        const int64_t row = tab[t].row_base + slot;
        for (int cc = c; cc < D4; cc += LPR) {
            float4 w = weight[row * D4 + cc];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            // chunk partials in chunk order (a fixed order: reproducible), sixteen then four loads in flight (a tiny table's run is
            // thousands of lookups = a hundred partials behind ONE lane group: at four in flight the kernel was its latency chain)
            int64_t p = p0;
            for (; p + 15 * SEG_CH < end; p += 16 * SEG_CH) {
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int64_t q = p + u * SEG_CH;
                    const int64_t pi = 2 * (q / SEG_CH) + (q == p0 ? 1 : 0);
                    v[u] = partials[((int64_t)t * pstride + pi) * D4 + cc];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; p + 3 * SEG_CH < end; p += 4 * SEG_CH) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t q = p + u * SEG_CH;
                    const int64_t pi = 2 * (q / SEG_CH) + (q == p0 ? 1 : 0);
                    v[u] = partials[((int64_t)t * pstride + pi) * D4 + cc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; p < end; p += SEG_CH) {
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
//              longlist [T*(n/SEG_CH+2)] i64 | long-run counter | runend [T*n] i32 (end of a long run, at its head) |
//              once [T*n] u8 (by position in the batch: the lookup's slot occurs once in the batch)
static int64_t bwd_pstride(int64_t n) { return 2 * (cdiv(n, SEG_CH) + 1); }
static uint64_t align256(uint64_t v) { return (v + 255) & ~(uint64_t)255; }

struct BwdWork {
    uint64_t *keysA, *keysB;
    int32_t* meta;
    float4* partials;
    int64_t* longlist;
    int32_t* longcount;
    int32_t* runend;
    uint8_t* once;          // [T*n] by position in the batch: this lookup's slot occurs once in the batch
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
    w.longcount = (int32_t*)wp; wp += 256;
    w.runend = (int32_t*)wp; wp += align256((uint64_t)T * n * 4);
    w.once = (uint8_t*)wp;
    return w;
}

extern "C" uint64_t cdlrm_embbag_bwd_work_bytes(int32_t T, int64_t n, int32_t dim) {
    return 2 * align256((uint64_t)T * n * 8) + align256((uint64_t)T * n * 4) +
           align256((uint64_t)T * bwd_pstride(n) * dim * 4) + align256((uint64_t)T * (n / SEG_CH + 2) * 8) + 256 +
           align256((uint64_t)T * n * 4) + align256((uint64_t)T * n);
}

// Keys per sorting workgroup.  The in-LDS sort is bound by vector-instruction issue on ONE CU (measured: 9 / 16 / 33 / 83 us
// for 1024 / 2048 / 4096 / 8192 keys per workgroup, tools/sort_scale.py), so mid-sized inputs are cut into 2048-key
// chunks that sort on different CUs and meet in global rank-merge passes (k_merge_pass, ~8 us each at 26 x 8192 keys);
// very long inputs (config c5: 65536 lookups per table) keep 8192-key chunks: there the merge passes dominate.
static int64_t sort_chunk(int64_t n) {
    if (n <= 2048) return n <= 1024 ? 1024 : 2048;
    return n <= 16384 ? 2048 : SORT_CHUNK;
}

// number of rank-merge passes decides which key buffer ends up sorted
static bool sorted_in_B(int64_t n) {
    int passes = 0;
    for (int64_t run = sort_chunk(n); run < n; run *= 2) ++passes;
    return passes & 1;
}

// the sort of `lists` lists of n slot ids each (a batch: one list per table; a window chunk: one per table and batch)
// (Measured for the look-ahead slices, which have a whole step of slack, and not taken: ONE workgroup per list of 8192 keys -- 83
//  us on 52 CUs, no merge passes, one launch instead of four -- 0.5401 against 0.5363 ms per c3 step; a tie at a batch of 2048.)
static int bwd_sort(int lists, int64_t n, const int32_t* slots, int nb, int64_t ld_in, int64_t batch_len, int nbt, int j0,
                    uint64_t* keysA, uint64_t* keysB, int32_t* meta, uint8_t* once, int32_t* longcount, hipStream_t s) {
    const int64_t chunk = sort_chunk(n);
    const int64_t nchunks = cdiv(n, chunk);
    const int npow2 = (int)chunk;
    static bool attr_set = false;
    if (!attr_set) {
        CDLRM_HIP_CHECK(hipFuncSetAttribute((const void*)k_sort_chunks<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            SORT_CHUNK * 8));
        attr_set = true;
    }
    // keys per thread: the chunk spread over the 1024 threads
    const int E = (int)(chunk / SORT_THREADS);
    const dim3 sgrid((unsigned)nchunks, (unsigned)lists);
    const size_t slds = (size_t)SORT_THREADS * E * 8;
    const int wm = nchunks == 1 ? 1 : 0;
#define SORT_CALL(E_) hipLaunchKernelGGL(k_sort_chunks<E_>, sgrid, dim3(SORT_THREADS), slds, s, slots, n, keysA, meta, npow2, wm, longcount, once, nb, ld_in, batch_len, nbt, j0)
    if (E == 1) SORT_CALL(1);
    else if (E == 2) SORT_CALL(2);
    else if (E == 4) SORT_CALL(4);
    else SORT_CALL(8);
#undef SORT_CALL
    uint64_t* cur = keysA;
    uint64_t* alt = keysB;
    for (int64_t run = chunk; run < n; run *= 2) {
        int64_t gx = cdiv(n, 256);
        if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(k_merge_pass, dim3((unsigned)gx, (unsigned)lists), dim3(256), 0, s, cur, alt, n, run, nb, nbt, j0);
        uint64_t* tmp = cur; cur = alt; alt = tmp;
    }
    if (nchunks > 1) {
        int64_t gx = cdiv(n, 256);
        if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(k_seg_meta, dim3((unsigned)gx, (unsigned)lists), dim3(256), 0, s, cur, n, meta, once, nb, nbt, j0);
    }
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_embbag_bwd_prepare(cdlrm_ctx* ctx, const int32_t* slots, int64_t n, void* work, void* stream) {
    CDLRM_REQUIRE(ctx && slots && work, "null argument");
    CDLRM_REQUIRE(((uintptr_t)work & 255) == 0, "work must be 256-byte aligned");
    CDLRM_REQUIRE(n < ((int64_t)1 << 31) && ctx->T < (1 << 20), "n < 2^31");
    if (n == 0) return 0;
#ifdef CDLRM_DEV
    if (g_cdlrm_debug[6] & 2) return 0;     // development build (tools/ab_step.py --attr debug:6): what the step costs WITHOUT the slot sort
#endif
    BwdWork w = carve(work, ctx->T, n, ctx->D);
    return bwd_sort(ctx->T, n, slots, 1, n, 0, 1, 0, w.keysA, w.keysB, w.meta, w.once, w.longcount, (hipStream_t)stream);
}

// ---- a window chunk's batches sorted at once (the look-ahead resolver's slot ids: cdlrm_window_resolve) ----------------------
// sorted layout: keys A [T*nb*n] u64 | keys B | meta [T*nb*n] i32 | once [T*nb*n] u8 | 256 B; list t * nb + j = table t, batch j
struct SortedWin {
    uint64_t *keysA, *keysB;
    int32_t* meta;
    uint8_t* once;
    int32_t* pad;
};
static SortedWin carve_sorted(void* sorted, int T, int nb, int64_t n) {
    SortedWin w;
    char* wp = (char*)sorted;
    const uint64_t e = (uint64_t)T * nb * n;
    w.keysA = (uint64_t*)wp; wp += align256(e * 8);
    w.keysB = (uint64_t*)wp; wp += align256(e * 8);
    w.meta = (int32_t*)wp; wp += align256(e * 4);
    w.once = (uint8_t*)wp; wp += align256(e);
    w.pad = (int32_t*)wp;
    return w;
}

extern "C" uint64_t cdlrm_embbag_bwd_sorted_bytes(int32_t num_tables, int32_t nb, int64_t n) {
    const uint64_t e = (uint64_t)num_tables * nb * n;
    return 2 * align256(e * 8) + align256(e * 4) + align256(e) + 256;
}

extern "C" int cdlrm_embbag_bwd_prepare_window(cdlrm_ctx* ctx, const int32_t* wslots, int64_t ld_w, int64_t batch_len,
                                               int32_t nb, int64_t n, int32_t j0, int32_t count, void* sorted, void* stream) {
    CDLRM_REQUIRE(ctx && wslots && sorted, "null argument");
    CDLRM_REQUIRE(((uintptr_t)sorted & 255) == 0, "sorted must be 256-byte aligned");
    CDLRM_REQUIRE(nb >= 1 && n >= 1 && n < ((int64_t)1 << 31) && (int64_t)ctx->T * nb < 65536, "1 <= nb, T * nb < 65536, n < 2^31");
    CDLRM_REQUIRE(j0 >= 0 && count >= 1 && j0 + count <= nb, "0 <= j0, 1 <= count, j0 + count <= nb");
    CDLRM_REQUIRE(batch_len >= n && ld_w >= (int64_t)(nb - 1) * batch_len + n, "a batch's n slot ids lie inside its batch_len columns");
    SortedWin w = carve_sorted(sorted, ctx->T, nb, n);
    return bwd_sort(ctx->T * count, n, wslots + (int64_t)j0 * batch_len, count, ld_w, batch_len, nb, j0, w.keysA, w.keysB, w.meta,
                    w.once, w.pad, (hipStream_t)stream);
}

extern "C" int cdlrm_embbag_bwd_sorted_views(cdlrm_ctx* ctx, void* sorted, int32_t nb, int64_t n, int32_t j,
                                             const uint64_t** keys, const int32_t** meta, const uint8_t** once) {
    CDLRM_REQUIRE(ctx && sorted && keys && meta && once && nb >= 1 && j >= 0 && j < nb && n >= 1, "bad argument");
    SortedWin w = carve_sorted(sorted, ctx->T, nb, n);
    *keys = (sorted_in_B(n) ? w.keysB : w.keysA) + (int64_t)j * n;
    *meta = w.meta + (int64_t)j * n;
    *once = w.once + (int64_t)j * n;
    return 0;
}

extern "C" int cdlrm_embbag_bwd_once_flags(cdlrm_ctx* ctx, void* work, int64_t n, const uint8_t** once) {
    CDLRM_REQUIRE(ctx && work && once && n >= 1, "bad argument");
    *once = carve(work, ctx->T, n, ctx->D).once;
    return 0;
}

// sums + row updates over sorted lists: `cur` / `meta` are table 0's list, table t's lies kstride elements further (a batch's own
// sort: n; a window chunk's: nb * n); aux_phase: the sorted slots are phase-0 aux slots (0 for a batch's own sort: k_take has
// added the phase); scratch: partials, long-run list and run ends of `work`
static int cdlrm_embbag_bwd_apply_core(cdlrm_ctx* ctx, const int64_t* offsets, int64_t n, int64_t n_bags, int64_t ld_off,
                     const float* grad, int64_t ld_bag, int64_t ld_table, float lr, void* work, const uint64_t* cur,
                     const int32_t* meta, int64_t kstride, int aux_phase, uint8_t* touched, void* stream, int skip_once) {
    CDLRM_REQUIRE(ctx && grad && work, "null argument");
    CDLRM_REQUIRE(ctx->weight, "cdlrm_ctx_bind_cache first");
    CDLRM_REQUIRE(((uintptr_t)grad & 15) == 0 && ld_bag % 4 == 0 && ld_table % 4 == 0 && ((uintptr_t)work & 255) == 0,
                  "aligned grad rows / work");
    CDLRM_REQUIRE(offsets != nullptr || n_bags == n, "Criteo layout needs n_bags == n");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return 0;
#ifdef CDLRM_DEV
    if (g_cdlrm_debug[6] & 1) return 0;     // development build: ... without the embedding update (an upper bound on what moving it buys)
#endif
    const int T = ctx->T, D4 = ctx->D / 4;
    const int lpr = lanes_per_row_b(D4);
    const int gpb = 256 / lpr;
    BwdWork w = carve(work, T, n, ctx->D);
    if (!cur) {
        cur = sorted_in_B(n) ? w.keysB : w.keysA;
        meta = w.meta;
        kstride = n;
    }
    const int ways = ctx->ways, aux_add = aux_phase * ctx->aux;
    float4* wt = reinterpret_cast<float4*>(ctx->weight);
    const int64_t aux_total = (int64_t)ctx->aux * ctx->aux_phases;
    if (skip_once || (g_cdlrm_debug[6] & 64)) {
        // a lane group per block of SEG_CH sorted positions (k_bwd_blocks): the form for the runs of >= 2 lookups that
        // cdlrm_embbag_bwd_apply_rest is left with (few heads per block).  With every run of one lookup in the list it is slower
        // than a lane group per position (c3: 48.8 against 44.0 us alone, 144 against 105 us inside the step: a block of a large
        // table holds 30 heads = 8 trips); cdlrm_debug_set(6, 64) selects it there too
        int64_t bx = cdiv(cdiv(n, SEG_CH), gpb);
        const int per_cu = g_cdlrm_debug[1] > 0 ? g_cdlrm_debug[1] : 12;
        const int64_t cap = cdiv((int64_t)256 * per_cu, T);
        if (g_cdlrm_debug[1] >= 0 && bx > cap) bx = cap;
        dim3 bgrid((unsigned)bx, (unsigned)T);
#define BLK_LEAN(L, HU_, RA_)                                                                                      \
    hipLaunchKernelGGL((k_bwd_blocks<L, true, HU_, RA_>), bgrid, dim3(256), 0, s, ctx->d_tab, D4, wt, cur, meta, offsets, n, \
                       n_bags, ld_off, grad, ld_bag, ld_table, lr, w.partials, w.pstride, w.longlist, w.longcount,      \
                       touched, aux_total, w.runend, skip_once, kstride, ways, aux_add)
#define BLK_CALL_LEAN(L) BLK_LEAN(L, 1, 8)
        // The runs of >= 2 lookups, one lookup per bag: ONE head at a time with four rows in flight (eight when the block holds one
        // head) -- 4 x 4 / 16 rows in flight need 239 registers, and beside the weight-gradient GEMMs it is the kernel's footprint on
        // a SIMD, not its own latency chain, that the step pays for (alone it finishes in 30 us either way).  c3, tools/ab_step.py,
        // one box, 4 rounds: heads x rows / lone-head rows 4x4/16: 0.5393 ms, 2x4/8: 0.5357, 1x4/8: 0.5332, 2x4/4: 0.5362,
        // 1x4/4: 0.5350.  cdlrm_debug_set(6, 128): the 4x4/16 form.
        if (!offsets && skip_once && !(g_cdlrm_debug[6] & 128)) {
            DISPATCH_LPR_B(lpr, BLK_CALL_LEAN)
        } else {
#define BLK_CALL(L)                                                                                                \
    if (offsets)                                                                                                   \
        hipLaunchKernelGGL((k_bwd_blocks<L, false>), bgrid, dim3(256), 0, s, ctx->d_tab, D4, wt, cur, meta, offsets, n,   \
                           n_bags, ld_off, grad, ld_bag, ld_table, lr, w.partials, w.pstride, w.longlist, w.longcount,  \
                           touched, aux_total, w.runend, skip_once, kstride, ways, aux_add);                       \
    else                                                                                                           \
        hipLaunchKernelGGL((k_bwd_blocks<L, true>), bgrid, dim3(256), 0, s, ctx->d_tab, D4, wt, cur, meta, offsets, n,    \
                           n_bags, ld_off, grad, ld_bag, ld_table, lr, w.partials, w.pstride, w.longlist, w.longcount,  \
                           touched, aux_total, w.runend, skip_once, kstride, ways, aux_add)
        DISPATCH_LPR_B(lpr, BLK_CALL)
#undef BLK_CALL
        }
#undef BLK_CALL_LEAN
#undef BLK_LEAN
    } else {
        int64_t gx = cdiv(n, gpb);
        if (gx > 65535) gx = 65535;
        // at most 12 workgroups per CU (the position loop strides): the kernel runs on a side queue beside the weight-gradient and
        // bottom-MLP GEMMs, and an unbounded grid (26 624 workgroups at c3, 213 k at c5) takes every free wave slot between their
        // launches.  Same box, same process, 6 x 90 steps each (tools/ab_step.py --attr debug:1): uncapped / 12 / 8 / 6 / 4 per CU
        // -> 0.5804 / 0.5754 / 0.5762 / 0.5977 / 0.6386 ms at c3 (10 .. 24: the same as 12), 3.762 / 3.722 ms at c5, nothing at a
        // per-rank batch of 1024 (3328 workgroups to begin with).  Results do not depend on the grid (a position is owned by
        // one lane group).  cdlrm_debug_set(1, n): n per CU, -1 uncapped
        const int per_cu = g_cdlrm_debug[1] > 0 ? g_cdlrm_debug[1] : 12;
        const int64_t cap = cdiv((int64_t)256 * per_cu, T);
        if (g_cdlrm_debug[1] >= 0 && gx > cap) gx = cap;
        dim3 grid((unsigned)gx, (unsigned)T);
#define BWD_CALL(L)                                                                                                \
    if (offsets)                                                                                                   \
        hipLaunchKernelGGL((k_bwd_chunks<L, false>), grid, dim3(256), 0, s, ctx->d_tab, D4, wt, cur, meta, offsets, n,    \
                           n_bags, ld_off, grad, ld_bag, ld_table, lr, w.partials, w.pstride, w.longlist, w.longcount,  \
                           touched, aux_total, w.runend, kstride, ways, aux_add);                                  \
    else                                                                                                           \
        hipLaunchKernelGGL((k_bwd_chunks<L, true>), grid, dim3(256), 0, s, ctx->d_tab, D4, wt, cur, meta, offsets, n,     \
                           n_bags, ld_off, grad, ld_bag, ld_table, lr, w.partials, w.pstride, w.longlist, w.longcount,  \
                           touched, aux_total, w.runend, kstride, ways, aux_add)
        DISPATCH_LPR_B(lpr, BWD_CALL)
#undef BWD_CALL
    }
    int64_t lx = cdiv((int64_t)T * (n / SEG_CH + 1), gpb);
    if (lx > 1024) lx = 1024;
#define LONG_CALL(L) hipLaunchKernelGGL(k_bwd_long<L>, dim3((unsigned)lx), dim3(256), 0, s, ctx->d_tab, D4, wt, cur, n, lr, w.partials, w.pstride, w.longlist, w.longcount, touched, aux_total, w.runend, kstride, ways, aux_add)
    DISPATCH_LPR_B(lpr, LONG_CALL)
#undef LONG_CALL
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_embbag_bwd_apply(cdlrm_ctx* ctx, const int64_t* offsets, int64_t n, int64_t n_bags, int64_t ld_off,
                                      const float* grad, int64_t ld_bag, int64_t ld_table, float lr, void* work,
                                      uint8_t* touched, void* stream) {
    return cdlrm_embbag_bwd_apply_core(ctx, offsets, n, n_bags, ld_off, grad, ld_bag, ld_table, lr, work, nullptr, nullptr, 0, 0, touched, stream, 0);
}

// The apply behind cdlrm_gather_interact_bwd_sgd: slots the batch reads once were updated there (their gradient rows were
// never written); what is left are the runs of two and more lookups.
extern "C" int cdlrm_embbag_bwd_apply_rest(cdlrm_ctx* ctx, const int64_t* offsets, int64_t n, int64_t n_bags, int64_t ld_off,
                                           const float* grad, int64_t ld_bag, int64_t ld_table, float lr, void* work,
                                           uint8_t* touched, void* stream) {
    return cdlrm_embbag_bwd_apply_core(ctx, offsets, n, n_bags, ld_off, grad, ld_bag, ld_table, lr, work, nullptr, nullptr, 0, 0, touched, stream, 1);
}

// The apply over a window chunk's sorted lists (cdlrm_embbag_bwd_prepare_window; keys / meta: batch j's views, tstride = nb * n):
// no per-batch sort.  `work` only lends its scratch (partial sums, long-run list); its long-run counter must be zero -- every
// apply leaves it zero, a fresh buffer is zero-filled by the caller.  One lookup per bag (the Criteo layout).
extern "C" int cdlrm_embbag_bwd_apply_sorted(cdlrm_ctx* ctx, int64_t n, const float* grad, int64_t ld_bag, int64_t ld_table,
                                             float lr, void* work, const uint64_t* keys, const int32_t* meta, int64_t tstride,
                                             int32_t aux_phase, int32_t rest, uint8_t* touched, void* stream) {
    CDLRM_REQUIRE(ctx && keys && meta && tstride >= n, "null argument / tstride");
    CDLRM_REQUIRE(aux_phase >= 0 && aux_phase < (ctx->aux_phases > 0 ? ctx->aux_phases : 1), "aux_phase outside the geometry's aux_phases");
    return cdlrm_embbag_bwd_apply_core(ctx, nullptr, n, n, 0, grad, ld_bag, ld_table, lr, work, keys, meta, tstride, aux_phase, touched, stream,
                     rest ? 1 : 0);
}

extern "C" int cdlrm_embbag_bwd_sgd(cdlrm_ctx* ctx, const int32_t* slots, const int64_t* offsets, int64_t n,
                                    int64_t n_bags, int64_t ld_off, const float* grad, int64_t ld_bag,
                                    int64_t ld_table, float lr, void* work, uint8_t* touched, void* stream) {
    int rc = cdlrm_embbag_bwd_prepare(ctx, slots, n, work, stream);
    if (rc) return rc;
    return cdlrm_embbag_bwd_apply(ctx, offsets, n, n_bags, ld_off, grad, ld_bag, ld_table, lr, work, touched, stream);
}
