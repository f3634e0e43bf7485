// Layer chains in ONE launch (local batches <= 2048: the per-rank shapes of a multi-GPU run).
//
// At M = 1024 a Linear layer is 3-4 us of MFMA work inside a launch that cannot take less than ~5 us on its queue plus a
// load latency: the top MLP's three forward GEMMs cost 32 us as three launches.  Here a chain of layers -- each one's
// output is the next one's input -- runs as one persistent kernel over (layer, 32-row block, 32-column tile) work items,
// and a row block moves on to the next layer as soon as ITS tiles of the previous layer are done (a counter per layer and
// row block), not when the whole grid is.
//
// The hand-off stays inside one XCD.  The part has 8 XCDs with an L2 each, coherent only through memory: an agent-scope
// release / acquire between workgroups is an L2 write-back + invalidate (measured on the head kernel's "last workgroup
// sums" tail: 16 us).  So row block rb belongs to XCD rb % 8 through ALL layers: every workgroup reads the XCD it runs on
// (HW_REG_XCC_ID) and takes work from that XCD's queue only.  A tile's stores are complete in the shared L2 when the
// wave's vmcnt reaches 0; the consumer has never touched those lines in this launch (L1 is invalidated at launch), so
// plain loads see them.  Counters and queue heads are atomics (performed beyond the L1).
//
// No deadlock whatever subset of the grid is resident: a queue lists its items in dependency order (layer by layer), an
// item is only ever claimed by a RUNNING workgroup, and what that workgroup may wait for was claimed earlier.  Waits are
// bounded anyway (an error flag instead of a hang).  The workgroup that leaves last resets the counters for the next
// launch and checks that every queue was emptied (an XCD without workgroups would leave its row blocks undone).
//
// MEASURED (tools/bench_kernels.py --only chain, MI355X, top MLP 480-512-512-256): bit-identical to the per-layer launches,
// and SLOWER -- forward 53.9 us as one launch against 30.9 us as three at M = 1024 (67.0 / 52.5 at 2048, 40.2 / 25.1 at
// 512); the input-gradient chain likewise (47.4 / 32.1).  A tile's body is ~4 us; claiming it (atomic with return +
// barrier), polling its row block's counter and publishing it (store acknowledgements + barrier + atomic) are three
// device-scope round trips of 1-2 us each: the hand-off costs what a launch boundary costs (~5 us), per TILE instead of
// per layer.  At these sizes a layer is 3.7 us of the whole chip's MFMA time; any synchronisation between layers that
// goes through memory is of that order.  So the training step does not use the chain; the entry points, the test and
// the benchmark stay as the record of the experiment (CDLRM_MLP_CHAIN=0 makes the entry points launch layer by layer).
#pragma once

#define CHAIN_MAX_OPS 4
#define CHAIN_XCDS 8
#define CHAIN_Q_PITCH 32            // ints between two queue heads (a 128-byte line each)
#define CHAIN_EXITED (CHAIN_XCDS * CHAIN_Q_PITCH)
#define CHAIN_ERR (CHAIN_EXITED + 1)
#define CHAIN_DONE0 (CHAIN_EXITED + CHAIN_Q_PITCH)
#define CHAIN_MAX_RBS 64            // M <= 2048
#define CHAIN_DBG0 (CHAIN_DONE0 + CHAIN_MAX_OPS * CHAIN_MAX_RBS)
#define CHAIN_SYNC_INTS (CHAIN_DBG0 + 512)      // + diagnostics words (CDLRM_CHAIN_KEEP & 4)

struct ChainArgs {
    int32_t n_ops, rbs;             // layers; row blocks of 32 rows
    int32_t ntn[CHAIN_MAX_OPS];     // 32-column tiles per layer
    int32_t mode[CHAIN_MAX_OPS];    // staged_body MODE per layer
    int32_t spin_limit;
    int32_t keep;                   // diagnostics: leave the counters as they are at the end
    int32_t* sync;                  // CHAIN_SYNC_INTS ints, zero before the first launch (the kernel leaves them zero)
    GemmArgs g[CHAIN_MAX_OPS];
};

__device__ __forceinline__ int chain_xcd() {
    // HW_REG_XCC_ID (hwreg 20), bits [3:0]: the XCD this wave runs on
    return (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15) % CHAIN_XCDS;
}

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256) k_gemm_chain(ChainArgs c) {
    extern __shared__ __attribute__((aligned(16))) float st_lds[];
    __shared__ int s_item, s_xcd;
    if (threadIdx.x == 0) s_xcd = chain_xcd();          // ONE reading for the workgroup (every thread decodes with it)
    __syncthreads();
    const int xcd = s_xcd;
#define CHAIN_DBG(k, v)                                                                             \
    if ((c.keep & 4) && threadIdx.x == 0 && blockIdx.x < 128) atomicExch(c.sync + CHAIN_DBG0 + blockIdx.x * 4 + (k), (v));
    CHAIN_DBG(0, 100 + xcd)
    int32_t* q = c.sync + xcd * CHAIN_Q_PITCH;
    int32_t* done = c.sync + CHAIN_DONE0;
    const int my_rbs = (c.rbs - xcd + CHAIN_XCDS - 1) / CHAIN_XCDS;         // row blocks xcd, xcd + 8, ...
    for (int guard = 0; guard < (1 << 16); ++guard) {    // (bounded: a workgroup never takes more items than exist)
        if (threadIdx.x == 0) s_item = atomicAdd(q, 1);
        __syncthreads();
        int t = s_item;
        CHAIN_DBG(1, t)
        CHAIN_DBG(2, 1)
        CHAIN_DBG(3, guard)
        int op = 0;
        while (op < c.n_ops && t >= my_rbs * c.ntn[op]) {
            t -= my_rbs * c.ntn[op];
            ++op;
        }
        if (op >= c.n_ops) break;                                           // (uniform: s_item is shared)
        const int rb = xcd + CHAIN_XCDS * (t / c.ntn[op]), nt = t % c.ntn[op];
        if (op > 0) {
            if (threadIdx.x == 0) {
                const int need = c.ntn[op - 1];
                int32_t* ctr = done + (op - 1) * CHAIN_MAX_RBS + rb;
                int spins = 0;
                while (atomicAdd(ctr, 0) < need) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > c.spin_limit) {
                        atomicOr(c.sync + CHAIN_ERR, 1);
                        break;
                    }
                }
            }
            __syncthreads();
        }
        CHAIN_DBG(2, 2)
        const GemmArgs& g = c.g[op];
        if (c.keep & 2) { /* diagnostics: the hand-off logic alone */ }
        else if (c.mode[op] == 0) staged_body<A_KC, B_KC, 0>(g, (unsigned)nt, (unsigned)rb, 0u, st_lds);
        else if (c.mode[op] == 1) staged_body<A_KC, B_KC, 1>(g, (unsigned)nt, (unsigned)rb, 0u, st_lds);
        else staged_body<A_KC, B_KC, 2>(g, (unsigned)nt, (unsigned)rb, 0u, st_lds);
        // this tile's stores have reached the L2 (every wave waits for its own, then the workgroup meets)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(done + op * CHAIN_MAX_RBS + rb, 1);
        CHAIN_DBG(2, 3)
    }
    CHAIN_DBG(2, 8)
    __shared__ int s_last;
    if (threadIdx.x == 0) s_last = atomicAdd(c.sync + CHAIN_EXITED, 1) == (int)gridDim.x - 1;
    __syncthreads();
    CHAIN_DBG(2, 9)
    if (s_last && !(c.keep & 1)) {
        // last one out: every queue must have been emptied by workgroups of ITS XCD; reset for the next launch (plain stores:
        // nobody else touches the words any more, the end of the kernel publishes them)
        if (threadIdx.x == 0) {
            int total = 0;
            for (int o = 0; o < c.n_ops; ++o) total += c.ntn[o];
            int bad = 0;
            for (int x = 0; x < CHAIN_XCDS; ++x) {
                const int rbx = (c.rbs - x + CHAIN_XCDS - 1) / CHAIN_XCDS;
                if (atomicAdd(c.sync + x * CHAIN_Q_PITCH, 0) < rbx * total) bad = 1;
            }
            if (bad) atomicOr(c.sync + CHAIN_ERR, 2);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < CHAIN_DBG0; i += blockDim.x)
            if (i != CHAIN_ERR) c.sync[i] = 0;
    }
}

// the chain applies to a layer: the staged kernel's conditions on an un-split contraction
template <bool A_KC, bool B_KC>
static inline bool chain_layer_ok(const GemmArgs& g) {
    return g.M <= 32 * CHAIN_MAX_RBS && direct_staged<A_KC, B_KC>(g, g.K);
}

template <bool A_KC, bool B_KC>
static int launch_chain(ChainArgs& c, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        staged_lds_attr(k_gemm_chain<A_KC, B_KC>);
        attr = true;
    }
    static int grid = 0;
    if (grid == 0) {
        const char* e = getenv("CDLRM_CHAIN_GRID");
        grid = e && atoi(e) > 0 ? atoi(e) : 512;     // 2 workgroups per CU (73.7 KB of LDS each), 64 per XCD
    }
    static int spin = 0, keep = 0;
    if (spin == 0) {
        const char* e = getenv("CDLRM_CHAIN_SPIN");
        spin = e && atoi(e) > 0 ? atoi(e) : (1 << 16);
        keep = getenv("CDLRM_CHAIN_KEEP") ? atoi(getenv("CDLRM_CHAIN_KEEP")) : 0;
    }
    c.spin_limit = spin;
    c.keep = keep;
    hipLaunchKernelGGL((k_gemm_chain<A_KC, B_KC>), dim3((unsigned)grid), dim3(256), ST_LDS_BYTES, s, c);
    CDLRM_LAUNCH_CHECK();
    return 0;
}
