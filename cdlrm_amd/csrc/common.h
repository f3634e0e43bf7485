// Internal helpers shared by the gfx950 kernels of libcdlrm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/cdlrm_hip.h"

#define CDLRM_WAVE 64
#define BM_WPB 1024   // bitmap words per scan block; table bitmaps are padded to this

// Per-table descriptor, resident in HBM, read through the scalar path (wave-uniform table id).
struct TableDesc {
    int64_t n_rows;     // rows of the host master table
    int64_t P;          // sets
    int64_t tag_base;   // element offset of this table's [P, ways] tags
    int64_t row_base;   // first cache row of this table in the flat weight buffer
    int64_t set_base;   // first set of this table in flat per-set arrays (prot)
    int64_t bm_base;    // first 64-bit word of this table's window bitmap
    int64_t bm_words;   // ceil(n_rows / 64)
    int64_t rows;       // ways * P + aux
};

struct cdlrm_ctx {
    int T = 0, D = 0, ways = 0, aux = 0, aux_phases = 1, device = 0;
    std::vector<TableDesc> h_tab;
    TableDesc* d_tab = nullptr;
    int64_t total_rows = 0, total_tags = 0, total_sets = 0, total_bm_words = 0;
    int64_t* tags = nullptr;
    float* weight = nullptr;
    float** d_host_rows = nullptr;       // device array [T] of device-visible pointers
    std::vector<float*> h_host_rows;
    float** d_ptr_fetch = nullptr;       // device arrays [T] for per-call pointer tables
    float** d_ptr_wb = nullptr;
    // window-resident rows of the non-cached window indices (cdlrm_ctx_bind_victims); null = read the host tables
    const int64_t* vict_idx = nullptr;
    const int64_t* vict_off = nullptr;
    const float* vict_rows = nullptr;
    int* d_err = nullptr;                // device error word
    void* ev_start = nullptr;            // cdlrm_ctx_time_next_gather: events of the NEXT cdlrm_embbag_fwd launch
    void* ev_stop = nullptr;
    int64_t* d_scan = nullptr;           // block sums for the scans of the window plan (plan stream)
    int64_t scan_cap = 0;
    // block sums for the table-agg / sync-to-rank-0 compaction: those run on the MAIN stream while the plan of the
    // next window may be scanning on the plan stream -- one scratch per path, never shared across streams
    int64_t* d_scan_agg = nullptr;
    int64_t scan_agg_cap = 0;
    int64_t* d_small = nullptr;          // small device scratch (counters), 256 int64
    int64_t* h_pinned = nullptr;         // pinned host staging for _sync reads, 1024 int64
};

void cdlrm_set_error(const char* fmt, ...);

#define CDLRM_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            cdlrm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (int)e__;                                                               \
        }                                                                                  \
    } while (0)

#define CDLRM_REQUIRE(cond, msg)                                                           \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            cdlrm_set_error("%s: requirement failed: %s (%s)", __func__, #cond, msg);      \
            return CDLRM_EINVAL;                                                           \
        }                                                                                  \
    } while (0)

#define CDLRM_HIDDEN_DATA __attribute__((visibility("hidden")))
#define CDLRM_LAUNCH_CHECK() CDLRM_HIP_CHECK(hipGetLastError())

// Development switches (cdlrm_debug_set, tools/ab_step.py --attr debug:<key>): A/B a kernel path against the one it replaced
// on ONE box in ONE process -- box-to-box spread is larger than most single-kernel gains.  All zero in production.
//   0: 13-wide forward on the LDS-tiled kernel   1: workgroups per CU of the embedding backward's chunk kernel (0: 12, -1: one per
//   8 positions)   2: scalar slab reduction   3: per-op host clocks of a tape   4 / 5: workgroups per CU of the fused gather +
//   interaction forward / backward
//   6: GEMM kernel selectors (bits): 16 = 128x64 instead of 128x128 k_gemm2 tiles at M >= 65536; 32 = never the wide kernel
//      (k_gemm3); 256 = the wide kernel for every eligible launch, CDLRM_GEMM_ALONE or not; 512 / 1024 = the split-M weight
//      gradients (all / the 512-wide ones) on the wide kernel.  Different contraction order inside a 16-deep group than
//      k_gemm2: equal to fp32 rounding, not bit for bit.  64 = the embedding backward's sums (cdlrm_embbag_bwd_apply) with a lane group
//      per block of 32 sorted positions (k_bwd_blocks, the kernel of cdlrm_embbag_bwd_apply_rest) instead of per position
//      (k_bwd_chunks): bit-identical.  128 = cdlrm_embbag_bwd_apply_rest / _apply_sorted(rest) with four heads x four rows
//      (sixteen for a lone head) in flight instead of one head x four (eight): bit-identical.
//      -DCDLRM_DEV builds ONLY (timing experiments that SKIP work; the shipped library refuses them): 1 = no embedding
//      update, 2 = no slot sort
//   7: bits: 1 = the epilogues before round 5 (operands fetched behind, not ahead of, their use); 2 = the fused gather +
//      interaction forward on ONE slab slice (before round 6's double buffering) -- both bit-identical to the default
// No key makes the shipped library skip work: a number measured with any of them set is a number for the same arithmetic.
extern CDLRM_HIDDEN_DATA int g_cdlrm_debug[8];

// Completion events attached to a launch (cdlrm_event_attach_next).  An event RECORDED on the training queue is a marker
// packet of its own and leaves a 6-8 us bubble there; handed to hipExtLaunchKernel as the stop event of
// the kernel it is meant to follow, the same event completes with that kernel and costs the queue nothing.  The pending
// event is per host thread (the thread that issues the launch); a launch site that can carry it takes it with
// cdlrm_take_stop_event, an entry point that could not place it records it the ordinary way before it returns.
// Only the two entry points the engine attaches events to (cdlrm_linear_bwd, cdlrm_interact_bwd) open a CdlrmStopScope; a
// launch site takes the pending event only inside such a scope, and the scope's destructor -- i.e. EVERY exit path of the
// entry point: B == 0, a failed requirement, a failed launch -- records whatever is still pending, so an event can neither be
// dropped nor ride on a later, unrelated GEMM of the thread (cdlrm_linear_fwd / cdlrm_mlp_wgrad share the launch sites).
#define CDLRM_HIDDEN __attribute__((visibility("hidden")))
// per host thread (the thread that issues the launch); lives in tape.hip behind a hidden accessor, so the library exports
// nothing but the entry points include/cdlrm_hip.h declares
struct CdlrmStopState {
    hipEvent_t event;
    hipStream_t stream;
    int depth;
};
CDLRM_HIDDEN CdlrmStopState* cdlrm_stop_state();
static inline hipEvent_t cdlrm_take_stop_event(hipStream_t s) {
    CdlrmStopState* st = cdlrm_stop_state();
    hipEvent_t e = st->event;
    if (e && st->depth > 0 && st->stream == s) {
        st->event = nullptr;
        return e;
    }
    return nullptr;
}
struct CdlrmStopScope {
    hipEvent_t held = nullptr;
    hipStream_t held_stream = nullptr;
    CdlrmStopScope() { ++cdlrm_stop_state()->depth; }
    // a call with several launches: keep the event away from them, it is recorded behind the last one (on exit)
    void hold(hipStream_t s) {
        held = cdlrm_take_stop_event(s);
        held_stream = s;
    }
    ~CdlrmStopScope() {
        CdlrmStopState* st = cdlrm_stop_state();
        if (held) (void)hipEventRecord(held, held_stream);
        if (--st->depth == 0 && st->event) {      // no launch carried it
            (void)hipEventRecord(st->event, st->stream);
            st->event = nullptr;
        }
    }
};
#define CDLRM_LAUNCH_EV(kernel, grid, block, lds, stream, ...)                                                     \
    do {                                                                                                           \
        hipEvent_t se__ = cdlrm_take_stop_event(stream);                                                           \
        if (se__) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, se__, 0, __VA_ARGS__);          \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                    \
    } while (0)
// hipGetLastError() is sticky per host thread: an error some OTHER caller of the runtime left behind (the host
// framework probes pointers and events with calls that are allowed to fail) would be reported by our next launch
// check.  Entry points that launch without a preceding checked call drop such a stale error first
// (CDLRM_DEBUG_STALE=1 prints it).
#define CDLRM_CLEAR_STALE()                                                                                   \
    do {                                                                                                      \
        hipError_t stale_ = hipGetLastError();                                                                \
        if (stale_ != hipSuccess && getenv("CDLRM_DEBUG_STALE"))                                              \
            fprintf(stderr, "[cdlrm] stale HIP error before %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(stale_)); \
    } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
__device__ __forceinline__ int64_t cdiv_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// ---- device helpers -----------------------------------------------------------------------------

__device__ __forceinline__ int64_t mod_sets(int64_t idx, int64_t P) {
    // idx % P for non-negative idx; 32-bit fast path (Criteo ids and set counts fit 32 bits)
    if (((uint64_t)idx | (uint64_t)P) >> 32) return idx % P;
    return (int64_t)((uint32_t)idx % (uint32_t)P);
}

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Exclusive scan of one int per thread over the whole block (blockDim.x multiple of 64, <= 1024).
// Returns the exclusive prefix; *total receives the block sum.  smem: >= 17 ints.
__device__ __forceinline__ int block_excl_scan(int v, int* smem, int* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = wave_incl_scan(v);
    if (lane == 63) smem[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        int w = lane < nw ? smem[lane] : 0;
        int wi = wave_incl_scan(w);
        if (lane < nw) smem[lane] = wi - w;
        if (lane == nw - 1) smem[16] = wi;
    }
    __syncthreads();
    int res = smem[wid] + inc - v;
    *total = smem[16];
    __syncthreads();
    return res;
}

// first k in [lo, hi) with a[k] >= key
__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* a, int64_t lo, int64_t hi, uint64_t key) {
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// number of entries of off[0..m) that are <= pos, minus 1  (bag of lookup `pos`)
__device__ __forceinline__ int64_t bag_of(const int64_t* off, int64_t m, int64_t pos) {
    int64_t lo = 0, hi = m;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (off[mid] <= pos) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}
// table owning flat position p given offsets off[0..T]
__device__ __forceinline__ int table_of(const int64_t* off, int T, int64_t p) {
    int lo = 0, hi = T;   // find largest k with off[k] <= p
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// scan utilities implemented in scan.hip
int cdlrm_scan_reserve(cdlrm_ctx* ctx, int64_t nblocks);
int cdlrm_scan_reserve_agg(cdlrm_ctx* ctx, int64_t nblocks);
