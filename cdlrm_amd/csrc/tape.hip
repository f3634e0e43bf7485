// Launch tapes: the recorded call sequence of one training step, replayed by ONE library call.
//
// The engine (cdlrm_amd/engine.py, _step_taped) records the ~45 calls of a step once per control path -- kernel launches
// through this library's entry points, event records and stream waits -- and re-issues them every step.  Replayed from
// Python, each call costs 3-5 us of interpreter and ctypes marshalling: 0.22 ms per step, which at a per-rank batch of
// 1024 (8192 over 8 GPUs) is MORE than the GPU needs for the step.  A tape holds the same calls as (entry point,
// arguments) records; arguments that change from step to step (the batch's tensors) are "cells" the host patches before
// a replay.
//
// Typed calls (round 3; the first version called every entry point through ONE mismatched function-pointer type, which
// only the x86-64 System V register convention made work): every entry point a tape may hold is REGISTERED below with its
// true type.  The host hands over a call's integer-class arguments (pointers, int32, int64: one int64 each, in parameter
// order) and its float arguments separately; `invoke<R, A...>` walks the parameter types A... of the registered function,
// takes the next integer or the next float for each, converts it to exactly that parameter type and calls the function
// through a pointer of its own type.  An address that is not registered, or an argument count that does not match the
// registered signature, is refused by cdlrm_tape_add (the engine then replays that step's tape from Python).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"

#define TAPE_MAX_INT 24
#define TAPE_MAX_FLT 8
#define TAPE_MAX_LANES 4

typedef int (*tape_invoke_fn)(void* fn, const int64_t* ia, const float* fa);

template <typename T>
static inline T tape_take(const int64_t*& ia, const float*& fa) {
    if constexpr (std::is_floating_point<T>::value) {
        return (T)*fa++;
    } else if constexpr (std::is_pointer<T>::value) {
        return reinterpret_cast<T>((intptr_t)*ia++);
    } else {
        static_assert(std::is_integral<T>::value, "tape entry points take pointers, integers and floats only");
        return static_cast<T>(*ia++);
    }
}

template <typename R, typename... A>
static int tape_invoke(void* fn, const int64_t* ia, const float* fa) {
    // braced initialisation evaluates its elements left to right: arguments are taken in parameter order
    std::tuple<A...> args{tape_take<A>(ia, fa)...};
    return (int)std::apply(reinterpret_cast<R (*)(A...)>(fn), args);    // void* -> the function's OWN type: a round trip
}

struct TapeEntry {
    void* fn;
    tape_invoke_fn invoke;
    int32_t n_int, n_flt;
    const char* name;
    int32_t stream_arg;     // position (in the full parameter list) of the hipStream_t the call issues on; -1: none
};

template <typename R, typename... A>
static TapeEntry tape_entry(R (*f)(A...), const char* name) {
    static_assert(std::is_same<R, int>::value, "tape entry points return int");
    constexpr int n_flt = (0 + ... + (std::is_floating_point<A>::value ? 1 : 0));
    constexpr int n_int = (int)sizeof...(A) - n_flt;
    static_assert(n_int <= TAPE_MAX_INT && n_flt <= TAPE_MAX_FLT, "too many arguments for a tape call");
    return TapeEntry{reinterpret_cast<void*>(f), &tape_invoke<R, A...>, n_int, n_flt, name, -1};
}
// the library's convention: a launching entry point takes its stream as the LAST parameter (void*)
template <typename... A> struct tape_last_is_ptr : std::false_type {};
template <typename A0> struct tape_last_is_ptr<A0> : std::is_same<A0, void*> {};
template <typename A0, typename... A> struct tape_last_is_ptr<A0, A...> : tape_last_is_ptr<A...> {};
template <typename R, typename... A>
static TapeEntry tape_entry_s(R (*f)(A...), const char* name) {
    static_assert(tape_last_is_ptr<A...>::value, "a launching entry point ends on its stream (void*)");
    TapeEntry e = tape_entry(f, name);
    e.stream_arg = (int32_t)sizeof...(A) - 1;
    return e;
}
template <typename R, typename... A>
static TapeEntry tape_entry_at(R (*f)(A...), const char* name, int32_t stream_arg) {
    TapeEntry e = tape_entry(f, name);
    e.stream_arg = stream_arg;
    return e;
}

extern "C" int cdlrm_tape_probe(float f0, int64_t a0, float f1, int32_t a1, void* a2, int64_t a3, float f2, int32_t a4,
                                int64_t a5, int64_t a6, void* a7, int32_t a8, float f3, int64_t a9);
extern "C" int cdlrm_tape_probe_log(int64_t tag, int64_t spin);

#define TAPE_FN(f) tape_entry_s(&f, #f)                 /* issues on the stream its last parameter names */
#define TAPE_FN_AT(f, pos) tape_entry_at(&f, #f, pos)   /* stream at that parameter position; -1: the call issues nothing */
// every entry point a recorded training / evaluation step can issue (engine.py: _fwd_bwd, step, evaluate)
static const std::vector<TapeEntry>& tape_registry() {
    static const std::vector<TapeEntry> reg = {
        TAPE_FN(cdlrm_embbag_probe), TAPE_FN(cdlrm_embbag_take), TAPE_FN(cdlrm_embbag_fwd), TAPE_FN(cdlrm_embbag_bwd_sgd),
        TAPE_FN(cdlrm_embbag_bwd_prepare), TAPE_FN(cdlrm_embbag_bwd_apply), TAPE_FN(cdlrm_embbag_bwd_apply_rest),
        TAPE_FN(cdlrm_gather_interact_bwd_sgd), TAPE_FN(cdlrm_embbag_bwd_prepare_window), TAPE_FN(cdlrm_embbag_bwd_apply_sorted),
        TAPE_FN(cdlrm_window_resolve),
        TAPE_FN(cdlrm_mark_rows), TAPE_FN(cdlrm_interact_fwd), TAPE_FN(cdlrm_interact_bwd), TAPE_FN(cdlrm_gather_interact_fwd),
        TAPE_FN(cdlrm_gather_interact_bwd), TAPE_FN(cdlrm_linear_fwd),
        TAPE_FN(cdlrm_linear_bwd), TAPE_FN(cdlrm_mlp_wgrad), TAPE_FN(cdlrm_mlp_wgrad_sgd), TAPE_FN(cdlrm_bce_fwd_bwd),
        TAPE_FN(cdlrm_loss_fwd_bwd), TAPE_FN(cdlrm_head_fwd_bwd), TAPE_FN(cdlrm_head_finish), TAPE_FN(cdlrm_act_bwd),
        TAPE_FN(cdlrm_sgd_step), TAPE_FN(cdlrm_sgd_step2), TAPE_FN(cdlrm_scale_div),
        TAPE_FN_AT(cdlrm_ctx_time_next_gather, -1),      // arms the NEXT gather launch: no stream of its own
        TAPE_FN_AT(cdlrm_event_record, 1), TAPE_FN_AT(cdlrm_stream_wait_event, 0), TAPE_FN_AT(cdlrm_event_attach_next, 1),
        TAPE_FN(cdlrm_agg_compact), TAPE_FN(cdlrm_agg_gather), TAPE_FN(cdlrm_agg_scatter),
        TAPE_FN_AT(cdlrm_tape_probe, -1), TAPE_FN_AT(cdlrm_tape_probe_log, -1),
    };
    return reg;
}
#undef TAPE_FN_AT
#undef TAPE_FN

struct TapeOp {
    void* fn;
    tape_invoke_fn invoke;
    int32_t n_int, n_flt;
    int64_t iargs[TAPE_MAX_INT];
    int32_t cell[TAPE_MAX_INT];     // -1: literal, else index of the cell whose value is the argument
    float fargs[TAPE_MAX_FLT];
    int32_t lane = 0;       // 0: issued by the replaying thread, 1 .. 3: by that helper thread (cdlrm_tape_set_lanes)
    int32_t dep = -1;       // tape index of an op of ANOTHER lane that has to be issued before this one (-1: none)
    const char* name = "";
    // development (cdlrm_debug_set(3, 1); tools/host_time.py): host time of this op's call / of its wait for `dep`, summed
    int64_t call_ns = 0, wait_ns = 0, calls = 0, max_ns = 0;
};

struct cdlrm_tape {
    std::vector<TapeOp> ops;
    std::vector<int64_t> cells;
    // multi-lane replay
    int lanes = 1;                      // lanes in use (1: single-threaded replay)
    int device = 0;
    std::atomic<int> done[TAPE_MAX_LANES];           // highest tape index each lane has issued (INT_MAX: the lane is through)
    std::atomic<int> finished[TAPE_MAX_LANES];       // helper lanes: this replay's ops are all issued
    int rc[TAPE_MAX_LANES];
    char err[TAPE_MAX_LANES][256];
};

static inline int tape_call(const TapeOp& o, const int64_t* a) { return o.invoke(o.fn, a, o.fargs); }

extern "C" cdlrm_tape* cdlrm_tape_create(int32_t n_cells) {
    if (n_cells < 0 || n_cells > 64) { cdlrm_set_error("cdlrm_tape_create: 0..64 cells"); return nullptr; }
    cdlrm_tape* t = new (std::nothrow) cdlrm_tape();
    if (!t) { cdlrm_set_error("cdlrm_tape_create: out of memory"); return nullptr; }
    t->cells.assign((size_t)n_cells, 0);
    return t;
}

extern "C" void cdlrm_tape_destroy(cdlrm_tape* t) { delete t; }

extern "C" int cdlrm_tape_add(cdlrm_tape* t, void* fn, int32_t n_int, const int64_t* iargs, const int32_t* cell,
                              int32_t n_flt, const float* fargs) {
    CDLRM_REQUIRE(t && fn, "null argument");
    CDLRM_REQUIRE(n_int >= 0 && n_int <= TAPE_MAX_INT && n_flt >= 0 && n_flt <= TAPE_MAX_FLT, "too many arguments for a tape call");
    const TapeEntry* e = nullptr;
    for (const TapeEntry& r : tape_registry())
        if (r.fn == fn) { e = &r; break; }
    CDLRM_REQUIRE(e, "not a registered tape entry point (csrc/tape.hip: tape_registry)");
    if (e->n_int != n_int || e->n_flt != n_flt) {
        cdlrm_set_error("cdlrm_tape_add: %s takes %d integer-class + %d float arguments, got %d + %d", e->name, e->n_int,
                        e->n_flt, n_int, n_flt);
        return CDLRM_EINVAL;
    }
    TapeOp o;
    o.fn = fn;
    o.invoke = e->invoke;
    o.name = e->name;
    o.n_int = n_int; o.n_flt = n_flt;
    for (int i = 0; i < TAPE_MAX_INT; ++i) {
        o.iargs[i] = i < n_int ? iargs[i] : 0;
        o.cell[i] = i < n_int ? cell[i] : -1;
        CDLRM_REQUIRE(o.cell[i] >= -1 && o.cell[i] < (int)t->cells.size(), "cell index outside the tape's cells");
    }
    for (int i = 0; i < TAPE_MAX_FLT; ++i) o.fargs[i] = i < n_flt ? fargs[i] : 0.f;
    t->ops.push_back(o);
    return 0;
}

extern "C" int64_t* cdlrm_tape_cells(cdlrm_tape* t) { return t ? t->cells.data() : nullptr; }

extern "C" int64_t cdlrm_tape_length(cdlrm_tape* t) { return t ? (int64_t)t->ops.size() : -1; }

// development: what op k of a tape is and what it has cost the issuing thread (cdlrm_debug_set(3, 1) switches the clocks on):
// out[0] = lane, out[1] = calls timed, out[2] = ns inside the call, out[3] = ns waiting for another lane's op, out[4] = the
// longest single call.  Returns the
// entry point's name (nullptr: no such op).
extern "C" const char* cdlrm_tape_op_info(cdlrm_tape* t, int64_t k, int64_t* out) {
    if (!t || !out || k < 0 || k >= (int64_t)t->ops.size()) return nullptr;
    const TapeOp& o = t->ops[(size_t)k];
    out[0] = o.lane; out[1] = o.calls; out[2] = o.call_ns; out[3] = o.wait_ns; out[4] = o.max_ns;
    return o.name;
}

// position of the stream parameter of a registered entry point (-1: the call issues nothing itself, -2: not registered)
extern "C" int32_t cdlrm_tape_stream_arg(void* fn) {
    for (const TapeEntry& e : tape_registry())
        if (e.fn == fn) return e.stream_arg;
    return -2;
}

// ---- multi-lane replay -------------------------------------------------------------------------------------------------
// At a per-rank batch of 1024 a step is ~45 runtime calls of ~4.5 us of HOST time each: the thread that issues them, not
// the GPU, sets the step time (0.20 ms of issue against ~0.17 ms of dependent GPU work).  A tape can therefore be split by
// stream: lane 0 -- everything on the training queue -- is issued by the replaying thread, lanes 1 .. 3 -- the side queues:
// embedding backward, the next batch's take / sort, the deferred weight gradients -- by helper threads of the process, one
// per lane, at the same time.  What orders the lanes on the HOST is exactly what orders the streams on the GPU, the events:
// a hipStreamWaitEvent must be issued behind the hipEventRecord it is meant to see and in front of the next record of the
// same event, so every op that touches an event carries `dep`, the latest earlier op of ANOTHER lane on that event, and is
// held back until that one has been issued (program order inside a lane does the rest: the touches of one event are totally
// ordered).  Dependencies point backwards in tape order: no cycles, no lost wake-ups.
static inline void tape_pause() { __builtin_ia32_pause(); }

static inline int64_t tape_now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int tape_run_lane(cdlrm_tape* t, int lane) {
    const int64_t* cells = t->cells.data();
    const int n = (int)t->ops.size();
    const bool timed = g_cdlrm_debug[3] != 0;
    int rc = 0;
    for (int k = 0; k < n; ++k) {
        TapeOp& o = t->ops[k];
        if (o.lane != lane) continue;
        const int64_t t0 = timed ? tape_now_ns() : 0;
        if (o.dep >= 0) {
            const std::atomic<int>& other = t->done[t->ops[o.dep].lane];
            while (other.load(std::memory_order_acquire) < o.dep) tape_pause();
        }
        const int64_t t1 = timed ? tape_now_ns() : 0;
        int64_t a[TAPE_MAX_INT];
        for (int i = 0; i < TAPE_MAX_INT; ++i) a[i] = o.cell[i] >= 0 ? cells[o.cell[i]] : o.iargs[i];
        rc = tape_call(o, a);
        if (timed) { const int64_t d = tape_now_ns() - t1; o.wait_ns += t1 - t0; o.call_ns += d; ++o.calls; if (d > o.max_ns) o.max_ns = d; }
        if (rc) break;
        t->done[lane].store(k, std::memory_order_release);
    }
    t->done[lane].store(INT_MAX, std::memory_order_release);       // (also on failure: the other lanes must not wait for ever)
    return rc;
}

// a helper thread of the process (one per lane > 0): spins for work while steps are being replayed, sleeps when none has come
// for a while
struct TapeHelper {
    int lane = 1;
    std::atomic<cdlrm_tape*> job{nullptr};
    std::atomic<int> sleeping{0};
    std::mutex m;
    std::condition_variable cv;
    void loop() {
        int idle = 0;
        int device = -1;
        for (;;) {
            cdlrm_tape* t = job.exchange(nullptr, std::memory_order_acquire);
            if (!t) {
                if (++idle < (1 << 16)) { tape_pause(); continue; }
                // going to sleep: announce it, THEN look for a job once more (sequentially consistent on both sides: the
                // submitter publishes the job, THEN looks at `sleeping` -- one of the two always sees the other; the bounded
                // wait is a second line of defence, not the mechanism)
                // An idle helper stays asleep: it re-checks for a job every 5 ms (the second line of defence) WITHOUT going back
                // to the spin phase -- a process that replayed a multi-lane tape once would otherwise keep up to three threads
                // spinning ~25 % of a core each through evaluate(), save and the CPU baseline.  Only a job restarts the spin.
                std::unique_lock<std::mutex> lk(m);
                sleeping.store(1, std::memory_order_seq_cst);
                while (!job.load(std::memory_order_seq_cst)) cv.wait_for(lk, std::chrono::milliseconds(5));
                sleeping.store(0, std::memory_order_seq_cst);
                idle = 0;
                continue;
            }
            idle = 0;
            if (t->device >= 0 && t->device != device) { (void)hipSetDevice(t->device); device = t->device; }
            if (t->device >= 0) (void)hipGetLastError();
            t->rc[lane] = tape_run_lane(t, lane);
            if (t->rc[lane]) snprintf(t->err[lane], sizeof(t->err[lane]), "%s", cdlrm_last_error());
            t->finished[lane].store(1, std::memory_order_release);
        }
    }
    void submit(cdlrm_tape* t) {
        job.store(t, std::memory_order_seq_cst);
        if (sleeping.load(std::memory_order_seq_cst)) { std::lock_guard<std::mutex> lk(m); cv.notify_one(); }
    }
};

static TapeHelper* tape_helper(int lane) {
    static TapeHelper* h[TAPE_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    static std::mutex m;
    std::lock_guard<std::mutex> lk(m);
    if (!h[lane]) {                     // never destroyed: the thread outlives every static destructor
        TapeHelper* p = new TapeHelper();
        p->lane = lane;
        std::thread([p] { p->loop(); }).detach();
        h[lane] = p;
    }
    return h[lane];
}

// lane / dep per op (arrays of cdlrm_tape_length entries).  Checked here: lanes are 0 .. 3, a dependency points at an EARLIER
// op of ANOTHER lane.  A tape whose ops are all in lane 0 stays single-threaded.
extern "C" int cdlrm_tape_set_lanes(cdlrm_tape* t, const int32_t* lane, const int32_t* dep, int64_t n) {
    CDLRM_REQUIRE(t && lane && dep && n == (int64_t)t->ops.size(), "one lane / dep entry per recorded op");
    int top = 0;
    for (int64_t k = 0; k < n; ++k) {
        CDLRM_REQUIRE(lane[k] >= 0 && lane[k] < TAPE_MAX_LANES, "lane: 0 .. 3");
        CDLRM_REQUIRE(dep[k] >= -1 && dep[k] < k && (dep[k] < 0 || lane[dep[k]] != lane[k]), "dep: an earlier op of another lane");
        top = lane[k] > top ? lane[k] : top;
    }
    for (int64_t k = 0; k < n; ++k) { t->ops[k].lane = lane[k]; t->ops[k].dep = dep[k]; }
    t->lanes = top + 1;
    if (hipGetDevice(&t->device) != hipSuccess) { t->device = -1; (void)hipGetLastError(); }     // (a box without a GPU: tests)
    return 0;
}

// Re-issues the recorded calls -- in order, or in lanes (cdlrm_tape_set_lanes); stops a lane at its first call that fails and
// returns that code (cdlrm_last_error() describes it).
extern "C" int cdlrm_tape_replay(cdlrm_tape* t) {
    CDLRM_REQUIRE(t, "null tape");
    if (t->lanes <= 1) {
        const int64_t* cells = t->cells.data();
        const bool timed = g_cdlrm_debug[3] != 0;
        for (TapeOp& o : t->ops) {
            int64_t a[TAPE_MAX_INT];
            for (int i = 0; i < TAPE_MAX_INT; ++i) a[i] = o.cell[i] >= 0 ? cells[o.cell[i]] : o.iargs[i];
            const int64_t t0 = timed ? tape_now_ns() : 0;
            const int rc = tape_call(o, a);
            if (timed) { const int64_t d = tape_now_ns() - t0; o.call_ns += d; ++o.calls; if (d > o.max_ns) o.max_ns = d; }
            if (rc) return rc;
        }
        return 0;
    }
    // one multi-lane replay at a time per process: a helper thread takes one job (two trainer threads in one process would
    // otherwise overwrite each other's; uncontended in the one-process-per-GPU layout)
    static std::mutex replay_mutex;
    std::lock_guard<std::mutex> hold(replay_mutex);
    for (int l = 0; l < t->lanes; ++l) {
        t->done[l].store(-1, std::memory_order_relaxed);
        t->finished[l].store(0, std::memory_order_relaxed);
        t->rc[l] = 0;
    }
    for (int l = 1; l < t->lanes; ++l) tape_helper(l)->submit(t);
    const int rc0 = tape_run_lane(t, 0);
    for (int l = 1; l < t->lanes; ++l)
        while (!t->finished[l].load(std::memory_order_acquire)) tape_pause();
    if (rc0) return rc0;
    for (int l = 1; l < t->lanes; ++l)
        if (t->rc[l]) { cdlrm_set_error("%s", t->err[l]); return t->rc[l]; }
    return 0;
}

// the stream-ordering calls of a step, as entry points a tape can hold
extern "C" int cdlrm_event_record(void* event, void* stream) {
    CDLRM_REQUIRE(event, "null event");
    CDLRM_HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return 0;
}

CdlrmStopState* cdlrm_stop_state() {
    static thread_local CdlrmStopState st{nullptr, nullptr, 0};
    return &st;
}

// `event` completes with the NEXT kernel this thread launches on `stream` through cdlrm_linear_bwd or cdlrm_interact_bwd
// (attached to the launch as its stop event: no marker packet on the queue); where that call cannot attach it -- a kernel
// path without the plumbing, more than one launch -- the call records it behind its launches instead.  Same ordering
// guarantees as cdlrm_event_record issued right behind that call.
extern "C" int cdlrm_event_attach_next(void* event, void* stream) {
    CDLRM_REQUIRE(event, "null event");
    CdlrmStopState* st = cdlrm_stop_state();
    if (st->event) {      // (never the case in the training step: one attach per consuming call)
        hipEvent_t old = st->event;
        st->event = nullptr;
        CDLRM_HIP_CHECK(hipEventRecord(old, st->stream));
    }
    st->event = (hipEvent_t)event;
    st->stream = (hipStream_t)stream;
    return 0;
}

extern "C" int cdlrm_stream_wait_event(void* stream, void* event) {
    CDLRM_REQUIRE(event, "null event");
    CDLRM_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return 0;
}

// A HIP stream at an explicit priority (hipDeviceGetStreamPriorityRange: numerically lower = more urgent; the value is
// clamped into the device's range).  The look-ahead plan runs on the LEAST urgent one: its scans share the GPU with the
// training step (main_no_ddp.py runs the same work in a separate Prefetcher process on CPU cores, cache_manager.py:66-115),
// and where they compete for CUs the step goes first.
extern "C" void* cdlrm_stream_create(int32_t priority) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; }
    int p = priority;
    if (p > least) p = least;
    if (p < greatest) p = greatest;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, p) != hipSuccess) {
        cdlrm_set_error("cdlrm_stream_create: hipStreamCreateWithPriority failed");
        return nullptr;
    }
    return (void*)s;
}

extern "C" int cdlrm_stream_destroy(void* stream) {
    if (stream) CDLRM_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

// timing events of the library's own (a torch event has no HIP handle before its first record; a tape needs the handle
// when it is built)
extern "C" void* cdlrm_event_create(int32_t timing) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming) != hipSuccess) {
        cdlrm_set_error("cdlrm_event_create: hipEventCreateWithFlags failed");
        return nullptr;
    }
    return (void*)e;
}

extern "C" int cdlrm_event_destroy(void* event) {
    if (event) CDLRM_HIP_CHECK(hipEventDestroy((hipEvent_t)event));
    return 0;
}

extern "C" int cdlrm_event_elapsed_us(void* start, void* stop, float* us) {
    CDLRM_REQUIRE(start && stop && us, "null argument");
    float ms = 0.f;
    CDLRM_HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
    CDLRM_HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
    *us = ms * 1e3f;
    return 0;
}

// ---- self-test of the typed call: interleaved float / int32 / pointer / int64 parameters, a cell in two places -------
static int64_t g_probe_sum;
static double g_probe_fsum;

extern "C" int cdlrm_tape_probe(float f0, int64_t a0, float f1, int32_t a1, void* a2, int64_t a3, float f2, int32_t a4,
                                int64_t a5, int64_t a6, void* a7, int32_t a8, float f3, int64_t a9) {
    g_probe_sum = a0 + 3 * (int64_t)a1 + 5 * (int64_t)(intptr_t)a2 + 7 * a3 + 11 * (int64_t)a4 + 13 * a5 + 17 * a6 +
                  19 * (int64_t)(intptr_t)a7 + 23 * (int64_t)a8 + 29 * a9;
    g_probe_fsum = (double)f0 + 2.0 * f1 + 4.0 * f2 + 8.0 * f3;
    return 0;
}

// a probe that logs the ORDER in which replayed calls are issued (two-lane tests; no GPU involved)
static std::atomic<int64_t> g_log_n{0};
static int64_t g_log[4096];
extern "C" int cdlrm_tape_probe_log(int64_t tag, int64_t spin) {
    for (volatile int64_t i = 0; i < spin; ++i) {}
    const int64_t k = g_log_n.fetch_add(1);
    if (k < 4096) g_log[k] = tag;
    return 0;
}
extern "C" int64_t cdlrm_tape_probe_log_take(int64_t* out, int64_t cap) {
    const int64_t n = g_log_n.exchange(0);
    for (int64_t i = 0; i < n && i < cap && i < 4096; ++i) out[i] = g_log[i];
    return n;
}

extern "C" int cdlrm_tape_selftest(void) {
    cdlrm_tape* t = cdlrm_tape_create(2);
    if (!t) return CDLRM_EINVAL;
    const int64_t ia[10] = {1000000000007LL, -5, 0x7f00deadbeefLL, 0, 77, -(1LL << 40), 12345, 0, -9, 1LL << 50};
    int32_t cell[10] = {-1, -1, -1, 0, -1, -1, -1, 1, -1, -1};
    const float fa[4] = {1.5f, -2.25f, 1024.f, 0.125f};
    int rc = cdlrm_tape_add(t, (void*)cdlrm_tape_probe, 10, ia, cell, 4, fa);
    if (!rc) {
        cdlrm_tape_cells(t)[0] = 424242424242LL;
        cdlrm_tape_cells(t)[1] = 0x7e00cafe0000LL;
        g_probe_sum = 0; g_probe_fsum = 0;
        rc = cdlrm_tape_replay(t);
        const int64_t want = ia[0] + 3 * ia[1] + 5 * ia[2] + 7 * 424242424242LL + 11 * ia[4] + 13 * ia[5] + 17 * ia[6] +
                             19 * 0x7e00cafe0000LL + 23 * ia[8] + 29 * ia[9];
        const double wantf = 1.5 + 2.0 * -2.25 + 4.0 * 1024.0 + 8.0 * 0.125;
        if (!rc && (g_probe_sum != want || g_probe_fsum != wantf)) {
            cdlrm_set_error("cdlrm_tape_selftest: the typed call does not reach its arguments");
            rc = CDLRM_EINVAL;
        }
    }
    cdlrm_tape_destroy(t);
    return rc;
}
