// Context, binding and error plumbing of libcdlrm_hip.so (no kernels of the hot path live here).
#include <stdarg.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <thread>

#include "common.h"

static thread_local char g_err[512] = "";

void cdlrm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int g_cdlrm_debug[8] = {0, 0, 0, 0, 0, 0, 0, 0};
extern "C" int cdlrm_debug_set(int32_t key, int32_t value) {
    CDLRM_REQUIRE(key >= 0 && key < 8, "key: 0 .. 7");
#ifndef CDLRM_DEV
    CDLRM_REQUIRE(!(key == 6 && (value & 3)), "work-skipping switches (key 6, bits 1 and 2) exist in -DCDLRM_DEV builds only");
#endif
    g_cdlrm_debug[key] = value;
    return 0;
}
extern "C" int cdlrm_abi_version(void) { return CDLRM_ABI_VERSION; }
extern "C" const char* cdlrm_last_error(void) { return g_err; }

extern "C" int cdlrm_ctx_create(const cdlrm_geometry* geo, cdlrm_ctx** out) {
    CDLRM_REQUIRE(geo && out, "null argument");
    CDLRM_REQUIRE(geo->num_tables >= 1 && geo->num_tables <= 1024, "1..1024 tables");
    CDLRM_REQUIRE(geo->dim >= 4 && geo->dim % 4 == 0, "dim must be a positive multiple of 4");
    CDLRM_REQUIRE(geo->num_ways >= 1 && geo->num_ways <= 64, "1..64 ways");
    CDLRM_REQUIRE(geo->aux_rows >= 0, "aux_rows >= 0");
    CDLRM_REQUIRE(geo->aux_phases >= 0 && geo->aux_phases <= 2, "aux_phases 0..2");
    CDLRM_HIP_CHECK(hipSetDevice(geo->device));
    cdlrm_ctx* c = new cdlrm_ctx();
    c->T = geo->num_tables; c->D = geo->dim; c->ways = geo->num_ways; c->aux = geo->aux_rows;
    c->device = geo->device;
    c->aux_phases = geo->aux_phases < 1 ? 1 : geo->aux_phases;
    int64_t tag = 0, row = 0, set = 0, bm = 0;
    for (int k = 0; k < c->T; ++k) {
        TableDesc d;
        d.n_rows = geo->table_rows[k];
        d.P = geo->cache_sets[k];
        if (d.n_rows < 1 || d.P < 1 || d.P > d.n_rows ||
            d.P * (int64_t)c->ways + (int64_t)c->aux * c->aux_phases >= ((int64_t)1 << 31)) {
            delete c;
            cdlrm_set_error("cdlrm_ctx_create: table %d has unsupported geometry (n=%lld P=%lld)", k,
                            (long long)geo->table_rows[k], (long long)geo->cache_sets[k]);
            return CDLRM_EINVAL;
        }
        d.tag_base = tag; d.row_base = row; d.set_base = set; d.bm_base = bm;
        d.bm_words = ((d.n_rows + 63) / 64 + BM_WPB - 1) / BM_WPB * BM_WPB;
        d.rows = d.P * c->ways + (int64_t)c->aux * c->aux_phases;
        tag += d.P * c->ways; row += d.rows; set += d.P; bm += d.bm_words;
        c->h_tab.push_back(d);
    }
    c->total_tags = tag; c->total_rows = row; c->total_sets = set; c->total_bm_words = bm;
    c->h_host_rows.assign(c->T, nullptr);
    CDLRM_HIP_CHECK(hipMalloc(&c->d_tab, sizeof(TableDesc) * c->T));
    CDLRM_HIP_CHECK(hipMemcpy(c->d_tab, c->h_tab.data(), sizeof(TableDesc) * c->T, hipMemcpyHostToDevice));
    CDLRM_HIP_CHECK(hipMalloc(&c->d_host_rows, sizeof(float*) * c->T));
    CDLRM_HIP_CHECK(hipMemset(c->d_host_rows, 0, sizeof(float*) * c->T));
    CDLRM_HIP_CHECK(hipMalloc(&c->d_ptr_fetch, sizeof(float*) * c->T));
    CDLRM_HIP_CHECK(hipMalloc(&c->d_ptr_wb, sizeof(float*) * c->T));
    CDLRM_HIP_CHECK(hipMalloc(&c->d_err, sizeof(int)));
    CDLRM_HIP_CHECK(hipMemset(c->d_err, 0, sizeof(int)));
    CDLRM_HIP_CHECK(hipMalloc(&c->d_small, sizeof(int64_t) * 256));
    CDLRM_HIP_CHECK(hipMemset(c->d_small, 0, sizeof(int64_t) * 256));
    CDLRM_HIP_CHECK(hipHostMalloc((void**)&c->h_pinned, sizeof(int64_t) * 4096, hipHostMallocDefault));
    int rc = cdlrm_scan_reserve(c, 1 << 16);
    if (rc) { delete c; return rc; }
    *out = c;
    return 0;
}

extern "C" int cdlrm_ctx_destroy(cdlrm_ctx* c) {
    if (!c) return 0;
    // Called from the host language's garbage collector at arbitrary points, possibly between two launches of ANOTHER
    // context's step: nothing may escape -- no C++ exception across the C ABI (the runtime threw
    // std::bad_variant_access out of a free issued while a failed step was being unwound), no sticky error for
    // somebody else's next launch check (hipGetLastError() is per thread, not per context).
    try {
        (void)hipDeviceSynchronize();          // kernels that still read the scratch buffers
        if (c->d_tab) (void)hipFree(c->d_tab);
        if (c->d_host_rows) (void)hipFree(c->d_host_rows);
        if (c->d_ptr_fetch) (void)hipFree(c->d_ptr_fetch);
        if (c->d_ptr_wb) (void)hipFree(c->d_ptr_wb);
        if (c->d_err) (void)hipFree(c->d_err);
        if (c->d_small) (void)hipFree(c->d_small);
        if (c->d_scan) (void)hipFree(c->d_scan);
        if (c->d_scan_agg) (void)hipFree(c->d_scan_agg);
        if (c->h_pinned) (void)hipHostFree(c->h_pinned);
        (void)hipGetLastError();
    } catch (...) {
    }
    delete c;
    return 0;
}

extern "C" int cdlrm_ctx_bind_cache(cdlrm_ctx* c, int64_t* tags, float* weight) {
    CDLRM_REQUIRE(c && tags && weight, "null argument");
    CDLRM_REQUIRE(((uintptr_t)weight & 15) == 0 && ((uintptr_t)tags & 15) == 0, "16-byte aligned buffers");
    c->tags = tags;
    c->weight = weight;
    return 0;
}

extern "C" int cdlrm_ctx_bind_host_tables(cdlrm_ctx* c, float* const* host_rows) {
    CDLRM_REQUIRE(c && host_rows, "null argument");
    for (int k = 0; k < c->T; ++k) {
        CDLRM_REQUIRE(host_rows[k] != nullptr && ((uintptr_t)host_rows[k] & 15) == 0, "16-byte aligned host tables");
        c->h_host_rows[k] = host_rows[k];
    }
    CDLRM_HIP_CHECK(hipMemcpy(c->d_host_rows, c->h_host_rows.data(), sizeof(float*) * c->T, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int cdlrm_host_register(void* host_ptr, uint64_t bytes, void** device_alias) {
    CDLRM_REQUIRE(host_ptr && bytes && device_alias, "null argument");
    CDLRM_HIP_CHECK(hipHostRegister(host_ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    CDLRM_HIP_CHECK(hipHostGetDevicePointer(device_alias, host_ptr, 0));
    return 0;
}

extern "C" int cdlrm_host_unregister(void* host_ptr) {
    CDLRM_REQUIRE(host_ptr, "null argument");
    CDLRM_HIP_CHECK(hipHostUnregister(host_ptr));
    return 0;
}

// Host-side row gather for the plan's bulk fetches: dst[j, :] = tables[t(j)][idx[j], :], t(j) from off[0..T].
// The window plan can move its winners' and victims' rows this way -- CPU threads gather into a pinned staging buffer,
// one DMA copies it to HBM -- instead of letting GPU waves read the host tables: a copy-engine transfer beside training
// costs the training kernels ~2 %, shader reads over PCIe cost them ~2x for as long as they run (measured).
// This is what the reference's Prefetcher process does on the CPU (model_no_ddp.py:80-87) -- here it overlaps training.
static void host_gather_range(const float* const* tables, const int64_t* idx, const int64_t* off, int T, int D, float* dst,
                              int64_t a, int64_t b, int background) {
    // A worker thread of the plan's row gather runs at the LOWEST scheduling weight (nice 19, this thread only): the gather has
    // ~2 s of slack per window, the thread that issues the training step has none, and the box grants the process 16 CPUs
    // (cgroup quota; os.cpu_count() says 256) -- at equal weight beside 32 gather threads the issuing thread would get half a
    // core for the 0.1-0.2 s a gather lasts.
    if (background) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), 19);
    int t = 0;
    const size_t row_bytes = (size_t)D * sizeof(float);
    constexpr int AHEAD = 16;
    int ta = 0;                                   // table of the row being prefetched
    for (int64_t j = a; j < b; ++j) {
        const int64_t jp = j + AHEAD;
        if (jp < b) {
            while (ta < T - 1 && jp >= off[ta + 1]) ++ta;
            const char* p = (const char*)(tables[ta] + idx[jp] * D);
            for (size_t o = 0; o < row_bytes; o += 64) __builtin_prefetch(p + o, 0, 0);
        }
        while (t < T - 1 && j >= off[t + 1]) ++t;
        memcpy(dst + j * D, tables[t] + idx[j] * D, row_bytes);
    }
}

extern "C" int cdlrm_host_gather_rows(const float* const* tables, const int64_t* idx, const int64_t* off, int32_t T,
                                      int32_t D, float* dst, int32_t nthreads) {
    CDLRM_REQUIRE(tables && idx && off && dst && T >= 1 && D >= 1, "bad argument");
    const int64_t n = off[T];
    if (n <= 0) return 0;
    for (int k = 0; k < T; ++k) CDLRM_REQUIRE(off[k] <= off[k + 1] && (off[k] == off[k + 1] || tables[k]), "bad offsets / table");
    int nt = nthreads < 1 ? 1 : nthreads;
    if ((int64_t)nt > (n + 4095) / 4096) nt = (int)((n + 4095) / 4096);
    if (nt <= 1) {
        host_gather_range(tables, idx, off, T, D, dst, 0, n, 0);
        return 0;
    }
    std::vector<std::thread> th;
    const int64_t per = (n + nt - 1) / nt;
    for (int i = 0; i < nt; ++i) {
        const int64_t a = i * per, b = std::min<int64_t>(n, a + per);
        if (a >= b) break;
        th.emplace_back(host_gather_range, tables, idx, off, (int)T, (int)D, dst, a, b, 1);
    }
    for (auto& x : th) x.join();
    return 0;
}

extern "C" int cdlrm_ctx_check_sync(cdlrm_ctx* c, void* stream) {
    CDLRM_REQUIRE(c, "null ctx");
    hipStream_t s = (hipStream_t)stream;
    int* h = (int*)(c->h_pinned + 4000);
    CDLRM_HIP_CHECK(hipMemcpyAsync(h, c->d_err, sizeof(int), hipMemcpyDeviceToHost, s));
    CDLRM_HIP_CHECK(hipMemsetAsync(c->d_err, 0, sizeof(int), s));
    CDLRM_HIP_CHECK(hipStreamSynchronize(s));
    if (*h != 0) {
        cdlrm_set_error("device error word = %d (1: index outside its table, 2: more misses than aux rows, "
                        "4: plan capacity exceeded)", *h);
        return CDLRM_ERANGE;
    }
    return 0;
}
