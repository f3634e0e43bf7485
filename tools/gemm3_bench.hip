// Development aid (round 6): the wide FP32-MFMA GEMM (cdlrm_amd/csrc/gemm_wide.h: k_gemm3) against k_gemm2 on the forward /
// dgrad shapes of configs c3 and c5: results against k_gemm2 (relative, the k order differs) and against an fp64 host sum on
// sampled elements; timings from interleaved rounds in one process; with -DG3_STAMP where a workgroup's cycles go.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I cdlrm_amd/csrc -I include tools/gemm3_bench.hip -o build_tmp/g3
//   hipcc ... -DG3_STAMP ... -o build_tmp/g3s
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <functional>
#include <vector>
#include "gemm_wide.h"

int g_cdlrm_debug[8] = {0, 0, 0, 0, 0, 0, 0, 0};
void cdlrm_set_error(const char* fmt, ...) {}
CdlrmStopState* cdlrm_stop_state() {
    static thread_local CdlrmStopState st{nullptr, nullptr, 0};
    return &st;
}

static std::vector<float> host_rand(size_t n, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)((s >> 8) & 0xffff) / 32768.f - 1.f; }
    return h;
}
static float* to_dev(const std::vector<float>& h) {
    float* d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); return d;
}
static double time_us(const std::function<void()>& f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipEventDestroy(e0); hipEventDestroy(e1); return ms * 1e3 / reps;
}
template <bool A_KC, bool B_KC, int TM, int TN>
static void launch2(const GemmArgs& g) {
    dim3 grid((unsigned)cdiv(g.N, 64 * TN), (unsigned)cdiv(g.M, 64 * TM), 1);
    hipLaunchKernelGGL((k_gemm2<A_KC, B_KC, TM, TN>), grid, dim3(256), 0, 0, g);
}

#ifdef G3_STAMP
static void stamp_report(const char* what, unsigned nwg) {
    std::vector<unsigned long long> h(8 * 4096);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g3_stamps), h.size() * 8);
    nwg = std::min(nwg, 4096u);
    std::vector<double> pro, loop, epi, drain, clk;
    unsigned long long first = ~0ull, last = 0;
    for (unsigned w = 0; w < nwg; ++w) {
        const unsigned long long* s = &h[w * 8];
        pro.push_back((double)(s[1] - s[0])); loop.push_back((double)(s[2] - s[1])); epi.push_back((double)(s[3] - s[2]));
        drain.push_back((double)(s[4] - s[3]));
        clk.push_back((double)(s[4] - s[0]) / ((double)(s[6] - s[5]) * 10.0));     // cycles per ns (memrealtime: 100 MHz)
        first = std::min(first, s[5]); last = std::max(last, s[6]);
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
    printf("   stamps %-22s prologue %6.0f  loop %7.0f  epilogue %6.0f (max %6.0f)  drain %6.0f (max %6.0f)  clock %.2f GHz  first start -> last end %.1f us\n",
           what, med(pro), med(loop), med(epi), mx(epi), med(drain), mx(drain), med(clk), (double)(last - first) / 100.0);
}
#endif

template <bool B_KC>
static void run(const char* name, int64_t M, int N, int64_t K) {
    std::vector<float> hA = host_rand((size_t)M * K, 1), hB = host_rand((size_t)N * K, 2), hbias = host_rand(N, 3),
                       hmask = host_rand((size_t)M * N, 4);
    float *A = to_dev(hA), *B = to_dev(hB), *bias = to_dev(hbias), *mask = to_dev(hmask);
    float *C0, *C1; hipMalloc(&C0, (size_t)M * N * 4); hipMalloc(&C1, (size_t)M * N * 4);
    GemmArgs g = gemm_args();
    g.A = A; g.lda = K; g.B = B; g.ldb = B_KC ? K : N; g.ldc = N; g.slab = (int64_t)M * N; g.M = M; g.N = N; g.K = K; g.kchunk = K;
    g.vecA = g.vecB = 1;
    if (B_KC) { g.bias = bias; g.act = 1; } else { g.mask = mask; g.ldmask = N; g.mask_act = 1; }
    g.fastep = 1;
    GemmArgs g0 = g, g1 = g; g0.C = C0; g1.C = C1;
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    launch2<true, B_KC, 2, 1>(g0); hipDeviceSynchronize();
    hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost);
    const double fl = 2.0 * M * N * K;
    auto one = [&](const char* tile, std::function<void()> f, unsigned nwg) {
        hipMemset(C1, 0xff, (size_t)M * N * 4);
        f(); hipDeviceSynchronize();
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { printf("%s %s: %s\n", name, tile, hipGetErrorString(e)); return; }
        hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost);
        double maxrel = 0, maxabs = 0;
        size_t nbad = 0;
        for (size_t i = 0; i < h0.size(); ++i) {
            const double d = fabs((double)h0[i] - (double)h1[i]);
            maxabs = std::max(maxabs, d);
            const double r = d / std::max(1.0, fabs((double)h0[i]));
            maxrel = std::max(maxrel, r);
            if (r > 1e-3) {
                if (nbad < 6) printf("   bad at m=%zu n=%zu: %g vs %g\n", i / N, i % N, h1[i], h0[i]);
                ++nbad;
            }
        }
        if (nbad) printf("   %zu bad elements of %zu\n", nbad, h0.size());
        // fp64 reference on sampled elements
        double max64 = 0;
        unsigned s = 99;
        for (int q = 0; q < 2000; ++q) {
            s = s * 1664525u + 1013904223u; const int64_t m = (s >> 4) % M;
            s = s * 1664525u + 1013904223u; const int n = (s >> 4) % N;
            double acc = 0;
            for (int64_t k = 0; k < K; ++k) acc += (double)hA[m * K + k] * (double)(B_KC ? hB[(size_t)n * K + k] : hB[(size_t)k * N + n]);
            if (B_KC) { acc += hbias[n]; acc = acc > 0 ? acc : 0; } else { acc = hmask[(size_t)m * N + n] > 0.f ? acc : 0; }
            max64 = std::max(max64, fabs(acc - (double)h1[(size_t)m * N + n]) / std::max(1.0, fabs(acc)));
        }
        std::vector<double> t; for (int r = 0; r < 5; ++r) t.push_back(time_us(f, M > 8192 ? 10 : 30));
        std::sort(t.begin(), t.end());
        printf("%-20s M=%6lld N=%4d K=%4lld %-18s %8.1f us %6.1f TF/s  vs k_gemm2 rel %.2e  vs fp64 rel %.2e\n", name, (long long)M, N,
               (long long)K, tile, t[2], fl / t[2] / 1e6, maxrel, max64);
#ifdef G3_STAMP
        if (nwg) { f(); hipDeviceSynchronize(); stamp_report(tile, nwg); }
#endif
        fflush(stdout);
    };
    one("gemm2 128x64", [&]() { launch2<true, B_KC, 2, 1>(g1); }, 0);
    one("gemm2 128x128", [&]() { launch2<true, B_KC, 2, 2>(g1); }, 0);
    one("gemm3 128x128 S3", [&]() { launch_gemm3<true, B_KC, 4, 4, 3>(g1, 1, 0); }, (unsigned)(cdiv(N, 128) * cdiv(M, 128)));
    one("gemm3 128x128 S4", [&]() { launch_gemm3<true, B_KC, 4, 4, 4>(g1, 1, 0); }, (unsigned)(cdiv(N, 128) * cdiv(M, 128)));
    one("gemm2 64x64", [&]() { launch2<true, B_KC, 1, 1>(g1); }, 0);
    one("gemm3 64x128 S3", [&]() { launch_gemm3<true, B_KC, 2, 4, 3>(g1, 1, 0); }, (unsigned)(cdiv(N, 128) * cdiv(M, 64)));
    one("gemm3 64x128 S4", [&]() { launch_gemm3<true, B_KC, 2, 4, 4>(g1, 1, 0); }, (unsigned)(cdiv(N, 128) * cdiv(M, 64)));
    hipFree(A); hipFree(B); hipFree(bias); hipFree(mask); hipFree(C0); hipFree(C1);
}
// weight-gradient layout: dW[N, K] = dZ[M, N]^T X[M, K], split over the batch into `splits` slabs (+ the bias gradient's partials)
static void run_wgrad(const char* name, int64_t M, int N, int K, int splits) {
    std::vector<float> hZ = host_rand((size_t)M * N, 5), hX = host_rand((size_t)M * K, 6);
    float *dZ = to_dev(hZ), *X = to_dev(hX);
    const size_t cnt = (size_t)N * K;
    float *S0, *S1, *c0, *c1;
    hipMalloc(&S0, cnt * splits * 4); hipMalloc(&S1, cnt * splits * 4); hipMalloc(&c0, (size_t)splits * N * 4); hipMalloc(&c1, (size_t)splits * N * 4);
    GemmArgs g = gemm_args();
    g.A = dZ; g.lda = N; g.B = X; g.ldb = K; g.ldc = K; g.slab = (int64_t)cnt; g.M = N; g.N = K; g.K = M;
    g.kchunk = cdiv(cdiv(M, splits), 32) * 32; g.vecA = g.vecB = 1;
    const int zs = (int)cdiv(M, g.kchunk);
    GemmArgs g0 = g, g1 = g; g0.C = S0; g0.colsum = c0; g1.C = S1; g1.colsum = c1;
    auto l2 = [&](const GemmArgs& a) {
        dim3 grid((unsigned)cdiv(a.N, 64), (unsigned)cdiv(a.M, 128), (unsigned)zs);
        hipLaunchKernelGGL((k_gemm2<false, false, 2, 1>), grid, dim3(256), 0, 0, a);
    };
    auto l3 = [&](const GemmArgs& a) { launch_gemm3<false, false, 4, 4, 3>(a, zs, 0); };
    l2(g0); hipMemset(S1, 0xff, cnt * zs * 4); hipMemset(c1, 0xff, (size_t)zs * N * 4); l3(g1); hipDeviceSynchronize();
    std::vector<float> h0(cnt * zs), h1(cnt * zs), k0((size_t)zs * N), k1((size_t)zs * N);
    hipMemcpy(h0.data(), S0, h0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), S1, h1.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(k0.data(), c0, k0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(k1.data(), c1, k1.size() * 4, hipMemcpyDeviceToHost);
    double maxrel = 0, maxcs = 0;
    for (size_t i = 0; i < h0.size(); ++i) maxrel = std::max(maxrel, fabs((double)h0[i] - h1[i]) / std::max(1.0, fabs((double)h0[i])));
    for (size_t i = 0; i < k0.size(); ++i) maxcs = std::max(maxcs, fabs((double)k0[i] - k1[i]) / std::max(1.0, fabs((double)k0[i])));
    // fp64 on sampled elements of the summed slabs
    double max64 = 0; unsigned s = 7;
    for (int q = 0; q < 500; ++q) {
        s = s * 1664525u + 1013904223u; const int n = (s >> 4) % N;
        s = s * 1664525u + 1013904223u; const int k = (s >> 4) % K;
        double ref = 0, got = 0;
        for (int64_t m = 0; m < M; ++m) ref += (double)hZ[m * N + n] * (double)hX[m * K + k];
        for (int z = 0; z < zs; ++z) got += h1[(size_t)z * cnt + (size_t)n * K + k];
        max64 = std::max(max64, fabs(ref - got) / std::max(1.0, fabs(ref)));
    }
    std::vector<double> t2, t3;
    for (int r = 0; r < 5; ++r) { t2.push_back(time_us([&]() { l2(g0); }, 30)); t3.push_back(time_us([&]() { l3(g1); }, 30)); }
    std::sort(t2.begin(), t2.end()); std::sort(t3.begin(), t3.end());
    const double fl = 2.0 * M * N * K;
    printf("%-20s M=%6lld N=%4d K=%4d splits %3d  gemm2 128x64 %7.1f us %6.1f TF/s | gemm3 %7.1f us %6.1f TF/s  slabs rel %.2e  colsum rel %.2e  fp64 rel %.2e\n",
           name, (long long)M, N, K, zs, t2[2], fl / t2[2] / 1e6, t3[2], fl / t3[2] / 1e6, maxrel, maxcs, max64);
#ifdef G3_STAMP
    l3(g1); hipDeviceSynchronize(); stamp_report("wgrad gemm3", (unsigned)(cdiv(N, 128) * cdiv(K, 128) * zs));
#endif
    fflush(stdout);
    hipFree(dZ); hipFree(X); hipFree(S0); hipFree(S1); hipFree(c0); hipFree(c1);
}
int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'n') {       // the narrow layers: 64x128 tiles
        run<true>("forward 256<-512", 8192, 256, 512);
        run<false>("dgrad 256<-128", 8192, 256, 128);
        run<true>("forward 128<-256", 8192, 128, 256);
        run<true>("forward 512<-512", 4096, 512, 512);
        run<false>("dgrad 512<-512", 4096, 512, 512);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'w') {
        run_wgrad("wgrad 512x512", 8192, 512, 512, 16);
        run_wgrad("wgrad 512x480", 8192, 512, 480, 16);
        run_wgrad("wgrad 256x512", 8192, 256, 512, 32);
        run_wgrad("wgrad 128x256", 8192, 128, 256, 128);
        run_wgrad("wgrad 512x512", 65536, 512, 512, 16);
        return 0;
    }
    const bool quick = argc > 1;
    for (int64_t M : {(int64_t)8192, (int64_t)65536}) {
        run<true>("forward 512<-512", M, 512, 512);
        run<false>("dgrad 512<-512", M, 512, 512);
        if (quick) continue;
        run<true>("forward 512<-480", M, 512, 480);
        run<false>("dgrad 480<-512", M, 480, 512);
        run<false>("dgrad 512<-256", M, 512, 256);
    }
    return 0;
}
