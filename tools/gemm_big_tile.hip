// Experiment (round 5): the LDS-DMA GEMM (gemm_glds.h: k_gemm2) on LARGER macro tiles -- per wave 128x64 / 64x128 / 128x128
// accumulator tiles (the vendor library's pick for these shapes is a 256x256 macro tile, 128x128 per wave, one workgroup per CU,
// profiles/r05_gemm_vs_vendor.json) -- against the production 128x64 tile, forward and dgrad layouts, M = 8192 and 65536.
// ... and the persistent form of the kernel (tools/gemm2p_experiment.h).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I cdlrm_amd/csrc -I include -I tools tools/gemm_big_tile.hip -o build_tmp/gbt
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <functional>
#include <vector>
#include "gemm2p_experiment.h"

int g_cdlrm_debug[8] = {0, 0, 0, 0, 0, 0, 0, 0};
void cdlrm_set_error(const char* fmt, ...) {}
CdlrmStopState* cdlrm_stop_state() {
    static thread_local CdlrmStopState st{nullptr, nullptr, 0};
    return &st;
}

static float* dev_rand(size_t n, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)((s >> 8) & 0xffff) / 32768.f - 1.f; }
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
static double time_us(const std::function<void()>& f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipEventDestroy(e0); hipEventDestroy(e1); return ms * 1e3 / reps;
}
template <bool A_KC, bool B_KC, int TM, int TN>
static void launch(const GemmArgs& g) {
    dim3 grid((unsigned)cdiv(g.N, 64 * TN), (unsigned)cdiv(g.M, 64 * TM), 1);
    hipLaunchKernelGGL((k_gemm2<A_KC, B_KC, TM, TN>), grid, dim3(256), 0, 0, g);
}
template <bool A_KC, bool B_KC, int TM, int TN>
static void launch_p(const GemmArgs& g, unsigned slots) {
    const unsigned ntx = (unsigned)cdiv(g.N, 64 * TN), nty = (unsigned)cdiv(g.M, 64 * TM);
    const unsigned ntiles = ntx * nty;
    hipLaunchKernelGGL((k_gemm2p<A_KC, B_KC, TM, TN>), dim3(std::min(slots, ntiles)), dim3(256), 0, 0, g, ntx, ntiles);
}
template <bool A_KC, bool B_KC>
static void run(const char* name, int64_t M, int N, int64_t K) {
    float* A = dev_rand((size_t)M * K, 1);
    float* B = dev_rand((size_t)N * K, 2);
    float* bias = dev_rand(N, 3);
    float* mask = dev_rand((size_t)M * N, 4);
    float *C0, *C1; hipMalloc(&C0, (size_t)M * N * 4); hipMalloc(&C1, (size_t)M * N * 4);
    GemmArgs g = gemm_args();
    g.A = A; g.lda = K; g.B = B; g.ldb = B_KC ? K : N; g.ldc = N; g.slab = (int64_t)M * N; g.M = M; g.N = N; g.K = K; g.kchunk = K;
    g.vecA = g.vecB = 1;
    if (B_KC) { g.bias = bias; g.act = 1; } else { g.mask = mask; g.ldmask = N; g.mask_act = 1; }
    g.fastep = 0;               // (the reference launch and the plain variants: the epilogue before round 5)
    GemmArgs g0 = g, g1 = g; g0.C = C0; g1.C = C1;
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    launch<A_KC, B_KC, 2, 1>(g0); hipDeviceSynchronize();
    hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost);
    const double fl = 2.0 * M * N * K;
    auto one = [&](const char* tile, std::function<void()> f) {
        hipMemset(C1, 0xff, (size_t)M * N * 4);
        f(); hipDeviceSynchronize();
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { printf("%s %s: %s\n", name, tile, hipGetErrorString(e)); return; }
        hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost);
        double maxd = 0; for (size_t i = 0; i < h0.size(); ++i) maxd = std::max(maxd, (double)fabsf(h0[i] - h1[i]));
        std::vector<double> t; for (int r = 0; r < 5; ++r) t.push_back(time_us(f, M > 8192 ? 10 : 30));
        std::sort(t.begin(), t.end());
        printf("%-28s M=%6lld N=%4d K=%4lld tile %-14s %8.1f us %6.1f TF/s  maxdiff vs 128x64 %g\n", name, (long long)M, N, (long long)K, tile,
               t[2], fl / t[2] / 1e6, maxd);
        fflush(stdout);
    };
    one("128x64", [&]() { launch<A_KC, B_KC, 2, 1>(g1); });
    GemmArgs gf = g1;
    gf.fastep = 1;              // the loads-first epilogue of full tiles (production default)
    one("128x64 fastep", [&]() { launch<A_KC, B_KC, 2, 1>(gf); });
    one("64x64", [&]() { launch<A_KC, B_KC, 1, 1>(g1); });
    one("64x64 fastep", [&]() { launch<A_KC, B_KC, 1, 1>(gf); });
    one("128x128", [&]() { launch<A_KC, B_KC, 2, 2>(g1); });
    one("p128x64 G512", [&]() { launch_p<A_KC, B_KC, 2, 1>(g1, 512); });
    one("p128x64 G768", [&]() { launch_p<A_KC, B_KC, 2, 1>(g1, 768); });
    one("p128x64 G256", [&]() { launch_p<A_KC, B_KC, 2, 1>(g1, 256); });
    one("p128x128 G256", [&]() { launch_p<A_KC, B_KC, 2, 2>(g1, 256); });
    one("p128x128 G512", [&]() { launch_p<A_KC, B_KC, 2, 2>(g1, 512); });
    one("p64x64 G1024", [&]() { launch_p<A_KC, B_KC, 1, 1>(g1, 1024); });
    hipFree(A); hipFree(B); hipFree(bias); hipFree(mask); hipFree(C0); hipFree(C1);
}
int main() {
    for (int64_t M : {(int64_t)65536, (int64_t)16384, (int64_t)8192}) {
        run<true, true>("forward 512<-512", M, 512, 512);
        run<true, false>("dgrad 512<-512", M, 512, 512);
        run<true, true>("forward 512<-480", M, 512, 480);
        run<true, false>("dgrad 480<-512", M, 480, 512);
        run<true, true>("forward 256<-512", M, 256, 512);
    }
    return 0;
}
