// Development aid: time the LDS-free small-batch GEMM (k_gemm_direct, cdlrm_amd/csrc/gemm.h) stand-alone on the three
// operand layouts of a Linear layer (forward, dgrad, wgrad), with ablation switches.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDABL=<n> -I cdlrm_amd/csrc tools/gemm_direct_ablate.hip -o /tmp/gd
//   /tmp/gd [batch] [out_features] [in_features]
// DABL: 0 full kernel, 1 no global loads inside the loop, 2 no MFMAs (loads + adds)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#ifndef DABL
#define DABL 0
#endif
#define DIRECT_ABLATE DABL
#include "gemm.h"

void cdlrm_set_error(const char* fmt, ...) {}
int g_cdlrm_debug[8] = {0, 0, 0, 0, 0, 0, 0, 0};
CdlrmStopState* cdlrm_stop_state() {
    static thread_local CdlrmStopState st{nullptr, nullptr, 0};
    return &st;
}

template <typename F>
static double time_us(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 100; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / 100;
}

int main(int argc, char** argv) {
    int64_t B = 1024; int N = 512, K = 512;
    if (argc > 3) { B = atol(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    float *X, *W, *Y, *dY, *dX, *dW, *b;
    hipMalloc(&X, B * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&Y, B * N * 4); hipMalloc(&b, N * 4);
    hipMalloc(&dY, B * N * 4); hipMalloc(&dX, B * K * 4); hipMalloc(&dW, (size_t)N * K * 4);
    size_t big = (size_t)(B > N ? B : N) * (K > N ? K : N);
    float* h = (float*)malloc(big * 4);
    for (size_t i = 0; i < big; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(X, h, B * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h, (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(dY, h, B * N * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, N * 4);
    const double fl = 2.0 * B * N * K;
    {
        GemmArgs g = gemm_args();
        g.A = X; g.lda = K; g.B = W; g.ldb = K; g.C = Y; g.ldc = N; g.M = B; g.N = N; g.K = K; g.kchunk = K;
        g.bias = b; g.act = 1; g.vecA = g.vecB = 1;
        const double us = time_us([&]() { launch_gemm_direct<true, true>(g, 1, 0); });
        printf("DABL=%d fwd   %ldx%dx%d  %7.1f us %6.1f TF\n", DABL, (long)B, N, K, us, fl / us / 1e6);
    }
    {
        GemmArgs g = gemm_args();
        g.A = dY; g.lda = N; g.B = W; g.ldb = K; g.C = dX; g.ldc = K; g.M = B; g.N = K; g.K = N; g.kchunk = N;
        g.vecA = g.vecB = 1; g.mask = X; g.ldmask = K; g.mask_act = 1;
        const double us = time_us([&]() { launch_gemm_direct<true, false>(g, 1, 0); });
        printf("DABL=%d dgrad %ldx%dx%d  %7.1f us %6.1f TF\n", DABL, (long)B, N, K, us, fl / us / 1e6);
    }
    {
        GemmArgs g = gemm_args();
        g.A = dY; g.lda = N; g.B = X; g.ldb = K; g.C = dW; g.ldc = K; g.M = N; g.N = K; g.K = B; g.kchunk = B;
        g.vecA = g.vecB = 1; g.colsum = b;
        const double us = time_us([&]() { launch_gemm_direct<false, false>(g, 1, 0); });
        printf("DABL=%d wgrad %ldx%dx%d  %7.1f us %6.1f TF\n", DABL, (long)B, N, K, us, fl / us / 1e6);
    }
    return 0;
}
