// Development aid: time the FP32-MFMA GEMM of cdlrm_amd/csrc/gemm.h stand-alone, with ablation switches
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DABL=<n> -I cdlrm_amd/csrc tools/gemm_ablate.hip -o /tmp/ga && /tmp/ga
// ABL: 0 full kernel, 1 no global loads in the loop, 2 no LDS stores/barriers in the loop, 3 neither (MFMA+LDS reads)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#ifndef ABL
#define ABL 0
#endif
#define GEMM_ABLATE ABL
#include "gemm.h"

void cdlrm_set_error(const char* fmt, ...) {}
int g_cdlrm_debug[8] = {0, 0, 0, 0, 0, 0, 0, 0};
CdlrmStopState* cdlrm_stop_state() {
    static thread_local CdlrmStopState st{nullptr, nullptr, 0};
    return &st;
}

int main(int argc, char** argv) {
    int64_t M = 8192; int N = 512, K = 512;
    if (argc > 3) { M = atol(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    float *X, *W, *Y, *b;
    hipMalloc(&X, M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&Y, M * N * 4); hipMalloc(&b, N * 4);
    hipMemset(X, 0, M * K * 4); hipMemset(W, 0, (size_t)N * K * 4); hipMemset(b, 0, N * 4);
    // random-ish data (zeros clock higher: cdna_hip_programming.md rule 25)
    float* h = (float*)malloc(M * K * 4);
    for (int64_t i = 0; i < M * K; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(X, h, M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h, (size_t)N * K * 4, hipMemcpyHostToDevice);
    GemmArgs g = gemm_args();
    g.A = X; g.lda = K; g.B = W; g.ldb = K; g.C = Y; g.ldc = N; g.slab = 0;
    g.M = M; g.N = N; g.K = K; g.kchunk = K; g.bias = b; g.act = 1; g.vecA = 1; g.vecB = 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int tm = 2; tm >= 1; --tm)
        for (int tn = 2; tn >= 1; --tn) {
            dim3 grid((N + 64 * tn - 1) / (64 * tn), (M + 64 * tm - 1) / (64 * tm), 1);
            auto launch = [&]() {
                if (tm == 2 && tn == 2) hipLaunchKernelGGL((k_gemm<true, true, 2, 2, true, true>), grid, dim3(256), 0, 0, g);
                else if (tm == 2) hipLaunchKernelGGL((k_gemm<true, true, 2, 1, true, true>), grid, dim3(256), 0, 0, g);
                else if (tn == 2) hipLaunchKernelGGL((k_gemm<true, true, 1, 2, true, true>), grid, dim3(256), 0, 0, g);
                else hipLaunchKernelGGL((k_gemm<true, true, 1, 1, true, true>), grid, dim3(256), 0, 0, g);
            };
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / 50;
            printf("ABL=%d tile %dx%d  grid %u  %8.1f us  %6.1f TF\n", ABL, 64 * tm, 64 * tn, grid.x * grid.y, us,
                   2.0 * M * N * K / us / 1e6);
        }
    return 0;
}
