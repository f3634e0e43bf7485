// Development aid: the LDS-DMA GEMM (cdlrm_amd/csrc/gemm_glds.h: k_gemm2) against the register-staged one (gemm.h:
// k_gemm) on the MLP layer shapes of config c3, the three operand layouts (forward, dgrad, wgrad), every tile shape:
// results compared element by element, timings from interleaved rounds in one process.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I cdlrm_amd/csrc tools/gemm2_bench.hip -o build_tmp/g2 && build_tmp/g2
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <vector>

static int g_extra_lds = 0;      // development: extra dynamic LDS per workgroup (caps the workgroups a CU admits)
#define G2_EXTRA_LDS(tm, tn) g_extra_lds
#include "gemm_glds.h"

int g_cdlrm_debug[8] = {0, 0, 0, 0, 0, 0, 0, 0};
void cdlrm_set_error(const char* fmt, ...) {}
CdlrmStopState* cdlrm_stop_state() {
    static thread_local CdlrmStopState st{nullptr, nullptr, 0};
    return &st;
}

static int g_data_mode = 1;      // 0 zeros, 1 uniform [-1, 1), 2 ReLU-like (half of the values zero)
static float* dev_rand(size_t n, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        float v = (float)((s >> 8) & 0xffff) / 32768.f - 1.f;
        if (g_data_mode == 0) v = 0.f;
        if (g_data_mode == 2 && v < 0.f) v = 0.f;
        h[i] = v;
    }
    float* d;
    hipMalloc(&d, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
}

template <bool A_KC, bool B_KC>
static void old_launch(const GemmArgs& g, int tm, int tn, int splits) {
    dim3 grid((unsigned)cdiv(g.N, 64 * tn), (unsigned)cdiv(g.M, 64 * tm), (unsigned)splits);
    if (tm == 2 && tn == 2) launch_gemm_v<A_KC, B_KC, 2, 2>(g, grid, 0);
    else if (tm == 2 && tn == 1) launch_gemm_v<A_KC, B_KC, 2, 1>(g, grid, 0);
    else if (tm == 1 && tn == 2) launch_gemm_v<A_KC, B_KC, 1, 2>(g, grid, 0);
    else launch_gemm_v<A_KC, B_KC, 1, 1>(g, grid, 0);
}

static double time_us(const std::function<void()>& f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3 / reps;
}

// layout 0: forward  C[M,N] = A[M,K] B[N,K]^T (+bias, ReLU)       (A_KC, B_KC)
// layout 1: dgrad    C[M,N] = A[M,K] B[K,N]   (* ReLU'(mask))     (A_KC, !B_KC)
// layout 2: wgrad    C[M,N] = A[K,M]^T B[K,N] (+ colsum, split-K) (!A_KC, !B_KC)
template <bool A_KC, bool B_KC>
static int run_case(const char* name, int64_t M, int N, int64_t K, int splits, bool timing) {
    const int64_t lda = A_KC ? K : M, ldb = B_KC ? K : N;
    float* A = dev_rand((size_t)(A_KC ? M * K : K * M), 1);
    float* B = dev_rand((size_t)(B_KC ? (int64_t)N * K : K * N), 2);
    float* bias = dev_rand(N, 3);
    float* mask = dev_rand((size_t)M * N, 4);
    float *C0, *C1, *C2, *cs0, *cs1, *cs2;
    const size_t cn = (size_t)M * N * splits;
    hipMalloc(&C0, cn * 4); hipMalloc(&C1, cn * 4); hipMalloc(&C2, cn * 4);
    hipMalloc(&cs0, (size_t)M * splits * 4); hipMalloc(&cs1, (size_t)M * splits * 4); hipMalloc(&cs2, (size_t)M * splits * 4);
    GemmArgs g = gemm_args();
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.ldc = N; g.slab = (int64_t)M * N;
    g.M = M; g.N = N; g.K = K; g.kchunk = cdiv(cdiv(K, splits), 32) * 32;
    g.vecA = 1; g.vecB = 1;
    if (A_KC && B_KC) { g.bias = bias; g.act = 1; }
    if (A_KC && !B_KC) { g.mask = mask; g.ldmask = N; g.mask_act = 1; }
    const int zs = (int)cdiv(K, g.kchunk);
    int bad = 0;
    const int tiles[4][2] = {{1, 1}, {1, 2}, {2, 1}, {2, 2}};
    std::vector<float> h0(cn), h1(cn), hc0((size_t)M * zs), hc1((size_t)M * zs);
    for (int ti = 0; ti < 4; ++ti) {
        const int tm = tiles[ti][0], tn = tiles[ti][1];
        GemmArgs g0 = g, g1 = g, g2 = g;
        g0.C = C0; g1.C = C1; g2.C = C2;
        if (!A_KC) { g0.colsum = cs0; g1.colsum = cs1; g2.colsum = cs2; }
        hipMemset(C0, 0, cn * 4); hipMemset(C1, 0xff, cn * 4); hipMemset(C2, 0xee, cn * 4);
        old_launch<A_KC, B_KC>(g0, tm, tn, zs);
        if (!gemm2_applies<A_KC, B_KC>(g1)) { printf("%s: DMA kernel does not apply\n", name); return 1; }
        launch_gemm2<A_KC, B_KC>(g1, tm, tn, zs, 0);
        hipDeviceSynchronize();
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { printf("%s tile %dx%d: %s\n", name, 64 * tm, 64 * tn, hipGetErrorString(e)); return 1; }
        hipMemcpy(h0.data(), C0, (size_t)M * N * zs * 4, hipMemcpyDeviceToHost);
        hipMemcpy(h1.data(), C1, (size_t)M * N * zs * 4, hipMemcpyDeviceToHost);
        double maxd = 0, maxv = 0;
        for (size_t i = 0; i < (size_t)M * N * zs; ++i) {
            maxd = std::max(maxd, (double)fabsf(h0[i] - h1[i]));
            maxv = std::max(maxv, (double)fabsf(h0[i]));
        }
        double csd = 0;
        if (!A_KC) {
            hipMemcpy(hc0.data(), cs0, (size_t)M * zs * 4, hipMemcpyDeviceToHost);
            hipMemcpy(hc1.data(), cs1, (size_t)M * zs * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < (size_t)M * zs; ++i) csd = std::max(csd, (double)fabsf(hc0[i] - hc1[i]));

        }
        // same products, same k order inside a tile: the two kernels agree exactly (colsum: another summation order)
        const bool ok = maxd == 0.0 && csd <= 1e-3;
        if (!ok) ++bad;
        double t0 = 0, t1 = 0, t2 = 0;
        if (timing) {
            std::vector<double> a, b, c;
            for (int round = 0; round < 5; ++round) {
                a.push_back(time_us([&]() { old_launch<A_KC, B_KC>(g0, tm, tn, zs); }, 20));
                b.push_back(time_us([&]() { launch_gemm2<A_KC, B_KC>(g1, tm, tn, zs, 0); }, 20));
                c.push_back(0.0);
            }
            std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
            t0 = a[2]; t1 = b[2]; t2 = c[2];
        }
#ifdef GEMM2_STAMP
        {
            hipMemset(C1, 0, 16);
            launch_gemm2<A_KC, B_KC>(g1, tm, tn, zs, 0);
            hipDeviceSynchronize();
            static unsigned long long hs[8 * 4096];
            hipMemcpyFromSymbol(hs, HIP_SYMBOL(g2_stamps), sizeof(hs));
            const unsigned nwg = (unsigned)(cdiv(g.N, 64 * tn) * cdiv(g.M, 64 * tm) * zs);
            std::vector<double> pro, loop, epi, clk, tot;
            unsigned long long first = ~0ull, last = 0;
            for (unsigned w = 0; w < nwg && w < 4096; ++w) {
                const unsigned long long* q = hs + w * 8;
                pro.push_back((double)(q[1] - q[0])); loop.push_back((double)(q[2] - q[1])); epi.push_back((double)(q[3] - q[2]));
                tot.push_back((double)(q[3] - q[0]));
                clk.push_back((double)(q[3] - q[0]) / (double)(q[5] - q[4]) * 100.0);       // MHz: realtime ticks at 100 MHz
                first = std::min(first, q[4]); last = std::max(last, q[5]);
            }
            {   // workgroups per CU (XCC id, SE, SH, CU from the hardware id registers)
                std::vector<unsigned> cu;
                for (unsigned w = 0; w < nwg && w < 4096; ++w) {
                    const unsigned long long h = hs[w * 8 + 6];
                    const unsigned hw = (unsigned)h, xcc = (unsigned)(h >> 32) & 15;
                    cu.push_back((xcc << 12) | (hw & 0xff00));          // cu_id[11:8] sh_id[12] se_id[15:13]
                }
                std::sort(cu.begin(), cu.end());
                int hist[16] = {0};
                size_t i0 = 0;
                int ncu = 0;
                while (i0 < cu.size()) { size_t j0 = i0; while (j0 < cu.size() && cu[j0] == cu[i0]) ++j0; hist[std::min<size_t>(15, j0 - i0)]++; ++ncu; i0 = j0; }
                printf("    placement: %d CUs used;", ncu);
                for (int k = 1; k < 16; ++k) if (hist[k]) printf(" %d CUs x %d WGs;", hist[k], k);
                printf("\n");
            }
            auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            const int nt = (int)(g.kchunk / 32);
            printf("    stamps: prologue %.0f  loop %.0f (= %.1f per MFMA of %d)  epilogue %.0f  total %.0f cycles; clock %.0f MHz; "
                   "first start -> last end %.1f us\n", med(pro), med(loop), med(loop) / (nt * 16.0 * tm * tn), nt * 16 * tm * tn,
                   med(epi), med(tot), med(clk), (double)(last - first) / 100.0);
        }
#endif
        const double fl = 2.0 * M * N * K;
        printf("%-28s %6ld x %4d x %5ld z%-2d tile %3dx%-3d  max|d| %.2e (|C| %.1f) colsum d %.1e %s", name, (long)M, N, (long)K,
               zs, 64 * tm, 64 * tn, maxd, maxv, csd, ok ? "ok  " : "BAD ");
        (void)t2;
        if (timing) printf("  old %6.1f us %5.1f TF | dma %6.1f us %5.1f TF", t0, fl / t0 / 1e6, t1, fl / t1 / 1e6);
        printf("\n");
    }
    hipFree(A); hipFree(B); hipFree(bias); hipFree(mask); hipFree(C0); hipFree(C1); hipFree(C2); hipFree(cs0); hipFree(cs1); hipFree(cs2);
    return bad;
}

int main(int argc, char** argv) {
    const bool timing = argc < 2 || atoi(argv[1]) != 0;
    const int64_t M = argc > 2 ? atol(argv[2]) : 8192;
    g_data_mode = argc > 3 ? atoi(argv[3]) : 1;
    g_extra_lds = argc > 5 ? atoi(argv[5]) : 0;
    const bool quick = argc > 4 && atoi(argv[4]) != 0;
    int bad = 0;
    if (quick) {
        bad += run_case<true, true>("fwd 512<-512", M, 512, 512, 1, timing);
        bad += run_case<true, false>("dgrad 512<-512", M, 512, 512, 1, timing);
        bad += run_case<false, false>("wgrad 512x512", 512, 512, M, 16, timing);
        printf(bad ? "FAILED: %d cases\n" : "all cases agree\n", bad);
        return bad != 0;
    }
    // edge shapes: partial tiles in both directions
    bad += run_case<true, true>("fwd edge", 1000, 200, 96, 1, false);
    bad += run_case<true, false>("dgrad edge", 1000, 200, 96, 1, false);
    bad += run_case<false, false>("wgrad edge", 200, 100, 2048, 4, false);
    bad += run_case<false, false>("wgrad edge uneven split", 72, 36, 1000 / 32 * 32, 3, false);
    // c3 layers: bot 13-512-256-128, top 479(480)-512-512-256-1
    bad += run_case<true, true>("fwd 512<-512", M, 512, 512, 1, timing);
    bad += run_case<true, true>("fwd 256<-512", M, 256, 512, 1, timing);
    bad += run_case<true, true>("fwd 512<-480", M, 512, 480, 1, timing);
    bad += run_case<true, true>("fwd 128<-256", M, 128, 256, 1, timing);
    bad += run_case<true, false>("dgrad 512<-512", M, 512, 512, 1, timing);
    bad += run_case<true, false>("dgrad 512<-256", M, 512, 256, 1, timing);
    bad += run_case<true, false>("dgrad 480<-512", M, 480, 512, 1, timing);
    bad += run_case<true, false>("dgrad 256<-128", M, 256, 128, 1, timing);
    bad += run_case<false, false>("wgrad 512x512", 512, 512, M, 16, timing);
    bad += run_case<false, false>("wgrad 256x512", 256, 512, M, 32, timing);
    bad += run_case<false, false>("wgrad 512x480", 512, 480, M, 16, timing);
    bad += run_case<false, false>("wgrad 128x256", 128, 256, M, 64, timing);
    printf(bad ? "FAILED: %d cases\n" : "all cases agree\n", bad);
    return bad != 0;
}
