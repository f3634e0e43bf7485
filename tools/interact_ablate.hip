// Development aid: the interaction kernels of cdlrm_amd/csrc/dense.hip stand-alone with ablation switches
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DIA_ABL=<n> -I cdlrm_amd/csrc -I include tools/interact_ablate.hip -o /tmp/ia && /tmp/ia
// IA_ABL: 0 full, 1 no MFMA, 2 no global loads in the loop, 3 no output stores
#include <stdarg.h>
#include "dense.hip"
void cdlrm_set_error(const char* fmt, ...) {}
int g_cdlrm_debug[8] = {0, 0, 0, 0, 0, 0, 0, 0};
CdlrmStopState* cdlrm_stop_state() {
    static thread_local CdlrmStopState st{nullptr, nullptr, 0};
    return &st;
}

int main(int argc, char** argv) {
    int64_t B = 8192; int F = 27, D = 128;
    if (argc > 1) B = atol(argv[1]);
    if (argc > 2) D = atoi(argv[2]);
    if (argc > 3) F = atoi(argv[3]);
    const int np = F * (F - 1) / 2, ld = (D + np + 3) / 4 * 4;
    float *feat, *R, *dR, *dfeat;
    hipMalloc(&feat, B * F * D * 4); hipMalloc(&dfeat, B * F * D * 4); hipMalloc(&R, B * ld * 4); hipMalloc(&dR, B * ld * 4);
    float* h = (float*)malloc(B * F * D * 4);
    for (int64_t i = 0; i < B * F * D; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(feat, h, B * F * D * 4, hipMemcpyHostToDevice);
    hipMemcpy(dR, h, B * ld * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        auto go = [&]() {
            if (which == 0) cdlrm_interact_fwd(feat, B, F, D, 0, R, ld, 0);
            else cdlrm_interact_bwd(feat, dR, ld, B, F, D, 0, 1, dfeat, 0);
        };
        for (int i = 0; i < 5; ++i) go();
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 30; ++i) go();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("ABL %d %s %.1f us", IA_ABL, which ? "bwd" : "fwd", ms / 30 * 1e3);
        {   // FNV-1a of the result's bytes: variants of a kernel must agree bit for bit
            const size_t nb = which ? (size_t)B * F * D * 4 : (size_t)B * ld * 4;
            unsigned char* hb = (unsigned char*)malloc(nb);
            hipMemcpy(hb, which ? (void*)dfeat : (void*)R, nb, hipMemcpyDeviceToHost);
            unsigned long long hsh = 1469598103934665603ull;
            for (size_t i = 0; i < nb; ++i) hsh = (hsh ^ hb[i]) * 1099511628211ull;
            printf("   result hash %016llx\n", hsh);
            free(hb);
        }
#ifdef IA_STAMP
        if (which == 0) {
            long long st[64];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(g_ia_stamp), sizeof(st));
            for (int i = 0; i + 3 < 60 && st[i + 3] > st[0]; i += 4)
                printf("  sample %d: start +%lld | stage %lld | mfma %lld | out %lld  (memtime ticks)\n", i / 4, st[i] - st[0],
                       st[i + 1] - st[i], st[i + 2] - st[i + 1], st[i + 3] - st[i + 2]);
        }
#endif
    }
    return 0;
}
