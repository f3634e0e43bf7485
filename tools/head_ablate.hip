// Development aid: the output-head kernel of cdlrm_amd/csrc/dense.hip stand-alone with ablation switches
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHEAD_ABL=<n> -I cdlrm_amd/csrc -I include tools/head_ablate.hip -o /tmp/ha && /tmp/ha
// HEAD_ABL: 0 full, 1 no loss-reduction tail, 2 no dY stores
#include <stdarg.h>
#include "dense.hip"
void cdlrm_set_error(const char* fmt, ...) {}
int g_cdlrm_debug[8] = {0, 0, 0, 0, 0, 0, 0, 0};
CdlrmStopState* cdlrm_stop_state() {
    static thread_local CdlrmStopState st{nullptr, nullptr, 0};
    return &st;
}

int main(int argc, char** argv) {
    int64_t B = 8192; int K = 256;
    if (argc > 1) B = atol(argv[1]);
    float *Y, *dY, *w, *bias, *T, *Z, *dZ, *loss, *scratch;
    hipMalloc(&Y, B * K * 4); hipMalloc(&dY, B * K * 4); hipMalloc(&w, K * 4); hipMalloc(&bias, 4); hipMalloc(&T, B * 4);
    hipMalloc(&Z, B * 4); hipMalloc(&dZ, B * 4); hipMalloc(&loss, 64); hipMalloc(&scratch, cdlrm_head_scratch_floats() * 4);
    hipMemset(scratch, 0, cdlrm_head_scratch_floats() * 4);
    float* h = (float*)malloc(B * K * 4);
    for (int64_t i = 0; i < B * K; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(Y, h, B * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h, K * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 4);
    for (int64_t i = 0; i < B; ++i) h[i] = (float)(i & 1);
    hipMemcpy(T, h, B * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() { cdlrm_head_fwd_bwd(Y, K, w, bias, T, B, K, 0, 1.f, 1.f, 0.f, 1, Z, nullptr, dZ, dY, K, loss, scratch, 1, 0); };
    for (int i = 0; i < 5; ++i) go();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) go();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float l[3]; hipMemcpy(l, loss, 12, hipMemcpyDeviceToHost);
    printf("HEAD_ABL %d B %ld: %.1f us  (loss %.5f correct %.0f)\n", HEAD_ABL, (long)B, ms / 50 * 1e3, l[0], l[1]);
    return 0;
}
