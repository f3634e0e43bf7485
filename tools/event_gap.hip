// What does a completion event attached to a launch cost the queue that carries it?  (engine.py: attach_events -- the training
// queue's interaction backward and last dgrad GEMM complete an event for the side queues.)  A chain of N ~20 us kernels on one
// stream, every k-th launched with a stop event of the given creation flags; a second stream waits for each event.
//   hipcc -O3 --offload-arch=gfx950 tools/event_gap.hip -o gpurun_out/event_gap && gpurun_out/event_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

static double run(unsigned flags, int mode, int N, float* buf, hipStream_t s, hipStream_t s2) {
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) hipEventCreateWithFlags(&e, flags);
    hipStreamSynchronize(s);
    hipStreamSynchronize(s2);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
        if (mode == 1 && i % 4 == 3) {            // attached stop event, waited for by the other stream
            hipExtLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, s, nullptr, ev[i], 0, buf, 3000);
            hipStreamWaitEvent(s2, ev[i], 0);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s2, buf + (1 << 20), 300);
        } else if (mode == 2 && i % 4 == 3) {     // recorded behind the launch
            hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, s, buf, 3000);
            hipEventRecord(ev[i], s);
            hipStreamWaitEvent(s2, ev[i], 0);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s2, buf + (1 << 20), 300);
        } else {
            hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, s, buf, 3000);
        }
    }
    hipStreamSynchronize(s);
    hipStreamSynchronize(s2);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    for (auto& e : ev) hipEventDestroy(e);
    return us / N;
}

int main() {
    float* buf;
    hipMalloc(&buf, 64 << 20);
    hipMemset(buf, 0, 64 << 20);
    hipStream_t s, s2;
    int lo, hi;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int N = 400;
    run(hipEventDisableTiming, 0, N, buf, s, s2);
    printf("no events                                   %.2f us per launch\n", run(hipEventDisableTiming, 0, N, buf, s, s2));
    struct { const char* name; unsigned f; } fl[] = {{"DisableTiming", hipEventDisableTiming},
                                                      {"DisableTiming|ReleaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice},
                                                      {"DisableTiming|DisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence},
                                                      {"Default (timing)", hipEventDefault}};
    for (auto& f : fl) {
        double a = run(f.f, 1, N, buf, s, s2), b = run(f.f, 2, N, buf, s, s2);
        printf("%-34s attached %.2f   recorded %.2f us per launch (every 4th launch carries one)\n", f.name, a, b);
    }
    return 0;
}
