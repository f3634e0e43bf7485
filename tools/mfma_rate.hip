// Development aid: issue rate of the FP32 MFMA shapes on this part (cycles per instruction, one wave per SIMD and two).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o /tmp/mr && /tmp/mr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int n) {
    float a = threadIdx.x * 0.001f, b = 1.0f - a;
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
    v16f d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0; d1[i] = 0; }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, c2, 0, 0, 0);
        } else {
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, d1, 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = c0[0] + c1[1] + c2[2] + d0[3] + d1[4];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[KIND] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 64);
    const int n = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512}) {
        for (int kind = 0; kind < 2; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, n);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, n);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            const int per = kind == 0 ? 3 : 2;
            const double flops = (double)blocks * 4 * n * per * (kind == 0 ? 2048.0 : 4096.0);
            printf("%s  %d workgroups (%d waves/SIMD): %.1f memtime ticks / MFMA, %.3f ms, %.1f TFLOP/s, %.1f ns / MFMA / wave\n",
                   kind == 0 ? "16x16x4 " : "32x32x2 ", blocks, blocks / 256, (double)h[kind] / (n * per), ms, flops / ms / 1e9,
                   ms * 1e6 / (n * per));
        }
    }
    return 0;
}
