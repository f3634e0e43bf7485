// Criteo-shaped synthetic sparse indices, generated on the device in ONE pass (cdlrm_amd/synth.py).
//
// The reference has no Criteo-shaped synthetic generator (its RandomDataset is uniform multi-hot, dlrm_data_pytorch.py:763-805);
// bench.py needs one that keeps up with a 12 M samples/s step on BOTH sides of the look-ahead -- the trainer's batches and the
// window plan's second pass over the same indices (the reference's Prefetcher iterates a second loader over the same data,
// cache_manager.py:87-90).  As a chain of elementwise float64 torch kernels the generator cost a c5 window 0.8 s of GPU time on
// each side; here every index is a pure function of (key, position in the table's infinite lookup stream):
//     u = splitmix64(key + position) -> [0, 1);   rank = floor((top u + 1)^(1 / (1 - alpha))) - 1   (Zipf-like, alpha != 1)
//     index = (rank * 2654435761 + 40503) mod n      (the hot ranks scattered over the id space)
// so any window / chunk of the stream can be produced anywhere, in any order, by one launch per table.
#include "common.h"

__device__ __forceinline__ uint64_t synth_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) k_synth_indices(int64_t* __restrict__ out, int64_t count, int64_t n, double alpha,
                                                       double top, double inv, uint64_t key, int64_t first) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t z = synth_mix64(key + (uint64_t)(first + i));
        const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
        int64_t r;
        if (alpha <= 0.0) {
            r = (int64_t)floor(u * (double)n);                       // uniform: worst-case hit rate
            if (r > n - 1) r = n - 1;
        } else {
            // inv == 0 marks alpha == 1: x = (n + 1)^u
            const double x = inv == 0.0 ? exp(u * top) : pow(top * u + 1.0, inv);
            r = (int64_t)floor(x) - 1;
            r = r < 0 ? 0 : (r > n - 1 ? n - 1 : r);
            r = (int64_t)(((uint64_t)r * 2654435761ull + 40503ull) % (uint64_t)n);
        }
        out[i] = r;
    }
}

// out[i] = index of lookup `first + i` of the table's stream, i < count.  n_rows: table cardinality; alpha: Zipf exponent
// (<= 0: uniform); key: the (seed, table) key of the stream.
extern "C" int cdlrm_synth_indices(int64_t* out, int64_t count, int64_t first, int64_t n_rows, double alpha, uint64_t key,
                                   void* stream) {
    CDLRM_REQUIRE(out && count >= 0 && first >= 0 && n_rows >= 1, "bad argument");
    if (count == 0) return 0;
    double top = 0.0, inv = 0.0;
    if (alpha > 0.0) {
        if (fabs(alpha - 1.0) < 1e-9) {
            top = log((double)n_rows + 1.0);
        } else {
            top = pow((double)n_rows + 1.0, 1.0 - alpha) - 1.0;
            inv = 1.0 / (1.0 - alpha);
        }
    }
    int64_t gx = cdiv(count, 256 * 8);
    if (gx > 8192) gx = 8192;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_synth_indices, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, out, count, n_rows, alpha, top,
                       inv, key, first);
    CDLRM_LAUNCH_CHECK();
    return 0;
}
