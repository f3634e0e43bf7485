// K15: quotient-remainder embedding bag (tricks/qr_embedding_bag.py:156-174), stand-alone operator.
// The reference never wires it into the cached path, so this is the operator only:
//   q = (idx / c).long()  -- a float32 TRUE division followed by truncation (wrong above 2^24, kept on purpose)
//   r = idx % c
//   out = bag_sum(Wq, q) (*|+|concat) bag_sum(Wr, r)
// HBM-bound row gathers, 16 B per lane.
#include "common.h"

__device__ __forceinline__ int64_t qr_quotient(int64_t idx, int c) {
    // torch: int64 tensor / python int -> true_divide in the default float dtype (float32), then .long()
    return (int64_t)((float)idx / (float)c);
}

template <int LPR>
__global__ void __launch_bounds__(256) k_qr_fwd(const int64_t* __restrict__ idx, const int64_t* __restrict__ offsets,
                                                int64_t n, int64_t n_bags, const float4* __restrict__ Wq,
                                                const float4* __restrict__ Wr, int64_t rows_q, int c, int D4, int op,
                                                float* __restrict__ out, float4* __restrict__ eq_out,
                                                float4* __restrict__ er_out, int* err) {
    const int cl = threadIdx.x % LPR;
    const int gpb = blockDim.x / LPR;
    const int gid = threadIdx.x / LPR;
    const int64_t ld_out = (op == 2 ? 2 : 1) * (int64_t)D4 * 4;
    for (int64_t b = (int64_t)blockIdx.x * gpb + gid; b < n_bags; b += (int64_t)gridDim.x * gpb) {
        const int64_t lo = offsets[b];
        const int64_t hi = (b + 1 < n_bags) ? offsets[b + 1] : n;
        for (int cc = cl; cc < D4; cc += LPR) {
            float4 aq = make_float4(0.f, 0.f, 0.f, 0.f), ar = aq;
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t v = idx[i];
                int64_t q = qr_quotient(v, c);
                const int64_t r = v % c;
                if (q < 0 || q >= rows_q || r < 0) { atomicOr(err, 1); q = 0; }
                const float4 x = Wq[q * D4 + cc], y = Wr[(r < 0 ? 0 : r) * D4 + cc];
                aq.x += x.x; aq.y += x.y; aq.z += x.z; aq.w += x.w;
                ar.x += y.x; ar.y += y.y; ar.z += y.z; ar.w += y.w;
            }
            if (eq_out) eq_out[b * D4 + cc] = aq;
            if (er_out) er_out[b * D4 + cc] = ar;
            float* o = out + b * ld_out + cc * 4;
            if (op == 0) {
                *reinterpret_cast<float4*>(o) = make_float4(aq.x * ar.x, aq.y * ar.y, aq.z * ar.z, aq.w * ar.w);
            } else if (op == 1) {
                *reinterpret_cast<float4*>(o) = make_float4(aq.x + ar.x, aq.y + ar.y, aq.z + ar.z, aq.w + ar.w);
            } else {
                *reinterpret_cast<float4*>(o) = aq;
                *reinterpret_cast<float4*>(o + D4 * 4) = ar;
            }
        }
    }
}

// dense gradients of both tables: gWq[q] += dEq[bag], gWr[r] += dEr[bag]  (float atomics: the operator's tables are
// small and this path is not on the cached training step; sums are order-dependent in the last bits)
template <int LPR>
__global__ void __launch_bounds__(256) k_qr_bwd(const int64_t* __restrict__ idx, const int64_t* __restrict__ offsets,
                                                int64_t n, int64_t n_bags, const float4* __restrict__ eq,
                                                const float4* __restrict__ er, const float* __restrict__ gout,
                                                int64_t rows_q, int c, int D4, int op, float* __restrict__ gWq,
                                                float* __restrict__ gWr) {
    const int cl = threadIdx.x % LPR;
    const int gpb = blockDim.x / LPR;
    const int gid = threadIdx.x / LPR;
    const int64_t ld_out = (op == 2 ? 2 : 1) * (int64_t)D4 * 4;
    for (int64_t b = (int64_t)blockIdx.x * gpb + gid; b < n_bags; b += (int64_t)gridDim.x * gpb) {
        const int64_t lo = offsets[b];
        const int64_t hi = (b + 1 < n_bags) ? offsets[b + 1] : n;
        for (int cc = cl; cc < D4; cc += LPR) {
            const float* go = gout + b * ld_out + cc * 4;
            const float4 g = *reinterpret_cast<const float4*>(go);
            float4 dq, dr;
            if (op == 0) {
                const float4 a = eq[b * D4 + cc], r_ = er[b * D4 + cc];
                dq = make_float4(g.x * r_.x, g.y * r_.y, g.z * r_.z, g.w * r_.w);
                dr = make_float4(g.x * a.x, g.y * a.y, g.z * a.z, g.w * a.w);
            } else if (op == 1) {
                dq = g; dr = g;
            } else {
                dq = g; dr = *reinterpret_cast<const float4*>(go + D4 * 4);
            }
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t v = idx[i];
                int64_t q = qr_quotient(v, c);
                if (q < 0 || q >= rows_q) q = 0;
                const int64_t r = v % c;
                float* pq = gWq + (q * D4 + cc) * 4;
                float* pr = gWr + (r * D4 + cc) * 4;
                atomicAdd(pq + 0, dq.x); atomicAdd(pq + 1, dq.y); atomicAdd(pq + 2, dq.z); atomicAdd(pq + 3, dq.w);
                atomicAdd(pr + 0, dr.x); atomicAdd(pr + 1, dr.y); atomicAdd(pr + 2, dr.z); atomicAdd(pr + 3, dr.w);
            }
        }
    }
}

static int qr_lpr(int D4) { int l = pow2ceil(D4); return l > 64 ? 64 : (l < 4 ? 4 : l); }

#define QR_DISPATCH(lpr, CALL)                  \
    switch (lpr) {                              \
        case 4: { CALL(4); break; }             \
        case 8: { CALL(8); break; }             \
        case 16: { CALL(16); break; }           \
        case 32: { CALL(32); break; }           \
        default: { CALL(64); break; }           \
    }

extern "C" int cdlrm_qr_embbag_fwd(const int64_t* idx, const int64_t* offsets, int64_t n, int64_t n_bags, const float* Wq,
                                   const float* Wr, int64_t rows_q, int32_t collisions, int32_t dim, int32_t op,
                                   float* out, float* eq_out, float* er_out, int32_t* err_word, void* stream) {
    CDLRM_REQUIRE(idx && offsets && Wq && Wr && out && err_word, "null argument");
    CDLRM_REQUIRE(dim >= 4 && dim % 4 == 0 && collisions >= 1 && op >= 0 && op <= 2, "bad shape / operation");
    CDLRM_REQUIRE((((uintptr_t)Wq | (uintptr_t)Wr | (uintptr_t)out) & 15) == 0, "16-byte aligned rows");
    if (n_bags == 0) return 0;
    const int D4 = dim / 4, lpr = qr_lpr(D4), gpb = 256 / lpr;
    int64_t gx = cdiv(n_bags, gpb);
    if (gx > 8192) gx = 8192;
#define QRF(L) hipLaunchKernelGGL(k_qr_fwd<L>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, idx, offsets, n, n_bags, reinterpret_cast<const float4*>(Wq), reinterpret_cast<const float4*>(Wr), rows_q, collisions, D4, op, out, reinterpret_cast<float4*>(eq_out), reinterpret_cast<float4*>(er_out), err_word)
    QR_DISPATCH(lpr, QRF)
#undef QRF
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_qr_embbag_bwd(const int64_t* idx, const int64_t* offsets, int64_t n, int64_t n_bags, const float* eq,
                                   const float* er, const float* grad_out, int64_t rows_q, int32_t collisions,
                                   int32_t dim, int32_t op, float* gWq, float* gWr, void* stream) {
    CDLRM_REQUIRE(idx && offsets && grad_out && gWq && gWr, "null argument");
    CDLRM_REQUIRE(op != 0 || (eq && er), "mult backward needs the pooled operands");
    CDLRM_REQUIRE(dim >= 4 && dim % 4 == 0 && collisions >= 1 && op >= 0 && op <= 2, "bad shape / operation");
    if (n_bags == 0) return 0;
    const int D4 = dim / 4, lpr = qr_lpr(D4), gpb = 256 / lpr;
    int64_t gx = cdiv(n_bags, gpb);
    if (gx > 8192) gx = 8192;
#define QRB(L) hipLaunchKernelGGL(k_qr_bwd<L>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, idx, offsets, n, n_bags, reinterpret_cast<const float4*>(eq), reinterpret_cast<const float4*>(er), grad_out, rows_q, collisions, D4, op, gWq, gWr)
    QR_DISPATCH(lpr, QRB)
#undef QRB
    CDLRM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Plain stand-alone EmbeddingBag(mode="sum") over ONE table of any width (the mixed-dimension trick's
// PrEmbeddingBag.embs, tricks/md_embedding_bag.py:60-78: widths are powers of two down to 1, so no 16-byte rows
// can be assumed).  One thread per output element; backward = dense gradient by float atomics (stand-alone operator,
// not on the cached training step).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bag_fwd(const int64_t* __restrict__ idx, const int64_t* __restrict__ offsets,
                                                 int64_t n, int64_t n_bags, const float* __restrict__ W, int64_t rows, int D,
                                                 float* __restrict__ out, int* err) {
    const int64_t total = n_bags * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = e / D;
        const int c = (int)(e % D);
        const int64_t lo = offsets[b], hi = (b + 1 < n_bags) ? offsets[b + 1] : n;
        float acc = 0.f;
        for (int64_t i = lo; i < hi; ++i) {
            int64_t v = idx[i];
            if (v < 0 || v >= rows) { atomicOr(err, 1); v = 0; }
            acc += W[v * D + c];
        }
        out[e] = acc;
    }
}

__global__ void __launch_bounds__(256) k_bag_bwd(const int64_t* __restrict__ idx, const int64_t* __restrict__ offsets,
                                                 int64_t n, int64_t n_bags, const float* __restrict__ gout, int64_t rows,
                                                 int D, float* __restrict__ gW) {
    const int64_t total = n_bags * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = e / D;
        const int c = (int)(e % D);
        const int64_t lo = offsets[b], hi = (b + 1 < n_bags) ? offsets[b + 1] : n;
        const float g = gout[e];
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t v = idx[i];
            if (v >= 0 && v < rows) atomicAdd(gW + v * D + c, g);
        }
    }
}

extern "C" int cdlrm_bag_fwd(const int64_t* idx, const int64_t* offsets, int64_t n, int64_t n_bags, const float* W,
                             int64_t rows, int32_t dim, float* out, int32_t* err_word, void* stream) {
    CDLRM_REQUIRE(idx && offsets && W && out && err_word && dim >= 1 && rows >= 1, "bad argument");
    if (n_bags == 0) return 0;
    int64_t gx = cdiv(n_bags * dim, 256);
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(k_bag_fwd, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, idx, offsets, n, n_bags, W, rows,
                       (int)dim, out, err_word);
    CDLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdlrm_bag_bwd(const int64_t* idx, const int64_t* offsets, int64_t n, int64_t n_bags, const float* grad_out,
                             int64_t rows, int32_t dim, float* gW, void* stream) {
    CDLRM_REQUIRE(idx && offsets && grad_out && gW && dim >= 1 && rows >= 1, "bad argument");
    if (n_bags == 0) return 0;
    int64_t gx = cdiv(n_bags * dim, 256);
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(k_bag_bwd, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, idx, offsets, n, n_bags, grad_out, rows,
                       (int)dim, gW);
    CDLRM_LAUNCH_CHECK();
    return 0;
}
