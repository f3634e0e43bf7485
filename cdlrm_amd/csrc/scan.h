// Flag compaction (stream compaction by reduce-then-scan), shared by the window and agg paths.
#pragma once
#include "common.h"

// out[j] = position of the j-th non-zero flag (ascending); *d_count = number of non-zero flags.
// n is read from d_n[0] when d_n != nullptr (then n_max bounds the launch), else n = n_max.
// Entries beyond `cap` are dropped and the device error word gets bit 4.
// clear bit 0: zero the flags it has consumed; bit 1: entries beyond `cap` are dropped silently (no error bit).
// scratch: block sums, >= cdiv(n_max, 4096) + 1 int64 -- ctx->d_scan on the plan stream, ctx->d_scan_agg for the
// table-agg path on the main stream (two compactions in flight on two streams must not share it).
int cdlrm_compact_flags(cdlrm_ctx* ctx, uint8_t* flags, const int64_t* d_n, int64_t n_max,
                        int32_t* out32, int64_t* out64, int64_t cap, int64_t* d_count, int clear,
                        hipStream_t s, int64_t* scratch = nullptr, int64_t scratch_cap = 0);

// single-block exclusive scan over int64 block sums, in place; total -> *d_total (may be null)
__global__ void k_scan_tops(int64_t* sums, int64_t n, int64_t* d_total);
