// Deterministic two-stage column reductions: stage 1 (op-specific) writes
// partials[block][s][col] for S reduction kinds; this kernel sums over blocks.
#pragma once
#include "ebn_common.h"

static inline int64_t ebn_colred_blocks(int64_t R) {
  int64_t nb = ebn_ceil_div(R, 64);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return nb;
}

// out_s[k] = (accumulate ? out_s[k] : 0) + sum_b partials[b][s][k], s in [0,S), S <= 2.
// One 256-thread block per 32 output columns: thread (col = t%32, part = t/32) sums every 8th block
// (coalesced 128-byte rows of partials), then the 8 parts are combined in a fixed order through LDS
// -> deterministic, and ~40x faster than one thread per column walking all blocks.
static __global__ __launch_bounds__(256) void ebn_reduce_partials_kernel(const float* __restrict__ partials,
                                                                         int nblk, int S, int A,
                                                                         float* __restrict__ out0,
                                                                         float* __restrict__ out1,
                                                                         int accumulate) {
  __shared__ float sm[8][33];
  const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + col;  // flattened (s, k)
  const bool ok = idx < S * A;
  float acc = 0.f;
  if (ok)
    for (int bk = part; bk < nblk; bk += 8) acc += partials[static_cast<int64_t>(bk) * S * A + idx];
  sm[part][col] = acc;
  __syncthreads();
  if (part == 0 && ok) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) t += sm[p][col];
    const int s = idx / A, k = idx - s * A;
    float* o = (s == 0) ? out0 : out1;
    if (o != nullptr) o[k] = accumulate ? (o[k] + t) : t;
  }
}
