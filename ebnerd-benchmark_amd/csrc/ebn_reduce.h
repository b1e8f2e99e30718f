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

// out_s[k] = (accumulate ? out_s[k] : 0) + scale * sum_b partials[b][s][k], s in {0,1}
static __global__ __launch_bounds__(256) void ebn_reduce_partials_kernel(const float* __restrict__ partials,
                                                                         int nblk, int S, int A,
                                                                         float* __restrict__ out0,
                                                                         float* __restrict__ out1,
                                                                         int accumulate) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * A) return;
  const int s = idx / A, k = idx - s * A;
  float acc = 0.f;
  for (int bk = 0; bk < nblk; ++bk) acc += partials[(static_cast<int64_t>(bk) * S + s) * A + k];
  float* o = (s == 0) ? out0 : out1;
  if (o == nullptr) return;
  o[k] = accumulate ? (o[k] + acc) : acc;
}
