// Device bodies of the small finishing passes of a training step's backward -- each is the second stage of a deterministic
// reduction (fixed summation order) -- shared by their own kernels and by the merged pass ebn_grad_finish_f32 (ebn_finish.hip).
#pragma once
#include "ebn_adam_flat.h"
#include "ebn_common.h"

// C[row][col] = sum_z part[z][row][col] (+ beta * C) for the flattened elements i = first, first + stride, ...  (total = M * N
// < 2^31: 32-bit index arithmetic -- a 64-bit division per element costs more than the sum itself).  Optional epilogues of the
// split-K GEMM it finishes: rank-1-per-sequence term (rs != NULL), bias + ReLU (bias != NULL).
static __device__ __forceinline__ void ebn_splitk_sum_body(uint32_t first, uint32_t stride, const float* __restrict__ part, int splits,
                                                           uint32_t total, uint32_t n32, float beta, float* __restrict__ C, int64_t ldc,
                                                           const float* __restrict__ rs, const float* __restrict__ cv, int64_t ldcv,
                                                           int32_t L, const float* __restrict__ bias, const EbnAdamFlat* ad = nullptr) {
  for (uint32_t i = first; i < total; i += stride) {
    float s = 0.f;
    int z = 0;
    for (; z + 4 <= splits; z += 4) {  // 4 independent loads in flight
      const float v0 = part[static_cast<int64_t>(z) * total + i], v1 = part[static_cast<int64_t>(z + 1) * total + i];
      const float v2 = part[static_cast<int64_t>(z + 2) * total + i], v3 = part[static_cast<int64_t>(z + 3) * total + i];
      s += v0;
      s += v1;
      s += v2;
      s += v3;
    }
    for (; z < splits; ++z) s += part[static_cast<int64_t>(z) * total + i];
    const uint32_t row = i / n32;
    const uint32_t col = i - row * n32;
    float* c = C + static_cast<int64_t>(row) * ldc + col;
    if (rs != nullptr) s = fmaf(rs[row], cv[static_cast<int64_t>(row / static_cast<uint32_t>(L)) * ldcv + col], s);
    if (bias != nullptr) s = fmaxf(s + bias[col], 0.f);
    const float g = (beta != 0.f) ? (s + beta * *c) : s;
    *c = g;
    if (ad != nullptr) ebn_adam_flat_apply(*ad, ad->st->adam_alpha, c - ad->grad, g);  // (ebn_grad_finish_adam_f32: the optimizer where the gradient is formed)
  }
}

// d(q), d(b) = sum over impressions of the partials (fixed order b = 0, 1, ...); the LAST block of the pass: the batch loss.
// Body shared with the merged finishing pass (ebn_grad_finish_f32): `blk` / `nblk` = this block among the blocks of the pass; only
// the first 256 threads of a block work (the merged pass runs 1024-thread blocks); `sw` = 4 floats of LDS.
static __device__ __forceinline__ void ebn_user_head_finish_body(float* sw, int blk, int nblk, const float* __restrict__ partials, int64_t B, int A,
                                          float* __restrict__ dq, float* __restrict__ db, const float* __restrict__ loss_rows,
                                          float* __restrict__ loss_out) {
  const int tid = threadIdx.x;
  if (blk == nblk - 1) {
    float s = 0.f;
    if (tid < 256)
      for (int64_t i = tid; i < B; i += 256) s += loss_rows[i];
    s = ebn_wave_sum(s);
    if (tid < 256 && (tid & 63) == 0) sw[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) loss_out[0] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
    return;
  }
  if (tid >= 256) return;
  const int idx = blk * 256 + tid;  // flattened (s, k)
  if (idx >= 2 * A) return;
  const int s = idx / A, k = idx - s * A;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;  // four loads in flight; combined in a fixed order
  int64_t b = 0;
  for (; b + 4 <= B; b += 4) {
    acc0 += partials[((b + 0) * 2 + s) * A + k];
    acc1 += partials[((b + 1) * 2 + s) * A + k];
    acc2 += partials[((b + 2) * 2 + s) * A + k];
    acc3 += partials[((b + 3) * 2 + s) * A + k];
  }
  for (; b < B; ++b) acc0 += partials[(b * 2 + s) * A + k];
  (s == 0 ? dq : db)[k] = (acc0 + acc1) + (acc2 + acc3);
}

