// Deterministic two-stage column reductions: stage 1 (op-specific) writes
// partials[row_block][s][col] for up to 2 reduction kinds; stage 2 sums over the row blocks.
#pragma once
#include "ebn_adam_flat.h"
#include "ebn_common.h"

// Row blocks of stage 1: 32 rows each (capped), so that even a few hundred rows (one TimeDistributed call
// site of a batch-32 step) spread over row_blocks x ceil(C/64) workgroups instead of a handful.
static inline int64_t ebn_colred_blocks(int64_t R) {
  int64_t nb = ebn_ceil_div(R, 32);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return nb;
}

// Stage 1 skeleton.  grid = (row_blocks, ceil(C/64)), block = 256 threads = 64 columns x 4 row lanes.
// F::row(r, c, a0, a1) visits one element (it may also write element-wise outputs); the 4 row lanes are
// combined through LDS in a fixed order.
template <class F>
__global__ __launch_bounds__(256) void ebn_colred_stage1_kernel(F f, float* __restrict__ partials, int64_t R, int C,
                                                                int64_t rpb) {
  __shared__ float sm[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rpb;
  const int64_t r1 = (r0 + rpb < R) ? r0 + rpb : R;
  float a0 = 0.f, a1 = 0.f;
  if (c < C)
    for (int64_t r = r0 + rl; r < r1; r += 4) f.row(r, c, a0, a1);
  sm[0][rl][cl] = a0;
  sm[1][rl][cl] = a1;
  __syncthreads();
  if (rl == 0 && c < C) {
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * C + c] = (sm[0][0][cl] + sm[0][1][cl]) + (sm[0][2][cl] + sm[0][3][cl]);
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * C + c] = (sm[1][0][cl] + sm[1][1][cl]) + (sm[1][2][cl] + sm[1][3][cl]);
  }
}

template <class F>
static inline void ebn_colred_stage1(F f, float* partials, int64_t R, int C, hipStream_t s, int64_t* nb_out) {
  const int64_t nb = ebn_colred_blocks(R);
  const int64_t rpb = ebn_ceil_div(R, nb);
  EBN_LAUNCH((ebn_colred_stage1_kernel<F>), dim3(static_cast<unsigned>(nb), static_cast<unsigned>(ebn_ceil_div(C, 64))),
                     dim3(256), 0, s, f, partials, R, C, rpb);
  *nb_out = nb;
}

// Stage 2: v_s[k] = scale * sum_b partials[b][s][k]; out_s[k] = (accumulate ? out_s[k] : 0) + v_s[k];
// optional copies site_s[k] = v_s[k] (per-call-site values kept next to an accumulated total).
// One 1024-thread block per 32 flattened (s,k) outputs: thread (col = t%32, part = t/32) sums every 32nd block
// (coalesced 128-byte rows of partials; 8 independent loads in flight), then the 32 parts are combined in a fixed
// order through LDS.  (With 8 parts the ~94 dependent iterations of a 24,000-row call site cost 15 us.)
// (the body, shared with the merged finishing pass of a training step: ebn_grad_finish_f32 in ebn_finish.hip; `blk` = index of the
// 1024-thread block among the blocks of THIS reduction, `sm` = 32 x 33 floats of LDS)
static __device__ __forceinline__ void ebn_reduce_partials_body(float (*sm)[33], int blk, const float* __restrict__ partials, int nblk, int S, int A,
                                                              float scale, float* __restrict__ out0, float* __restrict__ out1, int accumulate,
                                                              float* __restrict__ site0, float* __restrict__ site1,
                                                              const EbnAdamFlat* ad = nullptr) {
  const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int idx = blk * 32 + col;  // flattened (s, k)
  const bool ok = idx < S * A;
  float acc = 0.f;
  if (ok) {
    const float* p = partials + idx;
    const int64_t stride = 2 * static_cast<int64_t>(A);
    // eight row blocks per round trip, unconditional loads with clamped indices (a 24 000-row call site leaves 750 row blocks = 24 per
    // thread: four at a time plus a scalar tail were eight or nine dependent round trips -- the critical path of the step's finishing launch)
    for (int bk = part; bk < nblk; bk += 256) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int b = bk + 32 * j;
        v[j] = p[(b < nblk ? b : part) * stride];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += (bk + 32 * j < nblk) ? v[j] : 0.f;
    }
  }
  sm[part][col] = acc;
  __syncthreads();
  if (part == 0 && ok) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 32; ++p) t += sm[p][col];
    t *= scale;
    const int s = idx / A, k = idx - s * A;
    float* o = (s == 0) ? out0 : out1;
    if (o != nullptr) {
      const float g = accumulate ? (o[k] + t) : t;
      o[k] = g;
      if (ad != nullptr) ebn_adam_flat_apply(*ad, ad->st->adam_alpha, (o + k) - ad->grad, g);
    }
    float* st = (s == 0) ? site0 : site1;
    if (st != nullptr) st[k] = t;
  }
}

static __global__ __launch_bounds__(1024) void ebn_reduce_partials_kernel(const float* __restrict__ partials,
                                                                         int nblk, int S, int A, float scale,
                                                                         float* __restrict__ out0,
                                                                         float* __restrict__ out1, int accumulate,
                                                                         float* __restrict__ site0,
                                                                         float* __restrict__ site1) {
  __shared__ float sm[32][33];
  ebn_reduce_partials_body(sm, blockIdx.x, partials, nblk, S, A, scale, out0, out1, accumulate, site0, site1);
}

static inline void ebn_reduce_partials(const float* partials, int64_t nb, int S, int A, float scale, float* out0,
                                       float* out1, int accumulate, float* site0, float* site1, hipStream_t s) {
  EBN_LAUNCH(ebn_reduce_partials_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(S * A, 32))), dim3(1024), 0, s,
                     partials, static_cast<int>(nb), S, A, scale, out0, out1, accumulate, site0, site1);
}
