// a4 / a7: the reference AttLayer2 (layers.py:55-81) after its x.W matmul (done by ebn_gemm):
//   U = tanh(xW + b); e = U.q; a = exp(e) (no max-subtraction); w = a/(sum a + 1e-7);
//   out = sum_l w_l x_l
// and its backward.  All of it is HBM-bound row streaming (each [R,A] / [R,E] activation is
// read once or twice), so rows are read with unit-stride lanes and reduced with wave shuffles;
// parameter gradients (dq, db) use deterministic two-stage column reductions, not atomics.
#include "ebn_common.h"
#include "ebn_reduce.h"

namespace {

constexpr int POOL_THREADS = 256;
constexpr int POOL_WAVES = POOL_THREADS / 64;
constexpr float KERAS_EPS = 1e-7f;  // K.epsilon(), layers.py:75-77

// one workgroup per sequence
__global__ __launch_bounds__(POOL_THREADS) void attpool_fwd_kernel(float* __restrict__ U,
                                                                    const float* __restrict__ b,
                                                                    const float* __restrict__ q,
                                                                    const float* __restrict__ X,
                                                                    float* __restrict__ out,
                                                                    float* __restrict__ w, int L, int E, int A) {
  extern __shared__ float sm[];  // e / w of this sequence: L floats
  const int64_t n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int l = wave; l < L; l += POOL_WAVES) {
    float* urow = U + (n * L + l) * A;
    float part = 0.f;
    for (int k = lane; k < A; k += 64) {
      const float u = tanhf(urow[k] + b[k]);
      urow[k] = u;
      part = fmaf(u, q[k], part);
    }
    part = ebn_wave_sum(part);
    if (lane == 0) sm[l] = part;
  }
  __syncthreads();
  if (wave == 0) {
    float s = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float a = expf(sm[l]);
      sm[l] = a;
      s += a;
    }
    s = ebn_wave_sum(s) + KERAS_EPS;
    for (int l = lane; l < L; l += 64) {
      const float wl = sm[l] / s;
      sm[l] = wl;
      w[n * L + l] = wl;
    }
  }
  __syncthreads();
  for (int c = tid; c < E; c += POOL_THREADS) {
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc = fmaf(sm[l], X[(n * L + l) * E + c], acc);
    out[n * E + c] = acc;
  }
}

__global__ __launch_bounds__(POOL_THREADS) void attpool_bwd_pool_kernel(const float* __restrict__ X,
                                                                         const float* __restrict__ w,
                                                                         const float* __restrict__ dout,
                                                                         float* __restrict__ dX,
                                                                         float* __restrict__ de, int L, int E) {
  extern __shared__ float sm[];  // dw[L]
  const int64_t n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* g = dout + n * E;
  for (int l = wave; l < L; l += POOL_WAVES) {
    const float* xr = X + (n * L + l) * E;
    float part = 0.f;
    for (int c = lane; c < E; c += 64) part = fmaf(g[c], xr[c], part);
    part = ebn_wave_sum(part);
    if (lane == 0) sm[l] = part;
  }
  __syncthreads();
  if (wave == 0) {
    float s = 0.f;
    for (int l = lane; l < L; l += 64) s = fmaf(w[n * L + l], sm[l], s);
    s = ebn_wave_sum(s);
    for (int l = lane; l < L; l += 64) de[n * L + l] = w[n * L + l] * (sm[l] - s);
  }
  for (int l = 0; l < L; ++l) {
    const float wl = w[n * L + l];
    float* dr = dX + (n * L + l) * E;
    for (int c = tid; c < E; c += POOL_THREADS) dr[c] = wl * g[c];
  }
}

// stage 1: block b owns rows [b*rpb, ...); thread = column. partials[b][0][k]=dq, [b][1][k]=db
__global__ __launch_bounds__(POOL_THREADS) void attpool_bwd_dpre_kernel(float* __restrict__ U,
                                                                         const float* __restrict__ q,
                                                                         const float* __restrict__ de,
                                                                         float* __restrict__ partials,
                                                                         int64_t R, int A, int64_t rpb) {
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rpb;
  const int64_t r1 = (r0 + rpb < R) ? r0 + rpb : R;
  for (int k = threadIdx.x; k < A; k += POOL_THREADS) {
    const float qk = q[k];
    float dq = 0.f, db = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const float u = U[r * A + k];
      const float der = de[r];
      dq = fmaf(der, u, dq);
      const float dp = der * qk * (1.0f - u * u);
      U[r * A + k] = dp;
      db += dp;
    }
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * A + k] = dq;
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * A + k] = db;
  }
}

}  // namespace

extern "C" int64_t ebn_attpool_partials_len(int64_t R, int32_t A) { return ebn_colred_blocks(R) * 2 * A; }

extern "C" int ebn_attpool_fwd_f32(float* U, const float* b, const float* q, const float* X, float* out,
                                   float* w, int64_t n_seq, int32_t L, int32_t E, int32_t A,
                                   ebn_stream_t stream) {
  EBN_REQUIRE(U && b && q && X && out && w, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_seq >= 0 && L > 0 && E > 0 && A > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(L <= 8192, EBN_ERR_UNSUPPORTED);
  if (n_seq == 0) return EBN_OK;
  hipLaunchKernelGGL(attpool_fwd_kernel, dim3(static_cast<unsigned>(n_seq)), dim3(POOL_THREADS),
                     L * sizeof(float), ebn_stream(stream), U, b, q, X, out, w, L, E, A);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_attpool_bwd_pool_f32(const float* X, const float* w, const float* dout, float* dX,
                                        float* de, int64_t n_seq, int32_t L, int32_t E,
                                        ebn_stream_t stream) {
  EBN_REQUIRE(X && w && dout && dX && de, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_seq >= 0 && L > 0 && E > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(L <= 8192, EBN_ERR_UNSUPPORTED);
  if (n_seq == 0) return EBN_OK;
  hipLaunchKernelGGL(attpool_bwd_pool_kernel, dim3(static_cast<unsigned>(n_seq)), dim3(POOL_THREADS),
                     L * sizeof(float), ebn_stream(stream), X, w, dout, dX, de, L, E);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_attpool_bwd_dpre_f32(float* U, const float* q, const float* de, float* dq, float* db,
                                        float* partials, int64_t R, int32_t A, int32_t accumulate,
                                        ebn_stream_t stream) {
  EBN_REQUIRE(U && q && de && dq && db && partials, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(R >= 0 && A > 0, EBN_ERR_BAD_ARG);
  if (R == 0) return EBN_OK;
  const int64_t nb = ebn_colred_blocks(R);
  const int64_t rpb = ebn_ceil_div(R, nb);
  hipLaunchKernelGGL(attpool_bwd_dpre_kernel, dim3(static_cast<unsigned>(nb)), dim3(POOL_THREADS), 0,
                     ebn_stream(stream), U, q, de, partials, R, A, rpb);
  EBN_CHECK_LAUNCH();
  ebn_reduce_partials(partials, nb, 2, A, 1.0f, dq, db, accumulate, nullptr, nullptr, ebn_stream(stream));
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
