// a11: pieces of the NRMSDocVec news encoder around its Dense matmuls (nrms_docvec.py:113-135):
// bias + ReLU, BatchNormalization (per-call-site batch statistics, Keras defaults momentum
// 0.99 / eps 1e-3 [KERAS-SEMANTICS]), fused Dropout, and their backward; plus axpy / sum
// helpers (L2 kernel-regulariser, loss reduction).  Rows per call site are few (B*H or B*C),
// so everything here is latency-bound; column statistics use deterministic two-stage
// reductions (two-pass variance).
#include "ebn_common.h"
#include "ebn_reduce.h"

namespace {

constexpr float BN_EPS = 1e-3f;
constexpr float BN_MOM = 0.99f;

__global__ __launch_bounds__(256) void bias_relu_kernel(const float* __restrict__ X, const float* __restrict__ bias,
                                                        float* __restrict__ Y, int64_t n, int C) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % C);
    Y[i] = fmaxf(X[i] + bias[c], 0.f);
  }
}

// dX = dY*(Y>0); partials[b][0][c] = column sums of dX
__global__ __launch_bounds__(256) void bias_relu_bwd_kernel(const float* __restrict__ Y, const float* __restrict__ dY,
                                                            float* __restrict__ dX, float* __restrict__ partials,
                                                            int64_t R, int C, int64_t rpb) {
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rpb;
  const int64_t r1 = (r0 + rpb < R) ? r0 + rpb : R;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const float g = (Y[r * C + c] > 0.f) ? dY[r * C + c] : 0.f;
      dX[r * C + c] = g;
      s += g;
    }
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * C + c] = s;
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * C + c] = 0.f;
  }
}

// partials[b][0][c] = sum_r (X[r,c] - (mean ? mean[c] : 0))^P  with P = 1 or 2
template <int P>
__global__ __launch_bounds__(256) void col_moment_kernel(const float* __restrict__ X, const float* __restrict__ mean,
                                                         float* __restrict__ partials, int64_t R, int C,
                                                         int64_t rpb) {
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rpb;
  const int64_t r1 = (r0 + rpb < R) ? r0 + rpb : R;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mu = mean ? mean[c] : 0.f;
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const float x = X[r * C + c] - mu;
      s += (P == 1) ? x : x * x;
    }
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * C + c] = s;
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * C + c] = 0.f;
  }
}

// v[c] *= scale (turn sums into means)
__global__ __launch_bounds__(256) void scale_vec_kernel(float* __restrict__ v, int C, float scale) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) v[c] *= scale;
}

// var[c] (in istd_out) -> istd; update moving stats
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ mean, float* __restrict__ var_istd,
                                                          float* __restrict__ mmean, float* __restrict__ mvar,
                                                          int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float var = var_istd[c];
  mmean[c] = mmean[c] * BN_MOM + mean[c] * (1.0f - BN_MOM);
  mvar[c] = mvar[c] * BN_MOM + var * (1.0f - BN_MOM);
  var_istd[c] = 1.0f / sqrtf(var + BN_EPS);
}

__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float* __restrict__ mmean,
                                                            const float* __restrict__ mvar, float* __restrict__ mean,
                                                            float* __restrict__ istd, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  mean[c] = mmean[c];
  istd[c] = 1.0f / sqrtf(mvar[c] + BN_EPS);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean,
                                                       const float* __restrict__ istd, float* __restrict__ Y,
                                                       float* __restrict__ xhat, int64_t n, int C,
                                                       const uint32_t* __restrict__ key_ptr, uint32_t thresh,
                                                       float scale, int64_t elem_offset) {
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % C);
    const float xh = (X[i] - mean[c]) * istd[c];
    if (xhat) xhat[i] = xh;
    float y = xh * gamma[c] + beta[c];
    if (do_drop) y *= ebn_drop_mult(key, static_cast<uint64_t>(i + elem_offset), thresh, scale);
    Y[i] = y;
  }
}

// partials[b][0][c] = sum dY' xhat (dgamma), [b][1][c] = sum dY' (dbeta); dY' = dY * dropout mult
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ dY, const float* __restrict__ xhat,
                                                           float* __restrict__ partials, int64_t R, int C,
                                                           int64_t rpb, const uint32_t* __restrict__ key_ptr,
                                                           uint32_t thresh, float scale, int64_t elem_offset) {
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rpb;
  const int64_t r1 = (r0 + rpb < R) ? r0 + rpb : R;
  for (int c = threadIdx.x; c < C; c += 256) {
    float sg = 0.f, sb = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      float g = dY[r * C + c];
      if (do_drop) g *= ebn_drop_mult(key, static_cast<uint64_t>(r * C + c + elem_offset), thresh, scale);
      sg = fmaf(g, xhat[r * C + c], sg);
      sb += g;
    }
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * C + c] = sg;
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * C + c] = sb;
  }
}

// training: dX = istd*gamma*(dY' - dbeta_site/R - xhat*dgamma_site/R); eval: dX = dY'*gamma*istd
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dY, const float* __restrict__ xhat,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ istd,
                                                           const float* __restrict__ dgamma_site,
                                                           const float* __restrict__ dbeta_site, float* __restrict__ dX,
                                                           int64_t n, int C, float inv_R, int training,
                                                           const uint32_t* __restrict__ key_ptr, uint32_t thresh,
                                                           float scale, int64_t elem_offset) {
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % C);
    float g = dY[i];
    if (do_drop) g *= ebn_drop_mult(key, static_cast<uint64_t>(i + elem_offset), thresh, scale);
    float v = g;
    if (training) v = g - dbeta_site[c] * inv_R - xhat[i] * dgamma_site[c] * inv_R;
    dX[i] = v * gamma[c] * istd[c];
  }
}

__global__ __launch_bounds__(256) void add_vec_kernel(float* __restrict__ dst, const float* __restrict__ src, int C,
                                                      int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) dst[c] = accumulate ? dst[c] + src[c] : src[c];
}

__global__ __launch_bounds__(256) void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y,
                                                   int64_t n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256)
    y[i] = fmaf(a, x[i], y[i]);
}

// single-block deterministic sum (n is small: loss rows, or a weight matrix for the L2 term)
template <bool SQ>
__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, int64_t n, float scale,
                                                   float* __restrict__ out, int accumulate) {
  __shared__ float sw[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float v = x[i];
    s += SQ ? v * v : v;
  }
  s = ebn_wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sw[w];
    t *= scale;
    out[0] = accumulate ? out[0] + t : t;
  }
}

inline unsigned grid_for(int64_t n) {
  int64_t g = ebn_ceil_div(n, 256);
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return static_cast<unsigned>(g);
}

}  // namespace

extern "C" int64_t ebn_colsum_partials_len(int64_t R, int32_t Ccols) {
  // 2 reduction kinds x blocks x C, plus 2*C floats of per-site scratch (dgamma/dbeta of one call site)
  return ebn_colred_blocks(R) * 2 * Ccols + 2 * static_cast<int64_t>(Ccols);
}

extern "C" int ebn_bias_relu_f32(const float* X, const float* bias, float* Y, int64_t R, int32_t Ccols,
                                 ebn_stream_t stream) {
  EBN_REQUIRE(X && bias && Y, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(R >= 0 && Ccols > 0, EBN_ERR_BAD_ARG);
  if (R == 0) return EBN_OK;
  hipLaunchKernelGGL(bias_relu_kernel, dim3(grid_for(R * Ccols)), dim3(256), 0, ebn_stream(stream), X, bias, Y,
                     R * Ccols, Ccols);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_bias_relu_bwd_f32(const float* Y, const float* dY, float* dX, float* dbias, float* partials,
                                     int64_t R, int32_t Ccols, int32_t accumulate, ebn_stream_t stream) {
  EBN_REQUIRE(Y && dY && dX && dbias && partials, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(R >= 0 && Ccols > 0, EBN_ERR_BAD_ARG);
  if (R == 0) return EBN_OK;
  hipStream_t s = ebn_stream(stream);
  const int64_t nb = ebn_colred_blocks(R);
  const int64_t rpb = ebn_ceil_div(R, nb);
  hipLaunchKernelGGL(bias_relu_bwd_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, s, Y, dY, dX, partials, R,
                     Ccols, rpb);
  EBN_CHECK_LAUNCH();
  hipLaunchKernelGGL(ebn_reduce_partials_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(2 * Ccols, 32))),
                     dim3(256), 0, s, partials, static_cast<int>(nb), 2, Ccols, dbias,
                     static_cast<float*>(nullptr), accumulate);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_batchnorm_fwd_f32(const float* X, const float* gamma, const float* beta, float* moving_mean,
                                     float* moving_var, float* Y, float* xhat, float* mean_out, float* istd_out,
                                     float* partials, int64_t R, int32_t Ccols, int32_t training,
                                     const ebn_step_state* st, int32_t site, float drop_p, int64_t elem_offset,
                                     ebn_stream_t stream) {
  EBN_REQUIRE(X && gamma && beta && moving_mean && moving_var && Y && mean_out && istd_out, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(R >= 0 && Ccols > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(!training || (partials && xhat), EBN_ERR_BAD_ARG);
  if (R == 0) return EBN_OK;
  hipStream_t s = ebn_stream(stream);
  const int C = Ccols;
  const unsigned cgrid = static_cast<unsigned>(ebn_ceil_div(C, 256));
  if (training) {
    const int64_t nb = ebn_colred_blocks(R);
    const int64_t rpb = ebn_ceil_div(R, nb);
    const unsigned rgrid = static_cast<unsigned>(ebn_ceil_div(2 * C, 32));
    hipLaunchKernelGGL((col_moment_kernel<1>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, s, X,
                       static_cast<const float*>(nullptr), partials, R, C, rpb);
    hipLaunchKernelGGL(ebn_reduce_partials_kernel, dim3(rgrid), dim3(256), 0, s, partials, static_cast<int>(nb), 2, C,
                       mean_out, static_cast<float*>(nullptr), 0);
    hipLaunchKernelGGL(scale_vec_kernel, dim3(cgrid), dim3(256), 0, s, mean_out, C, 1.0f / static_cast<float>(R));
    hipLaunchKernelGGL((col_moment_kernel<2>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, s, X, mean_out, partials,
                       R, C, rpb);
    hipLaunchKernelGGL(ebn_reduce_partials_kernel, dim3(rgrid), dim3(256), 0, s, partials, static_cast<int>(nb), 2, C,
                       istd_out, static_cast<float*>(nullptr), 0);
    hipLaunchKernelGGL(scale_vec_kernel, dim3(cgrid), dim3(256), 0, s, istd_out, C, 1.0f / static_cast<float>(R));
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cgrid), dim3(256), 0, s, mean_out, istd_out, moving_mean, moving_var,
                       C);
  } else {
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cgrid), dim3(256), 0, s, moving_mean, moving_var, mean_out, istd_out,
                       C);
  }
  EBN_CHECK_LAUNCH();
  const EbnDrop dr = training ? ebn_make_drop(st, site, drop_p) : ebn_make_drop(nullptr, -1, 0.f);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, X, gamma, beta, mean_out, istd_out, Y,
                     xhat, R * C, C, dr.key_ptr, dr.thresh, dr.scale, elem_offset);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_batchnorm_bwd_f32(const float* dY, const float* xhat, const float* gamma, const float* istd,
                                     float* dX, float* dgamma, float* dbeta, float* partials, int64_t R,
                                     int32_t Ccols, int32_t training, int32_t accumulate, const ebn_step_state* st,
                                     int32_t site, float drop_p, int64_t elem_offset, ebn_stream_t stream) {
  EBN_REQUIRE(dY && xhat && gamma && istd && dX && dgamma && dbeta && partials, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(R >= 0 && Ccols > 0, EBN_ERR_BAD_ARG);
  if (R == 0) return EBN_OK;
  hipStream_t s = ebn_stream(stream);
  const int C = Ccols;
  const EbnDrop dr = training ? ebn_make_drop(st, site, drop_p) : ebn_make_drop(nullptr, -1, 0.f);
  const int64_t nb = ebn_colred_blocks(R);
  const int64_t rpb = ebn_ceil_div(R, nb);
  float* site_dg = partials + nb * 2 * C;  // per-call-site sums (the batch-stat terms must not mix sites)
  float* site_db = site_dg + C;
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, s, dY, xhat, partials, R, C,
                     rpb, dr.key_ptr, dr.thresh, dr.scale, elem_offset);
  hipLaunchKernelGGL(ebn_reduce_partials_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(2 * C, 32))), dim3(256), 0,
                     s, partials, static_cast<int>(nb), 2, C, site_dg, site_db, 0);
  EBN_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, dY, xhat, gamma, istd, site_dg,
                     site_db, dX, R * C, C, 1.0f / static_cast<float>(R), training, dr.key_ptr, dr.thresh, dr.scale,
                     elem_offset);
  const unsigned cgrid = static_cast<unsigned>(ebn_ceil_div(C, 256));
  hipLaunchKernelGGL(add_vec_kernel, dim3(cgrid), dim3(256), 0, s, dgamma, site_dg, C, accumulate);
  hipLaunchKernelGGL(add_vec_kernel, dim3(cgrid), dim3(256), 0, s, dbeta, site_db, C, accumulate);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_axpy_f32(float a, const float* x, float* y, int64_t n, ebn_stream_t stream) {
  EBN_REQUIRE(x && y && n >= 0, EBN_ERR_BAD_ARG);
  if (n == 0) return EBN_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, ebn_stream(stream), a, x, y, n);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_sum_f32(const float* x, int64_t n, float scale, float* out, int32_t accumulate,
                           ebn_stream_t stream) {
  EBN_REQUIRE(x && out && n >= 0, EBN_ERR_BAD_ARG);
  hipLaunchKernelGGL((sum_kernel<false>), dim3(1), dim3(1024), 0, ebn_stream(stream), x, n, scale, out, accumulate);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_sumsq_f32(const float* x, int64_t n, float scale, float* out, int32_t accumulate,
                             ebn_stream_t stream) {
  EBN_REQUIRE(x && out && n >= 0, EBN_ERR_BAD_ARG);
  hipLaunchKernelGGL((sum_kernel<true>), dim3(1), dim3(1024), 0, ebn_stream(stream), x, n, scale, out, accumulate);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
