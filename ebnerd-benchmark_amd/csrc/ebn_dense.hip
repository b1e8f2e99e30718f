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

// ---- stage-1 functors of the column reductions (ebn_reduce.h) ------------------------------------------------
struct ReluBwd {  // dX = dY*(Y>0); a0 = column sums of dX
  const float* Y;
  const float* dY;
  float* dX;
  int C;
  __device__ void row(int64_t r, int c, float& a0, float&) const {
    const int64_t i = r * C + c;
    const float g = (Y[i] > 0.f) ? dY[i] : 0.f;
    dX[i] = g;
    a0 += g;
  }
};

template <int P>
struct ColMoment {  // a0 = sum (x - mean)^P
  const float* X;
  const float* mean;  // may be null (P == 1)
  int C;
  __device__ void row(int64_t r, int c, float& a0, float&) const {
    const float x = X[r * C + c] - (mean ? mean[c] : 0.f);
    a0 += (P == 1) ? x : x * x;
  }
};

struct BnBwdStats {  // a0 = sum dY' xhat (dgamma), a1 = sum dY' (dbeta); dY' = dY * dropout multiplier
  const float* dY;
  const float* xhat;
  int C;
  const uint32_t* key_ptr;
  uint32_t thresh;
  float scale;
  int64_t elem_offset;
  __device__ void row(int64_t r, int c, float& a0, float& a1) const {
    const int64_t i = r * C + c;
    float g = dY[i];
    if (key_ptr != nullptr) g *= ebn_drop_mult(*key_ptr, static_cast<uint64_t>(i + elem_offset), thresh, scale);
    a0 = fmaf(g, xhat[i], a0);
    a1 += g;
  }
};

// var (sum of squared deviations / R) from the partials -> moving statistics, istd.  One thread per column.
__global__ __launch_bounds__(256) void bn_var_finalize_kernel(const float* __restrict__ partials, int nblk, int C,
                                                              float inv_R, const float* __restrict__ mean,
                                                              float* __restrict__ istd, float* __restrict__ mmean,
                                                              float* __restrict__ mvar) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += partials[static_cast<int64_t>(b) * 2 * C + c];
  const float var = s * inv_R;
  mmean[c] = mmean[c] * BN_MOM + mean[c] * (1.0f - BN_MOM);
  mvar[c] = mvar[c] * BN_MOM + var * (1.0f - BN_MOM);
  istd[c] = 1.0f / sqrtf(var + BN_EPS);
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

__global__ __launch_bounds__(256) void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y,
                                                   int64_t n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256)
    y[i] = fmaf(a, x[i], y[i]);
}

// L2 kernel regulariser in one pass over W: g += 2*lambda*W, partial[block] = sum W^2 of the block's elements
__global__ __launch_bounds__(256) void l2_reg_kernel(const float* __restrict__ W, float* __restrict__ g, int64_t n,
                                                     float two_lambda, float* __restrict__ partial) {
  __shared__ float sw[4];
  float s = 0.f;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const float w = W[i];
    g[i] = fmaf(two_lambda, w, g[i]);
    s = fmaf(w, w, s);
  }
  s = ebn_wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

struct L2Seg4 {
  const float* W[4];
  float* g[4];
  int64_t n[4];
};
// blockIdx.y = segment; partial[seg * gridDim.x + block]
__global__ __launch_bounds__(256) void l2_reg4_kernel(L2Seg4 sg, float two_lambda, float* __restrict__ partial) {
  __shared__ float sw[4];
  const int seg = blockIdx.y;
  const float* __restrict__ W = sg.W[seg];
  float* __restrict__ g = sg.g[seg];
  const int64_t n = sg.n[seg];
  float s = 0.f;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const float w = W[i];
    g[i] = fmaf(two_lambda, w, g[i]);
    s = fmaf(w, w, s);
  }
  s = ebn_wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[seg * gridDim.x + blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

// deterministic sum (optionally of squares) of n floats by ONE 1024-thread block (n is small: loss rows, partials)
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

// ---- single-launch "column strip" variants for a call site of at most STRIP_ROWS rows -------------------------------
// A TimeDistributed call site of the DocVec encoder has B*H = 640 or B*C = 160 rows: the two-stage reductions above
// cost 3-5 launches of ~4.5 us each for a few hundred KB of data.  Here one 1024-thread workgroup owns 16 columns:
// thread (c = t & 15, lane = t >> 4) keeps its <= 16 rows (lane, lane + 64, ...) in registers, column sums go through
// LDS in a fixed order (deterministic), and statistics, finalisation and the element-wise pass share one launch.
constexpr int STRIP_COLS = 16;
constexpr int STRIP_LANES = 64;
constexpr int STRIP_PER = 16;
constexpr int STRIP_ROWS = STRIP_LANES * STRIP_PER;  // 1024

// sum over the 64 row lanes of each column; every thread gets the total of its column
__device__ __forceinline__ float strip_colsum(float v, float (*sm)[STRIP_COLS + 1], int c, int lane) {
  __syncthreads();  // sm may still be read from a previous call
  sm[lane][c] = v;
  __syncthreads();
  if (lane < 8) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sm[lane * 8 + k][c];
    sm[STRIP_LANES + lane][c] = t;
  }
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) tot += sm[STRIP_LANES + k][c];
  return tot;
}

__global__ __launch_bounds__(1024) void bn_fwd_strip_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ mmean,
                                                            float* __restrict__ mvar, float* __restrict__ Y,
                                                            float* __restrict__ xhat, float* __restrict__ mean_out,
                                                            float* __restrict__ istd_out, int R, int C,
                                                            const uint32_t* __restrict__ key_ptr, uint32_t thresh,
                                                            float scale, int64_t elem_offset) {
  __shared__ float sm[STRIP_LANES + 8][STRIP_COLS + 1];
  const int cl = threadIdx.x & (STRIP_COLS - 1), lane = threadIdx.x >> 4;
  const int c = blockIdx.x * STRIP_COLS + cl;
  const bool cok = c < C;
  const float inv_R = 1.0f / static_cast<float>(R);
  float x[STRIP_PER];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < STRIP_PER; ++j) {
    const int r = lane + STRIP_LANES * j;
    const bool ok = cok && r < R;
    const float v = X[ok ? static_cast<int64_t>(r) * C + c : 0];  // unconditional, clamped address
    x[j] = ok ? v : 0.f;
    s += x[j];
  }
  const float mean = strip_colsum(s, sm, cl, lane) * inv_R;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < STRIP_PER; ++j) {
    const int r = lane + STRIP_LANES * j;
    const float d = (r < R) ? x[j] - mean : 0.f;
    q = fmaf(d, d, q);
  }
  const float var = strip_colsum(q, sm, cl, lane) * inv_R;  // biased, two-pass
  const float istd = 1.0f / sqrtf(var + BN_EPS);
  if (!cok) return;
  if (lane == 0) {
    mmean[c] = mmean[c] * BN_MOM + mean * (1.0f - BN_MOM);
    mvar[c] = mvar[c] * BN_MOM + var * (1.0f - BN_MOM);
    mean_out[c] = mean;
    istd_out[c] = istd;
  }
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  const float g = gamma[c], b = beta[c];
#pragma unroll
  for (int j = 0; j < STRIP_PER; ++j) {
    const int r = lane + STRIP_LANES * j;
    if (r >= R) break;
    const int64_t i = static_cast<int64_t>(r) * C + c;
    const float xh = (x[j] - mean) * istd;
    xhat[i] = xh;
    float y = xh * g + b;
    if (do_drop) y *= ebn_drop_mult(key, static_cast<uint64_t>(i + elem_offset), thresh, scale);
    Y[i] = y;
  }
}

__global__ __launch_bounds__(1024) void bn_bwd_strip_kernel(const float* __restrict__ dY, const float* __restrict__ xhat,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ istd, float* __restrict__ dX,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int R,
                                                            int C, int training, int accumulate,
                                                            const uint32_t* __restrict__ key_ptr, uint32_t thresh,
                                                            float scale, int64_t elem_offset) {
  __shared__ float sm[STRIP_LANES + 8][STRIP_COLS + 1];
  const int cl = threadIdx.x & (STRIP_COLS - 1), lane = threadIdx.x >> 4;
  const int c = blockIdx.x * STRIP_COLS + cl;
  const bool cok = c < C;
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  float g[STRIP_PER], xh[STRIP_PER];
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int j = 0; j < STRIP_PER; ++j) {
    const int r = lane + STRIP_LANES * j;
    const bool ok = cok && r < R;
    const int64_t i = ok ? static_cast<int64_t>(r) * C + c : 0;  // unconditional, clamped address
    float gv = dY[i];
    const float xv = xhat[i];
    if (do_drop) gv *= ebn_drop_mult(key, static_cast<uint64_t>(i + elem_offset), thresh, scale);
    g[j] = ok ? gv : 0.f;
    xh[j] = ok ? xv : 0.f;
    a0 = fmaf(g[j], xh[j], a0);
    a1 += g[j];
  }
  const float dg = strip_colsum(a0, sm, cl, lane);
  const float db = strip_colsum(a1, sm, cl, lane);
  if (!cok) return;
  if (lane == 0) {
    dgamma[c] = accumulate ? dgamma[c] + dg : dg;
    dbeta[c] = accumulate ? dbeta[c] + db : db;
  }
  const float inv_R = 1.0f / static_cast<float>(R);
  const float k = gamma[c] * istd[c];
#pragma unroll
  for (int j = 0; j < STRIP_PER; ++j) {
    const int r = lane + STRIP_LANES * j;
    if (r >= R) break;
    float v = g[j];
    if (training) v = g[j] - db * inv_R - xh[j] * dg * inv_R;
    dX[static_cast<int64_t>(r) * C + c] = v * k;
  }
}

__global__ __launch_bounds__(1024) void relu_bwd_strip_kernel(const float* __restrict__ Y, const float* __restrict__ dY,
                                                              float* __restrict__ dX, float* __restrict__ dbias, int R,
                                                              int C, int accumulate) {
  __shared__ float sm[STRIP_LANES + 8][STRIP_COLS + 1];
  const int cl = threadIdx.x & (STRIP_COLS - 1), lane = threadIdx.x >> 4;
  const int c = blockIdx.x * STRIP_COLS + cl;
  const bool cok = c < C;
  // all loads first, unconditionally addressed (clamped): a load that depends on a loaded value or sits under a
  // per-element branch is serialised by the compiler (vmcnt(0) per element)
  float y[STRIP_PER], g[STRIP_PER];
#pragma unroll
  for (int j = 0; j < STRIP_PER; ++j) {
    const int r = lane + STRIP_LANES * j;
    const bool ok = cok && r < R;
    const int64_t i = ok ? static_cast<int64_t>(r) * C + c : 0;
    y[j] = Y[i];
    g[j] = dY[i];
  }
  float a0 = 0.f;
#pragma unroll
  for (int j = 0; j < STRIP_PER; ++j) {
    const int r = lane + STRIP_LANES * j;
    if (cok && r < R) {
      const float v = (y[j] > 0.f) ? g[j] : 0.f;
      dX[static_cast<int64_t>(r) * C + c] = v;
      a0 += v;
    }
  }
  const float tot = strip_colsum(a0, sm, cl, lane);
  if (cok && lane == 0) dbias[c] = accumulate ? dbias[c] + tot : tot;
}

// ---- both TimeDistributed call sites of a training step in ONE launch -------------------------------------------------
// The encoder runs over the history block (rows [0,R0)) and the candidate block (rows [R0,R0+R1)) of one (R0+R1, C) row
// block: separate batch statistics and two moving-average updates (history first), but one workgroup per 16 columns
// walks both sites, so a Dense/BN layer costs one BN launch per direction instead of two (plus the ReLU backward of the
// Dense in front of it, which rides in the BN backward).
struct Bn2Sites {
  int R[2];
  float* mean_out[2];
  float* istd_out[2];
};

// column sums of TWO values per thread over the 64 row lanes in one pass (same fixed order as strip_colsum)
__device__ __forceinline__ void strip_colsum2(float v0, float v1, float (*sm)[2][STRIP_COLS + 1], int c, int lane, float& t0,
                                              float& t1) {
  __syncthreads();  // sm may still be read from a previous call
  sm[lane][0][c] = v0;
  sm[lane][1][c] = v1;
  __syncthreads();
  if (lane < 8) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a += sm[lane * 8 + k][0][c];
      b += sm[lane * 8 + k][1][c];
    }
    sm[STRIP_LANES + lane][0][c] = a;
    sm[STRIP_LANES + lane][1][c] = b;
  }
  __syncthreads();
  t0 = 0.f;
  t1 = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    t0 += sm[STRIP_LANES + k][0][c];
    t1 += sm[STRIP_LANES + k][1][c];
  }
}

// Both call sites in ONE pass over the R0 + R1 <= 1024 rows of the block: thread (column, row lane) keeps its <= 16 rows in
// registers whichever site they belong to, the per-site sums travel through LDS two at a time -- one load phase, four
// barrier rounds and one store phase for both sites (walking the sites one after the other paid each of them twice).
__global__ __launch_bounds__(1024) void bn2_fwd_strip_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ mmean,
                                                             float* __restrict__ mvar, float* __restrict__ Y,
                                                             float* __restrict__ xhat, Bn2Sites sites, int C,
                                                             const uint32_t* __restrict__ key_ptr, uint32_t thresh,
                                                             float scale) {
  __shared__ float sm[STRIP_LANES + 8][2][STRIP_COLS + 1];
  const int cl = threadIdx.x & (STRIP_COLS - 1), lane = threadIdx.x >> 4;
  const int c = blockIdx.x * STRIP_COLS + cl;
  const bool cok = c < C;
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  const int R0 = sites.R[0], R1 = sites.R[1], RN = R0 + R1;
  const float g = gamma[cok ? c : 0], b = beta[cok ? c : 0];
  const float mm0 = mmean[cok ? c : 0], mv0 = mvar[cok ? c : 0];
  float x[STRIP_PER];
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int j = 0; j < STRIP_PER; ++j) {
    const int r = lane + STRIP_LANES * j;
    const bool ok = cok && r < RN;
    const float v = X[ok ? static_cast<int64_t>(r) * C + c : 0];  // unconditional, clamped address
    x[j] = ok ? v : 0.f;
    s0 += (r < R0) ? x[j] : 0.f;
    s1 += (r < R0) ? 0.f : x[j];
  }
  float mean0, mean1;
  strip_colsum2(s0, s1, sm, cl, lane, mean0, mean1);
  mean0 = R0 > 0 ? mean0 / static_cast<float>(R0) : 0.f;
  mean1 = R1 > 0 ? mean1 / static_cast<float>(R1) : 0.f;
  float q0 = 0.f, q1 = 0.f;
#pragma unroll
  for (int j = 0; j < STRIP_PER; ++j) {
    const int r = lane + STRIP_LANES * j;
    const float d0 = (r < R0) ? x[j] - mean0 : 0.f;
    const float d1 = (r >= R0 && r < RN) ? x[j] - mean1 : 0.f;
    q0 = fmaf(d0, d0, q0);
    q1 = fmaf(d1, d1, q1);
  }
  float var0, var1;
  strip_colsum2(q0, q1, sm, cl, lane, var0, var1);  // biased, two-pass
  var0 = R0 > 0 ? var0 / static_cast<float>(R0) : 0.f;
  var1 = R1 > 0 ? var1 / static_cast<float>(R1) : 0.f;
  const float istd0 = 1.0f / sqrtf(var0 + BN_EPS), istd1 = 1.0f / sqrtf(var1 + BN_EPS);
  if (!cok) return;
  if (lane == 0) {
    float mm = mm0, mv = mv0;  // one moving-average update per call site, history first
    if (R0 > 0) {
      mm = mm * BN_MOM + mean0 * (1.0f - BN_MOM);
      mv = mv * BN_MOM + var0 * (1.0f - BN_MOM);
      sites.mean_out[0][c] = mean0;
      sites.istd_out[0][c] = istd0;
    }
    if (R1 > 0) {
      mm = mm * BN_MOM + mean1 * (1.0f - BN_MOM);
      mv = mv * BN_MOM + var1 * (1.0f - BN_MOM);
      sites.mean_out[1][c] = mean1;
      sites.istd_out[1][c] = istd1;
    }
    mmean[c] = mm;
    mvar[c] = mv;
  }
  for (int j = 0; j < STRIP_PER; ++j) {  // (runtime early exit, not a predicated unrolled loop: see bn2_relu_bwd_strip_kernel)
    const int r = lane + STRIP_LANES * j;
    if (r >= RN) break;
    const int64_t i = static_cast<int64_t>(r) * C + c;
    const float xh = (r < R0) ? (x[j] - mean0) * istd0 : (x[j] - mean1) * istd1;
    xhat[i] = xh;
    float y = xh * g + b;
    if (do_drop) y *= ebn_drop_mult(key, static_cast<uint64_t>(i), thresh, scale);
    Y[i] = y;
  }
}

// d(Dense pre-activation) = relu'(Rl) * BN-backward(dY) for both sites in one pass; dgamma / dbeta summed over the sites
// (the batch-statistics terms use each site's own sums); dbias = column sums of the result.
__global__ __launch_bounds__(1024) void bn2_relu_bwd_strip_kernel(const float* __restrict__ dY, const float* __restrict__ xhat,
                                                                  const float* __restrict__ Rl, const float* __restrict__ gamma,
                                                                  const float* __restrict__ istd0, const float* __restrict__ istd1,
                                                                  float* __restrict__ dX, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, float* __restrict__ dbias, int R0,
                                                                  int R1, int C, const uint32_t* __restrict__ key_ptr,
                                                                  uint32_t thresh, float scale) {
  __shared__ float sm[STRIP_LANES + 8][2][STRIP_COLS + 1];
  const int cl = threadIdx.x & (STRIP_COLS - 1), lane = threadIdx.x >> 4;
  const int c = blockIdx.x * STRIP_COLS + cl;
  const bool cok = c < C;
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  const int RN = R0 + R1;
  const float gm = gamma[cok ? c : 0];
  const float k0 = gm * istd0[cok ? c : 0], k1 = gm * istd1[cok ? c : 0];
  float g[STRIP_PER], xh[STRIP_PER], rl[STRIP_PER];
  float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;  // a0s = sum g*xhat of site s, a1s = sum g of site s
#pragma unroll
  for (int j = 0; j < STRIP_PER; ++j) {
    const int r = lane + STRIP_LANES * j;
    const bool ok = cok && r < RN;
    const int64_t i = ok ? static_cast<int64_t>(r) * C + c : 0;  // unconditional, clamped address
    float gv = dY[i];
    const float xv = xhat[i];
    rl[j] = Rl[i];
    if (do_drop) gv *= ebn_drop_mult(key, static_cast<uint64_t>(i), thresh, scale);
    g[j] = ok ? gv : 0.f;
    xh[j] = ok ? xv : 0.f;
    const float p = g[j] * xh[j];
    a00 += (r < R0) ? p : 0.f;
    a01 += (r < R0) ? 0.f : p;
    a10 += (r < R0) ? g[j] : 0.f;
    a11 += (r < R0) ? 0.f : g[j];
  }
  float dg0, dg1, db0, db1;
  strip_colsum2(a00, a01, sm, cl, lane, dg0, dg1);
  strip_colsum2(a10, a11, sm, cl, lane, db0, db1);
  const float ir0 = R0 > 0 ? 1.0f / static_cast<float>(R0) : 0.f, ir1 = R1 > 0 ? 1.0f / static_cast<float>(R1) : 0.f;
  float dbias_part = 0.f;
  if (cok) {
    for (int j = 0; j < STRIP_PER; ++j) {  // (runtime early exit: the fully unrolled, predicated form of this loop measured 2x slower)
      const int r = lane + STRIP_LANES * j;
      if (r >= RN) break;
      float v = (r < R0) ? (g[j] - db0 * ir0 - xh[j] * dg0 * ir0) * k0 : (g[j] - db1 * ir1 - xh[j] * dg1 * ir1) * k1;
      v = (rl[j] > 0.f) ? v : 0.f;
      dX[static_cast<int64_t>(r) * C + c] = v;
      dbias_part += v;
    }
  }
  float dbs, unused;
  strip_colsum2(dbias_part, 0.f, sm, cl, lane, dbs, unused);
  if (cok && lane == 0) {
    dgamma[c] = dg0 + dg1;
    dbeta[c] = db0 + db1;
    dbias[c] = dbs;
  }
}

inline unsigned grid_for(int64_t n) {
  int64_t g = ebn_ceil_div(n, 256);
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return static_cast<unsigned>(g);
}

constexpr int L2_BLOCKS = 256;

}  // namespace

extern "C" int64_t ebn_colsum_partials_len(int64_t R, int32_t Ccols) {
  // 2 reduction kinds x row blocks x C, plus 2*C floats of per-site scratch (dgamma/dbeta of one call site);
  // never less than the L2_BLOCKS floats ebn_l2_reg_f32 needs
  if (!ebn_dim_ok(R, Ccols)) return 0;
  const int64_t n = ebn_colred_blocks(R) * 2 * Ccols + 2 * static_cast<int64_t>(Ccols);
  return n > L2_BLOCKS ? n : L2_BLOCKS;
}

extern "C" int ebn_bias_relu_f32(const float* X, const float* bias, float* Y, int64_t R, int32_t Ccols,
                                 ebn_stream_t stream) {
  EBN_REQUIRE(X && bias && Y, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(R >= 0 && Ccols > 0, EBN_ERR_BAD_ARG);
  if (R == 0) return EBN_OK;
  EBN_LAUNCH(bias_relu_kernel, dim3(grid_for(R * Ccols)), dim3(256), 0, ebn_stream(stream), X, bias, Y,
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
  if (R <= STRIP_ROWS) {
    EBN_LAUNCH(relu_bwd_strip_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(Ccols, STRIP_COLS))), dim3(1024), 0, s,
                       Y, dY, dX, dbias, static_cast<int>(R), Ccols, accumulate);
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  int64_t nb;
  ebn_colred_stage1(ReluBwd{Y, dY, dX, Ccols}, partials, R, Ccols, s, &nb);
  EBN_CHECK_LAUNCH();
  ebn_reduce_partials(partials, nb, 1, Ccols, 1.0f, dbias, nullptr, accumulate, nullptr, nullptr, s);
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
  if (training && R <= STRIP_ROWS) {
    const EbnDrop dr = ebn_make_drop(st, site, drop_p);
    EBN_LAUNCH(bn_fwd_strip_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(C, STRIP_COLS))), dim3(1024), 0, s, X,
                       gamma, beta, moving_mean, moving_var, Y, xhat, mean_out, istd_out, static_cast<int>(R), C,
                       dr.key_ptr, dr.thresh, dr.scale, elem_offset);
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  if (training) {  // two-pass batch statistics of this call site: mean, then biased variance about it
    const float inv_R = 1.0f / static_cast<float>(R);
    int64_t nb;
    ebn_colred_stage1(ColMoment<1>{X, nullptr, C}, partials, R, C, s, &nb);
    ebn_reduce_partials(partials, nb, 1, C, inv_R, mean_out, nullptr, 0, nullptr, nullptr, s);
    ebn_colred_stage1(ColMoment<2>{X, mean_out, C}, partials, R, C, s, &nb);
    EBN_LAUNCH(bn_var_finalize_kernel, dim3(cgrid), dim3(256), 0, s, partials, static_cast<int>(nb), C, inv_R,
                       mean_out, istd_out, moving_mean, moving_var);
  } else {
    EBN_LAUNCH(bn_eval_stats_kernel, dim3(cgrid), dim3(256), 0, s, moving_mean, moving_var, mean_out, istd_out,
                       C);
  }
  EBN_CHECK_LAUNCH();
  const EbnDrop dr = training ? ebn_make_drop(st, site, drop_p) : ebn_make_drop(nullptr, -1, 0.f);
  EBN_LAUNCH(bn_apply_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, X, gamma, beta, mean_out, istd_out, Y,
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
  if (R <= STRIP_ROWS) {
    EBN_LAUNCH(bn_bwd_strip_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(C, STRIP_COLS))), dim3(1024), 0, s, dY,
                       xhat, gamma, istd, dX, dgamma, dbeta, static_cast<int>(R), C, training, accumulate, dr.key_ptr,
                       dr.thresh, dr.scale, elem_offset);
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  int64_t nb;
  ebn_colred_stage1(BnBwdStats{dY, xhat, C, dr.key_ptr, dr.thresh, dr.scale, elem_offset}, partials, R, C, s, &nb);
  float* site_dg = partials + nb * 2 * C;  // per-call-site sums (the batch-statistics terms must not mix sites)
  float* site_db = site_dg + C;
  ebn_reduce_partials(partials, nb, 2, C, 1.0f, dgamma, dbeta, accumulate, site_dg, site_db, s);
  EBN_CHECK_LAUNCH();
  EBN_LAUNCH(bn_bwd_apply_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, dY, xhat, gamma, istd, site_dg,
                     site_db, dX, R * C, C, 1.0f / static_cast<float>(R), training, dr.key_ptr, dr.thresh, dr.scale,
                     elem_offset);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_batchnorm2_fwd_f32(const float* X, const float* gamma, const float* beta, float* moving_mean,
                                      float* moving_var, float* Y, float* xhat, float* mean_out0, float* istd_out0,
                                      float* mean_out1, float* istd_out1, int64_t R0, int64_t R1, int32_t Ccols,
                                      const ebn_step_state* st, int32_t site, float drop_p, ebn_stream_t stream) {
  EBN_REQUIRE(X && gamma && beta && moving_mean && moving_var && Y && xhat, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(mean_out0 && istd_out0 && mean_out1 && istd_out1, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(R0 >= 0 && R1 >= 0 && Ccols > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(R0 + R1 <= STRIP_ROWS, EBN_ERR_UNSUPPORTED);  // larger blocks: one ebn_batchnorm_fwd_f32 per site
  if (R0 + R1 == 0) return EBN_OK;
  const EbnDrop dr = ebn_make_drop(st, site, drop_p);
  Bn2Sites sites{{static_cast<int>(R0), static_cast<int>(R1)}, {mean_out0, mean_out1}, {istd_out0, istd_out1}};
  EBN_LAUNCH(bn2_fwd_strip_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(Ccols, STRIP_COLS))), dim3(1024), 0,
                     ebn_stream(stream), X, gamma, beta, moving_mean, moving_var, Y, xhat, sites, Ccols, dr.key_ptr, dr.thresh,
                     dr.scale);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_batchnorm2_relu_bwd_f32(const float* dY, const float* xhat, const float* relu_out, const float* gamma,
                                           const float* istd0, const float* istd1, float* dX, float* dgamma, float* dbeta,
                                           float* dbias, int64_t R0, int64_t R1, int32_t Ccols, const ebn_step_state* st,
                                           int32_t site, float drop_p, ebn_stream_t stream) {
  EBN_REQUIRE(dY && xhat && relu_out && gamma && istd0 && istd1 && dX && dgamma && dbeta && dbias, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(R0 >= 0 && R1 >= 0 && Ccols > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(R0 + R1 <= STRIP_ROWS, EBN_ERR_UNSUPPORTED);
  if (R0 + R1 == 0) return EBN_OK;
  const EbnDrop dr = ebn_make_drop(st, site, drop_p);
  EBN_LAUNCH(bn2_relu_bwd_strip_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(Ccols, STRIP_COLS))), dim3(1024), 0,
                     ebn_stream(stream), dY, xhat, relu_out, gamma, istd0, istd1, dX, dgamma, dbeta, dbias, static_cast<int>(R0),
                     static_cast<int>(R1), Ccols, dr.key_ptr, dr.thresh, dr.scale);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

namespace {
struct Copy3 {
  const uint32_t* s[3];
  uint32_t* d[3];
  int64_t n[3];  // dwords
  ebn_step_state* st;  // non-null: also advance the step state (ebn_copy3_advance)
  double beta1, beta2;
};
__global__ __launch_bounds__(256) void copy3_kernel(Copy3 c) {
  if (c.st != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    // ebn_step_advance folded into the staging copy of a training step (one launch fewer on the step's critical path)
    ebn_step_state* st = c.st;
    const uint32_t t = st->step + 1u;
    st->step = t;
    const double b1t = pow(c.beta1, static_cast<double>(t));
    const double b2t = pow(c.beta2, static_cast<double>(t));
    st->adam_alpha = static_cast<float>(static_cast<double>(st->lr) * sqrt(1.0 - b2t) / (1.0 - b1t));
    for (uint32_t s = 0; s < EBN_N_SITES; ++s) st->drop_key[s] = ebn_dropout_key(st->seed, t, s);
  }
  const int which = blockIdx.y;
  const uint32_t* __restrict__ s = c.s[which];
  uint32_t* __restrict__ d = c.d[which];
  const int64_t n = c.n[which];
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) d[i] = s[i];
}
}  // namespace

static int copy3_launch(const void* s0, void* d0, int64_t n0, const void* s1, void* d1, int64_t n1, const void* s2,
                        void* d2, int64_t n2, ebn_step_state* st, double beta1, double beta2, ebn_stream_t stream) {
  const void* ss[3] = {s0, s1, s2};
  void* dd[3] = {d0, d1, d2};
  int64_t nn[3] = {n0, n1, n2};
  Copy3 c;
  c.st = st;
  c.beta1 = beta1;
  c.beta2 = beta2;
  int64_t most = 0;
  for (int i = 0; i < 3; ++i) {
    EBN_REQUIRE(nn[i] >= 0 && (nn[i] % 4) == 0, EBN_ERR_BAD_ARG);
    if (ss[i] == nullptr || nn[i] == 0) nn[i] = 0;
    else EBN_REQUIRE(dd[i] != nullptr, EBN_ERR_BAD_ARG);
    c.s[i] = static_cast<const uint32_t*>(ss[i]);
    c.d[i] = static_cast<uint32_t*>(dd[i]);
    c.n[i] = nn[i] / 4;
    if (c.n[i] > most) most = c.n[i];
  }
  if (most == 0 && st == nullptr) return EBN_OK;
  int64_t gx = ebn_ceil_div(most, 256);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  EBN_LAUNCH(copy3_kernel, dim3(static_cast<unsigned>(gx), 3), dim3(256), 0, ebn_stream(stream), c);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_copy3(const void* s0, void* d0, int64_t n0, const void* s1, void* d1, int64_t n1, const void* s2,
                         void* d2, int64_t n2, ebn_stream_t stream) {
  return copy3_launch(s0, d0, n0, s1, d1, n1, s2, d2, n2, nullptr, 0.0, 0.0, stream);
}

extern "C" int ebn_copy3_advance(const void* s0, void* d0, int64_t n0, const void* s1, void* d1, int64_t n1,
                                 const void* s2, void* d2, int64_t n2, ebn_step_state* st, double beta1, double beta2,
                                 ebn_stream_t stream) {
  EBN_REQUIRE(st, EBN_ERR_BAD_ARG);
  return copy3_launch(s0, d0, n0, s1, d1, n1, s2, d2, n2, st, beta1, beta2, stream);
}

extern "C" int ebn_axpy_f32(float a, const float* x, float* y, int64_t n, ebn_stream_t stream) {
  EBN_REQUIRE(x && y && n >= 0, EBN_ERR_BAD_ARG);
  if (n == 0) return EBN_OK;
  EBN_LAUNCH(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, ebn_stream(stream), a, x, y, n);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_l2_reg4_f32(const float* W0, float* g0, int64_t n0, const float* W1, float* g1, int64_t n1,
                               const float* W2, float* g2, int64_t n2, const float* W3, float* g3, int64_t n3,
                               float lambda, float* partials, float* loss, ebn_stream_t stream) {
  EBN_REQUIRE(partials && loss, EBN_ERR_BAD_ARG);
  const float* Ws[4] = {W0, W1, W2, W3};
  float* gs[4] = {g0, g1, g2, g3};
  const int64_t ns[4] = {n0, n1, n2, n3};
  L2Seg4 sg;
  int nseg = 0;
  int64_t most = 0;
  for (int i = 0; i < 4; ++i) {
    EBN_REQUIRE(ns[i] >= 0, EBN_ERR_BAD_ARG);
    if (Ws[i] == nullptr || ns[i] == 0) continue;
    EBN_REQUIRE(gs[i] != nullptr, EBN_ERR_BAD_ARG);
    sg.W[nseg] = Ws[i];
    sg.g[nseg] = gs[i];
    sg.n[nseg] = ns[i];
    if (ns[i] > most) most = ns[i];
    ++nseg;
  }
  if (nseg == 0) return EBN_OK;
  for (int i = nseg; i < 4; ++i) {
    sg.W[i] = nullptr;
    sg.g[i] = nullptr;
    sg.n[i] = 0;
  }
  hipStream_t s = ebn_stream(stream);
  int64_t grid = ebn_ceil_div(most, 256 * 4);
  if (grid > L2_BLOCKS) grid = L2_BLOCKS;
  if (grid < 1) grid = 1;
  EBN_LAUNCH(l2_reg4_kernel, dim3(static_cast<unsigned>(grid), static_cast<unsigned>(nseg)), dim3(256), 0, s, sg,
                     2.0f * lambda, partials);
  EBN_LAUNCH((sum_kernel<false>), dim3(1), dim3(1024), 0, s, partials, grid * nseg, lambda, loss, 1);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_l2_reg_f32(const float* W, float* gW, int64_t n, float lambda, float* partials, float* loss,
                              ebn_stream_t stream) {
  EBN_REQUIRE(W && gW && partials && loss && n >= 0, EBN_ERR_BAD_ARG);
  if (n == 0) return EBN_OK;
  hipStream_t s = ebn_stream(stream);
  int64_t grid = ebn_ceil_div(n, 256 * 4);
  if (grid > L2_BLOCKS) grid = L2_BLOCKS;
  if (grid < 1) grid = 1;
  EBN_LAUNCH(l2_reg_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, s, W, gW, n, 2.0f * lambda, partials);
  EBN_LAUNCH((sum_kernel<false>), dim3(1), dim3(1024), 0, s, partials, grid, lambda, loss, 1);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_sum_f32(const float* x, int64_t n, float scale, float* out, int32_t accumulate,
                           ebn_stream_t stream) {
  EBN_REQUIRE(x && out && n >= 0, EBN_ERR_BAD_ARG);
  EBN_LAUNCH((sum_kernel<false>), dim3(1), dim3(1024), 0, ebn_stream(stream), x, n, scale, out, accumulate);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_sumsq_f32(const float* x, int64_t n, float scale, float* out, int32_t accumulate,
                             ebn_stream_t stream) {
  EBN_REQUIRE(x && out && n >= 0, EBN_ERR_BAD_ARG);
  EBN_LAUNCH((sum_kernel<true>), dim3(1), dim3(1024), 0, ebn_stream(stream), x, n, scale, out, accumulate);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
