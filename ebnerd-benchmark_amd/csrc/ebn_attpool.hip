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
  if (dX == nullptr) return;
  for (int l = 0; l < L; ++l) {
    const float wl = w[n * L + l];
    float* dr = dX + (n * L + l) * E;
    for (int c = tid; c < E; c += POOL_THREADS) dr[c] = wl * g[c];
  }
}

// ---- float4 variants (A % 4 == 0, A <= 256, E % 4 == 0, E <= 1024, 16-byte aligned rows) ----------------------------
// Same arithmetic and the same summation orders as the scalar kernels above, restructured for memory-level
// parallelism: a wave first issues the loads of ALL the rows it owns (8 rows x 16 bytes per lane in flight), then does
// the math -- the scalar kernels walk their rows one dependent load at a time and sit at ~1/3 of the HBM rate.
constexpr int POOL_RB = 8;  // rows per wave per batch

// Loads are unconditional with a CLAMPED OFFSET (a select between two pointers, or a select on the loaded value right
// next to the load, is turned into a branch around the load by hipcc, and every such load then waits vmcnt(0) on its
// own).  ld4_clamped returns garbage-but-finite data for !ok; ld4_or_zero adds the zeroing select.
__device__ __forceinline__ float4 ld4_clamped(const float* __restrict__ base, int64_t off, bool ok) {
  return *reinterpret_cast<const float4*>(base + (ok ? off : 0));
}
__device__ __forceinline__ float4 ld4_or_zero(const float* __restrict__ base, int64_t off, bool ok) {
  const float4 v = ld4_clamped(base, off, ok);
  float4 r;
  r.x = ok ? v.x : 0.f;
  r.y = ok ? v.y : 0.f;
  r.z = ok ? v.z : 0.f;
  r.w = ok ? v.w : 0.f;
  return r;
}

__global__ __launch_bounds__(POOL_THREADS) void attpool_fwd_vec_kernel(float* __restrict__ U,
                                                                        const float* __restrict__ b,
                                                                        const float* __restrict__ q,
                                                                        const float* __restrict__ X,
                                                                        float* __restrict__ out,
                                                                        float* __restrict__ w, int L, int E, int A) {
  extern __shared__ float sm[];  // e / w of this sequence: L floats
  __shared__ float4 part_acc[POOL_THREADS];
  const int64_t n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A4 = A >> 2, E4 = E >> 2;
  {
    const bool kok = lane < A4;
    const float4 bb = ld4_or_zero(b, lane * 4, kok), qq = ld4_or_zero(q, lane * 4, kok);
    for (int l0 = wave; l0 < L; l0 += POOL_WAVES * POOL_RB) {
      float4 u[POOL_RB];
#pragma unroll
      for (int j = 0; j < POOL_RB; ++j) {
        const int l = l0 + POOL_WAVES * j;
        u[j] = ld4_clamped(U, (n * L + l) * A + lane * 4, kok && l < L);  // rows >= L are never used; lanes >= A4 are masked below
      }
#pragma unroll
      for (int j = 0; j < POOL_RB; ++j) {
        const int l = l0 + POOL_WAVES * j;
        if (l >= L) break;  // wave-uniform
        float4 t;
        t.x = tanhf(u[j].x + bb.x);
        t.y = tanhf(u[j].y + bb.y);
        t.z = tanhf(u[j].z + bb.z);
        t.w = tanhf(u[j].w + bb.w);
        if (kok) *reinterpret_cast<float4*>(U + (n * L + l) * A + lane * 4) = t;
        // same order as the scalar kernel would give for 4 consecutive k of one lane is not required: the
        // oracle comparison is tolerance-based; the order here is fixed (deterministic)
        float part = kok ? fmaf(t.w, qq.w, fmaf(t.z, qq.z, fmaf(t.y, qq.y, t.x * qq.x))) : 0.f;
        part = ebn_wave_sum(part);
        if (lane == 0) sm[l] = part;
      }
    }
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
  // out[c] = sum_l w_l X[l][c]: thread (c4, part) sums the rows l = part (mod NP); parts combined in fixed order
  const int NP = (E4 <= POOL_THREADS) ? POOL_THREADS / E4 : 1;
  for (int c0 = 0; c0 < E4; c0 += POOL_THREADS) {
    const int c4 = c0 + tid % ((E4 <= POOL_THREADS) ? E4 : POOL_THREADS);
    const int part = (E4 <= POOL_THREADS) ? tid / E4 : 0;
    const bool ok = c4 < E4 && part < NP;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l0 = part; l0 < L; l0 += NP * POOL_RB) {
      float4 x[POOL_RB];
#pragma unroll
      for (int j = 0; j < POOL_RB; ++j) {
        const int l = l0 + NP * j;
        x[j] = ld4_clamped(X, (n * L + l) * E + c4 * 4, ok && l < L);  // weight 0 for l >= L, result unused for !ok
      }
#pragma unroll
      for (int j = 0; j < POOL_RB; ++j) {
        const int l = l0 + NP * j;
        const float wl = (l < L) ? sm[l] : 0.f;
        acc.x = fmaf(wl, x[j].x, acc.x);
        acc.y = fmaf(wl, x[j].y, acc.y);
        acc.z = fmaf(wl, x[j].z, acc.z);
        acc.w = fmaf(wl, x[j].w, acc.w);
      }
    }
    if (NP > 1) {
      part_acc[tid] = acc;
      __syncthreads();
      if (part == 0 && ok) {
        for (int pp = 1; pp < NP; ++pp) {
          const float4 o = part_acc[pp * E4 + c4];
          acc.x += o.x;
          acc.y += o.y;
          acc.z += o.z;
          acc.w += o.w;
        }
      }
      __syncthreads();
    }
    if (part == 0 && ok) *reinterpret_cast<float4*>(out + n * E + c4 * 4) = acc;
  }
}

// write_dx = 0: only de is produced (the caller folds w_l * dout into the GEMM that accumulates d(x), see
// ebn_gemm_f32_rank1)
__global__ __launch_bounds__(POOL_THREADS) void attpool_bwd_pool_vec_kernel(const float* __restrict__ X,
                                                                             const float* __restrict__ w,
                                                                             const float* __restrict__ dout,
                                                                             float* __restrict__ dX,
                                                                             float* __restrict__ de, int L, int E,
                                                                             int write_dx) {
  extern __shared__ float sm[];  // dw[L]
  const int64_t n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int E4 = E >> 2;
  const float* g = dout + n * E;
  constexpr int VL = 4;  // float4 per lane per row: E <= 1024
  float4 gg[VL];
#pragma unroll
  for (int v = 0; v < VL; ++v) gg[v] = ld4_or_zero(g, (lane + 64 * v) * 4, lane + 64 * v < E4);
  for (int l0 = wave; l0 < L; l0 += POOL_WAVES * POOL_RB) {
    float part[POOL_RB];
#pragma unroll
    for (int j = 0; j < POOL_RB; ++j) part[j] = 0.f;
#pragma unroll
    for (int v = 0; v < VL; ++v) {
      if (64 * v >= E4) break;  // uniform
      float4 x[POOL_RB];
#pragma unroll
      for (int j = 0; j < POOL_RB; ++j) {
        const int l = l0 + POOL_WAVES * j;
        x[j] = ld4_clamped(X, (n * L + l) * E + (lane + 64 * v) * 4, l < L && lane + 64 * v < E4);  // gg = 0 / row unused
      }
#pragma unroll
      for (int j = 0; j < POOL_RB; ++j)
        part[j] = fmaf(gg[v].w, x[j].w, fmaf(gg[v].z, x[j].z, fmaf(gg[v].y, x[j].y, fmaf(gg[v].x, x[j].x, part[j]))));
    }
#pragma unroll
    for (int j = 0; j < POOL_RB; ++j) {
      const int l = l0 + POOL_WAVES * j;
      const float t = ebn_wave_sum(part[j]);
      if (lane == 0 && l < L) sm[l] = t;
    }
  }
  __syncthreads();
  if (wave == 0) {
    float s = 0.f;
    for (int l = lane; l < L; l += 64) s = fmaf(w[n * L + l], sm[l], s);
    s = ebn_wave_sum(s);
    for (int l = lane; l < L; l += 64) de[n * L + l] = w[n * L + l] * (sm[l] - s);
  }
  if (!write_dx) return;
  for (int c4 = tid; c4 < E4; c4 += POOL_THREADS) {
    const float4 gv = *reinterpret_cast<const float4*>(g + c4 * 4);
    for (int l = 0; l < L; ++l) {
      const float wl = w[n * L + l];
      *reinterpret_cast<float4*>(dX + (n * L + l) * E + c4 * 4) = make_float4(wl * gv.x, wl * gv.y, wl * gv.z, wl * gv.w);
    }
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
    // 8 rows at a time: all loads of a batch are issued before the first store to U (a store into the array being read
    // keeps the compiler from hoisting the next row's load: one dependent round trip per row otherwise)
    for (int64_t rb = r0; rb < r1; rb += 8) {
      float u[8], der[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t r = (rb + j < r1) ? rb + j : r1 - 1;  // clamped, unconditional
        u[j] = U[r * A + k];
        der[j] = de[r];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (rb + j >= r1) break;
        dq = fmaf(der[j], u[j], dq);
        const float dp = der[j] * qk * (1.0f - u[j] * u[j]);
        U[(rb + j) * A + k] = dp;
        db += dp;
      }
    }
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * A + k] = dq;
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * A + k] = db;
  }
}

}  // namespace

// The same for A % 4 == 0, A <= 1024 and 16-byte aligned U: thread = (4 columns, row stripe); a thread's rows of a batch are requested
// together as 16-byte pieces (eight rows = 128 bytes in flight per thread where the kernel above has 32), the stripes are combined
// through LDS in a fixed order.
__global__ __launch_bounds__(POOL_THREADS) void attpool_bwd_dpre_vec_kernel(float* __restrict__ U, const float* __restrict__ q,
                                                                             const float* __restrict__ de, float* __restrict__ partials,
                                                                             int64_t R, int A, int64_t rpb) {
  extern __shared__ float sm[];  // [2][RS][A]
  const int A4 = A >> 2, RS = POOL_THREADS / A4;
  const int tid = threadIdx.x, c4 = tid % A4, rs = tid / A4;
  const bool active = rs < RS;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rpb;
  const int64_t r1 = (r0 + rpb < R) ? r0 + rpb : R;
  float4 dq = make_float4(0.f, 0.f, 0.f, 0.f), db = dq;
  if (active) {
    const float4 qk = *reinterpret_cast<const float4*>(q + c4 * 4);
    for (int64_t rb = r0 + rs; rb < r1; rb += static_cast<int64_t>(8) * RS) {
      float4 u[8];
      float der[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int64_t r = rb + static_cast<int64_t>(j) * RS;
        r = r < r1 ? r : r1 - 1;  // clamped, unconditional
        u[j] = *reinterpret_cast<const float4*>(U + r * A + c4 * 4);
        der[j] = de[r];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t r = rb + static_cast<int64_t>(j) * RS;
        if (r >= r1) break;
        const float d = der[j];
        dq.x = fmaf(d, u[j].x, dq.x);
        dq.y = fmaf(d, u[j].y, dq.y);
        dq.z = fmaf(d, u[j].z, dq.z);
        dq.w = fmaf(d, u[j].w, dq.w);
        const float4 dp = make_float4(d * qk.x * (1.0f - u[j].x * u[j].x), d * qk.y * (1.0f - u[j].y * u[j].y),
                                      d * qk.z * (1.0f - u[j].z * u[j].z), d * qk.w * (1.0f - u[j].w * u[j].w));
        *reinterpret_cast<float4*>(U + r * A + c4 * 4) = dp;
        db.x += dp.x;
        db.y += dp.y;
        db.z += dp.z;
        db.w += dp.w;
      }
    }
    *reinterpret_cast<float4*>(&sm[rs * A + c4 * 4]) = dq;
    *reinterpret_cast<float4*>(&sm[(RS + rs) * A + c4 * 4]) = db;
  }
  __syncthreads();
  for (int k = tid; k < A; k += POOL_THREADS) {
    float tq = 0.f, tb = 0.f;
    for (int i = 0; i < RS; ++i) {
      tq += sm[i * A + k];
      tb += sm[(RS + i) * A + k];
    }
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * A + k] = tq;
    partials[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * A + k] = tb;
  }
}

extern "C" int64_t ebn_attpool_partials_len(int64_t R, int32_t A) { return ebn_dim_ok(R, A) ? ebn_colred_blocks(R) * 2 * A : 0; }

extern "C" int ebn_attpool_fwd_f32(float* U, const float* b, const float* q, const float* X, float* out,
                                   float* w, int64_t n_seq, int32_t L, int32_t E, int32_t A,
                                   ebn_stream_t stream) {
  EBN_REQUIRE(U && b && q && X && out && w, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_seq >= 0 && L > 0 && E > 0 && A > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(L <= 8192, EBN_ERR_UNSUPPORTED);
  if (n_seq == 0) return EBN_OK;
  const bool vec = (A % 4 == 0) && A <= 256 && (E % 4 == 0) && E <= 1024 && ebn_aligned16(U) && ebn_aligned16(b) &&
                   ebn_aligned16(q) && ebn_aligned16(X) && ebn_aligned16(out);
  if (vec)
    EBN_LAUNCH(attpool_fwd_vec_kernel, dim3(static_cast<unsigned>(n_seq)), dim3(POOL_THREADS),
                       L * sizeof(float), ebn_stream(stream), U, b, q, X, out, w, L, E, A);
  else
    EBN_LAUNCH(attpool_fwd_kernel, dim3(static_cast<unsigned>(n_seq)), dim3(POOL_THREADS),
                       L * sizeof(float), ebn_stream(stream), U, b, q, X, out, w, L, E, A);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_attpool_bwd_pool_f32(const float* X, const float* w, const float* dout, float* dX,
                                        float* de, int64_t n_seq, int32_t L, int32_t E,
                                        ebn_stream_t stream) {
  EBN_REQUIRE(X && w && dout && de, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_seq >= 0 && L > 0 && E > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(L <= 8192, EBN_ERR_UNSUPPORTED);
  if (n_seq == 0) return EBN_OK;
  const bool vec = (E % 4 == 0) && E <= 1024 && ebn_aligned16(X) && ebn_aligned16(dout) && ebn_aligned16(dX);  // NULL is aligned
  if (vec)
    EBN_LAUNCH(attpool_bwd_pool_vec_kernel, dim3(static_cast<unsigned>(n_seq)), dim3(POOL_THREADS),
                       L * sizeof(float), ebn_stream(stream), X, w, dout, dX, de, L, E, dX != nullptr ? 1 : 0);
  else
    EBN_LAUNCH(attpool_bwd_pool_kernel, dim3(static_cast<unsigned>(n_seq)), dim3(POOL_THREADS),
                       L * sizeof(float), ebn_stream(stream), X, w, dout, dX, de, L, E);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_attpool_bwd_dpre_f32(float* U, const float* q, const float* de, float* dq, float* db,
                                        float* partials, int64_t R, int32_t A, int32_t accumulate,
                                        ebn_stream_t stream) {
  EBN_REQUIRE(U && q && de && partials && ((dq != nullptr) == (db != nullptr)), EBN_ERR_BAD_ARG);
  EBN_REQUIRE(R >= 0 && A > 0, EBN_ERR_BAD_ARG);
  if (R == 0) return EBN_OK;
  const int64_t nb = ebn_colred_blocks(R);
  const int64_t rpb = ebn_ceil_div(R, nb);
  if ((A % 4) == 0 && A >= 4 && A <= 1024 && ebn_aligned16(U) && ebn_aligned16(q)) {
    const size_t lds = static_cast<size_t>(2) * (POOL_THREADS / (A / 4)) * A * sizeof(float);  // <= 2 x 256 x 4 floats = 8 KB
    EBN_LAUNCH(attpool_bwd_dpre_vec_kernel, dim3(static_cast<unsigned>(nb)), dim3(POOL_THREADS), lds, ebn_stream(stream), U, q, de, partials,
                       R, A, rpb);
  } else {
    EBN_LAUNCH(attpool_bwd_dpre_kernel, dim3(static_cast<unsigned>(nb)), dim3(POOL_THREADS), 0,
                       ebn_stream(stream), U, q, de, partials, R, A, rpb);
  }
  EBN_CHECK_LAUNCH();
  if (dq == nullptr) return EBN_OK;  // the sum over the row blocks is left to ebn_grad_finish_f32 (EBN_FINISH_COLRED job over `partials`)
  ebn_reduce_partials(partials, nb, 2, A, 1.0f, dq, db, accumulate, nullptr, nullptr, ebn_stream(stream));
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

