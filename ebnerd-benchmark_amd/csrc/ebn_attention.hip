// a3 / a6: core of the reference SelfAttention layer after the Q/K/V projections
// (layers.py:231-252): S = QK^T/sqrt(d), P = softmax_rows(S), O = P^T V  -- the TRANSPOSED
// attention matrix multiplies V (adjoint_a=True, layers.py:249); no mask (layers.py:209-211).
//
// Sequences are tiny (L = title_size 30 or history_size 20..50, d = 16/20), so one
// (sequence, head) problem is an L x L tile that lives in LDS next to its Q/K/V tiles; this
// is latency/LDS-bound VALU work (6 % of the projection FLOPs), not MFMA work.  One 64-lane
// wave per workgroup; when L <= 32 the two 32-lane halves of the wave work on two independent
// (sequence, head) problems (ds_read_b32 serves the halves as separate lane groups, so their
// broadcast reads never conflict).  Lane i owns row i of S/P (softmax + dQ/dV), lane j owns
// column j (O, dK): row accesses use an odd row stride, column accesses are unit-stride.
#include "ebn_common.h"

// MFMA fast path (ebn_attention_mfma.hip): L <= 32, d in {16,20,32}, 16-byte friendly layouts.
int ebn_attn_mfma_fwd(const float* qkv, int64_t ld_qkv, float* out, int64_t ld_out, int64_t n_seq, int32_t L,
                      int32_t h, int32_t d, const EbnDrop& dr, hipStream_t s, bool* handled);
int ebn_attn_mfma_bwd(const float* qkv, int64_t ld_qkv, const float* dout, int64_t ld_dout, float* dqkv,
                      int64_t ld_dqkv, int64_t n_seq, int32_t L, int32_t h, int32_t d, const EbnDrop& dr,
                      hipStream_t s, bool* handled, const float* pool_w, const float* pool_d, int64_t ld_pool);

namespace {

constexpr int DMAX = 32;  // head_dim <= 32 (reference configs: 20, 16)

struct AttnArgs {
  const float* qkv;
  int64_t ld_qkv;
  const float* dout;  // bwd only
  int64_t ld_dout;
  float* out;  // fwd: out, bwd: dqkv
  int64_t ld_out;
  int64_t n_prob;  // n_seq * h
  int32_t L, h, d;
  const uint32_t* key_ptr;
  uint32_t thresh;
  float scale;
};

// cooperative tile load by the GW lanes of one group: tile[l*d + c] = src[(row0+l)*ld + col0 + c]
template <int GW>
__device__ __forceinline__ void load_tile(float* __restrict__ tile, const float* __restrict__ src,
                                          int64_t ld, int64_t row0, int col0, int L, int d, int gl) {
  const int n = L * d;
  for (int e = gl; e < n; e += GW) {
    const int l = e / d;
    const int c = e - l * d;
    tile[e] = src[(row0 + l) * ld + col0 + c];
  }
}

// Computes normalised P (row-major, stride LP) for one problem; lane gl = row i.
template <int GW>
__device__ __forceinline__ void softmax_scores(const float* __restrict__ sQ, const float* __restrict__ sK,
                                               float* __restrict__ sP, int L, int LP, int d, int gl) {
  if (gl < L) {
    float q[DMAX];
#pragma unroll
    for (int c = 0; c < DMAX; ++c) q[c] = (c < d) ? sQ[gl * d + c] : 0.f;
    const float inv = 1.0f / sqrtf(static_cast<float>(d));
    float mx = -INFINITY;
    for (int j = 0; j < L; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < DMAX; ++c)
        if (c < d) s = fmaf(q[c], sK[j * d + c], s);
      s *= inv;
      sP[gl * LP + j] = s;
      mx = fmaxf(mx, s);
    }
    float sum = 0.f;
    for (int j = 0; j < L; ++j) {
      const float p = expf(sP[gl * LP + j] - mx);
      sP[gl * LP + j] = p;
      sum += p;
    }
    for (int j = 0; j < L; ++j) sP[gl * LP + j] = sP[gl * LP + j] / sum;
  }
}

template <int GW>
__global__ __launch_bounds__(64) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int G = 64 / GW;
  const int lane = threadIdx.x;
  const int g = lane / GW;
  const int gl = lane % GW;
  const int L = a.L, d = a.d;
  const int LP = (L & 1) ? L + 2 : L + 1;  // odd row stride
  const int tile = L * d;
  const int per_group = 3 * tile + L * LP;
  float* base = smem + g * per_group;
  float* sQ = base;
  float* sK = base + tile;
  float* sV = base + 2 * tile;
  float* sP = base + 3 * tile;

  const int64_t prob = static_cast<int64_t>(blockIdx.x) * G + g;
  const bool active = prob < a.n_prob;
  const int64_t seq = active ? prob / a.h : 0;
  const int head = active ? static_cast<int>(prob - seq * a.h) : 0;
  const int E = a.h * d;
  const int64_t row0 = seq * L;
  if (active) {
    load_tile<GW>(sQ, a.qkv, a.ld_qkv, row0, head * d, L, d, gl);
    load_tile<GW>(sK, a.qkv, a.ld_qkv, row0, E + head * d, L, d, gl);
    load_tile<GW>(sV, a.qkv, a.ld_qkv, row0, 2 * E + head * d, L, d, gl);
  }
  __syncthreads();
  if (active) softmax_scores<GW>(sQ, sK, sP, L, LP, d, gl);
  __syncthreads();
  if (active && gl < L) {
    const int j = gl;
    float o[DMAX];
#pragma unroll
    for (int c = 0; c < DMAX; ++c) o[c] = 0.f;
    for (int i = 0; i < L; ++i) {
      const float p = sP[i * LP + j];
#pragma unroll
      for (int c = 0; c < DMAX; ++c)
        if (c < d) o[c] = fmaf(p, sV[i * d + c], o[c]);
    }
    const bool do_drop = a.key_ptr != nullptr;
    const uint32_t key = do_drop ? *a.key_ptr : 0u;
    float* dst = a.out + (row0 + j) * a.ld_out + head * d;
    const uint64_t e0 = static_cast<uint64_t>(row0 + j) * E + head * d;  // logical (n,l,e) index
#pragma unroll
    for (int c = 0; c < DMAX; ++c) {
      if (c < d) {
        float v = o[c];
        if (do_drop) v *= ebn_drop_mult(key, e0 + c, a.thresh, a.scale);
        dst[c] = v;
      }
    }
  }
}

template <int GW>
__global__ __launch_bounds__(64) void attn_bwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int G = 64 / GW;
  const int lane = threadIdx.x;
  const int g = lane / GW;
  const int gl = lane % GW;
  const int L = a.L, d = a.d;
  const int LP = (L & 1) ? L + 2 : L + 1;
  const int tile = L * d;
  const int per_group = 4 * tile + 2 * L * LP;
  float* base = smem + g * per_group;
  float* sQ = base;
  float* sK = base + tile;
  float* sV = base + 2 * tile;
  float* sG = base + 3 * tile;  // dO (after dropout mask)
  float* sP = base + 4 * tile;
  float* sD = sP + L * LP;  // dP then dS

  const int64_t prob = static_cast<int64_t>(blockIdx.x) * G + g;
  const bool active = prob < a.n_prob;
  const int64_t seq = active ? prob / a.h : 0;
  const int head = active ? static_cast<int>(prob - seq * a.h) : 0;
  const int E = a.h * d;
  const int64_t row0 = seq * L;
  const float inv = 1.0f / sqrtf(static_cast<float>(d));
  if (active) {
    load_tile<GW>(sQ, a.qkv, a.ld_qkv, row0, head * d, L, d, gl);
    load_tile<GW>(sK, a.qkv, a.ld_qkv, row0, E + head * d, L, d, gl);
    load_tile<GW>(sV, a.qkv, a.ld_qkv, row0, 2 * E + head * d, L, d, gl);
    const bool do_drop = a.key_ptr != nullptr;
    const uint32_t key = do_drop ? *a.key_ptr : 0u;
    for (int e = gl; e < tile; e += GW) {
      const int l = e / d;
      const int c = e - l * d;
      float v = a.dout[(row0 + l) * a.ld_dout + head * d + c];
      if (do_drop) v *= ebn_drop_mult(key, static_cast<uint64_t>(row0 + l) * E + head * d + c, a.thresh, a.scale);
      sG[e] = v;
    }
  }
  __syncthreads();
  if (active) softmax_scores<GW>(sQ, sK, sP, L, LP, d, gl);
  __syncthreads();
  float* dq_dst = a.out;  // dqkv
  if (active && gl < L) {
    const int i = gl;
    // dV[i,:] = sum_j P[i,j] dO[j,:]
    float acc[DMAX];
#pragma unroll
    for (int c = 0; c < DMAX; ++c) acc[c] = 0.f;
    for (int j = 0; j < L; ++j) {
      const float p = sP[i * LP + j];
#pragma unroll
      for (int c = 0; c < DMAX; ++c)
        if (c < d) acc[c] = fmaf(p, sG[j * d + c], acc[c]);
    }
    float* dv = dq_dst + (row0 + i) * a.ld_out + 2 * E + head * d;
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
      if (c < d) dv[c] = acc[c];
    // dP[i,j] = V[i,:].dO[j,:] ; dS = P*(dP - sum_j P dP)
    float v[DMAX];
#pragma unroll
    for (int c = 0; c < DMAX; ++c) v[c] = (c < d) ? sV[i * d + c] : 0.f;
    float rowdot = 0.f;
    for (int j = 0; j < L; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < DMAX; ++c)
        if (c < d) s = fmaf(v[c], sG[j * d + c], s);
      sD[i * LP + j] = s;
      rowdot = fmaf(sP[i * LP + j], s, rowdot);
    }
    for (int j = 0; j < L; ++j) sD[i * LP + j] = sP[i * LP + j] * (sD[i * LP + j] - rowdot);
    // dQ[i,:] = inv * sum_j dS[i,j] K[j,:]
#pragma unroll
    for (int c = 0; c < DMAX; ++c) acc[c] = 0.f;
    for (int j = 0; j < L; ++j) {
      const float s = sD[i * LP + j];
#pragma unroll
      for (int c = 0; c < DMAX; ++c)
        if (c < d) acc[c] = fmaf(s, sK[j * d + c], acc[c]);
    }
    float* dq = dq_dst + (row0 + i) * a.ld_out + head * d;
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
      if (c < d) dq[c] = acc[c] * inv;
  }
  __syncthreads();
  if (active && gl < L) {
    const int j = gl;
    // dK[j,:] = inv * sum_i dS[i,j] Q[i,:]
    float acc[DMAX];
#pragma unroll
    for (int c = 0; c < DMAX; ++c) acc[c] = 0.f;
    for (int i = 0; i < L; ++i) {
      const float s = sD[i * LP + j];
#pragma unroll
      for (int c = 0; c < DMAX; ++c)
        if (c < d) acc[c] = fmaf(s, sQ[i * d + c], acc[c]);
    }
    float* dk = dq_dst + (row0 + j) * a.ld_out + E + head * d;
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
      if (c < d) dk[c] = acc[c] * inv;
  }
}

// ---- long sequences: 64 < L <= 256 (a history_size beyond the EB-NeRD defaults) ------------------------------------
// One 256-thread workgroup per (sequence, head); thread t owns row t (softmax statistics, dV, dQ) and later column t
// (O, dK).  No L x L matrix is kept: scores are recomputed from the Q/K tiles in LDS (L*d FMAs per thread and pass,
// trivial next to the projections), only the per-row statistics max / 1/Z / rowdot live in LDS.
constexpr int LONG_THREADS = 256;
constexpr int LONG_LMAX = 256;

__device__ __forceinline__ void long_load_tile(float* __restrict__ tile, const float* __restrict__ src, int64_t ld,
                                               int64_t row0, int col0, int L, int d) {
  for (int e = threadIdx.x; e < L * d; e += LONG_THREADS) {
    const int l = e / d, c = e - l * d;
    tile[e] = src[(row0 + l) * ld + col0 + c];
  }
}

__device__ __forceinline__ float long_dot(const float (&x)[DMAX], const float* __restrict__ y, int d) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < DMAX; ++c)
    if (c < d) s = fmaf(x[c], y[c], s);
  return s;
}

// row statistics of thread i = threadIdx.x: sM[i] = max_j s_ij, sZ[i] = 1 / sum_j exp(s_ij - max)
__device__ __forceinline__ void long_row_stats(const float* __restrict__ sQ, const float* __restrict__ sK,
                                               float* __restrict__ sM, float* __restrict__ sZ, int L, int d, float inv) {
  const int i = threadIdx.x;
  if (i < L) {
    float q[DMAX];
#pragma unroll
    for (int c = 0; c < DMAX; ++c) q[c] = (c < d) ? sQ[i * d + c] : 0.f;
    float mx = -INFINITY;
    for (int j = 0; j < L; ++j) mx = fmaxf(mx, long_dot(q, sK + j * d, d) * inv);
    float sum = 0.f;
    for (int j = 0; j < L; ++j) sum += expf(long_dot(q, sK + j * d, d) * inv - mx);
    sM[i] = mx;
    sZ[i] = 1.0f / sum;
  }
}

__global__ __launch_bounds__(LONG_THREADS) void attn_long_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.L, d = a.d, tile = L * d;
  float* sQ = smem;
  float* sK = sQ + tile;
  float* sV = sK + tile;
  float* sM = sV + tile;
  float* sZ = sM + L;
  const int64_t prob = blockIdx.x;
  const int64_t seq = prob / a.h;
  const int head = static_cast<int>(prob - seq * a.h);
  const int E = a.h * d;
  const int64_t row0 = seq * L;
  const float inv = 1.0f / sqrtf(static_cast<float>(d));
  long_load_tile(sQ, a.qkv, a.ld_qkv, row0, head * d, L, d);
  long_load_tile(sK, a.qkv, a.ld_qkv, row0, E + head * d, L, d);
  long_load_tile(sV, a.qkv, a.ld_qkv, row0, 2 * E + head * d, L, d);
  __syncthreads();
  long_row_stats(sQ, sK, sM, sZ, L, d, inv);
  __syncthreads();
  const int j = threadIdx.x;
  if (j >= L) return;
  float k[DMAX], o[DMAX];
#pragma unroll
  for (int c = 0; c < DMAX; ++c) {
    k[c] = (c < d) ? sK[j * d + c] : 0.f;
    o[c] = 0.f;
  }
  for (int i = 0; i < L; ++i) {  // O[j,:] = sum_i P[i,j] V[i,:]   (the TRANSPOSED attention matrix, layers.py:249)
    const float p = expf(long_dot(k, sQ + i * d, d) * inv - sM[i]) * sZ[i];
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
      if (c < d) o[c] = fmaf(p, sV[i * d + c], o[c]);
  }
  const bool do_drop = a.key_ptr != nullptr;
  const uint32_t key = do_drop ? *a.key_ptr : 0u;
  float* dst = a.out + (row0 + j) * a.ld_out + head * d;
  const uint64_t e0 = static_cast<uint64_t>(row0 + j) * E + head * d;
#pragma unroll
  for (int c = 0; c < DMAX; ++c) {
    if (c < d) {
      float v = o[c];
      if (do_drop) v *= ebn_drop_mult(key, e0 + c, a.thresh, a.scale);
      dst[c] = v;
    }
  }
}

__global__ __launch_bounds__(LONG_THREADS) void attn_long_bwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.L, d = a.d, tile = L * d;
  float* sQ = smem;
  float* sK = sQ + tile;
  float* sV = sK + tile;
  float* sG = sV + tile;  // dO with the forward dropout mask applied
  float* sM = sG + tile;
  float* sZ = sM + L;
  float* sR = sZ + L;  // rowdot_i = sum_j P_ij dP_ij
  const int64_t prob = blockIdx.x;
  const int64_t seq = prob / a.h;
  const int head = static_cast<int>(prob - seq * a.h);
  const int E = a.h * d;
  const int64_t row0 = seq * L;
  const float inv = 1.0f / sqrtf(static_cast<float>(d));
  long_load_tile(sQ, a.qkv, a.ld_qkv, row0, head * d, L, d);
  long_load_tile(sK, a.qkv, a.ld_qkv, row0, E + head * d, L, d);
  long_load_tile(sV, a.qkv, a.ld_qkv, row0, 2 * E + head * d, L, d);
  {
    const bool do_drop = a.key_ptr != nullptr;
    const uint32_t key = do_drop ? *a.key_ptr : 0u;
    for (int e = threadIdx.x; e < tile; e += LONG_THREADS) {
      const int l = e / d, c = e - l * d;
      float v = a.dout[(row0 + l) * a.ld_dout + head * d + c];
      if (do_drop) v *= ebn_drop_mult(key, static_cast<uint64_t>(row0 + l) * E + head * d + c, a.thresh, a.scale);
      sG[e] = v;
    }
  }
  __syncthreads();
  long_row_stats(sQ, sK, sM, sZ, L, d, inv);
  const int t = threadIdx.x;
  if (t < L) {  // row pass 1: dV[i,:] = sum_j P[i,j] dO[j,:], rowdot_i
    const int i = t;
    float q[DMAX], v[DMAX], acc[DMAX];
#pragma unroll
    for (int c = 0; c < DMAX; ++c) {
      q[c] = (c < d) ? sQ[i * d + c] : 0.f;
      v[c] = (c < d) ? sV[i * d + c] : 0.f;
      acc[c] = 0.f;
    }
    const float mi = sM[i], zi = sZ[i];
    float rowdot = 0.f;
    for (int j = 0; j < L; ++j) {
      const float p = expf(long_dot(q, sK + j * d, d) * inv - mi) * zi;
      rowdot = fmaf(p, long_dot(v, sG + j * d, d), rowdot);
#pragma unroll
      for (int c = 0; c < DMAX; ++c)
        if (c < d) acc[c] = fmaf(p, sG[j * d + c], acc[c]);
    }
    sR[i] = rowdot;
    float* dv = a.out + (row0 + i) * a.ld_out + 2 * E + head * d;
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
      if (c < d) dv[c] = acc[c];
    // row pass 2: dQ[i,:] = inv * sum_j dS[i,j] K[j,:],  dS = P (dP - rowdot)
#pragma unroll
    for (int c = 0; c < DMAX; ++c) acc[c] = 0.f;
    for (int j = 0; j < L; ++j) {
      const float p = expf(long_dot(q, sK + j * d, d) * inv - mi) * zi;
      const float ds = p * (long_dot(v, sG + j * d, d) - rowdot);
#pragma unroll
      for (int c = 0; c < DMAX; ++c)
        if (c < d) acc[c] = fmaf(ds, sK[j * d + c], acc[c]);
    }
    float* dq = a.out + (row0 + i) * a.ld_out + head * d;
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
      if (c < d) dq[c] = acc[c] * inv;
  }
  __syncthreads();
  if (t < L) {  // column pass: dK[j,:] = inv * sum_i dS[i,j] Q[i,:]
    const int j = t;
    float k[DMAX], g[DMAX], acc[DMAX];
#pragma unroll
    for (int c = 0; c < DMAX; ++c) {
      k[c] = (c < d) ? sK[j * d + c] : 0.f;
      g[c] = (c < d) ? sG[j * d + c] : 0.f;
      acc[c] = 0.f;
    }
    for (int i = 0; i < L; ++i) {
      const float p = expf(long_dot(k, sQ + i * d, d) * inv - sM[i]) * sZ[i];
      const float ds = p * (long_dot(g, sV + i * d, d) - sR[i]);
#pragma unroll
      for (int c = 0; c < DMAX; ++c)
        if (c < d) acc[c] = fmaf(ds, sQ[i * d + c], acc[c]);
    }
    float* dk = a.out + (row0 + j) * a.ld_out + E + head * d;
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
      if (c < d) dk[c] = acc[c] * inv;
  }
}

template <class K>
void allow_big_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

int check_attn_args(const void* qkv, const void* out, int64_t n_seq, int32_t L, int32_t h, int32_t d) {
  EBN_REQUIRE(qkv && out, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_seq >= 0 && L > 0 && h > 0 && d > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(L <= LONG_LMAX && d <= DMAX, EBN_ERR_UNSUPPORTED);
  return EBN_OK;
}

}  // namespace

extern "C" int ebn_attn_fwd_f32(const float* qkv, int64_t ld_qkv, float* out, int64_t ld_out, int64_t n_seq,
                                int32_t L, int32_t h, int32_t d, const ebn_step_state* st, int32_t site,
                                float drop_p, ebn_stream_t stream) {
  int rc = check_attn_args(qkv, out, n_seq, L, h, d);
  if (rc != EBN_OK) return rc;
  if (n_seq == 0) return EBN_OK;
  EBN_REQUIRE(ld_qkv >= 3 * h * d && ld_out >= h * d, EBN_ERR_BAD_ARG);
  const EbnDrop dr = ebn_make_drop(st, site, drop_p);
  bool handled = false;
  rc = ebn_attn_mfma_fwd(qkv, ld_qkv, out, ld_out, n_seq, L, h, d, dr, ebn_stream(stream), &handled);
  if (rc != EBN_OK || handled) return rc;
  AttnArgs a{qkv, ld_qkv, nullptr, 0, out, ld_out, n_seq * h, L, h, d, dr.key_ptr, dr.thresh, dr.scale};
  // one-wave kernels for L <= 64 while their LDS image fits 64 KB; anything longer or fatter takes the recompute kernels
  const int LPs = (L & 1) ? L + 2 : L + 1;
  const bool short_ok = L <= 64 && static_cast<size_t>((L <= 32 ? 2 : 1) * (3 * L * d + L * LPs)) * sizeof(float) <= 65536;
  if (!short_ok) {
    const size_t lds = static_cast<size_t>(3 * L * d + 2 * L) * sizeof(float);
    EBN_REQUIRE(lds <= 160 * 1024 && a.n_prob <= 0x7FFFFFFF, EBN_ERR_UNSUPPORTED);
    allow_big_lds(attn_long_fwd_kernel, lds);
    EBN_LAUNCH(attn_long_fwd_kernel, dim3(static_cast<unsigned>(a.n_prob)), dim3(LONG_THREADS), lds, ebn_stream(stream), a);
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  const int LP = (L & 1) ? L + 2 : L + 1;
  const int per_group = 3 * L * d + L * LP;
  EBN_REQUIRE(static_cast<size_t>((L <= 32 ? 2 : 1) * per_group) * sizeof(float) <= 65536, EBN_ERR_UNSUPPORTED);
  if (L <= 32) {
    const int64_t grid = ebn_ceil_div(a.n_prob, 2);
    EBN_LAUNCH(attn_fwd_kernel<32>, dim3(static_cast<unsigned>(grid)), dim3(64),
                       2 * per_group * sizeof(float), ebn_stream(stream), a);
  } else {
    EBN_LAUNCH(attn_fwd_kernel<64>, dim3(static_cast<unsigned>(a.n_prob)), dim3(64),
                       per_group * sizeof(float), ebn_stream(stream), a);
  }
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_attn_bwd_f32(const float* qkv, int64_t ld_qkv, const float* dout, int64_t ld_dout,
                                float* dqkv, int64_t ld_dqkv, int64_t n_seq, int32_t L, int32_t h, int32_t d,
                                const ebn_step_state* st, int32_t site, float drop_p, ebn_stream_t stream) {
  int rc = check_attn_args(qkv, dqkv, n_seq, L, h, d);
  if (rc != EBN_OK) return rc;
  EBN_REQUIRE(dout, EBN_ERR_BAD_ARG);
  if (n_seq == 0) return EBN_OK;
  EBN_REQUIRE(ld_qkv >= 3 * h * d && ld_dqkv >= 3 * h * d && ld_dout >= h * d, EBN_ERR_BAD_ARG);
  const EbnDrop dr = ebn_make_drop(st, site, drop_p);
  bool handled = false;
  rc = ebn_attn_mfma_bwd(qkv, ld_qkv, dout, ld_dout, dqkv, ld_dqkv, n_seq, L, h, d, dr, ebn_stream(stream), &handled, nullptr,
                         nullptr, 0);
  if (rc != EBN_OK || handled) return rc;
  AttnArgs a{qkv, ld_qkv, dout, ld_dout, dqkv, ld_dqkv, n_seq * h, L, h, d, dr.key_ptr, dr.thresh, dr.scale};
  const int LPs = (L & 1) ? L + 2 : L + 1;
  const bool short_ok = L <= 64 && static_cast<size_t>((L <= 32 ? 2 : 1) * (4 * L * d + 2 * L * LPs)) * sizeof(float) <= 65536;
  if (!short_ok) {  // e.g. L = 64, d = 32: the forward image fits the one-wave kernel, the backward one does not
    const size_t lds = static_cast<size_t>(4 * L * d + 3 * L) * sizeof(float);
    EBN_REQUIRE(lds <= 160 * 1024 && a.n_prob <= 0x7FFFFFFF, EBN_ERR_UNSUPPORTED);
    allow_big_lds(attn_long_bwd_kernel, lds);
    EBN_LAUNCH(attn_long_bwd_kernel, dim3(static_cast<unsigned>(a.n_prob)), dim3(LONG_THREADS), lds, ebn_stream(stream), a);
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  const int LP = (L & 1) ? L + 2 : L + 1;
  const int per_group = 4 * L * d + 2 * L * LP;
  EBN_REQUIRE(static_cast<size_t>((L <= 32 ? 2 : 1) * per_group) * sizeof(float) <= 65536, EBN_ERR_UNSUPPORTED);
  if (L <= 32) {
    const int64_t grid = ebn_ceil_div(a.n_prob, 2);
    EBN_LAUNCH(attn_bwd_kernel<32>, dim3(static_cast<unsigned>(grid)), dim3(64),
                       2 * per_group * sizeof(float), ebn_stream(stream), a);
  } else {
    EBN_LAUNCH(attn_bwd_kernel<64>, dim3(static_cast<unsigned>(a.n_prob)), dim3(64),
                       per_group * sizeof(float), ebn_stream(stream), a);
  }
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

// Backward of the attention core when its output fed an AttLayer2 pooling (both encoders, nrms.py:137-156 / 108-111):
// d(Y) = dpre.W^T (in `dout`) + w (x) d(pooled), the second term added on the fly while dO is staged -- the GEMM that
// produces the first term then needs no rank-1 epilogue and the [R, E] gradient is not re-read for it.  MFMA path only:
// returns EBN_ERR_UNSUPPORTED for shapes it does not take (the caller then folds the term into its GEMM instead).
extern "C" int ebn_attn_bwd_pooled_f32(const float* qkv, int64_t ld_qkv, const float* dout, int64_t ld_dout,
                                       const float* pool_w, const float* pool_dout, int64_t ld_pool, float* dqkv,
                                       int64_t ld_dqkv, int64_t n_seq, int32_t L, int32_t h, int32_t d,
                                       const ebn_step_state* st, int32_t site, float drop_p, ebn_stream_t stream) {
  int rc = check_attn_args(qkv, dqkv, n_seq, L, h, d);
  if (rc != EBN_OK) return rc;
  EBN_REQUIRE(dout && pool_w && pool_dout, EBN_ERR_BAD_ARG);
  if (n_seq == 0) return EBN_OK;
  EBN_REQUIRE(ld_qkv >= 3 * h * d && ld_dqkv >= 3 * h * d && ld_dout >= h * d && ld_pool >= h * d, EBN_ERR_BAD_ARG);
  const EbnDrop dr = ebn_make_drop(st, site, drop_p);
  bool handled = false;
  rc = ebn_attn_mfma_bwd(qkv, ld_qkv, dout, ld_dout, dqkv, ld_dqkv, n_seq, L, h, d, dr, ebn_stream(stream), &handled, pool_w,
                         pool_dout, ld_pool);
  if (rc != EBN_OK) return rc;
  return handled ? EBN_OK : EBN_ERR_UNSUPPORTED;
}

// host-side query: does ebn_attn_bwd_pooled_f32 take this shape?  (leading dimensions / alignment aside)
extern "C" int ebn_attn_bwd_pooled_supported(int32_t L, int32_t d) {
  return (L > 0 && L <= 64 && (d == 16 || d == 20 || d == 32)) ? 1 : 0;
}
