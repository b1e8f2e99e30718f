// a3 / a6 fast path: the SelfAttention core (layers.py:231-252) for L <= 64 (one or 2 x 2 MFMA tiles), d in {16,20,32},
// entirely on the matrix cores -- one 64-lane wave per (sequence, head), no workgroup barriers.
//
// Memory side: the wave copies its L x d slices of Q, K, V (and dO) ONCE, as coalesced 16-byte loads, into a
// wave-private LDS region (rows >= L zero-filled) and builds every MFMA operand form from there; results go back
// through the same region and leave as coalesced 16-byte stores.  (Building the operand forms straight from global
// memory touched every 128-byte line from 5-16 separate load instructions and thrashed the 32 KB vector L1.)
//
// Everything is a 32x32 exact-fp32 MFMA tile (v_mfma_f32_32x32x2_f32), zero-padded from L x L / L x d.
// Two facts about that instruction drive the dataflow:
//   * its contraction index is (step, lane-half); ANY mapping of the logical k to (step, half) is valid
//     as long as A and B agree.  "Row-form" operands give half `hi` the k-range [hi*d/2, (hi+1)*d/2)
//     of a row (contiguous 8-byte loads, no selects).
//   * its result layout -- lane (n = lane&31, hi), register r <-> row crow(r,hi) = (r&3)+8(r>>2)+4hi --
//     is exactly a B operand whose k is (r, hi): a result tile feeds the next product from registers.
// Because the reference multiplies V by the TRANSPOSED attention matrix (O = P^T V, layers.py:249) the
// product contracts over the softmax-row index i, so P is needed with i on registers (S = QK^T:
// lane j, regs i); the softmax statistics (max_j, sum_j) are cheapest with j on registers
// (T = KQ^T: lane i, regs j -> in-lane reduction + one cross-half swap).  Both tiles are computed
// (same products, same order) and the per-row statistic c_i = max_i + log2(Z_i) moves between the two
// layouts with ds_bpermute (__shfl).  Forward = 10+10+16 MFMAs, backward = 88 MFMAs per problem:
// ~15 / ~37 us for the 16,000 (title, head) problems of a batch-32 step, vs 216 / 1300 us for the
// LDS/VALU kernel it replaces (profiles/r01_a_*).
#include "ebn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef EBN_ATTN_WAVES
#define EBN_ATTN_WAVES 4  // independent (sequence, head) problems per workgroup, one wave each
#endif
constexpr int ATT_WAVES = EBN_ATTN_WAVES;

struct MfmaAttnArgs {
  const float* qkv;
  int64_t ld_qkv;
  const float* dout;
  int64_t ld_dout;
  float* out;  // fwd: Y ; bwd: dqkv
  int64_t ld_out;
  int64_t n_prob;
  int32_t L, h;
  const uint32_t* key_ptr;
  uint32_t thresh;
  float scale;
  // backward only, optional: the AttLayer2 pooling term of d(Y), added while dO is staged --
  //   dO[r][c] = dout[r][c] + pool_w[row0 + r] * pool_d[seq][head*D + c]      (layers.py:79-81 backward: w (x) d(out))
  const float* pool_w;
  const float* pool_d;
  int64_t ld_pool;
};

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// LDS row stride (floats): a multiple of 4 (16-byte rows) that is not a multiple of 16, so that the 32 rows a
// row-form fetch touches spread over 8 bank groups.
template <int D>
struct Tile {
  static constexpr int STRIDE = (D % 16 == 0) ? D + 4 : D;
  static constexpr int FLOATS = 32 * STRIDE;
  static constexpr int VPR = D / 4;  // float4 per row
  static constexpr int VECS = 32 * VPR;
  static constexpr int ROUNDS = (VECS + 63) / 64;
};

// orders this wave's LDS writes before its later LDS reads (and vice versa); no other wave touches the region
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// global M[L][D] (row stride ld) -> lds[L][STRIDE].  DROP: multiply by the dropout mask of
// element index e0 + r*E + c (the mask the forward pass applied to the matching output element).
// Staging is split into "issue every load" and "write to LDS": loads that sit under a per-element guard, or behind
// the LDS store of the previous piece, are serialised by the compiler (one global round trip after the other -- with
// 9-12 pieces per wave that was most of a wave's lifetime).  The loads are unconditional with a clamped row.
template <int D>
struct Staged {
  float4 v[Tile<D>::ROUNDS];
};

template <int D>
__device__ __forceinline__ void stage_load(Staged<D>& st, const float* __restrict__ base, int64_t ld, int L, int lane) {
  using T = Tile<D>;
#pragma unroll
  for (int t = 0; t < T::ROUNDS; ++t) {
    int v = lane + 64 * t;
    if (T::VECS % 64 != 0 && v >= T::VECS) v = 0;
    const int r = v / T::VPR, c4 = v - r * T::VPR;
    const int rc = r < L ? r : 0;
    st.v[t] = *reinterpret_cast<const float4*>(base + static_cast<int64_t>(rc) * ld + c4 * 4);
  }
}

// the pooling term's operands for the float4 pieces a lane stages (loaded up front with the tile itself)
template <int D>
struct PoolTerm {
  float w[Tile<D>::ROUNDS];
  float4 p[Tile<D>::ROUNDS];
};

template <int D>
__device__ __forceinline__ void pool_load(PoolTerm<D>& pt, const float* __restrict__ w_rows, const float* __restrict__ d_cols,
                                          int L, int lane) {
  using T = Tile<D>;
#pragma unroll
  for (int t = 0; t < T::ROUNDS; ++t) {
    int v = lane + 64 * t;
    if (T::VECS % 64 != 0 && v >= T::VECS) v = 0;
    const int r = v / T::VPR, c4 = v - r * T::VPR;
    pt.w[t] = w_rows[r < L ? r : 0];
    pt.p[t] = *reinterpret_cast<const float4*>(d_cols + c4 * 4);
  }
}

template <int D>
__device__ __forceinline__ void pool_add(Staged<D>& st, const PoolTerm<D>& pt) {
#pragma unroll
  for (int t = 0; t < Tile<D>::ROUNDS; ++t) {
    st.v[t].x = fmaf(pt.w[t], pt.p[t].x, st.v[t].x);
    st.v[t].y = fmaf(pt.w[t], pt.p[t].y, st.v[t].y);
    st.v[t].z = fmaf(pt.w[t], pt.p[t].z, st.v[t].z);
    st.v[t].w = fmaf(pt.w[t], pt.p[t].w, st.v[t].w);
  }
}

template <int D, bool DROP>
__device__ __forceinline__ void stage_store(float* __restrict__ lds, const Staged<D>& st, int L, int lane, uint32_t key,
                                            uint64_t e0, int E, uint32_t thresh, float scale) {
  using T = Tile<D>;
#pragma unroll
  for (int t = 0; t < T::ROUNDS; ++t) {
    const int v = lane + 64 * t;
    if (T::VECS % 64 != 0 && v >= T::VECS) break;
    const int r = v / T::VPR, c4 = v - r * T::VPR;
    if (r >= L) continue;
    float4 y = st.v[t];
    if (DROP) {
      const uint64_t e = e0 + static_cast<uint64_t>(r) * E + c4 * 4;  // multiple of 4: two aligned pairs
      const uint32_t h0 = ebn_dropout_pair_hash(key, e >> 1), h1 = ebn_dropout_pair_hash(key, (e >> 1) + 1);
      y.x *= ((h0 & 0xFFFFu) >= thresh) ? scale : 0.f;
      y.y *= ((h0 >> 16) >= thresh) ? scale : 0.f;
      y.z *= ((h1 & 0xFFFFu) >= thresh) ? scale : 0.f;
      y.w *= ((h1 >> 16) >= thresh) ? scale : 0.f;
    }
    *reinterpret_cast<float4*>(lds + r * T::STRIDE + c4 * 4) = y;
  }
}

// lds[32][STRIDE] -> global M[L][D], optionally through the dropout mask
template <int D, bool DROP>
__device__ __forceinline__ void stage_out(const float* __restrict__ lds, float* __restrict__ base, int64_t ld, int L,
                                          int lane, uint32_t key, uint64_t e0, int E, uint32_t thresh, float scale) {
  using T = Tile<D>;
#pragma unroll
  for (int t = 0; t < T::ROUNDS; ++t) {
    const int v = lane + 64 * t;
    if (T::VECS % 64 != 0 && v >= T::VECS) break;
    const int r = v / T::VPR, c4 = v - r * T::VPR;
    if (r >= L) continue;
    float4 y = *reinterpret_cast<const float4*>(lds + r * T::STRIDE + c4 * 4);
    if (DROP) {
      const uint64_t e = e0 + static_cast<uint64_t>(r) * E + c4 * 4;
      const uint32_t h0 = ebn_dropout_pair_hash(key, e >> 1), h1 = ebn_dropout_pair_hash(key, (e >> 1) + 1);
      y.x *= ((h0 & 0xFFFFu) >= thresh) ? scale : 0.f;
      y.y *= ((h0 >> 16) >= thresh) ? scale : 0.f;
      y.z *= ((h1 & 0xFFFFu) >= thresh) ? scale : 0.f;
      y.w *= ((h1 >> 16) >= thresh) ? scale : 0.f;
    }
    *reinterpret_cast<float4*>(base + static_cast<int64_t>(r) * ld + c4 * 4) = y;
  }
}

// The LDS regions hold L rows (not 32): rows >= L are predicated to zero on the way into the registers.
// lane (row = lane&31, hi): x[s] = M[row][hi*KH + s].  8-byte LDS reads.
template <int D>
__device__ __forceinline__ void lds_row_form(float (&x)[D / 2], const float* __restrict__ lds, int L, int row, int hi) {
  constexpr int KH = D / 2;
  const bool ok = row < L;
  const float2* p = reinterpret_cast<const float2*>(lds + (ok ? row : 0) * Tile<D>::STRIDE + hi * KH);
#pragma unroll
  for (int s = 0; s < KH / 2; ++s) {
    const float2 v = p[s];
    x[2 * s] = ok ? v.x : 0.f;
    x[2 * s + 1] = ok ? v.y : 0.f;
  }
}

// lane (c = lane&31, hi): y[s] = M[crow(s,hi)][c]; zero for c >= D or a row >= L.
template <int D>
__device__ __forceinline__ void lds_col_form(float (&y)[16], const float* __restrict__ lds, int L, int c, int hi) {
  const int cc = (c < D) ? c : 0;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int r = crow(s, hi);
    const bool ok = (c < D) && (r < L);
    const float v = lds[(r < L ? r : 0) * Tile<D>::STRIDE + cc];
    y[s] = ok ? v : 0.f;
  }
}

template <int KH>
__device__ __forceinline__ f32x16 mm_rows(const float (&a)[KH], const float (&b)[KH]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  return acc;
}

// out^T tile = sum_s A_col[s] (x) Z[s]  with Z a result-layout tile used as the B operand
__device__ __forceinline__ f32x16 mm_col_tile(const float (&a)[16], const f32x16& z) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], z[s], acc, 0, 0, 0);
  return acc;
}

// result tile whose lane owns row `row` and whose registers own columns crow(r,hi) -> lds[row][c], float4 groups
template <int D>
__device__ __forceinline__ void tile_rows_to_lds(float* __restrict__ lds, const f32x16& acc, int L, int row, int hi,
                                                 float mul) {
  if (row >= L) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c0 = 8 * g + 4 * hi;
    if (c0 < D) {
      float4 v = make_float4(acc[4 * g] * mul, acc[4 * g + 1] * mul, acc[4 * g + 2] * mul, acc[4 * g + 3] * mul);
      *reinterpret_cast<float4*>(lds + row * Tile<D>::STRIDE + c0) = v;
    }
  }
}

// Softmax in the exp2 domain: `inv2` = log2(e)/sqrt(d), so exp(x/sqrt(d) - max) = exp2(x*inv2 - max2) and one
// v_exp_f32 per element replaces the ~20-instruction expf expansion (the softmax of 2 x 32 x 32 scores was the VALU
// bottleneck of these kernels).  Arguments are <= 0, flush-to-zero below 2^-126 is the correct limit.
//
// Statistics of row i, held by lane i (both halves end with the same values): t[r] = T[j=crow(r,hi)][i].
// Returns c = max2 + log2(Z), so that P_ij = exp2(t_ij*inv2 - c) in either layout; turns t into P_ij (zero for j >= L).
__device__ __forceinline__ float softmax_in_lane(f32x16& t, int L, int hi, float inv2) {
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    t[r] *= inv2;
    if (crow(r, hi) < L) m = fmaxf(m, t[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float z = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) z += (crow(r, hi) < L) ? __builtin_amdgcn_exp2f(t[r] - m) : 0.f;
  z += __shfl_xor(z, 32, 64);
  const float c = m + __builtin_amdgcn_logf(z);  // v_log_f32 = log2; z in [1, 32]
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = (crow(r, hi) < L) ? __builtin_amdgcn_exp2f(t[r] - c) : 0.f;
  return c;
}

// s[r] = S[i=crow(r,hi)][j] -> P_ij using c_i fetched from lane i
__device__ __forceinline__ void softmax_from_stats(f32x16& s, int L, int hi, float inv2, float c) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = crow(r, hi);
    const float ci = __shfl(c, i, 64);
    s[r] = (i < L) ? __builtin_amdgcn_exp2f(s[r] * inv2 - ci) : 0.f;
  }
}

// forward only needs the statistics from the T layout
__device__ __forceinline__ float softmax_stats_only(const f32x16& t, int L, int hi, float inv2) {
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (crow(r, hi) < L) m = fmaxf(m, t[r] * inv2);
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float z = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) z += (crow(r, hi) < L) ? __builtin_amdgcn_exp2f(t[r] * inv2 - m) : 0.f;
  z += __shfl_xor(z, 32, 64);
  return m + __builtin_amdgcn_logf(z);
}

// LC: sequence length known at compile time (0 = use a.L).  title_size = 30 and history_size = 20 are what every
// BASELINE config runs: with L a constant most of the row / column validity masks of a 32-wide tile fold away (only
// registers 14, 15 of the upper lane half can be rows >= 30), which is a fifth of this VALU-issue-bound kernel.
template <int D, int LC>
__global__ __launch_bounds__(64 * ATT_WAVES) void attn_mfma_fwd_kernel(MfmaAttnArgs a) {
  using T = Tile<D>;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // ATT_WAVES x 3 regions x L x STRIDE
  const int lane = threadIdx.x & 63;
  const int64_t prob = static_cast<int64_t>(blockIdx.x) * ATT_WAVES + (threadIdx.x >> 6);
  if (prob >= a.n_prob) return;  // wave-uniform; no workgroup barriers in this kernel
  const int L = LC ? LC : a.L, E = a.h * D;
  const int region = L * T::STRIDE;
  float* sq = smem + (threadIdx.x >> 6) * 3 * region;
  float* sk = sq + region;
  float* sv = sk + region;
  const int row = lane & 31, hi = lane >> 5;
  const int64_t seq = prob / a.h;
  const int head = static_cast<int>(prob - seq * a.h);
  const int64_t row0 = seq * L;
  const float* qb = a.qkv + row0 * a.ld_qkv + head * D;
  const float inv2 = 1.44269504088896341f / sqrtf(static_cast<float>(D));

  {
    Staged<D> tq, tk, tv;
    stage_load<D>(tq, qb, a.ld_qkv, L, lane);
    stage_load<D>(tk, qb + E, a.ld_qkv, L, lane);
    stage_load<D>(tv, qb + 2 * E, a.ld_qkv, L, lane);
    stage_store<D, false>(sq, tq, L, lane, 0u, 0u, 0, 0u, 0.f);
    stage_store<D, false>(sk, tk, L, lane, 0u, 0u, 0, 0u, 0.f);
    stage_store<D, false>(sv, tv, L, lane, 0u, 0u, 0, 0u, 0.f);
  }
  wave_lds_sync();

  float qr[D / 2], kr[D / 2];
  lds_row_form<D>(qr, sq, L, row, hi);
  lds_row_form<D>(kr, sk, L, row, hi);
  float vc[16];
  lds_col_form<D>(vc, sv, L, row, hi);

  f32x16 Tt = mm_rows<D / 2>(kr, qr);  // T[j][i]: lane i, regs j
  f32x16 S = mm_rows<D / 2>(qr, kr);   // S[i][j]: lane j, regs i
  const float c = softmax_stats_only(Tt, L, hi, inv2);
  softmax_from_stats(S, L, hi, inv2, c);  // P[i][j]: lane j, regs i
  const f32x16 O = mm_col_tile(vc, S);    // O^T[c][j]: lane j, regs c

  wave_lds_sync();  // every operand read of sq is done: reuse it for the output rows
  tile_rows_to_lds<D>(sq, O, L, row, hi, 1.0f);
  wave_lds_sync();
  float* ob = a.out + row0 * a.ld_out + head * D;
  const uint64_t e0 = static_cast<uint64_t>(row0) * E + head * D;
  if (a.key_ptr != nullptr) stage_out<D, true>(sq, ob, a.ld_out, L, lane, *a.key_ptr, e0, E, a.thresh, a.scale);
  else stage_out<D, false>(sq, ob, a.ld_out, L, lane, 0u, 0u, 0, 0u, 0.f);
}

// Backward.  The "lane i" tiles (P, dP -> dV, dQ) are finished before the "lane j" tiles (-> dK) are started, so
// that at most two 32x32 tiles are live at a time: the kernel fits 128 registers = 4 waves per SIMD.
template <int D, int LC>
__global__ __launch_bounds__(64 * ATT_WAVES) void attn_mfma_bwd_kernel(MfmaAttnArgs a) {
  using T = Tile<D>;
  constexpr int KH = D / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // ATT_WAVES x 4 regions x L x STRIDE
  const int lane = threadIdx.x & 63;
  const int64_t prob = static_cast<int64_t>(blockIdx.x) * ATT_WAVES + (threadIdx.x >> 6);
  if (prob >= a.n_prob) return;
  const int L = LC ? LC : a.L, E = a.h * D;
  const int region = L * T::STRIDE;
  float* sq = smem + (threadIdx.x >> 6) * 4 * region;
  float* sk = sq + region;
  float* sv = sk + region;  // V, then the staging buffer of the three result tiles
  float* sg = sv + region;  // dO with the forward dropout mask applied
  const int row = lane & 31, hi = lane >> 5;
  const int64_t seq = prob / a.h;
  const int head = static_cast<int>(prob - seq * a.h);
  const int64_t row0 = seq * L;
  const float* qb = a.qkv + row0 * a.ld_qkv + head * D;
  const float* gb = a.dout + row0 * a.ld_dout + head * D;
  const float inv = 1.0f / sqrtf(static_cast<float>(D));
  const float inv2 = inv * 1.44269504088896341f;

  {
    Staged<D> tq, tk, tv, tg;
    stage_load<D>(tq, qb, a.ld_qkv, L, lane);
    stage_load<D>(tk, qb + E, a.ld_qkv, L, lane);
    stage_load<D>(tv, qb + 2 * E, a.ld_qkv, L, lane);
    stage_load<D>(tg, gb, a.ld_dout, L, lane);
    if (a.pool_w != nullptr) {  // wave-uniform
      PoolTerm<D> pt;
      pool_load<D>(pt, a.pool_w + row0, a.pool_d + seq * a.ld_pool + head * D, L, lane);
      pool_add<D>(tg, pt);
    }
    stage_store<D, false>(sq, tq, L, lane, 0u, 0u, 0, 0u, 0.f);
    stage_store<D, false>(sk, tk, L, lane, 0u, 0u, 0, 0u, 0.f);
    stage_store<D, false>(sv, tv, L, lane, 0u, 0u, 0, 0u, 0.f);
    if (a.key_ptr != nullptr)
      stage_store<D, true>(sg, tg, L, lane, *a.key_ptr, static_cast<uint64_t>(row0) * E + head * D, E, a.thresh, a.scale);
    else
      stage_store<D, false>(sg, tg, L, lane, 0u, 0u, 0, 0u, 0.f);
  }
  wave_lds_sync();

  float* ob = a.out + row0 * a.ld_out + head * D;
  float c, rowdot;
  float vr[KH];  // row form of V (lane = row): identical in both passes
  // ---- lane-i layout: P[i][j] and dP[i][j] with i on lanes -> dV, dQ
  {
    f32x16 P, dP;
    {
      float qr[KH], kr[KH];
      lds_row_form<D>(qr, sq, L, row, hi);
      lds_row_form<D>(kr, sk, L, row, hi);
      P = mm_rows<KH>(kr, qr);  // T[j][i]: lane i, regs j
    }
    c = softmax_in_lane(P, L, hi, inv2);
    {
      float gr[KH];
      lds_row_form<D>(vr, sv, L, row, hi);  // kept for the lane-j pass: sv is about to be reused for the results
      lds_row_form<D>(gr, sg, L, row, hi);
      dP = mm_rows<KH>(gr, vr);  // dP[i][j] = V[i].dO[j]: lane i, regs j
    }
    rowdot = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) rowdot = fmaf(P[r], dP[r], rowdot);
    rowdot += __shfl_xor(rowdot, 32, 64);  // sum_j P[i][j] dP[i][j] for i = lane&31

    float col[16];
    // dV^T[c][i] = sum_j dO[j][c] P[i][j]
    lds_col_form<D>(col, sg, L, row, hi);
    {
      const f32x16 dV = mm_col_tile(col, P);
      wave_lds_sync();  // the row-form reads of sv are done
      tile_rows_to_lds<D>(sv, dV, L, row, hi, 1.0f);
      wave_lds_sync();
      stage_out<D, false>(sv, ob + 2 * E, a.ld_out, L, lane, 0u, 0u, 0, 0u, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) P[r] = P[r] * (dP[r] - rowdot);  // dS[i][j]: lane i, regs j
    // dQ^T[c][i] = inv * sum_j K[j][c] dS[i][j]
    lds_col_form<D>(col, sk, L, row, hi);
    {
      const f32x16 dQ = mm_col_tile(col, P);
      wave_lds_sync();
      tile_rows_to_lds<D>(sv, dQ, L, row, hi, inv);
      wave_lds_sync();
      stage_out<D, false>(sv, ob, a.ld_out, L, lane, 0u, 0u, 0, 0u, 0.f);
    }
  }
  // ---- lane-j layout: the same two tiles with j on lanes -> dK
  {
    f32x16 P, dP;
    {
      float qr[KH], kr[KH];
      lds_row_form<D>(qr, sq, L, row, hi);
      lds_row_form<D>(kr, sk, L, row, hi);
      P = mm_rows<KH>(qr, kr);  // S[i][j]: lane j, regs i
    }
    softmax_from_stats(P, L, hi, inv2, c);
    {
      float gr[KH];
      lds_row_form<D>(gr, sg, L, row, hi);
      dP = mm_rows<KH>(vr, gr);  // lane j, regs i
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float rd = __shfl(rowdot, crow(r, hi), 64);
      P[r] = P[r] * (dP[r] - rd);  // dS[i][j]: lane j, regs i
    }
    // dK^T[c][j] = inv * sum_i Q[i][c] dS[i][j]
    float col[16];
    lds_col_form<D>(col, sq, L, row, hi);
    {
      const f32x16 dK = mm_col_tile(col, P);
      wave_lds_sync();
      tile_rows_to_lds<D>(sv, dK, L, row, hi, inv);
      wave_lds_sync();
      stage_out<D, false>(sv, ob + E, a.ld_out, L, lane, 0u, 0u, 0, 0u, 0.f);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// 32 < L <= 64 (history_size = 50 of BASELINE.json configs[3]: the news-level SelfAttention of the user encoder, nrms.py:
// 108-110).  Same dataflow with the L x L attention matrix as a 2 x 2 grid of 32 x 32 tiles: block ib of the softmax-row
// index i, block jb of the column index j.  Row statistics need both column blocks of a row (two tiles live), the
// P^T.V / dV / dQ / dK contractions accumulate over the two blocks of their contraction index.  Still one wave per
// (sequence, head) and no workgroup barrier; operands of both blocks are pulled into registers up front so the LDS
// regions can be reused for the results.
constexpr int ATT2_WAVES = 2;
constexpr int NB2 = 2;

__device__ __forceinline__ void mm_col_tile_acc(f32x16& acc, const float (&a)[16], const f32x16& z) {
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], z[s], acc, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero_tile() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// t[jb][r] = T[j = 32 jb + crow(r,hi)][i = this lane's row]; returns c_i = max2 + log2 Z over j < L (both blocks);
// with TO_P the tiles are turned into P_ij (zero for j >= L).
template <bool TO_P>
__device__ __forceinline__ float softmax2_in_lane(f32x16 (&t)[NB2], int L, int hi, float inv2) {
  float m = -INFINITY;
#pragma unroll
  for (int jb = 0; jb < NB2; ++jb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      t[jb][r] *= inv2;
      if (32 * jb + crow(r, hi) < L) m = fmaxf(m, t[jb][r]);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float z = 0.f;
#pragma unroll
  for (int jb = 0; jb < NB2; ++jb)
#pragma unroll
    for (int r = 0; r < 16; ++r) z += (32 * jb + crow(r, hi) < L) ? __builtin_amdgcn_exp2f(t[jb][r] - m) : 0.f;
  z += __shfl_xor(z, 32, 64);
  const float c = m + __builtin_amdgcn_logf(z);
  if (TO_P) {
#pragma unroll
    for (int jb = 0; jb < NB2; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) t[jb][r] = (32 * jb + crow(r, hi) < L) ? __builtin_amdgcn_exp2f(t[jb][r] - c) : 0.f;
  }
  return c;
}

// s[r] = S[i = 32 ib + crow(r,hi)][j = this lane's column] -> P_ij with c_i fetched from lane (i & 31) of block ib's stats
__device__ __forceinline__ void softmax2_from_stats(f32x16& s, int L, int ib, int hi, float inv2, float c_ib) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int il = crow(r, hi);
    const float ci = __shfl(c_ib, il, 64);
    s[r] = (32 * ib + il < L) ? __builtin_amdgcn_exp2f(s[r] * inv2 - ci) : 0.f;
  }
}

template <int D, int LC>
__global__ __launch_bounds__(64 * ATT2_WAVES) void attn_mfma2_fwd_kernel(MfmaAttnArgs a) {
  using T = Tile<D>;
  constexpr int KH = D / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // ATT2_WAVES x 3 regions x L x STRIDE
  const int lane = threadIdx.x & 63;
  const int64_t prob = static_cast<int64_t>(blockIdx.x) * ATT2_WAVES + (threadIdx.x >> 6);
  if (prob >= a.n_prob) return;
  const int L = LC ? LC : a.L, E = a.h * D;
  const int region = L * T::STRIDE;
  float* sq = smem + (threadIdx.x >> 6) * 3 * region;
  float* sk = sq + region;
  float* sv = sk + region;
  const int row = lane & 31, hi = lane >> 5;
  const int64_t seq = prob / a.h;
  const int head = static_cast<int>(prob - seq * a.h);
  const int64_t row0 = seq * L;
  const float* qb = a.qkv + row0 * a.ld_qkv + head * D;
  const float inv2 = 1.44269504088896341f / sqrtf(static_cast<float>(D));
  {
    Staged<D> tq[NB2], tk[NB2], tv[NB2];
#pragma unroll
    for (int b = 0; b < NB2; ++b) {
      stage_load<D>(tq[b], qb + 32 * b * a.ld_qkv, a.ld_qkv, L - 32 * b, lane);
      stage_load<D>(tk[b], qb + E + 32 * b * a.ld_qkv, a.ld_qkv, L - 32 * b, lane);
      stage_load<D>(tv[b], qb + 2 * E + 32 * b * a.ld_qkv, a.ld_qkv, L - 32 * b, lane);
    }
#pragma unroll
    for (int b = 0; b < NB2; ++b) {
      stage_store<D, false>(sq + 32 * b * T::STRIDE, tq[b], L - 32 * b, lane, 0u, 0u, 0, 0u, 0.f);
      stage_store<D, false>(sk + 32 * b * T::STRIDE, tk[b], L - 32 * b, lane, 0u, 0u, 0, 0u, 0.f);
      stage_store<D, false>(sv + 32 * b * T::STRIDE, tv[b], L - 32 * b, lane, 0u, 0u, 0, 0u, 0.f);
    }
  }
  wave_lds_sync();
  float qr[NB2][KH], kr[NB2][KH], vc[NB2][16];
#pragma unroll
  for (int b = 0; b < NB2; ++b) {
    lds_row_form<D>(qr[b], sq + 32 * b * T::STRIDE, L - 32 * b, row, hi);
    lds_row_form<D>(kr[b], sk + 32 * b * T::STRIDE, L - 32 * b, row, hi);
    lds_col_form<D>(vc[b], sv + 32 * b * T::STRIDE, L - 32 * b, row, hi);
  }
  float c[NB2];
#pragma unroll
  for (int ib = 0; ib < NB2; ++ib) {
    f32x16 t[NB2];
#pragma unroll
    for (int jb = 0; jb < NB2; ++jb) t[jb] = mm_rows<KH>(kr[jb], qr[ib]);  // T[j][i]: lane i, regs j
    c[ib] = softmax2_in_lane<false>(t, L, hi, inv2);
  }
  wave_lds_sync();  // every operand is in registers: sq becomes the output staging area
#pragma unroll
  for (int jb = 0; jb < NB2; ++jb) {
    f32x16 O = zero_tile();
#pragma unroll
    for (int ib = 0; ib < NB2; ++ib) {
      f32x16 S = mm_rows<KH>(qr[ib], kr[jb]);  // S[i][j]: lane j, regs i
      softmax2_from_stats(S, L, ib, hi, inv2, c[ib]);
      mm_col_tile_acc(O, vc[ib], S);  // O^T[c][j] += sum_{i in block ib} V[i][c] P[i][j]
    }
    tile_rows_to_lds<D>(sq + 32 * jb * T::STRIDE, O, L - 32 * jb, row, hi, 1.0f);
  }
  wave_lds_sync();
  float* ob = a.out + row0 * a.ld_out + head * D;
  const uint64_t e0 = static_cast<uint64_t>(row0) * E + head * D;
#pragma unroll
  for (int b = 0; b < NB2; ++b) {
    if (a.key_ptr != nullptr)
      stage_out<D, true>(sq + 32 * b * T::STRIDE, ob + 32 * b * a.ld_out, a.ld_out, L - 32 * b, lane, *a.key_ptr,
                         e0 + static_cast<uint64_t>(32 * b) * E, E, a.thresh, a.scale);
    else
      stage_out<D, false>(sq + 32 * b * T::STRIDE, ob + 32 * b * a.ld_out, a.ld_out, L - 32 * b, lane, 0u, 0u, 0, 0u, 0.f);
  }
}

template <int D, int LC>
__global__ __launch_bounds__(64 * ATT2_WAVES) void attn_mfma2_bwd_kernel(MfmaAttnArgs a) {
  using T = Tile<D>;
  constexpr int KH = D / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // ATT2_WAVES x 4 regions x L x STRIDE
  const int lane = threadIdx.x & 63;
  const int64_t prob = static_cast<int64_t>(blockIdx.x) * ATT2_WAVES + (threadIdx.x >> 6);
  if (prob >= a.n_prob) return;
  const int L = LC ? LC : a.L, E = a.h * D;
  const int region = L * T::STRIDE;
  float* sq = smem + (threadIdx.x >> 6) * 4 * region;
  float* sk = sq + region;
  float* sv = sk + region;  // V, then the staging buffer of the result tiles
  float* sg = sv + region;  // dO with the forward dropout mask applied
  const int row = lane & 31, hi = lane >> 5;
  const int64_t seq = prob / a.h;
  const int head = static_cast<int>(prob - seq * a.h);
  const int64_t row0 = seq * L;
  const float* qb = a.qkv + row0 * a.ld_qkv + head * D;
  const float* gb = a.dout + row0 * a.ld_dout + head * D;
  const float inv = 1.0f / sqrtf(static_cast<float>(D));
  const float inv2 = inv * 1.44269504088896341f;
#pragma unroll
  for (int b = 0; b < NB2; ++b) {  // (one block at a time: four staged matrices = 4 x ROUNDS float4 in flight)
    Staged<D> tq, tk, tv, tg;
    const int Lb = L - 32 * b;
    stage_load<D>(tq, qb + 32 * b * a.ld_qkv, a.ld_qkv, Lb, lane);
    stage_load<D>(tk, qb + E + 32 * b * a.ld_qkv, a.ld_qkv, Lb, lane);
    stage_load<D>(tv, qb + 2 * E + 32 * b * a.ld_qkv, a.ld_qkv, Lb, lane);
    stage_load<D>(tg, gb + 32 * b * a.ld_dout, a.ld_dout, Lb, lane);
    if (a.pool_w != nullptr) {
      PoolTerm<D> pt;
      pool_load<D>(pt, a.pool_w + row0 + 32 * b, a.pool_d + seq * a.ld_pool + head * D, Lb, lane);
      pool_add<D>(tg, pt);
    }
    stage_store<D, false>(sq + 32 * b * T::STRIDE, tq, Lb, lane, 0u, 0u, 0, 0u, 0.f);
    stage_store<D, false>(sk + 32 * b * T::STRIDE, tk, Lb, lane, 0u, 0u, 0, 0u, 0.f);
    stage_store<D, false>(sv + 32 * b * T::STRIDE, tv, Lb, lane, 0u, 0u, 0, 0u, 0.f);
    if (a.key_ptr != nullptr)
      stage_store<D, true>(sg + 32 * b * T::STRIDE, tg, Lb, lane, *a.key_ptr,
                           static_cast<uint64_t>(row0 + 32 * b) * E + head * D, E, a.thresh, a.scale);
    else
      stage_store<D, false>(sg + 32 * b * T::STRIDE, tg, Lb, lane, 0u, 0u, 0, 0u, 0.f);
  }
  wave_lds_sync();

  float* ob = a.out + row0 * a.ld_out + head * D;
  float c[NB2], rowdot[NB2];
  float vr[NB2][KH];  // row forms of V: read before sv turns into the staging buffer
#pragma unroll
  for (int b = 0; b < NB2; ++b) lds_row_form<D>(vr[b], sv + 32 * b * T::STRIDE, L - 32 * b, row, hi);
  wave_lds_sync();
  // ---- lane-i layout (i = 32 ib + lane row on lanes, j on registers) -> dV, dQ of row block ib
#pragma unroll
  for (int ib = 0; ib < NB2; ++ib) {
    f32x16 P[NB2], dP[NB2];
    {
      float qr[KH];
      lds_row_form<D>(qr, sq + 32 * ib * T::STRIDE, L - 32 * ib, row, hi);
#pragma unroll
      for (int jb = 0; jb < NB2; ++jb) {
        float kr[KH];
        lds_row_form<D>(kr, sk + 32 * jb * T::STRIDE, L - 32 * jb, row, hi);
        P[jb] = mm_rows<KH>(kr, qr);  // T[j][i]
      }
    }
    c[ib] = softmax2_in_lane<true>(P, L, hi, inv2);
    float rd = 0.f;
#pragma unroll
    for (int jb = 0; jb < NB2; ++jb) {
      float gr[KH];
      lds_row_form<D>(gr, sg + 32 * jb * T::STRIDE, L - 32 * jb, row, hi);
      dP[jb] = mm_rows<KH>(gr, vr[ib]);  // dP[i][j] = V[i].dO[j]
#pragma unroll
      for (int r = 0; r < 16; ++r) rd = fmaf(P[jb][r], dP[jb][r], rd);
    }
    rd += __shfl_xor(rd, 32, 64);
    rowdot[ib] = rd;
    float col[16];
    {  // dV^T[c][i] = sum_j dO[j][c] P[i][j]
      f32x16 dV = zero_tile();
#pragma unroll
      for (int jb = 0; jb < NB2; ++jb) {
        lds_col_form<D>(col, sg + 32 * jb * T::STRIDE, L - 32 * jb, row, hi);
        mm_col_tile_acc(dV, col, P[jb]);
      }
      wave_lds_sync();
      tile_rows_to_lds<D>(sv + 32 * ib * T::STRIDE, dV, L - 32 * ib, row, hi, 1.0f);
      wave_lds_sync();
      stage_out<D, false>(sv + 32 * ib * T::STRIDE, ob + 2 * E + 32 * ib * a.ld_out, a.ld_out, L - 32 * ib, lane, 0u, 0u, 0, 0u, 0.f);
    }
#pragma unroll
    for (int jb = 0; jb < NB2; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) P[jb][r] = P[jb][r] * (dP[jb][r] - rd);  // dS[i][j]
    {  // dQ^T[c][i] = inv * sum_j K[j][c] dS[i][j]
      f32x16 dQ = zero_tile();
#pragma unroll
      for (int jb = 0; jb < NB2; ++jb) {
        lds_col_form<D>(col, sk + 32 * jb * T::STRIDE, L - 32 * jb, row, hi);
        mm_col_tile_acc(dQ, col, P[jb]);
      }
      wave_lds_sync();
      tile_rows_to_lds<D>(sv + 32 * ib * T::STRIDE, dQ, L - 32 * ib, row, hi, inv);
      wave_lds_sync();
      stage_out<D, false>(sv + 32 * ib * T::STRIDE, ob + 32 * ib * a.ld_out, a.ld_out, L - 32 * ib, lane, 0u, 0u, 0, 0u, 0.f);
    }
  }
  // ---- lane-j layout (j = 32 jb + lane row on lanes, i on registers) -> dK of row block jb
#pragma unroll
  for (int jb = 0; jb < NB2; ++jb) {
    float kr[KH], gr[KH];
    lds_row_form<D>(kr, sk + 32 * jb * T::STRIDE, L - 32 * jb, row, hi);
    lds_row_form<D>(gr, sg + 32 * jb * T::STRIDE, L - 32 * jb, row, hi);
    f32x16 dK = zero_tile();
#pragma unroll
    for (int ib = 0; ib < NB2; ++ib) {
      f32x16 P, dP;
      {
        float qr[KH];
        lds_row_form<D>(qr, sq + 32 * ib * T::STRIDE, L - 32 * ib, row, hi);
        P = mm_rows<KH>(qr, kr);  // S[i][j]: lane j, regs i
      }
      softmax2_from_stats(P, L, ib, hi, inv2, c[ib]);
      dP = mm_rows<KH>(vr[ib], gr);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float rdi = __shfl(rowdot[ib], crow(r, hi), 64);
        P[r] = P[r] * (dP[r] - rdi);  // dS[i][j]
      }
      float col[16];
      lds_col_form<D>(col, sq + 32 * ib * T::STRIDE, L - 32 * ib, row, hi);
      mm_col_tile_acc(dK, col, P);  // dK^T[c][j] += sum_{i in block ib} Q[i][c] dS[i][j]
    }
    wave_lds_sync();
    tile_rows_to_lds<D>(sv + 32 * jb * T::STRIDE, dK, L - 32 * jb, row, hi, inv);
    wave_lds_sync();
    stage_out<D, false>(sv + 32 * jb * T::STRIDE, ob + E + 32 * jb * a.ld_out, a.ld_out, L - 32 * jb, lane, 0u, 0u, 0, 0u, 0.f);
  }
}

}  // namespace

// Returns 1 when the MFMA path handles (L, d, leading dims, alignment); the caller falls back otherwise.
static bool mfma_path_ok(int32_t L, int32_t d, int64_t lda, int64_t ldb, int64_t ldc, const void* p0,
                         const void* p1, const void* p2) {
  if (L > 64 || !(d == 16 || d == 20 || d == 32)) return false;  // L <= 32: one 32x32 tile; 32 < L <= 64: 2 x 2 tiles
  if ((lda % 4) || (ldb % 4) || (ldc % 4)) return false;
  return ebn_aligned16(p0) && ebn_aligned16(p1) && ebn_aligned16(p2);
}

// more than 64 KB of dynamic LDS per workgroup (d = 32, L = 32 backward: 72 KB) has to be granted per kernel
template <class K>
static void allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

template <int D, int LC>
static void launch_mfma2_lc(bool bwd, const MfmaAttnArgs& a, hipStream_t s) {
  const dim3 grid(static_cast<unsigned>(ebn_ceil_div(a.n_prob, ATT2_WAVES))), block(64 * ATT2_WAVES);
  const size_t lds = static_cast<size_t>(ATT2_WAVES) * (bwd ? 4 : 3) * a.L * Tile<D>::STRIDE * sizeof(float);
  if (bwd) {
    allow_lds(attn_mfma2_bwd_kernel<D, LC>, lds);
    hipLaunchKernelGGL((attn_mfma2_bwd_kernel<D, LC>), grid, block, lds, s, a);
  } else {
    allow_lds(attn_mfma2_fwd_kernel<D, LC>, lds);
    hipLaunchKernelGGL((attn_mfma2_fwd_kernel<D, LC>), grid, block, lds, s, a);
  }
}

template <int D>
static void launch_mfma2(bool bwd, const MfmaAttnArgs& a, hipStream_t s) {
  // history_size 50 (BASELINE.json configs[3]) as a compile-time constant: forward only (19.4 -> 16.4 us for the 640 problems
  // of a c4 step); the backward kernel measured SLOWER with it (29.4 -> 35.5 us)
  if (D == 20 && a.L == 50 && !bwd) launch_mfma2_lc<D, 50>(bwd, a, s);
  else launch_mfma2_lc<D, 0>(bwd, a, s);
}

int ebn_attn_mfma_fwd(const float* qkv, int64_t ld_qkv, float* out, int64_t ld_out, int64_t n_seq, int32_t L,
                      int32_t h, int32_t d, const EbnDrop& dr, hipStream_t s, bool* handled) {
  *handled = mfma_path_ok(L, d, ld_qkv, ld_out, ld_out, qkv, out, out);
  if (!*handled) return EBN_OK;
  MfmaAttnArgs a{qkv, ld_qkv, nullptr, 0, out, ld_out, n_seq * h, L, h, dr.key_ptr, dr.thresh, dr.scale, nullptr, nullptr, 0};
  if (L > 32) {
    if (d == 16) launch_mfma2<16>(false, a, s);
    else if (d == 20) launch_mfma2<20>(false, a, s);
    else launch_mfma2<32>(false, a, s);
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  const dim3 grid(static_cast<unsigned>(ebn_ceil_div(a.n_prob, ATT_WAVES))), block(64 * ATT_WAVES);
  const size_t lds = static_cast<size_t>(ATT_WAVES) * 3 * L * sizeof(float);  // x STRIDE below
  if (d == 16) hipLaunchKernelGGL((attn_mfma_fwd_kernel<16, 0>), grid, block, lds * Tile<16>::STRIDE, s, a);
  else if (d == 20 && L == 30) hipLaunchKernelGGL((attn_mfma_fwd_kernel<20, 30>), grid, block, lds * Tile<20>::STRIDE, s, a);
  else if (d == 20 && L == 20) hipLaunchKernelGGL((attn_mfma_fwd_kernel<20, 20>), grid, block, lds * Tile<20>::STRIDE, s, a);
  else if (d == 20) hipLaunchKernelGGL((attn_mfma_fwd_kernel<20, 0>), grid, block, lds * Tile<20>::STRIDE, s, a);
  else {
    allow_lds(attn_mfma_fwd_kernel<32, 0>, lds * Tile<32>::STRIDE);
    hipLaunchKernelGGL((attn_mfma_fwd_kernel<32, 0>), grid, block, lds * Tile<32>::STRIDE, s, a);
  }
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

int ebn_attn_mfma_bwd(const float* qkv, int64_t ld_qkv, const float* dout, int64_t ld_dout, float* dqkv,
                      int64_t ld_dqkv, int64_t n_seq, int32_t L, int32_t h, int32_t d, const EbnDrop& dr,
                      hipStream_t s, bool* handled, const float* pool_w, const float* pool_d, int64_t ld_pool) {
  *handled = mfma_path_ok(L, d, ld_qkv, ld_dout, ld_dqkv, qkv, dout, dqkv) &&
             (pool_w == nullptr || ((ld_pool % 4) == 0 && ebn_aligned16(pool_d)));
  if (!*handled) return EBN_OK;
  MfmaAttnArgs a{qkv, ld_qkv, dout, ld_dout, dqkv, ld_dqkv, n_seq * h, L, h, dr.key_ptr, dr.thresh, dr.scale, pool_w, pool_d, ld_pool};
  if (L > 32) {
    if (d == 16) launch_mfma2<16>(true, a, s);
    else if (d == 20) launch_mfma2<20>(true, a, s);
    else launch_mfma2<32>(true, a, s);
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  const dim3 grid(static_cast<unsigned>(ebn_ceil_div(a.n_prob, ATT_WAVES))), block(64 * ATT_WAVES);
  const size_t lds = static_cast<size_t>(ATT_WAVES) * 4 * L * sizeof(float);  // x STRIDE below
  if (d == 16) hipLaunchKernelGGL((attn_mfma_bwd_kernel<16, 0>), grid, block, lds * Tile<16>::STRIDE, s, a);
  else if (d == 20 && L == 30) hipLaunchKernelGGL((attn_mfma_bwd_kernel<20, 30>), grid, block, lds * Tile<20>::STRIDE, s, a);
  else if (d == 20 && L == 20) hipLaunchKernelGGL((attn_mfma_bwd_kernel<20, 20>), grid, block, lds * Tile<20>::STRIDE, s, a);
  else if (d == 20) hipLaunchKernelGGL((attn_mfma_bwd_kernel<20, 0>), grid, block, lds * Tile<20>::STRIDE, s, a);
  else {
    allow_lds(attn_mfma_bwd_kernel<32, 0>, lds * Tile<32>::STRIDE);
    hipLaunchKernelGGL((attn_mfma_bwd_kernel<32, 0>), grid, block, lds * Tile<32>::STRIDE, s, a);
  }
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
