// a3 / a6 fast path: the SelfAttention core (layers.py:231-252) for L <= 32, d in {16,20,32},
// entirely on the matrix cores -- one 64-lane wave per (sequence, head), no LDS, no barriers.
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
// (bitwise-consistent: same products, same order) and the per-row stats move between the two
// layouts with ds_bpermute (__shfl).  Forward = 10+10+16 MFMAs, backward = 88 MFMAs per problem:
// ~15 / ~37 us for the 16,000 (title, head) problems of a batch-32 step, vs 216 / 1300 us for the
// LDS/VALU kernel it replaces (profiles/r01_a_*).
#include "ebn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
};

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// lane (row = lane&31, hi): x[s] = M[row][hi*KH + s]; zero rows >= L.  8-byte loads.
template <int KH>
__device__ __forceinline__ void load_row_form(float (&x)[KH], const float* __restrict__ base, int64_t ld,
                                              int L, int row, int hi) {
  if (row < L) {
    const float2* p = reinterpret_cast<const float2*>(base + static_cast<int64_t>(row) * ld + hi * KH);
#pragma unroll
    for (int s = 0; s < KH / 2; ++s) {
      const float2 v = p[s];
      x[2 * s] = v.x;
      x[2 * s + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int s = 0; s < KH; ++s) x[s] = 0.f;
  }
}

// lane (c = lane&31, hi): y[s] = M[crow(s,hi)][c]; zero when the row is >= L or c >= d.
__device__ __forceinline__ void load_col_form(float (&y)[16], const float* __restrict__ base, int64_t ld,
                                              int L, int d, int c, int hi) {
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int row = crow(s, hi);
    y[s] = (row < L && c < d) ? base[static_cast<int64_t>(row) * ld + c] : 0.f;
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

// store a result tile whose lane owns row `row` and registers own columns crow(r,hi): float4 groups
template <int D>
__device__ __forceinline__ void store_tile_rows(float* __restrict__ dst, const f32x16& acc, int hi, float mul) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c0 = 8 * g + 4 * hi;
    if (c0 < D) {
      float4 v = make_float4(acc[4 * g] * mul, acc[4 * g + 1] * mul, acc[4 * g + 2] * mul, acc[4 * g + 3] * mul);
      *reinterpret_cast<float4*>(dst + c0) = v;
    }
  }
}

// softmax statistics of row i held by lane i (both halves end with the same values):
// t[r] = T[j=crow(r,hi)][i] * inv.  Returns m and 1/Z; turns t into P_ij (zero for j >= L).
__device__ __forceinline__ void softmax_in_lane(f32x16& t, int L, int hi, float inv, float& m, float& invz) {
  m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    t[r] *= inv;
    if (crow(r, hi) < L) m = fmaxf(m, t[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float z = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float e = (crow(r, hi) < L) ? expf(t[r] - m) : 0.f;
    t[r] = e;
    z += e;
  }
  z += __shfl_xor(z, 32, 64);
  invz = 1.0f / z;
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] *= invz;
}

// s[r] = S[i=crow(r,hi)][j] -> P_ji using the stats of row i fetched from lane i
__device__ __forceinline__ void softmax_from_stats(f32x16& s, int L, int hi, float inv, float m, float invz) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = crow(r, hi);
    const float mi = __shfl(m, i, 64);
    const float zi = __shfl(invz, i, 64);
    s[r] = (i < L) ? expf(s[r] * inv - mi) * zi : 0.f;
  }
}

template <int D>
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(MfmaAttnArgs a) {
  constexpr int KH = D / 2;
  const int lane = threadIdx.x & 63;
  const int64_t prob = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (prob >= a.n_prob) return;  // wave-uniform; no barriers in this kernel
  const int row = lane & 31, hi = lane >> 5;
  const int L = a.L, E = a.h * D;
  const int64_t seq = prob / a.h;
  const int head = static_cast<int>(prob - seq * a.h);
  const int64_t row0 = seq * L;
  const float* qb = a.qkv + row0 * a.ld_qkv + head * D;
  const float inv = 1.0f / sqrtf(static_cast<float>(D));

  float qr[KH], kr[KH];
  load_row_form<KH>(qr, qb, a.ld_qkv, L, row, hi);
  load_row_form<KH>(kr, qb + E, a.ld_qkv, L, row, hi);
  float vc[16];
  load_col_form(vc, qb + 2 * E, a.ld_qkv, L, D, row, hi);

  f32x16 T = mm_rows<KH>(kr, qr);  // T[j][i]: lane i, regs j
  f32x16 S = mm_rows<KH>(qr, kr);  // S[i][j]: lane j, regs i
  float m, invz;
  softmax_in_lane(T, L, hi, inv, m, invz);
  softmax_from_stats(S, L, hi, inv, m, invz);  // P[i][j]: lane j, regs i
  const f32x16 O = mm_col_tile(vc, S);         // O^T[c][j]: lane j, regs c

  if (row < L) {
    f32x16 o = O;
    if (a.key_ptr != nullptr) {
      const uint32_t key = *a.key_ptr;
      const uint64_t e0 = static_cast<uint64_t>(row0 + row) * E + head * D;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = crow(r, hi);
        if (c < D) o[r] *= ebn_drop_mult(key, e0 + c, a.thresh, a.scale);
      }
    }
    store_tile_rows<D>(a.out + (row0 + row) * a.ld_out + head * D, o, hi, 1.0f);
  }
}

template <int D>
__global__ __launch_bounds__(256) void attn_mfma_bwd_kernel(MfmaAttnArgs a) {
  constexpr int KH = D / 2;
  const int lane = threadIdx.x & 63;
  const int64_t prob = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (prob >= a.n_prob) return;
  const int row = lane & 31, hi = lane >> 5;
  const int L = a.L, E = a.h * D;
  const int64_t seq = prob / a.h;
  const int head = static_cast<int>(prob - seq * a.h);
  const int64_t row0 = seq * L;
  const float* qb = a.qkv + row0 * a.ld_qkv + head * D;
  const float* gb = a.dout + row0 * a.ld_dout + head * D;
  const float inv = 1.0f / sqrtf(static_cast<float>(D));
  const bool do_drop = a.key_ptr != nullptr;
  const uint32_t key = do_drop ? *a.key_ptr : 0u;

  float qr[KH], kr[KH], vr[KH], gr[KH];
  load_row_form<KH>(qr, qb, a.ld_qkv, L, row, hi);
  load_row_form<KH>(kr, qb + E, a.ld_qkv, L, row, hi);
  load_row_form<KH>(vr, qb + 2 * E, a.ld_qkv, L, row, hi);
  load_row_form<KH>(gr, gb, a.ld_dout, L, row, hi);
  if (do_drop && row < L) {
    const uint64_t e0 = static_cast<uint64_t>(row0 + row) * E + head * D + hi * KH;
#pragma unroll
    for (int s = 0; s < KH; ++s) gr[s] *= ebn_drop_mult(key, e0 + s, a.thresh, a.scale);
  }

  f32x16 Pij = mm_rows<KH>(kr, qr);  // T[j][i]: lane i, regs j
  f32x16 Pji = mm_rows<KH>(qr, kr);  // S[i][j]: lane j, regs i
  float m, invz;
  softmax_in_lane(Pij, L, hi, inv, m, invz);
  softmax_from_stats(Pji, L, hi, inv, m, invz);

  f32x16 dPij = mm_rows<KH>(gr, vr);  // dP[i][j] = V[i].dO[j]: lane i, regs j
  f32x16 dPji = mm_rows<KH>(vr, gr);  // lane j, regs i
  float rowdot = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) rowdot = fmaf(Pij[r], dPij[r], rowdot);
  rowdot += __shfl_xor(rowdot, 32, 64);  // sum_j P[i][j] dP[i][j] for i = lane&31

  // dV^T[c][i] = sum_j dO[j][c] P[i][j]
  float col[16];
  load_col_form(col, gb, a.ld_dout, L, D, row, hi);
  if (do_drop) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int j = crow(s, hi);
      if (j < L && row < D)
        col[s] *= ebn_drop_mult(key, static_cast<uint64_t>(row0 + j) * E + head * D + row, a.thresh, a.scale);
    }
  }
  float* ob = a.out + row0 * a.ld_out + head * D;
  {
    const f32x16 dV = mm_col_tile(col, Pij);
    if (row < L) store_tile_rows<D>(ob + static_cast<int64_t>(row) * a.ld_out + 2 * E, dV, hi, 1.0f);
  }
  // dS in both layouts
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    Pij[r] = Pij[r] * (dPij[r] - rowdot);  // dS[i][j]: lane i, regs j
    const float rd = __shfl(rowdot, crow(r, hi), 64);
    Pji[r] = Pji[r] * (dPji[r] - rd);  // dS[i][j]: lane j, regs i
  }
  // dQ^T[c][i] = inv * sum_j K[j][c] dS[i][j]
  load_col_form(col, qb + E, a.ld_qkv, L, D, row, hi);
  {
    const f32x16 dQ = mm_col_tile(col, Pij);
    if (row < L) store_tile_rows<D>(ob + static_cast<int64_t>(row) * a.ld_out, dQ, hi, inv);
  }
  // dK^T[c][j] = inv * sum_i Q[i][c] dS[i][j]
  load_col_form(col, qb, a.ld_qkv, L, D, row, hi);
  {
    const f32x16 dK = mm_col_tile(col, Pji);
    if (row < L) store_tile_rows<D>(ob + static_cast<int64_t>(row) * a.ld_out + E, dK, hi, inv);
  }
}

}  // namespace

// Returns 1 when the MFMA path handles (L, d, leading dims, alignment); the caller falls back otherwise.
static bool mfma_path_ok(int32_t L, int32_t d, int64_t lda, int64_t ldb, int64_t ldc, const void* p0,
                         const void* p1, const void* p2) {
  if (L > 32 || !(d == 16 || d == 20 || d == 32)) return false;
  if ((lda % 4) || (ldb % 4) || (ldc % 4)) return false;
  return ebn_aligned16(p0) && ebn_aligned16(p1) && ebn_aligned16(p2);
}

int ebn_attn_mfma_fwd(const float* qkv, int64_t ld_qkv, float* out, int64_t ld_out, int64_t n_seq, int32_t L,
                      int32_t h, int32_t d, const EbnDrop& dr, hipStream_t s, bool* handled) {
  *handled = mfma_path_ok(L, d, ld_qkv, ld_out, ld_out, qkv, out, out);
  if (!*handled) return EBN_OK;
  MfmaAttnArgs a{qkv, ld_qkv, nullptr, 0, out, ld_out, n_seq * h, L, h, dr.key_ptr, dr.thresh, dr.scale};
  const dim3 grid(static_cast<unsigned>(ebn_ceil_div(a.n_prob, 4))), block(256);
  if (d == 16) hipLaunchKernelGGL(attn_mfma_fwd_kernel<16>, grid, block, 0, s, a);
  else if (d == 20) hipLaunchKernelGGL(attn_mfma_fwd_kernel<20>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(attn_mfma_fwd_kernel<32>, grid, block, 0, s, a);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

int ebn_attn_mfma_bwd(const float* qkv, int64_t ld_qkv, const float* dout, int64_t ld_dout, float* dqkv,
                      int64_t ld_dqkv, int64_t n_seq, int32_t L, int32_t h, int32_t d, const EbnDrop& dr,
                      hipStream_t s, bool* handled) {
  *handled = mfma_path_ok(L, d, ld_qkv, ld_dout, ld_dqkv, qkv, dout, dqkv);
  if (!*handled) return EBN_OK;
  MfmaAttnArgs a{qkv, ld_qkv, dout, ld_dout, dqkv, ld_dqkv, n_seq * h, L, h, dr.key_ptr, dr.thresh, dr.scale};
  const dim3 grid(static_cast<unsigned>(ebn_ceil_div(a.n_prob, 4))), block(256);
  if (d == 16) hipLaunchKernelGGL(attn_mfma_bwd_kernel<16>, grid, block, 0, s, a);
  else if (d == 20) hipLaunchKernelGGL(attn_mfma_bwd_kernel<20>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(attn_mfma_bwd_kernel<32>, grid, block, 0, s, a);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
