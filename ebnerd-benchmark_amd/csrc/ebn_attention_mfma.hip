// a3 / a6 fast path: the SelfAttention core (layers.py:231-252) for L <= 64 (one or 2 x 2 MFMA tiles), d in {16,20,32},
// on the matrix cores -- one 64-lane wave per (sequence, head), no workgroup barriers.
//
// Everything is a 32x32 exact-fp32 MFMA tile (v_mfma_f32_32x32x2_f32), padded from L x L / L x d.
// Two facts about that instruction drive the dataflow:
//   * its contraction index is (step, lane-half); ANY mapping of the logical k to (step, half) is valid
//     as long as A and B agree.  "Row-form" operands give half `hi` the k-range [hi*d/2, (hi+1)*d/2)
//     of a row (contiguous 8-byte loads, no selects).
//   * its result layout -- lane (n = lane&31, hi), register r <-> row crow(r,hi) = (r&3)+8(r>>2)+4hi --
//     is exactly a B operand whose k is (r, hi): a result tile feeds the next product from registers.
// Because the reference multiplies V by the TRANSPOSED attention matrix (O = P^T V, layers.py:249) the product contracts
// over the softmax-row index i, so P is needed with i on registers, while the softmax statistics (max_j, sum_j) are
// cheapest with j on registers (T = K Q^T: lane i, regs j -> in-lane reductions + one cross-half swap).
//
// L <= 32 (the kernels of every BASELINE configuration's title level, and of history_size 20): T is computed ONCE, the
// softmax is done in-lane, and the tile is TRANSPOSED through wave-private LDS into the other layout (tile_transpose).
// Round 1 instead computed both layouts with swapped operands; on gfx950 the fp32 MFMA shares the vector ALU (it does not
// overlap VALU work), so a recomputed tile is 640 ALU cycles where the LDS pipe does the same job on the side.  Forward
// = 10 + 16 MFMAs, backward = 68.  Operands that are needed in row form only (Q, K forward; V backward) come straight
// from global memory; the others are copied once, as coalesced 16-byte loads, into wave-private LDS; result tiles leave
// from registers.  32 < L <= 64: the attn_mfma2_* kernels further down (2 x 2 tiles, both layouts computed, results staged
// through LDS).
#include <stdlib.h>

#include "ebn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ATT_WAVES = 4;  // independent (sequence, head) problems per workgroup, one wave each

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
  // group-form kernels: start-up stagger of the first round of workgroups (see stagger_first_round); 0 = off
  uint32_t stagger_units, stagger_shift, stagger_mod;
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
__device__ __forceinline__ void stage_load(Staged<D>& st, const float* __restrict__ base, uint32_t ld, int L, int lane) {
  using T = Tile<D>;
#pragma unroll
  for (int t = 0; t < T::ROUNDS; ++t) {
    int v = lane + 64 * t;
    if (T::VECS % 64 != 0 && v >= T::VECS) v = 0;
    const int r = v / T::VPR, c4 = v - r * T::VPR;
    const uint32_t rc = r < L ? r : 0;
    // wave-uniform base + 32-bit lane offset (<= 64 rows x ld < 2^24, checked by the launcher): scalar-base addressing,
    // no 64-bit multiply-add per piece
    st.v[t] = *reinterpret_cast<const float4*>(base + (rc * ld + static_cast<uint32_t>(c4 * 4)));
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
      // e0, r * E and c4 * 4 are multiples of 4: two aligned pairs; wave-uniform 64-bit part + 32-bit lane part
      const uint64_t pr = (e0 >> 1) + ((static_cast<uint32_t>(r) * static_cast<uint32_t>(E) + static_cast<uint32_t>(c4 * 4)) >> 1);
      const uint32_t h0 = ebn_dropout_pair_hash(key, pr), h1 = ebn_dropout_pair_hash(key, pr + 1);
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
__device__ __forceinline__ void stage_out(const float* __restrict__ lds, float* __restrict__ base, uint32_t ld, int L,
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
      const uint64_t pr = (e0 >> 1) + ((static_cast<uint32_t>(r) * static_cast<uint32_t>(E) + static_cast<uint32_t>(c4 * 4)) >> 1);
      const uint32_t h0 = ebn_dropout_pair_hash(key, pr), h1 = ebn_dropout_pair_hash(key, pr + 1);
      y.x *= ((h0 & 0xFFFFu) >= thresh) ? scale : 0.f;
      y.y *= ((h0 >> 16) >= thresh) ? scale : 0.f;
      y.z *= ((h1 & 0xFFFFu) >= thresh) ? scale : 0.f;
      y.w *= ((h1 >> 16) >= thresh) ? scale : 0.f;
    }
    *reinterpret_cast<float4*>(base + (static_cast<uint32_t>(r) * ld + static_cast<uint32_t>(c4 * 4))) = y;
  }
}

// The LDS regions hold L rows (not 32).  Rows >= L are NOT zeroed on the way into the registers: the row index is clamped
// (row 0 is read again), so every operand element is a finite value of the problem's own data, and zeros are only needed
// -- and only produced, by the softmax masks -- where a row >= L is a CONTRACTION index:
//   * row-form operands feed mm_rows, which contracts over the head dimension: a clamped lane only changes tile entries
//     whose lane or register index is >= L.  Register indices >= L are masked by the softmax (P = 0 there, and
//     dS = P (dP - rowdot) = 0 with a finite dP); lanes >= L only reach output columns >= L, which are never stored, and
//     their statistics are fetched (__shfl) only for masked entries;
//   * column-form operands feed mm_col_tile, which contracts over the row index against a P / dS tile that is zero there.
//     Columns >= D only produce output rows >= D (never stored): the column index is clamped as well.
// (Value selects on every element were a tenth of the VALU instructions of these VALU-issue-bound kernels.)
// lane (row = lane&31, hi): x[s] = M[row][hi*KH + s].  8-byte LDS reads.
template <int D>
__device__ __forceinline__ void lds_row_form(float (&x)[D / 2], const float* __restrict__ lds, int L, int row, int hi) {
  constexpr int KH = D / 2;
  const float2* p = reinterpret_cast<const float2*>(lds + (row < L ? row : 0) * Tile<D>::STRIDE + hi * KH);
#pragma unroll
  for (int s = 0; s < KH / 2; ++s) {
    const float2 v = p[s];
    x[2 * s] = v.x;
    x[2 * s + 1] = v.y;
  }
}

// lane (c = lane&31, hi): y[s] = M[crow(s,hi)][c] (row and column clamped, see above).
template <int D>
__device__ __forceinline__ void lds_col_form(float (&y)[16], const float* __restrict__ lds, int L, int c, int hi) {
  const float* p = lds + ((c < D) ? c : 0);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int r = crow(s, hi);
    y[s] = p[(r < L ? r : 0) * Tile<D>::STRIDE];
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

// The same tile straight to global memory (optionally through the dropout mask): lane (row, hi) owns the 16-byte pieces
// at columns 8g + 4hi of its row -- as many store instructions as the staged, 80-bytes-per-row form issues, without the
// LDS round trip (write, wait, read, wait) in front of them and without a staging region.
template <int D, bool DROP>
__device__ __forceinline__ void tile_rows_to_global(float* __restrict__ base, uint32_t ld, const f32x16& acc, int L, int row,
                                                    int hi, float mul, uint32_t key, uint64_t e0, int E, uint32_t thresh,
                                                    float scale) {
  if (row >= L) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c0 = 8 * g + 4 * hi;
    if (c0 < D) {
      float4 y = make_float4(acc[4 * g] * mul, acc[4 * g + 1] * mul, acc[4 * g + 2] * mul, acc[4 * g + 3] * mul);
      if (DROP) {
        const uint64_t pr = (e0 >> 1) + ((static_cast<uint32_t>(row) * static_cast<uint32_t>(E) + static_cast<uint32_t>(c0)) >> 1);
        const uint32_t h0 = ebn_dropout_pair_hash(key, pr), h1 = ebn_dropout_pair_hash(key, pr + 1);
        y.x *= ((h0 & 0xFFFFu) >= thresh) ? scale : 0.f;
        y.y *= ((h0 >> 16) >= thresh) ? scale : 0.f;
        y.z *= ((h1 & 0xFFFFu) >= thresh) ? scale : 0.f;
        y.w *= ((h1 >> 16) >= thresh) ? scale : 0.f;
      }
      *reinterpret_cast<float4*>(base + (static_cast<uint32_t>(row) * ld + static_cast<uint32_t>(c0))) = y;
    }
  }
}

// Row form straight from global memory: lane (row, hi) reads its D/2 consecutive floats (8-byte pieces: the half-row
// offset hi * D/2 floats is 40 bytes for D = 20).  For operands that are needed in row form only.
template <int D>
__device__ __forceinline__ void global_row_form(float (&x)[D / 2], const float* __restrict__ base, uint32_t ld, int L, int row, int hi) {
  constexpr int KH = D / 2;
  const float2* p = reinterpret_cast<const float2*>(base + (static_cast<uint32_t>(row < L ? row : 0) * ld + static_cast<uint32_t>(hi * KH)));
#pragma unroll
  for (int s = 0; s < KH / 2; ++s) {
    const float2 v = p[s];
    x[2 * s] = v.x;
    x[2 * s + 1] = v.y;
  }
}

// Softmax in the exp2 domain: `inv2` = log2(e)/sqrt(d), so exp(x/sqrt(d) - max) = exp2(x*inv2 - max2) and one
// v_exp_f32 per element replaces the ~20-instruction expf expansion (the softmax of 2 x 32 x 32 scores was the VALU
// bottleneck of these kernels).  Arguments are <= 0, flush-to-zero below 2^-126 is the correct limit.
//
// Row i is held by lane i (both lane halves together): t[r] = T[j=crow(r,hi)][i].  Turns t into P_ij (zero for j >= L).
__device__ __forceinline__ void softmax_in_lane(f32x16& t, int L, int hi, float inv2) {
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    t[r] *= inv2;
    if (crow(r, hi) < L) m = fmaxf(m, t[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float z = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    t[r] = (crow(r, hi) < L) ? __builtin_amdgcn_exp2f(t[r] - m) : 0.f;
    z += t[r];
  }
  z += __shfl_xor(z, 32, 64);
  const float rz = 1.0f / z;  // z in [1, 32]
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] *= rz;
}

// 32 x 32 tile transpose through wave-private LDS.  In: lane (a = lane&31, hi), register r <-> element (a, b = crow(r,hi)).
// Out: lane (b, hi), register r <-> element (a = crow(r,hi), b), with the registers of rows a >= L set to zero (they are
// contraction indices of the product that follows; the source lanes a >= L hold finite duplicates of row 0, see above).
// 16 ds_write_b32 (32 consecutive floats per lane half: conflict-free) and 4 ds_read_b128 (rows 8g + 4hi + 0..3 of the
// output are contiguous; row stride 36 floats spreads 16 lanes x 16 bytes over all 64 banks).
//
// Why transposes instead of computing a tile twice with the operands swapped (which is what round 1 did, moving only
// the per-row softmax statistics between the layouts): on gfx950 the exact-fp32 MFMA and the vector ALU do NOT overlap --
// tools/microbench/mfma_valu_overlap.hip: matrix loop 231 us, vector loop 135 us, both in one wave 446 us, split over
// two waves of a SIMD 361 us -- so these kernels cost (MFMAs x 64 cycles + VALU cycles) per SIMD, and a recomputed
// 32 x 32 x 20 tile is 640 of those cycles where the LDS pipe does the same job on the side.
constexpr int TP_STRIDE = 36;
constexpr int TP_FLOATS = 32 * TP_STRIDE;

__device__ __forceinline__ void tile_transpose(f32x16& t, float* __restrict__ buf, int L, int lane_row, int hi) {
  float* w = buf + 4 * hi * TP_STRIDE + lane_row;
#pragma unroll
  for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2)) * TP_STRIDE] = t[r];  // element (a, b) -> buf[b][a]
  wave_lds_sync();
  const float* rd = buf + lane_row * TP_STRIDE + 4 * hi;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(rd + 8 * g);
    t[4 * g] = (8 * g + 4 * hi + 0 < L) ? v.x : 0.f;
    t[4 * g + 1] = (8 * g + 4 * hi + 1 < L) ? v.y : 0.f;
    t[4 * g + 2] = (8 * g + 4 * hi + 2 < L) ? v.z : 0.f;
    t[4 * g + 3] = (8 * g + 4 * hi + 3 < L) ? v.w : 0.f;
  }
}


// LC: sequence length known at compile time (0 = use a.L).  title_size = 30 and history_size = 20 are what every
// BASELINE config runs: with L a constant most of the row / column validity masks of a 32-wide tile fold away (only
// registers 14, 15 of the upper lane half can be rows >= 30).
//
// One problem per wave.  (PERSISTENT waves that walk p, p + stride, ... with the tiles of the next problem prefetched into
// registers during the MFMA / softmax phase were measured: 36-63 more registers = one or two waves per SIMD fewer, forward
// 41 -> 39-41 us, backward 73 -> 84-92 us, far worse when the occupancy hint makes the prefetch registers spill.  These
// kernels are bound by the LENGTH of a wave's dependent chain -- LDS round trips, MFMA chains, cross-lane statistics --
// and what hides it is the number of resident waves, not a deeper pipeline inside one wave.)
struct AttnProb {
  int64_t row0;  // first row of the sequence
  uint32_t seq;
  uint32_t head;
};

// Workgroup -> problem order.  The hardware deals consecutive workgroups round-robin to the 8 XCDs (private L2s), and the heads
// of a title are neighbours in memory: head k's 80-byte row pieces share 128-byte lines with heads k - 1 / k + 1.  With the
// dispatch order as the problem order those neighbours sit on DIFFERENT XCDs, every shared line is fetched by two L2s and --
// what costs -- written as two partial lines from two L2s.  Remapped so that each XCD walks one contiguous run of problems
// (bijective for any grid size; the same formula as the GEMM's tile order).
__device__ __forceinline__ uint32_t xcd_chunked_block() {
  const uint32_t nwg = gridDim.x, orig = blockIdx.x;
  const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = orig & 7u, idx = orig >> 3;
  return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// The workgroups of a launch's FIRST dispatch round start together, and a workgroup of the group-form backward is three phases
// of fixed length -- fetch (memory), MFMA / softmax chain (ALU), store (memory) -- so the `mod` workgroups that share a CU stay
// in phase: all of them wait for memory together, then all compete for the ALU together (measured: 69 us where the memory
// phases alone take 45 and the MFMAs 28; a padded-LDS control with 3 instead of 4 workgroups per CU: 76 us; 5: 67 us).
// De-phased at start-up: workgroup b of the first round sleeps ((b >> shift) % mod) x units x 512 cycles before it begins;
// later rounds inherit the offsets of the slots they fill.  800 titles: 69 -> 62 us, 1760: 150 -> 139 us (units 6-8, shift 8 =
// the dispatcher deals 256 consecutive workgroups one per CU; shift 3: no gain); c4 step -16 us, c2 -3 us.
// (Measured instead of this and NOT kept, profiles/r04_tuning_notes.md: a persistent, software-pipelined form of the kernel that
// requests the next group's Q | K | dO pieces into 36 registers before it computes -- 128 VGPRs, bit-identical, 66.6 us.)
__device__ __forceinline__ void stagger_first_round(const MfmaAttnArgs& a) {
  if (a.stagger_units != 0u && blockIdx.x < (a.stagger_mod << a.stagger_shift)) {
    const uint32_t n = ((blockIdx.x >> a.stagger_shift) % a.stagger_mod) * a.stagger_units;
    for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
  }
}

// n_prob < 2^31 (checked by the launcher): one 32-bit scalar division per problem
__device__ __forceinline__ AttnProb attn_prob(int64_t prob, int h, int L) {
  AttnProb p;
  p.seq = static_cast<uint32_t>(prob) / static_cast<uint32_t>(h);
  p.head = static_cast<uint32_t>(prob) - p.seq * static_cast<uint32_t>(h);
  p.row0 = static_cast<int64_t>(p.seq) * L;
  return p;
}

template <int D>
__host__ __device__ constexpr int fwd_wave_floats(int L) {  // V's region, later the transpose buffer
  return L * Tile<D>::STRIDE > TP_FLOATS ? L * Tile<D>::STRIDE : TP_FLOATS;
}

template <int D, int LC>
__global__ __launch_bounds__(64 * ATT_WAVES) void attn_mfma_fwd_kernel(MfmaAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // ATT_WAVES x fwd_wave_floats
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform on its face: base pointers stay scalar
  const int64_t prob = static_cast<int64_t>(xcd_chunked_block()) * ATT_WAVES + wv;
  if (prob >= a.n_prob) return;  // wave-uniform; no workgroup barriers in this kernel
  const int L = LC ? LC : a.L, E = a.h * D;
  float* sv = smem + wv * fwd_wave_floats<D>(L);
  const int row = lane & 31, hi = lane >> 5;
  const float inv2 = 1.44269504088896341f / sqrtf(static_cast<float>(D));
  const bool drop = a.key_ptr != nullptr;
  const uint32_t key = drop ? *a.key_ptr : 0u;

  const AttnProb p = attn_prob(prob, a.h, L);
  const float* qb = a.qkv + p.row0 * a.ld_qkv + p.head * D;
  // Q and K are needed in row form only (T = K Q^T contracts over the head dimension): straight from global memory.
  // Only V, wanted in column form, goes through LDS.
  float qr[D / 2], kr[D / 2];
  {
    Staged<D> tv;
    stage_load<D>(tv, qb + 2 * E, a.ld_qkv, L, lane);
    global_row_form<D>(qr, qb, a.ld_qkv, L, row, hi);
    global_row_form<D>(kr, qb + E, a.ld_qkv, L, row, hi);
    stage_store<D, false>(sv, tv, L, lane, 0u, 0u, 0, 0u, 0.f);
  }
  wave_lds_sync();
  float vc[16];
  lds_col_form<D>(vc, sv, L, row, hi);

  f32x16 P = mm_rows<D / 2>(kr, qr);  // T[j][i]: lane i, regs j
  softmax_in_lane(P, L, hi, inv2);    // P[i][j]: lane i, regs j (rows of the softmax in-lane)
  wave_lds_sync();                    // V's column form is in registers: its region becomes the transpose buffer
  tile_transpose(P, sv, L, row, hi);  // P[i][j]: lane j, regs i -- the contraction index of P^T V on the registers
  const f32x16 O = mm_col_tile(vc, P);  // O^T[c][j]: lane j, regs c
  float* ob = a.out + p.row0 * a.ld_out + p.head * D;
  const uint64_t e0 = static_cast<uint64_t>(p.row0) * E + p.head * D;
  if (drop) tile_rows_to_global<D, true>(ob, a.ld_out, O, L, row, hi, 1.0f, key, e0, E, a.thresh, a.scale);
  else tile_rows_to_global<D, false>(ob, a.ld_out, O, L, row, hi, 1.0f, 0u, 0u, 0, 0u, 0.f);
}

// Forward, GROUP form (see the backward's group form further down for the measurements behind it): G consecutive heads of one
// sequence per workgroup, one wave per head; Q, K, V come in as contiguous G x 80-byte row runs through the whole workgroup,
// the output tiles go back the same way (dropout applied on the way out).  LDS per wave: Q | K (later the transpose buffer) | V
// (later the output tile).  Same MFMA order as attn_mfma_fwd_kernel: identical bytes.
template <int D>
__host__ __device__ constexpr int fwd_group_v_offset(int L) {
  return 2 * L * Tile<D>::STRIDE > TP_FLOATS ? 2 * L * Tile<D>::STRIDE : TP_FLOATS;
}
__host__ __device__ constexpr int group_tile_pad(int L);
template <int D>
__host__ __device__ constexpr int fwd_group_wave_floats(int L) {
  return fwd_group_v_offset<D>(L) + L * Tile<D>::STRIDE + group_tile_pad(L);
}

template <int D, int LC, int G>
__global__ __launch_bounds__(64 * G) void attn_mfma_fwd_group_kernel(MfmaAttnArgs a) {
  using T = Tile<D>;
  constexpr int NT = 64 * G;
  constexpr int GV = G * T::VPR;
  constexpr int ROUNDS = (32 * GV + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // G x fwd_group_wave_floats
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = LC ? LC : a.L, E = a.h * D;
  const int region = L * T::STRIDE;
  const int wave_floats = fwd_group_wave_floats<D>(L), v_off = fwd_group_v_offset<D>(L);
  const int row = lane & 31, hi = lane >> 5;
  const float inv2 = 1.44269504088896341f / sqrtf(static_cast<float>(D));
  const bool drop = a.key_ptr != nullptr;
  const uint32_t key = drop ? *a.key_ptr : 0u;

  const AttnProb p0 = attn_prob(static_cast<int64_t>(xcd_chunked_block()) * G, a.h, L);
  const uint32_t gcol = p0.head * D;
  const float* gq = a.qkv + p0.row0 * a.ld_qkv + gcol;
  const int nvec = L * GV;
  {
    float4 vq[ROUNDS], vk[ROUNDS], vv[ROUNDS];
    uint32_t dst[ROUNDS];
#pragma unroll
    for (int t = 0; t < ROUNDS; ++t) {
      const int idx = tid + NT * t;
      const int idc = idx < nvec ? idx : 0;
      const uint32_t r = static_cast<uint32_t>(idc) / static_cast<uint32_t>(GV), c4g = static_cast<uint32_t>(idc) - r * GV;
      const uint32_t head = c4g / T::VPR, c4 = c4g - head * T::VPR;
      dst[t] = head * wave_floats + r * T::STRIDE + c4 * 4;
      const uint32_t off = r * static_cast<uint32_t>(a.ld_qkv) + c4g * 4;
      vq[t] = *reinterpret_cast<const float4*>(gq + off);
      vk[t] = *reinterpret_cast<const float4*>(gq + E + off);
      vv[t] = *reinterpret_cast<const float4*>(gq + 2 * E + off);
    }
#pragma unroll
    for (int t = 0; t < ROUNDS; ++t) {
      if (tid + NT * t < nvec) {
        *reinterpret_cast<float4*>(smem + dst[t]) = vq[t];
        *reinterpret_cast<float4*>(smem + dst[t] + region) = vk[t];
        *reinterpret_cast<float4*>(smem + dst[t] + v_off) = vv[t];
      }
    }
  }
  __syncthreads();

  float* sq = smem + wv * wave_floats;
  float* sk = sq + region;
  float* sv = sq + v_off;
  f32x16 P;
  {
    float qr[D / 2], kr[D / 2];
    lds_row_form<D>(qr, sq, L, row, hi);
    lds_row_form<D>(kr, sk, L, row, hi);
    P = mm_rows<D / 2>(kr, qr);  // T[j][i]: lane i, regs j
  }
  float vc[16];
  lds_col_form<D>(vc, sv, L, row, hi);
  softmax_in_lane(P, L, hi, inv2);
  wave_lds_sync();                    // Q, K, V are in registers: Q|K becomes the transpose buffer, V's tile takes the output
  tile_transpose(P, sq, L, row, hi);  // P[i][j]: lane j, regs i
  {
    const f32x16 O = mm_col_tile(vc, P);  // O^T[c][j]: lane j, regs c
    tile_rows_to_lds<D>(sv, O, L, row, hi, 1.0f);
  }
  __syncthreads();

  float* go = a.out + p0.row0 * a.ld_out + gcol;
#pragma unroll
  for (int t = 0; t < ROUNDS; ++t) {
    const int idx = tid + NT * t;
    if (idx < nvec) {
      const uint32_t r = static_cast<uint32_t>(idx) / static_cast<uint32_t>(GV), c4g = static_cast<uint32_t>(idx) - r * GV;
      const uint32_t head = c4g / T::VPR, c4 = c4g - head * T::VPR;
      float4 y = *reinterpret_cast<const float4*>(smem + head * wave_floats + v_off + r * T::STRIDE + c4 * 4);
      if (drop) {
        const uint64_t pr = ((static_cast<uint64_t>(p0.row0) * E + gcol) >> 1) + ((r * static_cast<uint32_t>(E) + c4g * 4) >> 1);
        const uint32_t h0 = ebn_dropout_pair_hash(key, pr), h1 = ebn_dropout_pair_hash(key, pr + 1);
        y.x *= ((h0 & 0xFFFFu) >= a.thresh) ? a.scale : 0.f;
        y.y *= ((h0 >> 16) >= a.thresh) ? a.scale : 0.f;
        y.z *= ((h1 & 0xFFFFu) >= a.thresh) ? a.scale : 0.f;
        y.w *= ((h1 >> 16) >= a.thresh) ? a.scale : 0.f;
      }
      *reinterpret_cast<float4*>(go + (r * static_cast<uint32_t>(a.ld_out) + c4g * 4)) = y;
    }
  }
}

// Backward.  Everything but d(K) is computed in the lane-i layout (softmax row i on the lanes, j on the registers: row
// statistics and sum_j P dP are in-lane); d(K) contracts over i and gets d(S) through one tile transpose.
template <int D>
__host__ __device__ constexpr int bwd_wave_floats(int L) {  // Q | K | dO regions; K, dO later the transpose buffer
  return 3 * L * Tile<D>::STRIDE > L * Tile<D>::STRIDE + TP_FLOATS ? 3 * L * Tile<D>::STRIDE : L * Tile<D>::STRIDE + TP_FLOATS;
}

template <int D, int LC>
__global__ __launch_bounds__(64 * ATT_WAVES) void attn_mfma_bwd_kernel(MfmaAttnArgs a) {
  using T = Tile<D>;
  constexpr int KH = D / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // ATT_WAVES x bwd_wave_floats
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform on its face: base pointers stay scalar
  const int64_t prob = static_cast<int64_t>(xcd_chunked_block()) * ATT_WAVES + wv;
  if (prob >= a.n_prob) return;
  const int L = LC ? LC : a.L, E = a.h * D;
  const int region = L * T::STRIDE;
  float* sq = smem + wv * bwd_wave_floats<D>(L);
  float* sk = sq + region;
  float* sg = sk + region;  // dO with the pooling term added and the forward dropout mask applied
  const int row = lane & 31, hi = lane >> 5;
  const float inv = 1.0f / sqrtf(static_cast<float>(D));
  const float inv2 = inv * 1.44269504088896341f;
  const bool drop = a.key_ptr != nullptr, pooled = a.pool_w != nullptr;  // wave-uniform
  const uint32_t key = drop ? *a.key_ptr : 0u;

  const AttnProb p = attn_prob(prob, a.h, L);
  const float* qb = a.qkv + p.row0 * a.ld_qkv + p.head * D;
  float vr[KH];  // row form of V (lane = row), the only form V is needed in: straight from global memory
  {
    Staged<D> tq, tk, tg;
    stage_load<D>(tq, qb, a.ld_qkv, L, lane);
    stage_load<D>(tk, qb + E, a.ld_qkv, L, lane);
    stage_load<D>(tg, a.dout + p.row0 * a.ld_dout + p.head * D, a.ld_dout, L, lane);
    global_row_form<D>(vr, qb + 2 * E, a.ld_qkv, L, row, hi);
    if (pooled) {
      PoolTerm<D> pt;
      pool_load<D>(pt, a.pool_w + p.row0, a.pool_d + p.seq * a.ld_pool + p.head * D, L, lane);
      pool_add<D>(tg, pt);
    }
    stage_store<D, false>(sq, tq, L, lane, 0u, 0u, 0, 0u, 0.f);
    stage_store<D, false>(sk, tk, L, lane, 0u, 0u, 0, 0u, 0.f);
    if (drop) stage_store<D, true>(sg, tg, L, lane, key, static_cast<uint64_t>(p.row0) * E + p.head * D, E, a.thresh, a.scale);
    else stage_store<D, false>(sg, tg, L, lane, 0u, 0u, 0, 0u, 0.f);
  }
  wave_lds_sync();

  float* ob = a.out + p.row0 * a.ld_out + p.head * D;
  f32x16 P, dP;
  {
    float qr[KH], kr[KH];
    lds_row_form<D>(qr, sq, L, row, hi);
    lds_row_form<D>(kr, sk, L, row, hi);
    P = mm_rows<KH>(kr, qr);  // T[j][i]: lane i, regs j
  }
  softmax_in_lane(P, L, hi, inv2);  // P[i][j]: lane i, regs j
  {
    float gr[KH];
    lds_row_form<D>(gr, sg, L, row, hi);
    dP = mm_rows<KH>(gr, vr);  // dP[i][j] = V[i].dO[j]: lane i, regs j
  }
  float rowdot = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) rowdot = fmaf(P[r], dP[r], rowdot);
  rowdot += __shfl_xor(rowdot, 32, 64);  // sum_j P[i][j] dP[i][j] for i = lane&31

  float col[16];
  lds_col_form<D>(col, sg, L, row, hi);
  {  // dV^T[c][i] = sum_j dO[j][c] P[i][j]
    const f32x16 dV = mm_col_tile(col, P);
    tile_rows_to_global<D, false>(ob + 2 * E, a.ld_out, dV, L, row, hi, 1.0f, 0u, 0u, 0, 0u, 0.f);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) P[r] = P[r] * (dP[r] - rowdot);  // dS[i][j]: lane i, regs j
  lds_col_form<D>(col, sk, L, row, hi);
  {  // dQ^T[c][i] = inv * sum_j K[j][c] dS[i][j]
    const f32x16 dQ = mm_col_tile(col, P);
    tile_rows_to_global<D, false>(ob, a.ld_out, dQ, L, row, hi, inv, 0u, 0u, 0, 0u, 0.f);
  }
  wave_lds_sync();                    // K and dO have been read for the last time: their regions become the transpose buffer
  tile_transpose(P, sk, L, row, hi);  // dS[i][j]: lane j, regs i
  lds_col_form<D>(col, sq, L, row, hi);
  {  // dK^T[c][j] = inv * sum_i Q[i][c] dS[i][j]
    const f32x16 dK = mm_col_tile(col, P);
    tile_rows_to_global<D, false>(ob + E, a.ld_out, dK, L, row, hi, inv, 0u, 0u, 0, 0u, 0.f);
  }
}


// Backward, GROUP form: a workgroup = G consecutive heads of ONE sequence, one wave per head as above, but the operands come
// in and the results go out through the whole workgroup.  Measured on the per-wave kernel above (profiles/r03_tuning_notes.md):
// its loads alone take 25 us, its stores alone 23 us, both together WITHOUT any arithmetic 66 us of the kernel's 75 -- the
// memory pattern (80-byte row pieces, 16-byte result pieces scattered over 30 rows), not the MFMA chain, is what it costs,
// with no excess HBM traffic (FETCH/WRITE_SIZE = the algorithmic bytes): it is request count and DRAM page locality.  Here the
// G x D columns of the group are one contiguous run per row (320 bytes for G = 4), fetched as consecutive 16-byte lanes into
// the per-head LDS tiles, and the three result tiles of every head go back through LDS the same way: half the L2 requests
// (13.2 M against 29.6 M per 3200 titles), 75 -> 65 us per 800 titles on HBM-resident data.  G = 4 keeps one wave per SIMD;
// G = 2 (72 us), G = 5 (83 us) and G = 10 (97 us) lose more to request count or to the two workgroup barriers.
// LDS per wave: THREE tiles, Q | K | dO (28.8 KB per workgroup of four heads: five workgroups per CU; round 3's fourth tile
// for V, 38.4 KB and four workgroups, measured 2 % slower).  V is needed in row form only and comes straight from global memory
// into registers, as in the per-wave kernel.  Results: d(V) over dO once dO's row and column forms are in registers; Q|K become
// the transpose buffer once their forms are in registers, d(Q) waits in registers across the transpose and then takes Q's tile,
// d(K) K's.  Bit-identical to the per-wave kernel.
template <int D>
__host__ __device__ constexpr int bwd_group_g_offset(int L) {  // offset of the dO tile: behind Q | K and behind the transpose buffer
  return 2 * L * Tile<D>::STRIDE > TP_FLOATS ? 2 * L * Tile<D>::STRIDE : TP_FLOATS;
}
// Distance between the tiles of neighbouring heads.  The cooperative float4 stores / reads walk 5 lanes of one head's row, then
// jump to the next head's tile: with the tiles 1800 floats apart (L = 30) the 8-lane store groups and 16-lane read groups hit
// the same banks from two heads -- 564 extra LDS cycles per workgroup of 684 in all (tools/lds/bank_sim.py; counters: 3.2 M conflict
// cycles per launch).  12 floats of padding: 156.  (Found by search for L = 30; other lengths keep the tight layout.)
__host__ __device__ constexpr int group_tile_pad(int L) { return L == 30 ? 12 : 0; }
template <int D>
__host__ __device__ constexpr int bwd_group_wave_floats(int L) {
  return bwd_group_g_offset<D>(L) + L * Tile<D>::STRIDE + group_tile_pad(L);
}

template <int D, int LC, int G>
__global__ __launch_bounds__(64 * G) void attn_mfma_bwd_group_kernel(MfmaAttnArgs a) {
  using T = Tile<D>;
  constexpr int KH = D / 2;
  constexpr int NT = 64 * G;
  constexpr int GV = G * T::VPR;                    // float4 per row of the group
  constexpr int ROUNDS = (32 * GV + NT - 1) / NT;   // L <= 32
  extern __shared__ __attribute__((aligned(16))) float smem[];  // G x bwd_group_wave_floats
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = LC ? LC : a.L, E = a.h * D;
  const int region = L * T::STRIDE;
  const int wave_floats = bwd_group_wave_floats<D>(L), g_off = bwd_group_g_offset<D>(L);
  const int row = lane & 31, hi = lane >> 5;
  const float inv = 1.0f / sqrtf(static_cast<float>(D));
  const float inv2 = inv * 1.44269504088896341f;
  const bool drop = a.key_ptr != nullptr, pooled = a.pool_w != nullptr;  // uniform
  const uint32_t key = drop ? *a.key_ptr : 0u;

  stagger_first_round(a);
  const AttnProb p0 = attn_prob(static_cast<int64_t>(xcd_chunked_block()) * G, a.h, L);  // head p0.head .. + G - 1 of sequence p0.seq
  const uint32_t gcol = p0.head * D;
  const float* gq = a.qkv + p0.row0 * a.ld_qkv + gcol;
  const float* gd = a.dout + p0.row0 * a.ld_dout + gcol;
  const int nvec = L * GV;
  float vr[KH];
  global_row_form<D>(vr, gq + 2 * E + wv * D, a.ld_qkv, L, row, hi);
  {
    float4 vq[ROUNDS], vk[ROUNDS], vg[ROUNDS];
    uint32_t dst[ROUNDS];  // LDS float offset of the piece inside its head's Q tile
#pragma unroll
    for (int t = 0; t < ROUNDS; ++t) {
      const int idx = tid + NT * t;
      const int idc = idx < nvec ? idx : 0;
      const uint32_t r = static_cast<uint32_t>(idc) / static_cast<uint32_t>(GV), c4g = static_cast<uint32_t>(idc) - r * GV;
      const uint32_t head = c4g / T::VPR, c4 = c4g - head * T::VPR;
      dst[t] = head * wave_floats + r * T::STRIDE + c4 * 4;
      const uint32_t off = r * static_cast<uint32_t>(a.ld_qkv) + c4g * 4;
      vq[t] = *reinterpret_cast<const float4*>(gq + off);
      vk[t] = *reinterpret_cast<const float4*>(gq + E + off);
      vg[t] = *reinterpret_cast<const float4*>(gd + (r * static_cast<uint32_t>(a.ld_dout) + c4g * 4));
      if (pooled) {
        const float w = a.pool_w[p0.row0 + r];
        const float4 pd = *reinterpret_cast<const float4*>(a.pool_d + p0.seq * a.ld_pool + gcol + c4g * 4);
        vg[t].x = fmaf(w, pd.x, vg[t].x);
        vg[t].y = fmaf(w, pd.y, vg[t].y);
        vg[t].z = fmaf(w, pd.z, vg[t].z);
        vg[t].w = fmaf(w, pd.w, vg[t].w);
      }
      if (drop) {  // the forward mask of element (row0 + r, gcol + 4 c4g): two aligned pairs
        const uint64_t pr = ((static_cast<uint64_t>(p0.row0) * E + gcol) >> 1) + ((r * static_cast<uint32_t>(E) + c4g * 4) >> 1);
        const uint32_t h0 = ebn_dropout_pair_hash(key, pr), h1 = ebn_dropout_pair_hash(key, pr + 1);
        vg[t].x *= ((h0 & 0xFFFFu) >= a.thresh) ? a.scale : 0.f;
        vg[t].y *= ((h0 >> 16) >= a.thresh) ? a.scale : 0.f;
        vg[t].z *= ((h1 & 0xFFFFu) >= a.thresh) ? a.scale : 0.f;
        vg[t].w *= ((h1 >> 16) >= a.thresh) ? a.scale : 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < ROUNDS; ++t) {
      if (tid + NT * t < nvec) {
        *reinterpret_cast<float4*>(smem + dst[t]) = vq[t];
        *reinterpret_cast<float4*>(smem + dst[t] + region) = vk[t];
        *reinterpret_cast<float4*>(smem + dst[t] + g_off) = vg[t];
      }
    }
  }
  __syncthreads();

  float* sq = smem + wv * wave_floats;
  float* sk = sq + region;
  float* sg = sq + g_off;
  f32x16 P, dP;
  {
    float qr[KH], kr[KH];
    lds_row_form<D>(qr, sq, L, row, hi);
    lds_row_form<D>(kr, sk, L, row, hi);
    P = mm_rows<KH>(kr, qr);  // T[j][i]: lane i, regs j
  }
  softmax_in_lane(P, L, hi, inv2);  // P[i][j]: lane i, regs j
  {
    float gr[KH];
    lds_row_form<D>(gr, sg, L, row, hi);
    dP = mm_rows<KH>(gr, vr);  // dP[i][j] = V[i].dO[j]: lane i, regs j
  }
  float rowdot = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) rowdot = fmaf(P[r], dP[r], rowdot);
  rowdot += __shfl_xor(rowdot, 32, 64);

  float col[16], colq[16];
  lds_col_form<D>(col, sg, L, row, hi);
  lds_col_form<D>(colq, sq, L, row, hi);
  wave_lds_sync();  // dO's row and column forms are in registers: its tile takes d(V)
  {  // dV^T[c][i] = sum_j dO[j][c] P[i][j]
    const f32x16 dV = mm_col_tile(col, P);
    tile_rows_to_lds<D>(sg, dV, L, row, hi, 1.0f);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) P[r] = P[r] * (dP[r] - rowdot);  // dS[i][j]: lane i, regs j
  lds_col_form<D>(col, sk, L, row, hi);
  const f32x16 dQ = mm_col_tile(col, P);  // dQ^T[c][i] = inv * sum_j K[j][c] dS[i][j]; stays in registers over the transpose
  wave_lds_sync();                    // Q and K have been read for the last time: their tiles become the transpose buffer
  tile_transpose(P, sq, L, row, hi);  // dS[i][j]: lane j, regs i
  wave_lds_sync();
  tile_rows_to_lds<D>(sq, dQ, L, row, hi, inv);
  {  // dK^T[c][j] = inv * sum_i Q[i][c] dS[i][j]
    const f32x16 dK = mm_col_tile(colq, P);
    tile_rows_to_lds<D>(sk, dK, L, row, hi, inv);
  }
  __syncthreads();

  float* go = a.out + p0.row0 * a.ld_out + gcol;
#pragma unroll
  for (int t = 0; t < ROUNDS; ++t) {
    const int idx = tid + NT * t;
    if (idx < nvec) {
      const uint32_t r = static_cast<uint32_t>(idx) / static_cast<uint32_t>(GV), c4g = static_cast<uint32_t>(idx) - r * GV;
      const uint32_t head = c4g / T::VPR, c4 = c4g - head * T::VPR;
      const float* src = smem + head * wave_floats + r * T::STRIDE + c4 * 4;
      const float4 q4 = *reinterpret_cast<const float4*>(src);
      const float4 k4 = *reinterpret_cast<const float4*>(src + region);
      const float4 v4 = *reinterpret_cast<const float4*>(src + g_off);
      float* d = go + (r * static_cast<uint32_t>(a.ld_out) + c4g * 4);
      *reinterpret_cast<float4*>(d) = q4;
      *reinterpret_cast<float4*>(d + E) = k4;
      *reinterpret_cast<float4*>(d + 2 * E) = v4;
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
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t prob = static_cast<int64_t>(xcd_chunked_block()) * ATT2_WAVES + wv;
  if (prob >= a.n_prob) return;
  const int L = LC ? LC : a.L, E = a.h * D;
  const int region = L * T::STRIDE;
  float* sq = smem + wv * 3 * region;
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
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t prob = static_cast<int64_t>(xcd_chunked_block()) * ATT2_WAVES + wv;
  if (prob >= a.n_prob) return;
  const int L = LC ? LC : a.L, E = a.h * D;
  const int region = L * T::STRIDE;
  float* sq = smem + wv * 4 * region;
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

constexpr int BWD_GROUP = 4;  // heads per workgroup of the group-form backward: one wave per SIMD (5 or 10 measured slower)
constexpr int64_t FWD_GROUP_MIN_QKV_BYTES = 200'000'000;
constexpr int64_t BWD_GROUP_MIN_PROBLEMS = 4096;  // about one resident round of waves on 256 CUs
// The start-up stagger of the group backward de-phases the workgroups that share a CU: it needs the number of CUs the first
// dispatch round is dealt over (a power of two: the shift) -- read once from the device; any other topology runs without it.
static uint32_t stagger_shift_of_device() {
  static const uint32_t shift = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0u;
    if (cus < 16 || (cus & (cus - 1)) != 0) return 0u;
    uint32_t s = 0;
    while ((1 << s) < cus) ++s;
    return s;
  }();
  return shift;
}

static bool bwd_group_off() {  // EBN_ATTN_BWD_PER_WAVE=1: the one-wave-per-head backward everywhere (validation / tuning)
  static const bool off = [] { const char* e = getenv("EBN_ATTN_BWD_PER_WAVE"); return e && e[0] == '1'; }();
  return off;
}

// Returns 1 when the MFMA path handles (L, d, leading dims, alignment); the caller falls back otherwise.
static bool mfma_path_ok(int32_t L, int32_t d, int64_t lda, int64_t ldb, int64_t ldc, const void* p0,
                         const void* p1, const void* p2) {
  if (L > 64 || !(d == 16 || d == 20 || d == 32)) return false;  // L <= 32: one 32x32 tile; 32 < L <= 64: 2 x 2 tiles
  if ((lda % 4) || (ldb % 4) || (ldc % 4)) return false;
  if (lda >= (1 << 24) || ldb >= (1 << 24) || ldc >= (1 << 24)) return false;  // 32-bit lane offsets: 64 rows x ld
  return ebn_aligned16(p0) && ebn_aligned16(p1) && ebn_aligned16(p2);
}

// more than 64 KB of dynamic LDS per workgroup (d = 32, L = 32 backward: 72 KB) has to be granted per kernel
template <class K>
static void allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

template <class K>
static void launch_waves(K kernel, size_t lds, const MfmaAttnArgs& a, hipStream_t s) {
#ifdef EBN_ATTN_EXP_PAD_LDS  /* tuning experiment (tools/build_variant.sh): EBN_ATTN_PAD_LDS bytes of unused LDS per workgroup = fewer resident waves per CU */
  static const size_t pad = getenv("EBN_ATTN_PAD_LDS") ? static_cast<size_t>(atol(getenv("EBN_ATTN_PAD_LDS"))) : 0;
  lds += pad;
#endif
  allow_lds(kernel, lds);
  EBN_LAUNCH(kernel, dim3(static_cast<unsigned>(ebn_ceil_div(a.n_prob, ATT_WAVES))), dim3(64 * ATT_WAVES), lds, s, a);
}

template <int D, int LC>
static void launch_mfma2_lc(bool bwd, const MfmaAttnArgs& a, hipStream_t s) {
  const dim3 grid(static_cast<unsigned>(ebn_ceil_div(a.n_prob, ATT2_WAVES))), block(64 * ATT2_WAVES);
  const size_t lds = static_cast<size_t>(ATT2_WAVES) * (bwd ? 4 : 3) * a.L * Tile<D>::STRIDE * sizeof(float);
  if (bwd) {
    allow_lds(attn_mfma2_bwd_kernel<D, LC>, lds);
    EBN_LAUNCH((attn_mfma2_bwd_kernel<D, LC>), grid, block, lds, s, a);
  } else {
    allow_lds(attn_mfma2_fwd_kernel<D, LC>, lds);
    EBN_LAUNCH((attn_mfma2_fwd_kernel<D, LC>), grid, block, lds, s, a);
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
  *handled = mfma_path_ok(L, d, ld_qkv, ld_out, ld_out, qkv, out, out) && n_seq * h < (int64_t{1} << 31);
  if (!*handled) return EBN_OK;
  MfmaAttnArgs a{qkv, ld_qkv, nullptr, 0, out, ld_out, n_seq * h, L, h, dr.key_ptr, dr.thresh, dr.scale, nullptr, nullptr, 0, 0u, 0u, 1u};
  if (L > 32) {
    if (d == 16) launch_mfma2<16>(false, a, s);
    else if (d == 20) launch_mfma2<20>(false, a, s);
    else launch_mfma2<32>(false, a, s);
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  // Group form as in the backward, but only where Q|K|V no longer sits in the 256 MB memory-side cache the projection GEMM left it
  // in: per 800 titles 31.3 us against 39.2 us one wave per head on HBM-resident data (3200 titles), 36.5 against 32.5 us on
  // cache-resident data (800 titles; in the c2 step 35.5 against 34.8 us); the crossover lies between 172 and 230 MB of Q|K|V
  const bool big = a.n_prob * L * 3 * d * static_cast<int64_t>(sizeof(float)) >= FWD_GROUP_MIN_QKV_BYTES;
  if (d == 20 && (h % BWD_GROUP) == 0 && big && !bwd_group_off()) {
    const size_t lds = static_cast<size_t>(BWD_GROUP) * fwd_group_wave_floats<20>(L) * sizeof(float);
    const dim3 grid(static_cast<unsigned>(a.n_prob / BWD_GROUP)), block(64 * BWD_GROUP);
    if (L == 30) {
      allow_lds(attn_mfma_fwd_group_kernel<20, 30, BWD_GROUP>, lds);
      EBN_LAUNCH((attn_mfma_fwd_group_kernel<20, 30, BWD_GROUP>), grid, block, lds, s, a);
    } else {
      allow_lds(attn_mfma_fwd_group_kernel<20, 0, BWD_GROUP>, lds);
      EBN_LAUNCH((attn_mfma_fwd_group_kernel<20, 0, BWD_GROUP>), grid, block, lds, s, a);
    }
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  const size_t wb = static_cast<size_t>(ATT_WAVES) * sizeof(float);
  if (d == 16) launch_waves(attn_mfma_fwd_kernel<16, 0>, wb * fwd_wave_floats<16>(L), a, s);
  else if (d == 20 && L == 30) launch_waves(attn_mfma_fwd_kernel<20, 30>, wb * fwd_wave_floats<20>(L), a, s);
  else if (d == 20 && L == 20) launch_waves(attn_mfma_fwd_kernel<20, 20>, wb * fwd_wave_floats<20>(L), a, s);
  else if (d == 20) launch_waves(attn_mfma_fwd_kernel<20, 0>, wb * fwd_wave_floats<20>(L), a, s);
  else launch_waves(attn_mfma_fwd_kernel<32, 0>, wb * fwd_wave_floats<32>(L), a, s);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

int ebn_attn_mfma_bwd(const float* qkv, int64_t ld_qkv, const float* dout, int64_t ld_dout, float* dqkv,
                      int64_t ld_dqkv, int64_t n_seq, int32_t L, int32_t h, int32_t d, const EbnDrop& dr,
                      hipStream_t s, bool* handled, const float* pool_w, const float* pool_d, int64_t ld_pool) {
  *handled = mfma_path_ok(L, d, ld_qkv, ld_dout, ld_dqkv, qkv, dout, dqkv) && n_seq * h < (int64_t{1} << 31) &&
             (pool_w == nullptr || ((ld_pool % 4) == 0 && ebn_aligned16(pool_d)));
  if (!*handled) return EBN_OK;
  MfmaAttnArgs a{qkv, ld_qkv, dout, ld_dout, dqkv, ld_dqkv, n_seq * h, L, h, dr.key_ptr, dr.thresh, dr.scale, pool_w, pool_d, ld_pool, 0u, 0u, 1u};
  if (L > 32) {
    if (d == 16) launch_mfma2<16>(true, a, s);
    else if (d == 20) launch_mfma2<20>(true, a, s);
    else launch_mfma2<32>(true, a, s);
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  // G heads of a sequence per workgroup, cooperative row traffic -- where the launch is memory-bound: the 640 problems of the
  // user-level attention (one round of waves, latency-bound) measured 8.2 us in this form against 6.6 us one wave per head
  if (d == 20 && (h % BWD_GROUP) == 0 && a.n_prob >= BWD_GROUP_MIN_PROBLEMS && !bwd_group_off()) {
    const size_t lds = static_cast<size_t>(BWD_GROUP) * bwd_group_wave_floats<20>(L) * sizeof(float);
    const dim3 grid(static_cast<unsigned>(a.n_prob / BWD_GROUP)), block(64 * BWD_GROUP);
    a.stagger_shift = stagger_shift_of_device();  // 8 on the 256 CUs of an MI355X; 0 = unknown topology: no stagger
    a.stagger_units = a.stagger_shift != 0u ? 7u : 0u;  // x 512 cycles per step
    a.stagger_mod = 5u;  // workgroups per CU: 28.8 KB of LDS each
    if (L == 30) {
      allow_lds(attn_mfma_bwd_group_kernel<20, 30, BWD_GROUP>, lds);
      EBN_LAUNCH((attn_mfma_bwd_group_kernel<20, 30, BWD_GROUP>), grid, block, lds, s, a);
    } else {
      allow_lds(attn_mfma_bwd_group_kernel<20, 0, BWD_GROUP>, lds);
      EBN_LAUNCH((attn_mfma_bwd_group_kernel<20, 0, BWD_GROUP>), grid, block, lds, s, a);
    }
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  const size_t wb = static_cast<size_t>(ATT_WAVES) * sizeof(float);
  if (d == 16) launch_waves(attn_mfma_bwd_kernel<16, 0>, wb * bwd_wave_floats<16>(L), a, s);
  else if (d == 20 && L == 30) launch_waves(attn_mfma_bwd_kernel<20, 30>, wb * bwd_wave_floats<20>(L), a, s);
  else if (d == 20 && L == 20) launch_waves(attn_mfma_bwd_kernel<20, 20>, wb * bwd_wave_floats<20>(L), a, s);
  else if (d == 20) launch_waves(attn_mfma_bwd_kernel<20, 0>, wb * bwd_wave_floats<20>(L), a, s);
  else launch_waves(attn_mfma_bwd_kernel<32, 0>, wb * bwd_wave_floats<32>(L), a, s);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
