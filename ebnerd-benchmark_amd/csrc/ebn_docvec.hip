// a11, fused: the training step of the NRMSDocVec news encoder (nrms_docvec.py:113-135)
//
//     x -> [Dense(u, relu, l2) -> BatchNormalization -> Dropout] x n -> Dense(E, relu)
//
// over the B*(H+C) <= 2048 document vectors of a step, as ONE launch per Dense layer and direction.  The separate passes of
// ebn_dense.hip (BatchNormalization forward / backward, ReLU backward: one 16-column strip per workgroup, 32 workgroups on a
// 256-CU chip, 6-11 us each) disappear into the matmuls on either side of them:
//
//   * the matmul that PRODUCES a layer's relu output also adds, per 32-row tile and column, the tile's sum and its sum of
//     squares into per-site FIXED-POINT accumulators: 64-bit integer atomics, so the totals do
//     not depend on the order in which the tiles arrive -- bitwise reproducible like the embedding-gradient accumulator of
//     ebn_embedding.hip (one atomic per tile, column and sum: a few ten thousand per launch);
//   * the matmul that CONSUMES the normalised activations turns the accumulators into per-site batch statistics in its
//     prologue (float64 for E[x^2] - mean^2) -- every workgroup redundantly, so there is no cross-workgroup dependency inside
//     a launch -- and applies gamma * (x - mean) * istd + beta and the dropout mask to its A operand on the way from
//     registers to LDS.  The workgroups of a row tile also write the transformed operand out once, slab by slab in turns
//     (it is the A operand of the weight gradient in backward); workgroup (0, 0) writes the statistics and updates the
//     moving averages (history site first, then the candidate site: two TimeDistributed call sites, nrms_docvec.py:88-90,
//     176-178).
//   * backward is the mirror image: d(pre-activation) = relu' * BatchNorm-backward(dropout-backward(dXn)) is formed on the A
//     operand of dXn_{l-1} = dP_l . W_l^T from the masked gradient dy_l the previous launch stored and the two column sums
//     (sum dy, sum dy * xhat) it accumulated the same way.
// The accumulators are re-zeroed inside the step: the backward ones by the first forward launch, the forward ones by the last
// backward launch (a step that ran only half way leaves them dirty: ebn_dvn_fwd_train_f32 documents the contract).
//
// Tile: 32 rows x 64 columns x 128-deep slabs, 8 waves (2 row blocks x 2 column halves x 2 halves of each slab's k range, two
// 16x16 MFMA blocks per wave, two accumulators per block; the k halves are added through LDS in a fixed order), FOUR slabs of
// loads in flight (four register sets, two LDS buffers).  The LDS image (row stride 136, 3-bit column swizzle) is the one of
// gemm_small_vec_kernel (ebn_gemm.hip).  Row tiles never straddle the two call sites (each site is tiled on its own), so a tile
// has ONE set of statistics.
#include "ebn_common.h"
#include "ebn_tn_finale.h"

typedef float ebn_f32x4 __attribute__((ext_vector_type(4)));
typedef int ebn_i32x4 __attribute__((ext_vector_type(4)));
__device__ ebn_f32x4 ebn_raw_buffer_load_x4(ebn_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float BN_EPS = 1e-3f;
constexpr float BN_MOM = 0.99f;

constexpr int TM = 32, TN = 64, TK = 128, LD = TK + 8, NTHR = 512, NSET = 4;
constexpr int SA = TM * TK / 4 / NTHR;  // float4 per thread and slab of the A tile (2)
constexpr int SB = TN * TK / 4 / NTHR;  // ... of the B tile (4)
constexpr int TILE_A = TM * LD, TILE_B = TN * LD;
constexpr int MAX_K = 1024;     // widest layer whose per-column constants fit next to the tiles in LDS
constexpr int MAX_TILES = 4096;  // row tiles of both sites together (131 072 rows: only the launch geometry bounds it)
// fixed-point scales of the accumulators: forward sums of relu outputs / of their squares, backward sums of gradients.  Ranges: |sum x| <
// 2^31, sum x^2 < 2^35, |sum dy| < 2^23 per call site and column -- orders of magnitude beyond a model that has not diverged.  A tile sum
// outside the range, or not finite, is NOT added: it raises args->range_flag (sticky; fix_addend below) and the host layer raises
// FloatingPointError at its next flag check -- the statistics never silently wrap
constexpr float FIX_SUM = 4294967296.0f /* 2^32 */, FIX_SQ = 268435456.0f /* 2^28 */, FIX_GRAD = 1099511627776.0f /* 2^40 */;
constexpr int L2_SLOTS = MAX_K / 64;  // column tiles of the widest regularised kernel

__device__ __forceinline__ ebn_i32x4 make_rsrc(const float* base) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  ebn_i32x4 r;
  r.x = static_cast<int>(static_cast<uint32_t>(a));
  r.y = static_cast<int>(static_cast<uint32_t>(a >> 32) & 0xFFFFu);
  r.z = -1;
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ float4 bload4(ebn_i32x4 r, uint32_t lane_bytes, uint32_t slab_bytes) {
  const ebn_f32x4 t = ebn_raw_buffer_load_x4(r, static_cast<int>(lane_bytes), static_cast<int>(slab_bytes), 0);
  return make_float4(t.x, t.y, t.z, t.w);
}

enum { AX_PLAIN = 0, AX_BN = 1, AX_RELU = 2, AX_DBN = 3 };
enum { EPI_RELU = 0, EPI_RELU_STATS = 1, EPI_DY = 2 };

struct PanelArgs {
  int n0, n1;          // rows of the two call sites
  int K, Nout;         // contraction length, output width
  const float* A;      // AX_PLAIN: the operand; AX_BN: relu output R of the layer in front; AX_RELU: d(output); AX_DBN: dy
  const float* A2;     // AX_RELU: the relu output the gradient is masked with; AX_DBN: R of the same layer
  const float* B;      // B_KC ? [Nout][K] : [K][Nout]
  int ldb;
  float* C;            // (N, Nout)
  float* Aout;         // transformed A operand, written out once (slab kt by column tile kt mod #column tiles); may be null
  const long long* in_acc;  // accumulators of the A-side layer: [2 sums][2 sites][K]
  const float* gamma;  // of the A-side BatchNormalization
  const float* beta;
  float* mean_io;      // [2][K] per-site batch mean (AX_BN writes, AX_DBN reads)
  float* istd_io;
  float* mmean;        // moving statistics (AX_BN, workgroup (0,0))
  float* mvar;
  float* ggamma;       // AX_DBN, workgroup (0,0)
  float* gbeta;
  const uint32_t* key_in;  // dropout key of the A-side layer (AX_BN); null = no dropout
  uint32_t thresh;
  float scale;
  const float* bias;       // EPI_RELU*
  long long* out_acc;      // EPI_RELU_STATS / EPI_DY: [2 sums][2 sites][Nout]
  const float* Rout;       // EPI_DY: relu output of the layer whose dy this launch produces
  const float* mean_out;   // EPI_DY: [2][Nout]
  const float* istd_out;
  const uint32_t* key_out; // EPI_DY: dropout key of that layer; null = no dropout
  float* l2_part;          // forward of a regularised Dense: sum of squares of the weight columns of column tile tx (may be null)
  float* zero;             // a span this launch re-zeroes for later launches of the step (nobody reads it meanwhile); may be null
  int zero_n;
  int32_t* range_flag;     // set when a tile's column sum leaves the accumulators' range or is not finite (may be null)
};

// One tile's contribution to a fixed-point accumulator: the scaled sum as a 64-bit integer -- or 0 and the sticky range flag when it is
// not finite or so large that `tiles` such addends could wrap the 64-bit total (the run has diverged: the statistics formed from the
// integer sums would be garbage where the separate-pass kernels propagate NaN; the host layer raises FloatingPointError on the flag).
__device__ __forceinline__ unsigned long long fix_addend(float scaled, float limit, int32_t* flag) {
  if (!(fabsf(scaled) < limit)) {
    if (flag != nullptr) *flag = 1;
    return 0ull;
  }
  return static_cast<unsigned long long>(__float2ll_rn(scaled));
}

// swizzled column of the LDS image: whole float4s move, by the row's (row >> 2) & 7 (tools/lds/bank_sim.py)
__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 7) << 2; }

// keep decisions of the four consecutive elements idx .. idx + 3 (idx % 4 == 0): two pair hashes
__device__ __forceinline__ void keep4(uint32_t key, uint32_t idx, uint32_t thresh, bool (&k)[4]) {
  const uint32_t h0 = ebn_dropout_pair_hash(key, static_cast<uint64_t>(idx >> 1));
  const uint32_t h1 = ebn_dropout_pair_hash(key, static_cast<uint64_t>((idx >> 1) + 1u));
  k[0] = (h0 & 0xFFFFu) >= thresh;
  k[1] = (h0 >> 16) >= thresh;
  k[2] = (h1 & 0xFFFFu) >= thresh;
  k[3] = (h1 >> 16) >= thresh;
}

template <int AX, bool B_KC, int EPI>
__global__ __launch_bounds__(NTHR) void dvn_panel_kernel(PanelArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][A tile | B tile] | constants [3][Kpad] | reduction scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = (wave >> 1) & 1, kg = wave >> 2;  // row block, column half, half of every slab's k range
  const int tiles0 = (p.n0 + TM - 1) / TM;
  const int ty = blockIdx.y, tx = blockIdx.x;
  const int site = ty >= tiles0 ? 1 : 0;
  const int row0 = site ? p.n0 + (ty - tiles0) * TM : ty * TM;
  const int row_end = site ? p.n0 + p.n1 : p.n0;
  const int K = p.K, Nout = p.Nout, n0c = tx * TN;
  const int nk = (K + TK - 1) / TK, nk_full = K / TK;
  const int Kpad = nk * TK;
  float* cst = smem + 2 * (TILE_A + TILE_B);
  float* red = cst + ((AX == AX_BN || AX == AX_DBN) ? 3 * Kpad : 0);  // [4][64]

  // ---- lane offsets of the staged pieces ---------------------------------------------------------------------------
  const int k4 = (tid & 31) * 4;  // A (and a k-contiguous B): the thread's k offset inside a slab
  uint32_t oa[SA], ob[SB];
  int arow[SA];
#pragma unroll
  for (int i = 0; i < SA; ++i) {
    const int rl = (tid >> 5) + (NTHR / 32) * i;
    int rg = row0 + rl;
    rg = rg < row_end ? rg : row_end - 1;
    arow[i] = row0 + rl;
    oa[i] = static_cast<uint32_t>(((rg - row0) * K + k4) * 4);
  }
#pragma unroll
  for (int i = 0; i < SB; ++i) {
    if (B_KC) {
      int n = n0c + (tid >> 5) + (NTHR / 32) * i;
      n = n < Nout ? n : Nout - 1;
      ob[i] = static_cast<uint32_t>(((n - n0c) * p.ldb + k4) * 4);
    } else {
      int n4 = n0c + (tid & 15) * 4;
      n4 = n4 < Nout ? n4 : Nout - 4;
      ob[i] = static_cast<uint32_t>((((tid >> 4) + (NTHR / 16) * i) * p.ldb + (n4 - n0c)) * 4);
    }
  }
  const ebn_i32x4 arsrc = make_rsrc(p.A + static_cast<int64_t>(row0) * K);
  const ebn_i32x4 a2rsrc = make_rsrc((AX == AX_RELU || AX == AX_DBN) ? p.A2 + static_cast<int64_t>(row0) * K : p.A);
  const ebn_i32x4 brsrc = make_rsrc(B_KC ? p.B + static_cast<int64_t>(n0c) * p.ldb : p.B + n0c);
  const uint32_t step_a = TK * 4, step_b = static_cast<uint32_t>((B_KC ? TK : TK * p.ldb) * 4);
  constexpr bool TWO_A = (AX == AX_RELU || AX == AX_DBN);

  // FOUR register sets: the loads of slab kt + 4 are requested before slab kt is multiplied.  The operands of a step were written
  // by the launch in front, on other XCDs: they come from the memory-side cache, two microseconds away, while a slab is multiplied
  // in less than one.
  float4 ra[NSET][SA], ra2[NSET][TWO_A ? SA : 1], rb[NSET][SB];
#define DVN_LOAD(SET, KT)                                                                                  \
  do {                                                                                                     \
    const uint32_t sa__ = static_cast<uint32_t>(KT) * step_a, sb__ = static_cast<uint32_t>(KT) * step_b;   \
    if ((KT) < nk_full) {                                                                                  \
      _Pragma("unroll") for (int i = 0; i < SA; ++i) ra[SET][i] = bload4(arsrc, oa[i], sa__);              \
      if (TWO_A) { _Pragma("unroll") for (int i = 0; i < SA; ++i) ra2[SET][i] = bload4(a2rsrc, oa[i], sa__); } \
      _Pragma("unroll") for (int i = 0; i < SB; ++i) rb[SET][i] = bload4(brsrc, ob[i], sb__);              \
    } else { /* partial last slab: pieces at k >= K read offset 0 and become zero */                      \
      const int krem__ = K - (KT) * TK;                                                                    \
      const bool oka__ = k4 < krem__;                                                                      \
      _Pragma("unroll") for (int i = 0; i < SA; ++i) {                                                     \
        const float4 t__ = bload4(arsrc, oka__ ? oa[i] + sa__ : 0u, 0u);                                   \
        ra[SET][i] = make_float4(oka__ ? t__.x : 0.f, oka__ ? t__.y : 0.f, oka__ ? t__.z : 0.f, oka__ ? t__.w : 0.f); \
        if (TWO_A) {                                                                                       \
          const float4 u__ = bload4(a2rsrc, oka__ ? oa[i] + sa__ : 0u, 0u);                                \
          ra2[SET][i] = make_float4(oka__ ? u__.x : 0.f, oka__ ? u__.y : 0.f, oka__ ? u__.z : 0.f, oka__ ? u__.w : 0.f); \
        }                                                                                                  \
      }                                                                                                    \
      _Pragma("unroll") for (int i = 0; i < SB; ++i) {                                                     \
        const bool okb__ = (B_KC ? k4 : (tid >> 4) + (NTHR / 16) * i) < krem__;                            \
        const float4 t__ = bload4(brsrc, okb__ ? ob[i] + sb__ : 0u, 0u);                                   \
        rb[SET][i] = make_float4(okb__ ? t__.x : 0.f, okb__ ? t__.y : 0.f, okb__ ? t__.z : 0.f, okb__ ? t__.w : 0.f); \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)

  // ---- per-column constants of the A transform (every workgroup, redundantly: no cross-workgroup dependency).  Their inputs are
  // requested BEFORE the tile pieces: loads return in order, and behind four slabs of tile data they would arrive last -----------
  constexpr bool CST = (AX == AX_BN || AX == AX_DBN);
  constexpr int CPT = CST ? MAX_K / NTHR : 1;  // columns per thread
  const uint32_t key_in = (AX == AX_BN && p.key_in != nullptr) ? *p.key_in : 0u;
  const bool drop_in = (AX == AX_BN) && p.key_in != nullptr;
  long long qa[CPT][4];
  float qg[CPT], qb[CPT], qm[CPT], qi[CPT];
  if (CST) {
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int c = tid + j * NTHR;
      const int cc = c < K ? c : K - 1;
      // [sum kind][site][K]: all four words of the column (workgroup (0,0) needs both sites, the others only their own)
      qa[j][0] = p.in_acc[cc];
      qa[j][1] = p.in_acc[K + cc];
      qa[j][2] = p.in_acc[2 * K + cc];
      qa[j][3] = p.in_acc[3 * K + cc];
      qg[j] = p.gamma[cc];
      qb[j] = (AX == AX_BN) ? p.beta[cc] : 0.f;
      qm[j] = (AX == AX_DBN) ? p.mean_io[site * K + cc] : 0.f;
      qi[j] = (AX == AX_DBN) ? p.istd_io[site * K + cc] : 0.f;
    }
  }

  DVN_LOAD(0, 0);
  if (nk > 1) DVN_LOAD(1, 1);
  if (nk > 2) DVN_LOAD(2, 2);
  if (nk > 3) DVN_LOAD(3, 3);

  if (CST) {
    const bool first_wg = (tx == 0 && ty == 0);
    const float n_s[2] = {static_cast<float>(p.n0), static_cast<float>(p.n1)};
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int c = tid + j * NTHR;
      if (c >= Kpad) break;
      float c0 = 0.f, c1 = 0.f, c2 = 0.f;
      const long long a00 = qa[j][0], a01 = qa[j][1], a10 = qa[j][2], a11 = qa[j][3];
      if (AX == AX_BN) {
        float mean[2], var[2], istd[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const double inv_n = n_s[s] > 0.f ? 1.0 / static_cast<double>(n_s[s]) : 0.0;
          const double m = static_cast<double>(s ? a01 : a00) * (1.0 / static_cast<double>(FIX_SUM)) * inv_n;
          const double e2 = static_cast<double>(s ? a11 : a10) * (1.0 / static_cast<double>(FIX_SQ)) * inv_n;
          const double v = e2 - m * m;  // biased variance
          mean[s] = static_cast<float>(m);
          var[s] = static_cast<float>(v > 0.0 ? v : 0.0);
          istd[s] = 1.0f / sqrtf(var[s] + BN_EPS);
        }
        const float gs = qg[j] * istd[site];
        c0 = gs * p.scale;
        c1 = (qb[j] - mean[site] * gs) * p.scale;
        if (first_wg && c < K) {
          float mm = p.mmean[c], mv = p.mvar[c];  // one moving-average update per call site, history first
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (n_s[s] > 0.f) {
              mm = mm * BN_MOM + mean[s] * (1.0f - BN_MOM);
              mv = mv * BN_MOM + var[s] * (1.0f - BN_MOM);
              p.mean_io[s * K + c] = mean[s];
              p.istd_io[s * K + c] = istd[s];
            }
          }
          p.mmean[c] = mm;
          p.mvar[c] = mv;
        }
      } else {
        const float sd[2] = {static_cast<float>(static_cast<double>(a00) * (1.0 / static_cast<double>(FIX_GRAD))),
                             static_cast<float>(static_cast<double>(a01) * (1.0 / static_cast<double>(FIX_GRAD)))};
        const float sx[2] = {static_cast<float>(static_cast<double>(a10) * (1.0 / static_cast<double>(FIX_GRAD))),
                             static_cast<float>(static_cast<double>(a11) * (1.0 / static_cast<double>(FIX_GRAD)))};
        const float inv = n_s[site] > 0.f ? 1.0f / n_s[site] : 0.f;
        const float s1 = sd[site] * inv, s2 = sx[site] * inv;
        const float istd = qi[j], mean = qm[j];
        const float k = qg[j] * istd;
        c0 = k;
        c1 = k * s2 * istd;                    // multiplies R
        c2 = k * s2 * (mean * istd) - k * s1;  // constant term
        if (first_wg && c < K) {
          p.ggamma[c] = sx[0] + sx[1];
          p.gbeta[c] = sd[0] + sd[1];
        }
      }
      const bool in = c < K;
      cst[c] = in ? c0 : 0.f;
      cst[Kpad + c] = in ? c1 : 0.f;
      cst[2 * Kpad + c] = in ? c2 : 0.f;
    }
    __syncthreads();
  }
  if (p.zero != nullptr) {  // a contiguous span, dealt to the workgroups of the launch
    const int nwg = gridDim.x * gridDim.y, per = (p.zero_n + nwg - 1) / nwg;
    const int z0 = (ty * gridDim.x + tx) * per;
    for (int i = z0 + tid; i < z0 + per && i < p.zero_n; i += NTHR) p.zero[i] = 0.f;
  }

  // ---- registers -> (transform) -> LDS ---------------------------------------------------------------------------------
  // the transformed operand is written out ONCE, slab kt by the workgroup of column tile kt mod (column tiles): spread over the row
  // tile's workgroups instead of making the first column tile's the launch's stragglers
  const bool side = p.Aout != nullptr;
  const int ntx = static_cast<int>(gridDim.x);
  // kernel_regularizer=l2: the first row of tiles adds up the squares of the weight pieces it stages anyway
  const bool l2_wg = !B_KC && p.l2_part != nullptr && ty == 0;
  const bool do_l2 = l2_wg && n0c + (tid & 15) * 4 < Nout;
  float l2acc = 0.f;
#define DVN_STORE(SET, BUF, KT)                                                                            \
  do {                                                                                                     \
    float* sa__ = smem + (BUF) * (TILE_A + TILE_B);                                                        \
    float* sb__ = sa__ + TILE_A;                                                                           \
    const int kk__ = (KT) * TK + k4;                                                                       \
    float4 cs0__ = make_float4(0.f, 0.f, 0.f, 0.f), cs1__ = cs0__, cs2__ = cs0__;                          \
    if (AX == AX_BN || AX == AX_DBN) {                                                                     \
      cs0__ = *reinterpret_cast<const float4*>(cst + kk__);                                                \
      cs1__ = *reinterpret_cast<const float4*>(cst + Kpad + kk__);                                         \
      if (AX == AX_DBN) cs2__ = *reinterpret_cast<const float4*>(cst + 2 * Kpad + kk__);                   \
    }                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < SA; ++i) {                                                       \
      float4 v__ = ra[SET][i];                                                                             \
      if (AX == AX_BN) {                                                                                   \
        v__ = make_float4(fmaf(v__.x, cs0__.x, cs1__.x), fmaf(v__.y, cs0__.y, cs1__.y), fmaf(v__.z, cs0__.z, cs1__.z), \
                          fmaf(v__.w, cs0__.w, cs1__.w));                                                  \
        if (drop_in) {                                                                                     \
          bool kp__[4];                                                                                    \
          keep4(key_in, static_cast<uint32_t>(arow[i]) * static_cast<uint32_t>(K) + static_cast<uint32_t>(kk__), p.thresh, kp__); \
          v__ = make_float4(kp__[0] ? v__.x : 0.f, kp__[1] ? v__.y : 0.f, kp__[2] ? v__.z : 0.f, kp__[3] ? v__.w : 0.f); \
        }                                                                                                  \
      } else if (AX == AX_RELU) {                                                                          \
        const float4 y__ = ra2[SET][i];                                                                    \
        v__ = make_float4(y__.x > 0.f ? v__.x : 0.f, y__.y > 0.f ? v__.y : 0.f, y__.z > 0.f ? v__.z : 0.f, y__.w > 0.f ? v__.w : 0.f); \
      } else if (AX == AX_DBN) {                                                                           \
        const float4 y__ = ra2[SET][i];                                                                    \
        v__ = make_float4(y__.x > 0.f ? fmaf(v__.x, cs0__.x, fmaf(-y__.x, cs1__.x, cs2__.x)) : 0.f,         \
                          y__.y > 0.f ? fmaf(v__.y, cs0__.y, fmaf(-y__.y, cs1__.y, cs2__.y)) : 0.f,         \
                          y__.z > 0.f ? fmaf(v__.z, cs0__.z, fmaf(-y__.z, cs1__.z, cs2__.z)) : 0.f,         \
                          y__.w > 0.f ? fmaf(v__.w, cs0__.w, fmaf(-y__.w, cs1__.w, cs2__.w)) : 0.f);        \
      }                                                                                                    \
      const int rl__ = (tid >> 5) + (NTHR / 32) * i;                                                       \
      *reinterpret_cast<float4*>(&sa__[rl__ * LD + (k4 ^ swz(rl__))]) = v__;                               \
      if (side && (KT) % ntx == tx && arow[i] < row_end && kk__ < K)                                       \
        *reinterpret_cast<float4*>(p.Aout + static_cast<int64_t>(arow[i]) * K + kk__) = v__;               \
    }                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < SB; ++i) {                                                       \
      if (B_KC) {                                                                                          \
        const int nl__ = (tid >> 5) + (NTHR / 32) * i;                                                     \
        *reinterpret_cast<float4*>(&sb__[nl__ * LD + (k4 ^ swz(nl__))]) = rb[SET][i];                      \
      } else {                                                                                             \
        const int n4__ = (tid & 15) * 4;                                                                   \
        const int kc__ = ((tid >> 4) + (NTHR / 16) * i) ^ swz(n4__);                                       \
        sb__[(n4__ + 0) * LD + kc__] = rb[SET][i].x;                                                       \
        sb__[(n4__ + 1) * LD + kc__] = rb[SET][i].y;                                                       \
        sb__[(n4__ + 2) * LD + kc__] = rb[SET][i].z;                                                       \
        sb__[(n4__ + 3) * LD + kc__] = rb[SET][i].w;                                                       \
        if (do_l2)                                                                                         \
          l2acc = fmaf(rb[SET][i].x, rb[SET][i].x, fmaf(rb[SET][i].y, rb[SET][i].y,                        \
                       fmaf(rb[SET][i].z, rb[SET][i].z, fmaf(rb[SET][i].w, rb[SET][i].w, l2acc))));        \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)

  // ---- multiply one slab: a wave owns 16 rows x 32 columns (two MFMA blocks sharing the A fragment) and HALF of the slab's
  // k range (32-deep groups 2 kg, 2 kg + 1): two waves per SIMD, the other one's MFMAs cover this one's LDS and barrier waits ----
  f32x4 acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc10 = acc00, acc11 = acc00;
  const int r16 = lane & 15, kq = lane >> 4;
  const int ar = wm * 16 + r16, br0 = wn * 32 + r16, br1 = br0 + 16;
  const int ca0 = (4 * kq) ^ swz(ar), ca1 = ca0 ^ 16;
  const int cb00 = (4 * kq) ^ swz(br0), cb01 = cb00 ^ 16;
  const int cb10 = (4 * kq) ^ swz(br1), cb11 = cb10 ^ 16;
#define DVN_READ(SET, g)                                                        \
  do {                                                                          \
    fa__[SET][0] = *reinterpret_cast<const float4*>(ap__ + 32 * (g) + ca0);     \
    fa__[SET][1] = *reinterpret_cast<const float4*>(ap__ + 32 * (g) + ca1);     \
    fb0__[SET][0] = *reinterpret_cast<const float4*>(bp0__ + 32 * (g) + cb00);  \
    fb0__[SET][1] = *reinterpret_cast<const float4*>(bp0__ + 32 * (g) + cb01);  \
    fb1__[SET][0] = *reinterpret_cast<const float4*>(bp1__ + 32 * (g) + cb10);  \
    fb1__[SET][1] = *reinterpret_cast<const float4*>(bp1__ + 32 * (g) + cb11);  \
  } while (0)
// -DEBN_DVN_ALT_ORDER (csrc/variants/dvn_alt_order.so, a TEST build only): every product of a block goes into ONE accumulator
// chain instead of two alternating ones -- a different, equally correct summation order of the same matmul.  tests/
// test_full_size_parity.py runs the full-size NRMSDocVec parity test against both builds: the comparison with the oracle must not
// depend on which side of 0 a ReLU input's rounding falls (verdict r5 item 3).
#ifdef EBN_DVN_ALT_ORDER
#define DVN_ACC01 acc00
#define DVN_ACC11 acc10
#else
#define DVN_ACC01 acc01
#define DVN_ACC11 acc11
#endif
#define DVN_MUL(SET)                                                                                   \
  do {                                                                                                 \
    acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].x, fb0__[SET][0].x, acc00, 0, 0, 0);     \
    acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].x, fb1__[SET][0].x, acc10, 0, 0, 0);     \
    DVN_ACC01 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].y, fb0__[SET][0].y, DVN_ACC01, 0, 0, 0); \
    DVN_ACC11 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].y, fb1__[SET][0].y, DVN_ACC11, 0, 0, 0); \
    acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].z, fb0__[SET][0].z, acc00, 0, 0, 0);     \
    acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].z, fb1__[SET][0].z, acc10, 0, 0, 0);     \
    DVN_ACC01 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].w, fb0__[SET][0].w, DVN_ACC01, 0, 0, 0); \
    DVN_ACC11 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].w, fb1__[SET][0].w, DVN_ACC11, 0, 0, 0); \
    acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].x, fb0__[SET][1].x, acc00, 0, 0, 0);     \
    acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].x, fb1__[SET][1].x, acc10, 0, 0, 0);     \
    DVN_ACC01 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].y, fb0__[SET][1].y, DVN_ACC01, 0, 0, 0); \
    DVN_ACC11 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].y, fb1__[SET][1].y, DVN_ACC11, 0, 0, 0); \
    acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].z, fb0__[SET][1].z, acc00, 0, 0, 0);     \
    acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].z, fb1__[SET][1].z, acc10, 0, 0, 0);     \
    DVN_ACC01 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].w, fb0__[SET][1].w, DVN_ACC01, 0, 0, 0); \
    DVN_ACC11 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].w, fb1__[SET][1].w, DVN_ACC11, 0, 0, 0); \
  } while (0)
#define DVN_MMA(BUF, KT)                                                                                   \
  do {                                                                                                     \
    const float* ap__ = smem + (BUF) * (TILE_A + TILE_B) + ar * LD;                                        \
    const float* bp0__ = smem + (BUF) * (TILE_A + TILE_B) + TILE_A + br0 * LD;                             \
    const float* bp1__ = smem + (BUF) * (TILE_A + TILE_B) + TILE_A + br1 * LD;                             \
    float4 fa__[2][2], fb0__[2][2], fb1__[2][2];                                                           \
    const int groups__ = (KT) < nk_full ? 4 : (K - (KT) * TK + 31) / 32;                                   \
    if (2 * kg + 1 < groups__) {                                                                           \
      DVN_READ(0, 2 * kg);                                                                                 \
      DVN_READ(1, 2 * kg + 1);                                                                             \
      DVN_MUL(0);                                                                                          \
      DVN_MUL(1);                                                                                          \
    } else if (2 * kg < groups__) {                                                                        \
      DVN_READ(0, 2 * kg);                                                                                 \
      DVN_MUL(0);                                                                                          \
    }                                                                                                      \
  } while (0)

  // slab kt: multiply from LDS buffer kt & 1, then move slab kt + 1 (requested three steps ago) from its register set into the
  // other buffer and request slab kt + 4 into the set slab kt left; the loop is unrolled by the four register sets
  // the bias of the epilogue, requested before the slab loop (behind it: a memory round trip of its own).  (The relu outputs and statistics
  // the dy epilogue needs are NOT: eight more loads in front of the loop cost the backward launches what they gain -- measured.)
  float epre_a[2] = {0.f, 0.f};
  if (EPI != EPI_DY && kg == 0) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = n0c + wn * 32 + b * 16 + (lane & 15);
      epre_a[b] = p.bias[col < Nout ? col : Nout - 1];
    }
  }
  DVN_STORE(0, 0, 0);
  __syncthreads();
#define DVN_STEP(J)                                                              \
  {                                                                              \
    if (kt + (J) + 4 < nk) DVN_LOAD((J), kt + (J) + 4);                          \
    DVN_MMA((J) & 1, kt + (J));                                                  \
    if (kt + (J) + 1 < nk) DVN_STORE(((J) + 1) & 3, ((J) + 1) & 1, kt + (J) + 1); \
    __syncthreads();                                                             \
    if (kt + (J) + 1 >= nk) break;                                               \
  }
  for (int kt = 0; kt < nk; kt += NSET) {
    DVN_STEP(0)
    DVN_STEP(1)
    DVN_STEP(2)
    DVN_STEP(3)
  }
#undef DVN_STEP
#undef DVN_LOAD
#undef DVN_STORE
#undef DVN_MMA
#undef DVN_READ
#undef DVN_MUL

  if (l2_wg) {  // block-uniform
    const float w = ebn_wave_sum(l2acc);
    if (lane == 0) red[wave] = w;
    __syncthreads();
    if (tid == 0) p.l2_part[tx] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    __syncthreads();
  }
  // ---- the two k halves of a block meet in LDS (fixed order: half 0 + half 1); waves 0..3 run the epilogue -------------------
  f32x4 accb[2] = {acc00 + acc01, acc10 + acc11};
  {
    float* xs = smem;  // the tile buffers are free: the slab loop ends behind a barrier
    if (kg == 1) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[((wave & 3) * 8 + b * 4 + r) * 64 + lane] = accb[b][r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) accb[b][r] += xs[(wave * 8 + b * 4 + r) * 64 + lane];
    }
  }
  const bool epw = kg == 0;
  // ---- epilogue --------------------------------------------------------------------------------------------------------
  // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
  float val[2][4];
  float s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f};
  const uint32_t key_out = (EPI == EPI_DY && p.key_out != nullptr) ? *p.key_out : 0u;
  if (epw) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = n0c + wn * 32 + b * 16 + r16;
      const bool cok = col < Nout;
      const int colc = cok ? col : Nout - 1;
      float mi = 0.f, is = 0.f;
      if (EPI == EPI_DY) {
        is = p.istd_out[site * Nout + colc];
        mi = p.mean_out[site * Nout + colc] * is;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wm * 16 + 4 * kq + r;
        const bool ok = cok && row < row_end;
        float v = accb[b][r];
        if (EPI != EPI_DY) {
          v = fmaxf(v + epre_a[b], 0.f);
          if (ok) p.C[static_cast<int64_t>(row) * Nout + col] = v;
          val[b][r] = ok ? v : 0.f;
          s0[b] += val[b][r];
          s1[b] = fmaf(val[b][r], val[b][r], s1[b]);
        } else {
          const int64_t i = static_cast<int64_t>(ok ? row : row0) * Nout + colc;
          const float rv = p.Rout[i];
          if (p.key_out != nullptr) v = ebn_dropout_keep(key_out, static_cast<uint64_t>(i), p.thresh) ? v * p.scale : 0.f;
          if (ok) p.C[i] = v;
          val[b][r] = ok ? v : 0.f;
          s0[b] += val[b][r];
          s1[b] = fmaf(val[b][r], fmaf(rv, is, -mi), s1[b]);
        }
      }
    }
  }
  if (EPI == EPI_RELU) return;
  // column sums over the tile's 32 rows: in-lane (4 rows), across the four lane quarters, across the two row waves
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    s0[b] += __shfl_xor(s0[b], 16, 64);
    s0[b] += __shfl_xor(s0[b], 32, 64);
    s1[b] += __shfl_xor(s1[b], 16, 64);
    s1[b] += __shfl_xor(s1[b], 32, 64);
  }
  const int cl0 = wn * 32 + r16;  // local column of block 0 (block 1: + 16)
  if (epw && kq == 0) {
    red[wm * 64 + cl0] = s0[0];
    red[wm * 64 + cl0 + 16] = s0[1];
    red[128 + wm * 64 + cl0] = s1[0];
    red[128 + wm * 64 + cl0 + 16] = s1[1];
  }
  __syncthreads();
  const float fix_limit = 4.0e18f / static_cast<float>(gridDim.y);  // 2^62 / (row tiles of the launch): the totals cannot wrap
  if (EPI == EPI_DY) {
    if (tid < TN && n0c + tid < Nout) {
      long long* acc = p.out_acc + site * Nout + n0c + tid;
      atomicAdd(reinterpret_cast<unsigned long long*>(acc), fix_addend((red[tid] + red[64 + tid]) * FIX_GRAD, fix_limit, p.range_flag));
      atomicAdd(reinterpret_cast<unsigned long long*>(acc + 2 * Nout), fix_addend((red[128 + tid] + red[192 + tid]) * FIX_GRAD, fix_limit, p.range_flag));
    }
    return;
  }
  // EPI_RELU_STATS: the tile's sum and sum of squares (32 addends each in fp32; the totals are exact integer sums of these)
  if (tid < TN && n0c + tid < Nout) {
    long long* acc = p.out_acc + site * Nout + n0c + tid;
    atomicAdd(reinterpret_cast<unsigned long long*>(acc), fix_addend((red[tid] + red[64 + tid]) * FIX_SUM, fix_limit, p.range_flag));
    atomicAdd(reinterpret_cast<unsigned long long*>(acc + 2 * Nout), fix_addend((red[128 + tid] + red[192 + tid]) * FIX_SQ, fix_limit, p.range_flag));
  }
}

// d(pre-activation) of the FIRST Dense layer: nothing multiplies it by a weight matrix in this encoder (the document vectors
// are inputs), so it is materialised element-wise from dy_0, R_0 and the accumulated column sums.  One workgroup = 32 rows x
// 256 columns.  As the last launch of the step's news-encoder backward it also re-zeroes the FORWARD accumulators and adds the
// L2 penalty of the regularised kernels to the loss.
constexpr int ATHR = 256;
struct ApplyArgs {
  int n0, n1, C;
  const float* dy;
  const float* R;
  const long long* acc;  // [2 sums][2 sites][C]
  const float* gamma;
  const float* mean;  // [2][C]
  const float* istd;
  float* dP;
  float* ggamma;
  float* gbeta;
  const float* l2_part;  // [n_l2][L2_SLOTS] sums of squares left by the forward launches; n_l2 = 0: no regulariser
  int n_l2;
  int l2_tiles[EBN_DVN_MAX_LAYERS];
  float l2;
  float* loss;
  float* zero;
  int zero_n;
};
__global__ __launch_bounds__(ATHR) void dvn_dbn_apply_kernel(ApplyArgs p) {
  __shared__ __attribute__((aligned(16))) float cst[3][256];
  const int tid = threadIdx.x;
  const int tiles0 = (p.n0 + TM - 1) / TM;
  const int ty = blockIdx.y, c0 = blockIdx.x * 256;
  const int site = ty >= tiles0 ? 1 : 0;
  const int row0 = site ? p.n0 + (ty - tiles0) * TM : ty * TM;
  const int row_end = site ? p.n0 + p.n1 : p.n0;
  const int C = p.C, c = c0 + tid;
  // the element loads do not depend on the constants: request them first
  const int c4 = (tid & 63) * 4;
  const bool cok4 = c0 + c4 < C;
  float4 g[8], y[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {  // clamped rows / columns, unconditional
    int row = row0 + (tid >> 6) + 4 * j;
    row = row < row_end ? row : row_end - 1;
    const int64_t i = static_cast<int64_t>(row) * C + (cok4 ? c0 + c4 : 0);
    g[j] = *reinterpret_cast<const float4*>(p.dy + i);
    y[j] = *reinterpret_cast<const float4*>(p.R + i);
  }
  float k0 = 0.f, k1 = 0.f, k2 = 0.f;
  {
    const int cc = c < C ? c : C - 1;
    const long long a00 = p.acc[cc], a01 = p.acc[C + cc], a10 = p.acc[2 * C + cc], a11 = p.acc[3 * C + cc];
    const float sd[2] = {static_cast<float>(static_cast<double>(a00) * (1.0 / static_cast<double>(FIX_GRAD))),
                         static_cast<float>(static_cast<double>(a01) * (1.0 / static_cast<double>(FIX_GRAD)))};
    const float sx[2] = {static_cast<float>(static_cast<double>(a10) * (1.0 / static_cast<double>(FIX_GRAD))),
                         static_cast<float>(static_cast<double>(a11) * (1.0 / static_cast<double>(FIX_GRAD)))};
    const int ns = site ? p.n1 : p.n0;
    const float inv = ns > 0 ? 1.0f / static_cast<float>(ns) : 0.f;
    const float s1 = sd[site] * inv, s2 = sx[site] * inv;
    const float istd = p.istd[site * C + cc], mean = p.mean[site * C + cc];
    const float k = p.gamma[cc] * istd;
    k0 = k;
    k1 = k * s2 * istd;
    k2 = k * s2 * (mean * istd) - k * s1;
    if (ty == 0 && c < C) {
      p.ggamma[c] = sx[0] + sx[1];
      p.gbeta[c] = sd[0] + sd[1];
    }
  }
  cst[0][tid] = k0;
  cst[1][tid] = k1;
  cst[2][tid] = k2;
  __syncthreads();
  if (blockIdx.x == 0 && ty == 0 && tid == 0 && p.loss != nullptr && p.n_l2 > 0) {  // loss += l2 * sum W^2, fixed order
    float t = 0.f;
    for (int l = 0; l < p.n_l2; ++l) {
      float tl = 0.f;
      for (int j = 0; j < p.l2_tiles[l]; ++j) tl += p.l2_part[l * L2_SLOTS + j];
      t += tl;
    }
    p.loss[0] += p.l2 * t;
  }
  if (p.zero != nullptr) {
    const int nwg = gridDim.x * gridDim.y, per = (p.zero_n + nwg - 1) / nwg;
    const int z0 = (ty * gridDim.x + blockIdx.x) * per;
    for (int i = z0 + tid; i < z0 + per && i < p.zero_n; i += ATHR) p.zero[i] = 0.f;
  }
  if (!cok4) return;
  const float4 a0 = *reinterpret_cast<const float4*>(&cst[0][c4]), a1 = *reinterpret_cast<const float4*>(&cst[1][c4]),
               a2 = *reinterpret_cast<const float4*>(&cst[2][c4]);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = row0 + (tid >> 6) + 4 * j;
    if (row >= row_end) continue;
    const float4 v = make_float4(y[j].x > 0.f ? fmaf(g[j].x, a0.x, fmaf(-y[j].x, a1.x, a2.x)) : 0.f,
                                 y[j].y > 0.f ? fmaf(g[j].y, a0.y, fmaf(-y[j].y, a1.y, a2.y)) : 0.f,
                                 y[j].z > 0.f ? fmaf(g[j].z, a0.z, fmaf(-y[j].z, a1.z, a2.z)) : 0.f,
                                 y[j].w > 0.f ? fmaf(g[j].w, a0.w, fmaf(-y[j].w, a1.w, a2.w)) : 0.f);
    *reinterpret_cast<float4*>(p.dP + static_cast<int64_t>(row) * C + c0 + c4) = v;
  }
}

// The prologue of a step on a device-resident batch of article-row numbers (dataloader.py:169-179: lookup_article_matrix[rows]):
// step-state advance + label copy + document-vector gather as ONE launch -- the rows are read straight from the (up to two)
// staged index segments, never unpacked.
struct StageArgs {
  const int32_t* s0;
  const int32_t* s1;
  int n0, n1;
  const uint32_t* lab_src;
  uint32_t* lab_dst;
  int n_lab;
  const float4* matrix;
  float4* X0;
  int vpr;  // float4 per row
  int64_t n_rows;
  int32_t* oob;
  ebn_step_state* st;
  double beta1, beta2;
};
__global__ __launch_bounds__(256) void dvn_stage_gather_kernel(StageArgs a) {
  if (a.st != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {  // ebn_step_advance
    ebn_step_state* st = a.st;
    const uint32_t t = st->step + 1u;
    st->step = t;
    const double b1t = pow(a.beta1, static_cast<double>(t));
    const double b2t = pow(a.beta2, static_cast<double>(t));
    st->adam_alpha = static_cast<float>(static_cast<double>(st->lr) * sqrt(1.0 - b2t) / (1.0 - b1t));
    for (uint32_t s = 0; s < EBN_N_SITES; ++s) st->drop_key[s] = ebn_dropout_key(st->seed, t, s);
  }
  const int gtid = blockIdx.x * 256 + threadIdx.x;
  if (gtid < a.n_lab) a.lab_dst[gtid] = a.lab_src[gtid];
  const int items = (a.n0 + a.n1) * a.vpr;
  // two items per thread, ids first, then the rows: unconditional loads with clamped addresses
  int it[2], row[2];
  int64_t id[2];
  bool ok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    it[u] = gtid + u * gridDim.x * 256;
    ok[u] = it[u] < items;
    const int itc = ok[u] ? it[u] : 0;
    row[u] = itc / a.vpr;
    id[u] = row[u] < a.n0 ? a.s0[row[u]] : a.s1[row[u] - a.n0];
  }
  float4 v[2];
  bool bad = false;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const bool in = id[u] >= 0 && id[u] < a.n_rows;
    bad |= ok[u] && !in;
    const int itc = ok[u] ? it[u] : 0;
    const float4 t = a.matrix[(in ? id[u] : 0) * a.vpr + (itc - row[u] * a.vpr)];
    v[u] = in ? t : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (bad && a.oob != nullptr) *a.oob = 1;
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (ok[u]) a.X0[it[u]] = v[u];
}

inline int row_tiles(const ebn_dvn_args* a) { return (a->n0 + TM - 1) / TM + (a->n1 + TM - 1) / TM; }

// stat layout (floats; 64-bit words 8-byte aligned because every block is a multiple of 8 floats -- widths are multiples of 4):
//   forward accumulators, all layers:  per layer int64 [2 sums][2 sites][u]  = 8 u floats
//   backward accumulators, all layers: the same
//   per layer mean [2][u] | istd [2][u]
//   L2 column-tile sums [EBN_DVN_MAX_LAYERS][L2_SLOTS]
inline int64_t sum_units(const ebn_dvn_args* a, int upto) {
  int64_t n = 0;
  for (int l = 0; l < upto; ++l) n += a->units[l];
  return n;
}
struct StatView {
  long long* fwd;
  long long* bwd;
  float* mean;
  float* istd;
};
inline StatView stat_view(const ebn_dvn_args* a, int l) {
  const int64_t U = sum_units(a, a->n_layers), before = sum_units(a, l), u = a->units[l];
  float* base = a->stat;
  return StatView{reinterpret_cast<long long*>(base + 8 * before), reinterpret_cast<long long*>(base + 8 * U + 8 * before),
                  base + 16 * U + 4 * before, base + 16 * U + 4 * before + 2 * u};
}
inline float* l2_view(const ebn_dvn_args* a) { return a->stat + 20 * sum_units(a, a->n_layers); }

inline size_t panel_lds(int K, bool with_cst) {
  const int Kpad = (K + TK - 1) / TK * TK;
  return static_cast<size_t>(2 * (TILE_A + TILE_B) + (with_cst ? 3 * Kpad : 0) + 256) * sizeof(float);
}

template <int AX, bool B_KC, int EPI>
int launch_panel(const PanelArgs& p, hipStream_t s) {
  constexpr bool CST = (AX == AX_BN || AX == AX_DBN);
  const size_t lds = panel_lds(MAX_K, true);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&dvn_panel_kernel<AX, B_KC, EPI>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  if (attr != hipSuccess) return static_cast<int>(attr);
  const dim3 grid(static_cast<unsigned>((p.Nout + TN - 1) / TN), static_cast<unsigned>((p.n0 + TM - 1) / TM + (p.n1 + TM - 1) / TM));
  EBN_LAUNCH((dvn_panel_kernel<AX, B_KC, EPI>), grid, dim3(NTHR), panel_lds(p.K, CST), s, p);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

int check_args(const ebn_dvn_args* a) {
  EBN_REQUIRE(a != nullptr, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(a->n_layers >= 1 && a->n_layers <= EBN_DVN_MAX_LAYERS, EBN_ERR_UNSUPPORTED);
  EBN_REQUIRE(a->n0 >= 0 && a->n1 >= 0 && a->n0 + a->n1 > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(row_tiles(a) <= MAX_TILES, EBN_ERR_UNSUPPORTED);
  EBN_REQUIRE(a->n0 <= TM * MAX_TILES && a->n1 <= TM * MAX_TILES, EBN_ERR_UNSUPPORTED);
  EBN_REQUIRE(a->din >= 4 && a->din % 4 == 0 && a->e_out >= 4 && a->e_out % 4 == 0, EBN_ERR_UNSUPPORTED);
  EBN_REQUIRE(static_cast<int64_t>(a->n0 + a->n1) * a->din < (int64_t{1} << 29), EBN_ERR_UNSUPPORTED);
  for (int l = 0; l < a->n_layers; ++l)
    EBN_REQUIRE(a->units[l] >= 4 && a->units[l] % 4 == 0 && a->units[l] <= MAX_K, EBN_ERR_UNSUPPORTED);
  EBN_REQUIRE(a->drop_p >= 0.f && a->drop_p < 1.f, EBN_ERR_BAD_ARG);
  return EBN_OK;
}

}  // namespace

extern "C" int ebn_dvn_supported(const ebn_dvn_args* a) { return check_args(a) == EBN_OK ? 1 : 0; }

extern "C" int64_t ebn_dvn_stat_floats(const ebn_dvn_args* a) {
  if (check_args(a) != EBN_OK) return 0;
  return 20 * sum_units(a, a->n_layers) + EBN_DVN_MAX_LAYERS * L2_SLOTS;
}

extern "C" int ebn_dvn_fwd_train_f32(const ebn_dvn_args* a, const ebn_step_state* st, ebn_stream_t stream) {
  const int rc = check_args(a);
  if (rc != EBN_OK) return rc;
  EBN_REQUIRE(a->X0 != nullptr && a->NE != nullptr && a->stat != nullptr, EBN_ERR_BAD_ARG);
  hipStream_t s = ebn_stream(stream);
  const int L = a->n_layers;
  for (int l = 0; l <= L; ++l) {
    EBN_REQUIRE(a->W[l] != nullptr && a->b[l] != nullptr, EBN_ERR_BAD_ARG);
    PanelArgs p{};
    p.n0 = a->n0;
    p.n1 = a->n1;
    p.K = l ? a->units[l - 1] : a->din;
    p.Nout = l < L ? a->units[l] : a->e_out;
    p.B = a->W[l];
    p.ldb = p.Nout;
    p.bias = a->b[l];
    p.C = l < L ? a->R[l] : a->NE;
    p.range_flag = a->range_flag;
    EBN_REQUIRE(p.C != nullptr, EBN_ERR_BAD_ARG);
    if (l < L) p.out_acc = stat_view(a, l).fwd;
    if (l < L && a->l2 > 0.f) p.l2_part = l2_view(a) + l * L2_SLOTS;
    if (l == 0) {
      p.A = a->X0;
      p.zero = reinterpret_cast<float*>(stat_view(a, 0).bwd);  // the backward accumulators of all layers: free until the backward
      p.zero_n = static_cast<int>(8 * sum_units(a, L));
    } else {
      const StatView sv = stat_view(a, l - 1);
      const EbnDrop d = ebn_make_drop(st, EBN_SITE_MLP0 + (l - 1), a->drop_p);
      EBN_REQUIRE(a->gamma[l - 1] && a->beta[l - 1] && a->moving_mean[l - 1] && a->moving_var[l - 1] && a->Xn[l - 1], EBN_ERR_BAD_ARG);
      p.A = a->R[l - 1];
      p.Aout = a->Xn[l - 1];
      p.in_acc = sv.fwd;
      p.gamma = a->gamma[l - 1];
      p.beta = a->beta[l - 1];
      p.mean_io = sv.mean;
      p.istd_io = sv.istd;
      p.mmean = a->moving_mean[l - 1];
      p.mvar = a->moving_var[l - 1];
      p.key_in = d.key_ptr;
      p.thresh = d.thresh;
      p.scale = d.scale;
    }
    int r;
    if (l == 0) r = launch_panel<AX_PLAIN, false, EPI_RELU_STATS>(p, s);
    else if (l < L) r = launch_panel<AX_BN, false, EPI_RELU_STATS>(p, s);
    else r = launch_panel<AX_BN, false, EPI_RELU>(p, s);
    if (r != EBN_OK) return r;
  }
  return EBN_OK;
}

// one launch of the backward: l = n_layers (output Dense) ... 1: dy_{l-1} = dropout-backward(dP_l . W_l^T), dP_l formed on the A operand
// and written out; l = 0: dP_0 element-wise (+ the L2 term of the loss, + re-zeroing of the forward accumulators)
static int dvn_bwd_layer(const ebn_dvn_args* a, int32_t l, const ebn_step_state* st, ebn_stream_t stream) {
  const int rc = check_args(a);
  if (rc != EBN_OK) return rc;
  EBN_REQUIRE(a->dNE != nullptr && a->NE != nullptr && a->stat != nullptr, EBN_ERR_BAD_ARG);
  hipStream_t s = ebn_stream(stream);
  const int L = a->n_layers;
  EBN_REQUIRE(l >= 0 && l <= L, EBN_ERR_BAD_ARG);
  if (l >= 1) {
    const StatView so = stat_view(a, l - 1);
    const EbnDrop d_out = ebn_make_drop(st, EBN_SITE_MLP0 + (l - 1), a->drop_p);
    PanelArgs p{};
    p.n0 = a->n0;
    p.n1 = a->n1;
    p.K = l < L ? a->units[l] : a->e_out;
    p.Nout = a->units[l - 1];
    p.B = a->W[l];  // (Nout, K) row-major: k contiguous
    p.ldb = p.K;
    p.C = a->dY[l - 1];
    p.Aout = a->dP[l];
    p.out_acc = so.bwd;
    p.Rout = a->R[l - 1];
    p.mean_out = so.mean;
    p.istd_out = so.istd;
    p.key_out = d_out.key_ptr;
    p.thresh = d_out.thresh;
    p.scale = d_out.scale;
    p.range_flag = a->range_flag;
    EBN_REQUIRE(p.B && p.C && p.Aout && p.Rout, EBN_ERR_BAD_ARG);
    if (l == L) {
      p.A = a->dNE;
      p.A2 = a->NE;
      return launch_panel<AX_RELU, true, EPI_DY>(p, s);
    }
    const StatView si = stat_view(a, l);
    EBN_REQUIRE(a->dY[l] && a->R[l] && a->gamma[l] && a->ggamma[l] && a->gbeta[l], EBN_ERR_BAD_ARG);
    p.A = a->dY[l];
    p.A2 = a->R[l];
    p.in_acc = si.bwd;
    p.gamma = a->gamma[l];
    p.mean_io = si.mean;
    p.istd_io = si.istd;
    p.ggamma = a->ggamma[l];
    p.gbeta = a->gbeta[l];
    return launch_panel<AX_DBN, true, EPI_DY>(p, s);
  }
  const StatView s0 = stat_view(a, 0);
  EBN_REQUIRE(a->dY[0] && a->R[0] && a->gamma[0] && a->ggamma[0] && a->gbeta[0] && a->dP[0], EBN_ERR_BAD_ARG);
  ApplyArgs q{a->n0, a->n1, a->units[0], a->dY[0], a->R[0], s0.bwd, a->gamma[0], s0.mean, s0.istd, a->dP[0], a->ggamma[0], a->gbeta[0]};
  q.zero = reinterpret_cast<float*>(s0.fwd);  // the forward accumulators of all layers, for the next step
  q.zero_n = static_cast<int>(8 * sum_units(a, L));
  q.l2_part = l2_view(a);
  q.n_l2 = a->l2 > 0.f ? L : 0;
  for (int i = 0; i < L; ++i) q.l2_tiles[i] = (a->units[i] + TN - 1) / TN;
  q.l2 = a->l2;
  q.loss = a->loss;
  const dim3 grid(static_cast<unsigned>((a->units[0] + 255) / 256), static_cast<unsigned>(row_tiles(a)));
  EBN_LAUNCH(dvn_dbn_apply_kernel, grid, dim3(ATHR), 0, s, q);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_dvn_bwd_f32(const ebn_dvn_args* a, const ebn_step_state* st, ebn_stream_t stream) {
  const int rc = check_args(a);
  if (rc != EBN_OK) return rc;
  for (int l = a->n_layers; l >= 0; --l) {
    const int r = dvn_bwd_layer(a, l, st, stream);
    if (r != EBN_OK) return r;
  }
  return EBN_OK;
}

// The closing launch of a ONE-RANK training step (see ebn_tn_finale.h): the Dense weight gradients of ebn_gemm_tn_group_f32 with Adam in
// the tiles' epilogues, Adam over the remaining ranges of the flat parameter buffer, the user head's finishing sums and the batch loss
// including the L2 term (the caller ran ebn_dvn_bwd_f32 with args->loss == NULL and ebn_user_head_train_f32 with dq == db == NULL).
extern "C" int ebn_dvn_finale_f32(const ebn_dvn_args* a, const ebn_tn_problem* problems, int32_t n, const ebn_dvn_finale* f,
                                  const ebn_step_state* st, ebn_stream_t stream) {
  const int rc = check_args(a);
  if (rc != EBN_OK) return rc;
  EBN_REQUIRE(problems != nullptr && f != nullptr && st != nullptr && a->stat != nullptr, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(f->theta && f->grad && f->m && f->v && f->numel > 0 && f->numel <= EBN_DIM_MAX * 16, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(f->n_rest >= 0 && f->n_rest <= EBN_DVN_FINALE_MAX_REST, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(f->head_partials && f->dq && f->db && f->loss_rows && f->loss_out && f->B >= 1 && f->A >= 1 && f->A < (1 << 24), EBN_ERR_BAD_ARG);
  auto inside = [&](const float* ptr, int64_t count) {  // [ptr, ptr + count) lies in the flat gradient buffer
    return ptr >= f->grad && count >= 0 && (ptr - f->grad) + count <= f->numel;
  };
  EBN_REQUIRE(n >= 1 && n <= EBN_TN_GROUP_MAX, EBN_ERR_BAD_ARG);
  for (int i = 0; i < n; ++i) {
    const ebn_tn_problem& q = problems[i];
    EBN_REQUIRE(q.C != nullptr && q.M >= 1 && q.N >= 1 && q.ldc >= q.N, EBN_ERR_BAD_ARG);
    EBN_REQUIRE(inside(q.C, (q.M - 1) * q.ldc + q.N) && (q.colsum == nullptr || inside(q.colsum, q.N)), EBN_ERR_BAD_ARG);
  }
  EBN_REQUIRE(inside(f->dq, f->A) && inside(f->db, f->A), EBN_ERR_BAD_ARG);
  EbnTnFinale t{};
  t.adam = EbnAdamFlat{f->grad, f->theta, f->m, f->v, st, static_cast<float>(1.0 - f->beta1), static_cast<float>(1.0 - f->beta2),
                       static_cast<float>(f->eps), f->grad_scale};
  t.n_rest = f->n_rest;
  t.rest_total = 0;
  for (int i = 0; i < f->n_rest; ++i) {
    EBN_REQUIRE(f->rest_off[i] >= 0 && f->rest_len[i] >= 0 && f->rest_off[i] + f->rest_len[i] <= f->numel, EBN_ERR_BAD_ARG);
    t.rest_off[i] = f->rest_off[i];
    t.rest_len[i] = f->rest_len[i];
    t.rest_total += f->rest_len[i];
  }
  t.head_partials = f->head_partials;
  t.B = f->B;
  t.A = f->A;
  t.dq = f->dq;
  t.db = f->db;
  t.loss_rows = f->loss_rows;
  t.loss_out = f->loss_out;
  t.l2_part = l2_view(a);
  t.n_l2 = a->l2 > 0.f ? a->n_layers : 0;
  t.l2_slots = L2_SLOTS;
  for (int i = 0; i < a->n_layers; ++i) t.l2_tiles[i] = (a->units[i] + TN - 1) / TN;
  t.l2 = a->l2;
  return ebn_tn_group_finale_launch(problems, n, t, ebn_stream(stream));
}

extern "C" int ebn_docvec_stage_gather_f32(const int32_t* idx0, int64_t n0, const int32_t* idx1, int64_t n1, const float* labels_src,
                                           float* labels_dst, int64_t n_labels, const float* matrix, int64_t n_rows, int32_t din, float* X0,
                                           int32_t* oob_flag, ebn_step_state* st, double beta1, double beta2, ebn_stream_t stream) {
  EBN_REQUIRE(n0 >= 0 && n1 >= 0 && n_labels >= 0 && n_rows > 0 && din > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE((n0 == 0 || idx0) && (n1 == 0 || idx1) && (n_labels == 0 || (labels_src && labels_dst)) && matrix && X0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE((din % 4) == 0 && ebn_aligned16(matrix) && ebn_aligned16(X0), EBN_ERR_ALIGN);
  EBN_REQUIRE((n0 + n1) * (din / 4) < (int64_t{1} << 30) && n_labels < (int64_t{1} << 30), EBN_ERR_UNSUPPORTED);
  StageArgs a{idx0, idx1, static_cast<int>(n0), static_cast<int>(n1), reinterpret_cast<const uint32_t*>(labels_src),
              reinterpret_cast<uint32_t*>(labels_dst), static_cast<int>(n_labels), reinterpret_cast<const float4*>(matrix),
              reinterpret_cast<float4*>(X0), din / 4, n_rows, oob_flag, st, beta1, beta2};
  const int64_t items = (n0 + n1) * (din / 4);
  int64_t blocks = ebn_ceil_div(items, 512);
  if (blocks * 256 < n_labels) blocks = ebn_ceil_div(n_labels, 256);
  if (blocks < 1) blocks = 1;
  EBN_LAUNCH(dvn_stage_gather_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, ebn_stream(stream), a);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
