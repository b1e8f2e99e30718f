// Exact-fp32 MFMA GEMM for the projection matmuls of the hot path
// (K.dot at layers.py:65,214,220,226; Dense at nrms_docvec.py:116,130; and their backward).
//
//   C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C       (row-major)
//
// MFMA-bound (v_mfma_f32_32x32x2_f32: 64 cycles / 4096 FLOP per SIMD -> 157 TFLOP/s chip
// peak, bitwise an fp32 fma chain -> keeps the 1e-4 parity budget of north_star).
// Block tile BM x BN x 16 (128x128 or 64x64 with the 4 waves as 2x2; 256x64 with the waves stacked 4x1), each wave
// owning its share in 32x32 MFMA tiles.
// LDS images of a 16-deep K slab, chosen per operand by how it is stored in memory:
//   mn-contiguous ([K][mn]):  S[k][mn]; operand fetch is a conflict-free ds_read_b32 of 32 consecutive floats of one k row;
//   k-contiguous ([mn][K]):   S4[mn][kq ^ ((mn>>2)&3)] of float4 -- one conflict-free ds_read_b128 per four MFMA steps.
// Full slabs of 16-byte-aligned operands go global -> LDS directly (buffer_load_dwordx4 ... lds into an unpadded image; for
// a k-contiguous operand the XOR swizzle is applied to the source column a lane fetches); a partial last slab and unaligned
// operands go through registers (one ds_write_b128 per fetched float4).
// The slab's 16 k are assigned to the MFMA's (step, lane half) as k = 8j + 4*half + w for step 4j + w, identically
// for A and B, which is what makes a k-contiguous float4 feed four consecutive steps.  The next slab is fetched
// while the current one is multiplied (2 LDS buffers, one barrier per slab).  The slab loop carries NO vector-ALU work
// (scalar slab offsets, literal LDS buffer indices): on gfx950 the fp32 MFMA does not overlap VALU instructions
// (tools/microbench/mfma_valu_overlap.hip).  Skinny outputs (weight gradients: M,N ~ 1e3, K ~ 2e4) use split-K with
// deterministic slab partials in the caller's workspace -- never atomics.
#include <stdlib.h>

#include "ebn_common.h"
#include "ebn_finish.h"
#include "ebn_tn_finale.h"

// Buffer-load intrinsics bound by name (the __amdgpu_buffer_rsrc_t builtins make the HOST pass drop the launch stub of
// a kernel that uses them).  Declared outside the anonymous namespace: they are external symbols of the compiler.
typedef float ebn_f32x4 __attribute__((ext_vector_type(4)));
typedef int ebn_i32x4 __attribute__((ext_vector_type(4)));
__device__ ebn_f32x4 ebn_raw_buffer_load_x4(ebn_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void ebn_raw_buffer_load_lds(ebn_i32x4 rsrc, __attribute__((address_space(3))) void* lds, int size, int voffset,
                                        int soffset, int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

// ebn_gemm_direct.hip
int ebn_gemm_direct_tn_slices(int64_t M, int64_t N, int64_t K);
int ebn_gemm_direct_tn_launch(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B, int64_t ldb,
                              float* part, hipStream_t s);
bool ebn_gemm_direct_wanted(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K);
int ebn_gemm_direct_launch(int32_t transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B,
                           int64_t ldb, float* C, int64_t ldc, hipStream_t s);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));


typedef float f32x4n __attribute__((ext_vector_type(4)));
// 16-byte load through an explicitly GLOBAL pointer (global_load_dwordx4, never flat_load)
__device__ __forceinline__ float4 gload4(const __attribute__((address_space(1))) float* p) {
  const f32x4n t = *reinterpret_cast<const __attribute__((address_space(1))) f32x4n*>(p);
  return make_float4(t.x, t.y, t.z, t.w);
}
// Raw buffer resource over [base, base + 4 GB): buffer_load takes a scalar (SGPR) byte offset next to the lane's 32-bit
// byte offset, so a slab loop advances ONE scalar per operand and carries no per-lane address.  (With plain pointers the
// loop-strength-reduction pass turns base + lane offset back into a per-lane 64-bit pointer that is bumped on the
// vector ALU every slab -- and on gfx950 the fp32 MFMA does not overlap vector-ALU work.)
typedef ebn_i32x4 i32x4n;
__device__ __forceinline__ i32x4n make_rsrc(const float* base) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  i32x4n r;
  r.x = static_cast<int>(static_cast<uint32_t>(a));
  r.y = static_cast<int>(static_cast<uint32_t>(a >> 32) & 0xFFFFu);  // stride 0: raw buffer
  r.z = -1;                                                            // 4 GB of records: no bounds use
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ float4 bload4(i32x4n r, uint32_t lane_bytes, uint32_t slab_bytes) {
  const f32x4n t = ebn_raw_buffer_load_x4(r, static_cast<int>(lane_bytes), static_cast<int>(slab_bytes), 0);
  return make_float4(t.x, t.y, t.z, t.w);
}

// Waves per SIMD the register allocation must allow (LDS admits three 41.5 KB workgroups per CU).  It also selects the
// VGPR form of the MFMAs (accumulators in ordinary registers): without it the compiler picks the AGPR form and, around
// the two-slab loop body, copies all 64 accumulators between the two register files every iteration.
constexpr int GEMM_WPE = 3;
constexpr int BK = 16;
constexpr int PAD = 4;
constexpr int GEMM_THREADS = 256;

// One operand tile of R_MN x BK (mn = m or n index). KCONTIG: memory is [mn][k] (k fastest);
// else memory is [k][mn] (mn fastest).
template <int BMN, bool KCONTIG, bool VEC, int PADX = PAD>
struct TileLoader {
  static constexpr int VECS = BMN * BK / 4;            // float4 per tile
  static constexpr int PER_THREAD = VECS / GEMM_THREADS;  // 2 (BMN=128) or 1 (BMN=64)
  static_assert(VECS % GEMM_THREADS == 0, "tile/thread mismatch");

  // returns 4 consecutive elements along the contiguous axis, zero-filled outside the matrix.
  // vec_ok (host-checked: 16-byte aligned base, ld % 4 == 0, extent of the contiguous axis % 4 == 0)
  // makes every float4 all-in or all-out, so the load is BRANCH-FREE: an out-of-range lane reads the
  // (always valid) first 16 bytes of the matrix and the value is replaced by a select.  Guarded
  // per-element loads make hipcc branch around each one and serialise them on vmcnt(0).
  __device__ static __forceinline__ float4 load(const float* __restrict__ P, int64_t ld, int64_t mn0,
                                                int64_t k0, int64_t MN, int64_t Kdim, int v) {
    int64_t gmn, gk;
    if (KCONTIG) {
      gmn = mn0 + v / (BK / 4);
      gk = k0 + (v % (BK / 4)) * 4;
    } else {
      gk = k0 + v / (BMN / 4);
      gmn = mn0 + (v % (BMN / 4)) * 4;
    }
    const int64_t off = KCONTIG ? (gmn * ld + gk) : (gk * ld + gmn);
    if (VEC) {
      // clamp the OFFSET (not the pointer) and zero per component: a pointer select next to a whole-value select is
      // turned back into a branch around the load, and each such load then waits vmcnt(0) on its own
      const bool valid = (gmn < MN) && (gk < Kdim);
      const float4 r = *reinterpret_cast<const float4*>(P + (valid ? off : 0));
      float4 z;
      z.x = valid ? r.x : 0.f;
      z.y = valid ? r.y : 0.f;
      z.z = valid ? r.z : 0.f;
      z.w = valid ? r.w : 0.f;
      return z;
    }
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KCONTIG) {
      if (gmn < MN) {
        const float* p = P + off;
        if (gk + 0 < Kdim) r.x = p[0];
        if (gk + 1 < Kdim) r.y = p[1];
        if (gk + 2 < Kdim) r.z = p[2];
        if (gk + 3 < Kdim) r.w = p[3];
      }
    } else {
      if (gk < Kdim) {
        const float* p = P + off;
        if (gmn + 0 < MN) r.x = p[0];
        if (gmn + 1 < MN) r.y = p[1];
        if (gmn + 2 < MN) r.z = p[2];
        if (gmn + 3 < MN) r.w = p[3];
      }
    }
    return r;
  }

  // LDS image: S[k][mn], row stride BMN+PAD
  __device__ static __forceinline__ void store(float* __restrict__ S, int v, float4 r) {
    constexpr int LD = BMN + PADX;
    if (KCONTIG) {
      // k-contiguous operand: image S4[mn][kq ^ ((mn >> 2) & 3)] of float4 (4 consecutive k): ONE ds_write_b128 per
      // float4 (8 lanes of a store group cover 128 contiguous bytes), read back with ds_read_b128 -- the XOR spreads
      // the 16 rows of a read group over all 64 banks
      const int mn = v / (BK / 4);
      const int kq = v % (BK / 4);
      *reinterpret_cast<float4*>(&S[(mn * 4 + (kq ^ ((mn >> 2) & 3))) * 4]) = r;
    } else {
      const int k = v / (BMN / 4);
      const int mq = v % (BMN / 4);
      *reinterpret_cast<float4*>(&S[k * LD + mq * 4]) = r;
    }
  }
};

// Optional rank-1-per-sequence term of the epilogue: C[row][col] += rs[row] * cv[row / L][col]  (EPI = 1).
// It is the d(x) of the AttLayer2 pooling, w[n,l] * dout[n,:] (layers.py:79-81 backward), folded into the GEMM that
// produces the other half of d(x), so that neither kernel has to write and re-read the [R,E] gradient.
struct GemmEpi {
  const float* rs;
  const float* cv;
  int64_t ldcv;
  int32_t L;
  const float* bias;  // EPI = 2: C = max(acc + bias[col], 0) -- Dense(units, activation="relu") forward
  // small-output tiles only (ebn_gemm_tn_group_f32): column sums of a [K][N] B operand, written by the first row of tiles;
  // C += two_lambda * l2w[row][col] (leading dimension ldc)
  float* colsum = nullptr;
  const float* l2w = nullptr;
  float two_lambda = 0.f;
  // EPI = 3 (ebn_gemm_f32_rowmap): row r of the A operand is row rowmap[r] of a (rowmap_rows, K) table -- the embedding gather of
  // nrms.py:125-134 fused into the projection's A-operand fetch (inference: no dropout between them); ids outside the table read
  // row 0 and raise *oob
  const int32_t* rowmap = nullptr;
  int64_t rowmap_rows = 0;
  int32_t* oob = nullptr;
#ifdef EBN_GEMM_EXP_STAGGER
  // tuning experiment (tools/build_variant.sh -DEBN_GEMM_EXP_STAGGER, EBN_GEMM_STAGGER_PCT at run time): the three workgroups that
  // share a CU in the first dispatch round start a third of a tile's duration apart (percent of that third)
  uint32_t stagger_pct = 0;
#endif
};

// TA: A stored [K,M]; TB: B stored [N,K].
// gridDim.z = split-K factor; when > 1 each z-slice writes alpha*partial to
// Cpart + z*M*N (dense ld = N) and a reduce kernel finishes; else writes C directly.
template <int BM, int BN, int WAVES_M, bool TA, bool TB, bool VEC, int EPI = 0>
__global__ __launch_bounds__(GEMM_THREADS, GEMM_WPE) void gemm_f32_kernel(
    int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A, int64_t lda,
    const float* __restrict__ B, int64_t ldb, float beta, float* __restrict__ C, int64_t ldc,
    int64_t k_per_split, float* __restrict__ Cpart, GemmEpi epi) {
  constexpr int WAVES_N = 4 / WAVES_M;    // the 4 waves form a WAVES_M x WAVES_N grid over the block tile
  constexpr int WTM = BM / WAVES_M;       // rows / columns owned by one wave
  constexpr int WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32;            // 32x32 MFMA tiles per wave along m / n
  constexpr int TN = WTN / 32;
  // GLDS: the 16-deep slab of an operand goes global -> LDS directly (buffer_load ... lds: wave-uniform LDS base + lane x
  // 16 B), no staging registers, no ds_write pass:
  //   * stored [K][mn]: 16 contiguous row pieces, dropped into an UNPADDED [k][mn] image (the pad only ever served the
  //     scalar stores of the other layout; the operand fetch reads 32 consecutive floats of one k row either way);
  //   * stored [mn][K]: the float4 image S4[mn][kq ^ ((mn >> 2) & 3)] is 64 contiguous bytes per row, so lane l of an
  //     instruction fills chunk l and the XOR swizzle is applied to the SOURCE column it fetches -- the four lanes of a
  //     row still read the row's 64 contiguous bytes, permuted.
  // Used for every layout of 16-byte-aligned operands.  (While the k-contiguous images were read through float4-typed
  // loads the NT GEMMs lost with it -- the compiler then waits vmcnt(0) in front of the first LDS read of every slab --
  // and were kept on the register path; with float-typed reads the NT input-gradient GEMM of c1 went 169 -> 156 us.)
  constexpr bool GLDS_A = VEC;
  constexpr bool GLDS_B = GLDS_A;
  constexpr int PADA = GLDS_A ? 0 : PAD, PADB = GLDS_B ? 0 : PAD;
  constexpr int LDA_S = BM + PADA;
  constexpr int LDB_S = BN + PADB;
  using LA = TileLoader<BM, !TA, VEC, PADA>;
  using LB = TileLoader<BN, TB, VEC, PADB>;

  // ONE __shared__ object: with two, hipcc cannot tell the glds destination from the buffer being read and waits
  // vmcnt(0) before the first ds_read of every slab (cdna_hip_programming.md, glds trap (a))
  constexpr bool A_KC = !TA, B_KC = TB;  // k-contiguous operands use the float4 image
  constexpr int A_FLOATS = A_KC ? BM * BK : BK * LDA_S, B_FLOATS = B_KC ? BN * BK : BK * LDB_S;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_FLOATS + B_FLOATS)];
#define EBN_AS(b) (smem + (b) * A_FLOATS)
#define EBN_BS(b) (smem + 2 * A_FLOATS + (b) * B_FLOATS)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS destinations of the glds fetch, wave tile origin
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  // XCD-aware tile order: hardware deals consecutive workgroups round-robin to the 8 XCDs (private
  // L2s); remap so that each XCD walks a CONTIGUOUS run of tiles (neighbouring tiles share their A
  // row-panel / B column-panel in one L2).  Bijective for any grid size; speed only, never correctness.
  const int64_t GX = gridDim.x, GY = gridDim.y, GZ = gridDim.z;
  const int64_t per_z = GX * GY, nwg = per_z * GZ;
  // position in the hardware dispatch order
  const int64_t vb = static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x + per_z * blockIdx.z;
  int64_t m0 = 0, n0 = 0, kbeg = 0, kend = 0, zsplit = 0;
  int nk = 0, nk_full = 0, nk_main = 0;

  f32x16 acc[TM][TN];

  float4 ra[LA::PER_THREAD], rb[LB::PER_THREAD];

  // Fast tile fetch (VEC): one source pointer per float4 a thread owns, advanced by a constant per slab -- the slab
  // loop issues bare 16-byte loads, no index arithmetic, compares or selects (they cost ~11 % of the MFMA issue time
  // when done per slab).  Rows / columns past the matrix edge are CLAMPED to the last valid float4 instead of
  // zero-filled: they only feed output rows / columns the epilogue never stores.  Only a partial last slab
  // (K range not a multiple of BK) goes through the guarded, zero-filling loader.
  // address_space(1): these pointers are carried through the slab loop and re-selected at its back edge, where the
  // compiler loses track of their being GLOBAL pointers and emits flat_load -- which also counts on lgkmcnt, so every
  // s_waitcnt lgkmcnt(0) in front of the MFMAs (meant for the LDS operand reads) would wait for the prefetch of the next
  // slab as well and put the whole global-load latency on the MFMA critical path.
  // The source of a float4 is a buffer resource based at the tile's first row / column of this K split, a constant
  // 32-bit lane offset and a scalar slab offset: the slab loop issues bare buffer_loads and one s_add per operand.
  // (Offsets are BYTES in 32 bits: the launcher checks that 256 tile rows and the K range of an operand span < 4 GB.)
  uint32_t oa[LA::PER_THREAD], ob[LB::PER_THREAD];
  i32x4n arsrc, brsrc;
  const uint32_t step_a = static_cast<uint32_t>((TA ? BK * lda : BK) * 4);  // bytes per slab
  const uint32_t step_b = static_cast<uint32_t>((TB ? BK : BK * ldb) * 4);
  uint32_t sa = 0, sb = 0;                                                   // scalar slab offsets
  // ---- direct-to-LDS fetch (GLDS kernels): wave w issues IPW instructions per operand and slab.  [K][mn] operands:
  // instruction q covers RPI consecutive k rows (64 lanes x 16 B = RPI rows of BMN floats).  [mn][K] operands: instruction
  // q covers the 16 rows (64 chunks of 16 B) number (q * 4 + w).
  constexpr int A_LPR = BM / 4, A_RPI = 64 / (A_LPR < 64 ? A_LPR : 64), A_IPW = A_KC ? BM / 64 : (BK / A_RPI) / 4;
  constexpr int B_LPR = BN / 4, B_RPI = 64 / (B_LPR < 64 ? B_LPR : 64), B_IPW = B_KC ? BN / 64 : (BK / B_RPI) / 4;
  static_assert(BM <= 256 && BN <= 256 && A_IPW >= 1 && B_IPW >= 1, "glds tiling");
  uint32_t gao[GLDS_A ? A_IPW : 1], gbo[GLDS_B ? B_IPW : 1];  // lane byte offsets into arsrc / brsrc
  // Everything that depends on WHICH tile (and K split) this is: origin, K range, buffer resources, lane offsets.
  constexpr bool ROWMAP = EPI == 3;
  auto mapped_row = [&](int64_t row) -> int64_t {  // table row of operand row `row` (clamped into the table; the flag tells)
    const int64_t id = epi.rowmap[row];
    const bool in = id >= 0 && id < epi.rowmap_rows;
    if (!in && epi.oob != nullptr) *epi.oob = 1;
    return in ? id : 0;
  };
  auto tile_setup = [&](int64_t v) {
    int64_t lin = v;
    {
      const int64_t q = nwg / 8, r = nwg % 8, xcd = v % 8, idx = v / 8;
      lin = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    zsplit = lin / per_z;  // tiles of one K-split share their A/B K-range: keep them on one XCD too
    const int64_t tile_id = lin - zsplit * per_z;
    m0 = (tile_id / GX) * BM;
    n0 = (tile_id % GX) * BN;
    kbeg = zsplit * k_per_split;
    kend = (kbeg + k_per_split < K) ? (kbeg + k_per_split) : K;
    nk = static_cast<int>((kend - kbeg + BK - 1) / BK);
    nk_full = static_cast<int>((kend - kbeg) / BK);  // slabs that lie completely inside [kbeg, kend)
    nk_main = VEC ? nk_full : 0;
      arsrc = make_rsrc(ROWMAP ? A + kbeg : (TA ? A + kbeg * lda + m0 : A + m0 * lda + kbeg));
      brsrc = make_rsrc(TB ? B + n0 * ldb + kbeg : B + kbeg * ldb + n0);
      sa = 0;
      sb = 0;
    if (VEC) {
  #pragma unroll
      for (int i = 0; i < LA::PER_THREAD; ++i) {
        const int v = tid + i * GEMM_THREADS;
        if (!TA) {  // A is [M][K]
          int64_t row = m0 + v / (BK / 4);
          row = row < M ? row : M - 1;
          oa[i] = static_cast<uint32_t>(((ROWMAP ? mapped_row(row) : row - m0) * lda + (v % (BK / 4)) * 4) * 4);
        } else {  // A is [K][M]
          int64_t col = m0 + (v % (BM / 4)) * 4;
          col = col < M ? col : M - 4;
          oa[i] = static_cast<uint32_t>(((v / (BM / 4)) * lda + (col - m0)) * 4);
        }
      }
  #pragma unroll
      for (int i = 0; i < LB::PER_THREAD; ++i) {
        const int v = tid + i * GEMM_THREADS;
        if (TB) {  // B is [N][K]
          int64_t row = n0 + v / (BK / 4);
          row = row < N ? row : N - 1;
          ob[i] = static_cast<uint32_t>(((row - n0) * ldb + (v % (BK / 4)) * 4) * 4);
        } else {  // B is [K][N]
          int64_t col = n0 + (v % (BN / 4)) * 4;
          col = col < N ? col : N - 4;
          ob[i] = static_cast<uint32_t>(((v / (BN / 4)) * ldb + (col - n0)) * 4);
        }
      }
    }

    if (GLDS_A) {
  #pragma unroll
      for (int q = 0; q < A_IPW; ++q) {
        if (A_KC) {
          const int r = ((q * 4 + wave) * 64 + lane) >> 2, kq = (lane & 3) ^ ((r >> 2) & 3);
          int64_t row = m0 + r;
          row = row < M ? row : M - 1;
          gao[q] = static_cast<uint32_t>(((ROWMAP ? mapped_row(row) : row - m0) * lda + kq * 4) * 4);
        } else {
          const int krow = (wave * A_IPW + q) * A_RPI + lane / A_LPR;
          int64_t col = m0 + (lane % A_LPR) * 4;
          col = col < M ? col : M - 4;
          gao[q] = static_cast<uint32_t>((krow * lda + (col - m0)) * 4);
        }
      }
    }
    if (GLDS_B) {
  #pragma unroll
      for (int q = 0; q < B_IPW; ++q) {
        if (B_KC) {
          const int r = ((q * 4 + wave) * 64 + lane) >> 2, kq = (lane & 3) ^ ((r >> 2) & 3);
          int64_t row = n0 + r;
          row = row < N ? row : N - 1;
          gbo[q] = static_cast<uint32_t>(((row - n0) * ldb + kq * 4) * 4);
        } else {
          const int krow = (wave * B_IPW + q) * B_RPI + lane / B_LPR;
          int64_t col = n0 + (lane % B_LPR) * 4;
          col = col < N ? col : N - 4;
          gbo[q] = static_cast<uint32_t>((krow * ldb + (col - n0)) * 4);
        }
      }
    }
  };
  tile_setup(vb);
#ifdef EBN_GEMM_EXP_STAGGER
  if (epi.stagger_pct != 0u && vb < 768) {  // 256 consecutive workgroups go one per CU: b, b + 256, b + 512 share a CU
    // one tile on a CU shared by three = nk slabs x 3 x 2048 cycles; a third of that = nk x 2048 cycles = nk / 2 sleeps of 4096
    const uint32_t n = (static_cast<uint32_t>(vb >> 8) % 3u) * static_cast<uint32_t>(nk) * epi.stagger_pct / 200u;
    for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(64);
  }
#endif
  // LDS float offset of the 1 KB that instruction q of this wave fills
#define EBN_GLDS_A_DST(Q) (A_KC ? ((Q) * 4 + wave) * 256 : (wave * A_IPW + (Q)) * A_RPI * BM)
#define EBN_GLDS_B_DST(Q) (B_KC ? ((Q) * 4 + wave) * 256 : (wave * B_IPW + (Q)) * B_RPI * BN)
  // Tail slab of a VEC kernel (K range not a multiple of BK): the same buffer loads, with the pieces at k >= kend replaced
  // by zero -- a float4 is all-in or all-out (k-contiguous operands have K % 4 == 0; for the others a k row is in or out).
  // An out-of-range piece reads offset 0 of the resource (always valid).  Costs a handful of VALU instructions, once.
#define EBN_TAIL_PIECE(R, RSRC, OFF, SOFF, KREL)                                         \
  do {                                                                                   \
    const bool ok__ = (KREL) < krem__;                                                   \
    const float4 t__ = bload4(RSRC, ok__ ? (OFF) + (SOFF) : 0u, 0u);                      \
    (R).x = ok__ ? t__.x : 0.f; (R).y = ok__ ? t__.y : 0.f;                              \
    (R).z = ok__ ? t__.z : 0.f; (R).w = ok__ ? t__.w : 0.f;                              \
  } while (0)
  // Fetch of a FULL slab of a VEC kernel: per operand straight to LDS buffer BUF (glds) or into registers (stored by
  // EBN_STORE_FULL after the MFMAs).
#define EBN_FETCH_FULL(BUF)                                                              \
  do {                                                                                   \
    if (GLDS_A) {                                                                        \
      _Pragma("unroll") for (int q = 0; q < A_IPW; ++q)                                  \
        ebn_raw_buffer_load_lds(arsrc, (__attribute__((address_space(3))) void*)(EBN_AS(BUF) + EBN_GLDS_A_DST(q)), \
                                16, static_cast<int>(gao[q]), static_cast<int>(sa), 0, 0); \
    } else {                                                                             \
      _Pragma("unroll") for (int i = 0; i < LA::PER_THREAD; ++i) ra[i] = bload4(arsrc, oa[i], sa); \
    }                                                                                    \
    if (GLDS_B) {                                                                        \
      _Pragma("unroll") for (int q = 0; q < B_IPW; ++q)                                  \
        ebn_raw_buffer_load_lds(brsrc, (__attribute__((address_space(3))) void*)(EBN_BS(BUF) + EBN_GLDS_B_DST(q)), \
                                16, static_cast<int>(gbo[q]), static_cast<int>(sb), 0, 0); \
    } else {                                                                             \
      _Pragma("unroll") for (int i = 0; i < LB::PER_THREAD; ++i) rb[i] = bload4(brsrc, ob[i], sb); \
    }                                                                                    \
    sa += step_a;                                                                        \
    sb += step_b;                                                                        \
  } while (0)
#define EBN_STORE_FULL(BUF)                                                              \
  do {                                                                                   \
    if (!GLDS_A) {                                                                       \
      _Pragma("unroll") for (int i = 0; i < LA::PER_THREAD; ++i) LA::store(EBN_AS(BUF), tid + i * GEMM_THREADS, ra[i]); \
    }                                                                                    \
    if (!GLDS_B) {                                                                       \
      _Pragma("unroll") for (int i = 0; i < LB::PER_THREAD; ++i) LB::store(EBN_BS(BUF), tid + i * GEMM_THREADS, rb[i]); \
    }                                                                                    \
  } while (0)
  // A partial last slab, and every slab of a non-VEC kernel: both operands through registers.
#define EBN_FETCH_PART(KT)                                                               \
  do {                                                                                   \
  if (VEC) {                                                                             \
    const int krem__ = static_cast<int>(kend - kbeg) - (KT) * BK;                        \
    _Pragma("unroll") for (int i = 0; i < LA::PER_THREAD; ++i) {                         \
      const int v__ = tid + i * GEMM_THREADS;                                            \
      EBN_TAIL_PIECE(ra[i], arsrc, oa[i], sa, TA ? v__ / (BM / 4) : (v__ % (BK / 4)) * 4); \
    }                                                                                    \
    _Pragma("unroll") for (int i = 0; i < LB::PER_THREAD; ++i) {                         \
      const int v__ = tid + i * GEMM_THREADS;                                            \
      EBN_TAIL_PIECE(rb[i], brsrc, ob[i], sb, TB ? (v__ % (BK / 4)) * 4 : v__ / (BN / 4)); \
    }                                                                                    \
  } else {                                                                               \
    const int64_t k0__ = kbeg + static_cast<int64_t>(KT) * BK;                           \
    _Pragma("unroll") for (int i = 0; i < LA::PER_THREAD; ++i)                           \
        ra[i] = LA::load(A, lda, m0, k0__, M, kend, tid + i * GEMM_THREADS);             \
    _Pragma("unroll") for (int i = 0; i < LB::PER_THREAD; ++i)                           \
        rb[i] = LB::load(B, ldb, n0, k0__, N, kend, tid + i * GEMM_THREADS);             \
  }                                                                                      \
  } while (0)
#define EBN_STORE_PART(BUF)                                                              \
  do {                                                                                   \
    _Pragma("unroll") for (int i = 0; i < LA::PER_THREAD; ++i) LA::store(EBN_AS(BUF), tid + i * GEMM_THREADS, ra[i]); \
    _Pragma("unroll") for (int i = 0; i < LB::PER_THREAD; ++i) LB::store(EBN_BS(BUF), tid + i * GEMM_THREADS, rb[i]); \
  } while (0)

  // Pipeline: LDS buffer `cur` holds slab kt; slab kt+1 is fetched into registers while slab kt is multiplied
  // and written to LDS[cur^1] afterwards; one barrier per slab.  Variants measured and rejected on MI355X
  // (profiles/r01_gemm_tuning.md): two-slab-deep prefetch, BK=32, S[k/4][mn][4] image with ds_read_b64, glds-filled
  // float4 image, in-kernel split-K fix-up, s_setprio around the MFMA cluster / per-workgroup static priority.
  // Two phases.  MAIN: the full slabs of a VEC kernel, two per loop iteration with the LDS buffer index a literal, and
  // nothing but full-slab fetches inside (a partial-slab path merging into this loop makes the compiler wait vmcnt(0)
  // before the first LDS read of every slab -- i.e. for the glds fetch it has just issued).  REST: a partial last slab,
  // or every slab of a non-VEC kernel, through registers with a run-time buffer index.
  const int kl = lane >> 5;
  const int il = lane & 31;
  {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (nk_main > 0) {
    EBN_FETCH_FULL(0);
    EBN_STORE_FULL(0);
    __syncthreads();
  }
  // The MFMAs of one slab out of LDS buffer CUR.  MFMA contraction index = (instruction, lane half); the slab's 16 k are
  // assigned as   step 4j + w of half kl  <->  k = 8j + 4kl + w      (A and B agree, so any assignment is valid)
  // which makes the four values a lane feeds to steps 4j..4j+3 one float4 of a k-contiguous operand.
  // (The k-contiguous images are read through FLOAT-typed loads -- the compiler still merges the four into one
  // ds_read_b128 -- because a float4-typed LDS read next to the glds fetch of the other buffer makes it wait vmcnt(0), i.e.
  // for the fetch it has just issued, in front of the first read of every slab: 436-442 -> 428 us on the projection.)
#define EBN_MMA(CUR)  \
  {  \
    const float* as = A_KC ? EBN_AS((CUR)) + (wm * WTM + il) * 16 : EBN_AS((CUR)) + (4 * kl) * LDA_S + wm * WTM + il;  \
    const float* bs = B_KC ? EBN_BS((CUR)) + (wn * WTN + il) * 16 : EBN_BS((CUR)) + (4 * kl) * LDB_S + wn * WTN + il;  \
    const int sw = (il >> 2) & 3;  \
_Pragma("unroll")  \
    for (int j8 = 0; j8 < BK / 8; ++j8) {  \
      float a[TM][4], b[TN][4];  \
_Pragma("unroll")  \
      for (int i = 0; i < TM; ++i) {  \
        if (A_KC) {  \
          const float* pa__ = as + i * 32 * 16 + (((2 * j8 + kl) ^ sw) * 4);  \
          a[i][0] = pa__[0]; a[i][1] = pa__[1]; a[i][2] = pa__[2]; a[i][3] = pa__[3];  \
        } else {  \
_Pragma("unroll")  \
          for (int w = 0; w < 4; ++w) a[i][w] = as[(8 * j8 + w) * LDA_S + i * 32];  \
        }  \
      }  \
_Pragma("unroll")  \
      for (int j = 0; j < TN; ++j) {  \
        if (B_KC) {  \
          const float* pb__ = bs + j * 32 * 16 + (((2 * j8 + kl) ^ sw) * 4);  \
          b[j][0] = pb__[0]; b[j][1] = pb__[1]; b[j][2] = pb__[2]; b[j][3] = pb__[3];  \
        } else {  \
_Pragma("unroll")  \
          for (int w = 0; w < 4; ++w) b[j][w] = bs[(8 * j8 + w) * LDB_S + j * 32];  \
        }  \
      }  \
_Pragma("unroll")  \
      for (int w = 0; w < 4; ++w)  \
_Pragma("unroll")  \
        for (int i = 0; i < TM; ++i)  \
_Pragma("unroll")  \
          for (int j = 0; j < TN; ++j)  \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][w], b[j][w], acc[i][j], 0, 0, 0);  \
    }  \
  }
  // MAIN: slab KT out of buffer CUR (a literal: every LDS address is a per-thread base register plus an immediate -- with a
  // run-time buffer index the addresses were recomputed on the vector ALU every slab, a dozen VALU instructions that the
  // MFMAs do not hide); the next full slab goes straight into the other buffer (glds; its readers passed the previous
  // barrier) or into registers while this one is multiplied, and to LDS afterwards.  The barrier also drains the glds
  // queue (vmcnt): the slab must have landed before anyone reads it.
#define EBN_SLAB(CUR, KT)  \
  {  \
    const bool next__ = (KT) + 1 < nk_main;  \
    if (next__) EBN_FETCH_FULL((CUR) ^ 1);  \
    EBN_MMA(CUR)  \
    if (next__) EBN_STORE_FULL((CUR) ^ 1);  \
    __syncthreads();  \
  }
  for (int kt = 0; kt < nk_main; kt += 2) {
    EBN_SLAB(0, kt);
    if (kt + 1 < nk_main) EBN_SLAB(1, kt + 1);
  }
#undef EBN_SLAB
  // REST
  if (nk_main < nk) {
    int cur = nk_main & 1;
    EBN_FETCH_PART(nk_main);
    EBN_STORE_PART(cur);
    __syncthreads();
    for (int kt = nk_main; kt < nk; ++kt) {
      if (kt + 1 < nk) EBN_FETCH_PART(kt + 1);
      EBN_MMA(cur)
      if (kt + 1 < nk) EBN_STORE_PART(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  const int64_t em0 = m0, en0 = n0, ezs = zsplit;

  // epilogue. C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  if constexpr (EPI == 1 || EPI == 2) {
    // Epilogues that READ (the rank-1 operands, the bias): all reads of the 16 elements of a 32x32 tile are issued first,
    // unconditionally and with clamped indices; validity only guards the stores.  Reads placed next to the per-element
    // guards are serialised -- one global round trip per element, 128 per lane for the rank-1 form.  (The plain
    // epilogue below is kept as it was: restructuring it costs the big projections 2 % through register allocation.)
    const bool split = GZ > 1;
    float* out = split ? (Cpart + ezs * M * N) : C;
    const int64_t ldo = split ? N : ldc;
    const bool read_c = !split && beta != 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int64_t col = en0 + wn * WTN + j * 32 + (lane & 31);
        const bool col_ok = col < N;
        const int64_t colc = col_ok ? col : N - 1;
        const int64_t row_base = em0 + wm * WTM + i * 32 + 4 * (lane >> 5);
        float add[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] = 0.f;
        if (EPI == 1 && !split) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = row_base + (r & 3) + 8 * (r >> 2);
            const uint32_t rowc = static_cast<uint32_t>(row < M ? row : M - 1);  // M < 2^31 rows: 32-bit division
            add[r] = epi.rs[rowc] * epi.cv[static_cast<int64_t>(rowc / static_cast<uint32_t>(epi.L)) * epi.ldcv + colc];
          }
        }
        if (EPI == 2 && !split) {
          const float bv = epi.bias[colc];
#pragma unroll
          for (int r = 0; r < 16; ++r) add[r] = bv;
        }
        if (read_c) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = row_base + (r & 3) + 8 * (r >> 2);
            add[r] = fmaf(beta, out[(row < M ? row : M - 1) * ldo + colc], add[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row_base + (r & 3) + 8 * (r >> 2);
          float v = fmaf(alpha, acc[i][j][r], add[r]);
          if (EPI == 2 && !split) v = fmaxf(v, 0.f);
          if (col_ok && row < M) out[row * ldo + col] = v;
        }
      }
    }

  } else {
    const bool split = GZ > 1;
    float* out = split ? (Cpart + ezs * M * N) : C;
    const int64_t ldo = split ? N : ldc;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int64_t col = en0 + wn * WTN + j * 32 + (lane & 31);
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = em0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row >= M) continue;
          float v = alpha * acc[i][j][r];
          if (!split && beta != 0.f) v += beta * out[row * ldo + col];
          if (EPI == 1 && !split)  // 32-bit division: M < 2^31 rows
            v = fmaf(epi.rs[row], epi.cv[static_cast<int64_t>(static_cast<uint32_t>(row) / static_cast<uint32_t>(epi.L)) * epi.ldcv + col], v);
          if (EPI == 2 && !split) v = fmaxf(v + epi.bias[col], 0.f);
#if defined(EBN_GEMM_EXP_NOSTORE)  /* tuning experiment (tools/build_variant.sh): the tile's MFMAs without its C stores */
          if (v == 1.2345e38f) out[row * ldo + col] = v;
#elif defined(EBN_GEMM_EXP_NT)     /* tuning experiment: non-temporal C stores */
          __builtin_nontemporal_store(v, &out[row * ldo + col]);
#else
          out[row * ldo + col] = v;
#endif
        }
      }
    }

  }
  }
#undef EBN_MMA
#undef EBN_FETCH_FULL
#undef EBN_FETCH_PART
#undef EBN_STORE_FULL
#undef EBN_STORE_PART
#undef EBN_TAIL_PIECE
#undef EBN_GLDS_A_DST
#undef EBN_GLDS_B_DST
#undef EBN_AS
#undef EBN_BS
#undef EBN_STORE_SLAB
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits,
                                                            int64_t M, int64_t N, float beta,
                                                            float* __restrict__ C, int64_t ldc, GemmEpi epi) {
  // M * N < 2^31 (checked by the launcher): 32-bit index arithmetic -- a 64-bit division per element costs more than
  // the sum itself
  ebn_splitk_sum_body(blockIdx.x * 256u + threadIdx.x, gridDim.x * 256u, part, splits, static_cast<uint32_t>(M * N), static_cast<uint32_t>(N),
                      beta, C, ldc, epi.rs, epi.cv, epi.ldcv, epi.L, epi.bias);
}

template <int BM, int BN, int WAVES_M>
int launch_gemm(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                int64_t lda, const float* B, int64_t ldb, float beta, float* C, int64_t ldc, int vecA,
                int vecB, int splits, int64_t k_per_split, float* part, hipStream_t s, int site, GemmEpi epi) {
  dim3 grid(static_cast<unsigned>(ebn_ceil_div(N, BN)), static_cast<unsigned>(ebn_ceil_div(M, BM)),
            static_cast<unsigned>(splits));
  dim3 block(GEMM_THREADS);
#ifdef EBN_GEMM_EXP_STAGGER
  static const uint32_t stagger_pct = getenv("EBN_GEMM_STAGGER_PCT") ? static_cast<uint32_t>(atoi(getenv("EBN_GEMM_STAGGER_PCT"))) : 0u;
  epi.stagger_pct = splits == 1 ? stagger_pct : 0u;
#endif
#define EBN_GEMM_LAUNCH(TA, TB)                                                                            \
  do {                                                                                                    \
    if (vecA && vecB)                                                                                     \
      EBN_LAUNCH((gemm_f32_kernel<BM, BN, WAVES_M, TA, TB, true>), grid, block, 0, s, M, N, K, alpha, A, lda, \
                         B, ldb, beta, C, ldc, k_per_split, part, epi);                       \
    else if constexpr (BM == 64 && BN == 64) /* unaligned operands: the scalar-load form exists for the 64 x 64 tile only */ \
      EBN_LAUNCH((gemm_f32_kernel<BM, BN, WAVES_M, TA, TB, false>), grid, block, 0, s, M, N, K, alpha, A,  \
                         lda, B, ldb, beta, C, ldc, k_per_split, part, epi);                  \
    else                                                                                                  \
      return EBN_ERR_UNSUPPORTED;                                                                         \
  } while (0)
  if (epi.rowmap != nullptr) {  // caller guarantees !transA && !transB && vecA && vecB, no split-K; the 256 x 64 tile only
    if constexpr (BM == 256)
      EBN_LAUNCH((gemm_f32_kernel<BM, BN, WAVES_M, false, false, true, 3>), grid, block, 0, s, M, N, K, alpha, A, lda, B, ldb, beta,
                         C, ldc, k_per_split, part, epi);
    else
      return EBN_ERR_UNSUPPORTED;
  } else if (epi.bias != nullptr)  // caller guarantees !transA && !transB && vecA && vecB
    EBN_LAUNCH((gemm_f32_kernel<BM, BN, WAVES_M, false, false, true, 2>), grid, block, 0, s, M, N, K, alpha, A, lda,
                       B, ldb, beta, C, ldc, k_per_split, part, epi);
  else if (epi.rs != nullptr)  // caller guarantees !transA && transB && vecA && vecB
    EBN_LAUNCH((gemm_f32_kernel<BM, BN, WAVES_M, false, true, true, 1>), grid, block, 0, s, M, N, K, alpha, A, lda,
                       B, ldb, beta, C, ldc, k_per_split, part, epi);
  else if (!transA && !transB) EBN_GEMM_LAUNCH(false, false);
  else if (!transA && transB) EBN_GEMM_LAUNCH(false, true);
  else if (transA && !transB) EBN_GEMM_LAUNCH(true, false);
  else EBN_GEMM_LAUNCH(true, true);
#undef EBN_GEMM_LAUNCH
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}


// ---- small-output GEMMs: 32x32 block tile from v_mfma_f32_16x16x4_f32, 32-deep K slabs, one launch ---------------
// The user encoder (640 rows) and the NRMSDocVec MLP (800 rows) multiply matrices whose OUTPUT has only a few hundred
// 64x64 tiles: fewer workgroups than CUs.  The big-tile kernel above fills the chip by splitting K and paying a second
// (reduce) launch; at these sizes both launches sit at the ~5 us floor of a dependent chain.  Here a 256-thread
// workgroup owns a 32x32 tile (4 waves as 2x2, each one 16x16 MFMA tile), so a 640x1200 output is 760 workgroups and
// needs no split; a slab is 32 deep (fewer barriers per FLOP, 8 MFMAs per wave between them) and both operand tiles sit
// in LDS as [mn][k] with the lane's 8 contraction indices contiguous (k = 8*quarter + step, identical for A and B), i.e.
// two ds_read_b128 per operand and slab.  Exact fp32 like the big kernel (an fp32 fma chain per output element).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SBM = 32, SBN = 32;
constexpr int SBK = 128;       // slab depth: these GEMMs are load-latency bound (one or two workgroups per CU), so a slab
                               // carries 4 x 16 bytes per thread and operand -- 4x the bytes in flight of a 32-deep slab
constexpr int SLD = SBK + 4;   // LDS row stride (floats): 16-byte rows; the 16 rows of a b128 read group cover all 64 banks
constexpr int SLV = SBK + 8;   // row stride of the vector-path kernels' swizzled image (small_store_t)
constexpr int SPT = SBM * SBK / 4 / GEMM_THREADS;  // float4 per thread, operand and slab (4)

// One operand tile = SPT float4 per thread.  KC: memory is [mn][k] (k contiguous): item v -> row v/32, k 4*(v%32); else memory
// is [k][mn]: item v -> k v/8, mn 4*(v%8).  Out-of-range k is zero-filled (it feeds real outputs), out-of-range mn is clamped
// (it only feeds outputs that are never stored).  VEC: 16-byte loads (aligned base/ld, extents % 4 == 0).  All loads are
// unconditional with clamped offsets; validity is applied with selects.
template <bool KC, bool VEC>
__device__ __forceinline__ void small_load(float4 (&r)[SPT], const float* __restrict__ P, int64_t ld, int64_t mn0, int64_t k0,
                                           int64_t MN, int64_t Kend, int tid) {
#pragma unroll
  for (int i = 0; i < SPT; ++i) {
    const int v = tid + i * GEMM_THREADS;
    if (KC) {
      int64_t mn = mn0 + v / (SBK / 4);
      mn = mn < MN ? mn : MN - 1;
      const int64_t k = k0 + (v % (SBK / 4)) * 4;
      const float* p = P + mn * ld;
      if (VEC) {
        const bool ok = k < Kend;  // K % 4 == 0: a float4 is all in or all out
        const float4 t = *reinterpret_cast<const float4*>(p + (ok ? k : 0));
        r[i] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
      } else {
        const float t0 = p[k + 0 < Kend ? k + 0 : 0], t1 = p[k + 1 < Kend ? k + 1 : 0], t2 = p[k + 2 < Kend ? k + 2 : 0],
                    t3 = p[k + 3 < Kend ? k + 3 : 0];
        r[i] = make_float4(k + 0 < Kend ? t0 : 0.f, k + 1 < Kend ? t1 : 0.f, k + 2 < Kend ? t2 : 0.f, k + 3 < Kend ? t3 : 0.f);
      }
    } else {
      const int64_t k = k0 + v / (SBM / 4);
      const bool kok = k < Kend;
      const float* p = P + (kok ? k : 0) * ld;
      const int64_t mn = mn0 + (v % (SBM / 4)) * 4;
      if (VEC) {
        const float4 t = *reinterpret_cast<const float4*>(p + (mn + 3 < MN ? mn : MN - 4));  // MN % 4 == 0
        r[i] = make_float4(kok ? t.x : 0.f, kok ? t.y : 0.f, kok ? t.z : 0.f, kok ? t.w : 0.f);
      } else {
        const float t0 = p[mn + 0 < MN ? mn + 0 : MN - 1], t1 = p[mn + 1 < MN ? mn + 1 : MN - 1],
                    t2 = p[mn + 2 < MN ? mn + 2 : MN - 1], t3 = p[mn + 3 < MN ? mn + 3 : MN - 1];
        r[i] = make_float4(kok ? t0 : 0.f, kok ? t1 : 0.f, kok ? t2 : 0.f, kok ? t3 : 0.f);
      }
    }
  }
}

template <bool KC>
__device__ __forceinline__ void small_store(float* __restrict__ S, const float4 (&r)[SPT], int tid) {
#pragma unroll
  for (int i = 0; i < SPT; ++i) {
    const int v = tid + i * GEMM_THREADS;
    if (KC) {
      *reinterpret_cast<float4*>(&S[(v / (SBK / 4)) * SLD + (v % (SBK / 4)) * 4]) = r[i];
    } else {  // k = v/8, rows 4*(v%8)..+3: transposed scalar stores
      const int k = v / (SBM / 4), m4 = (v % (SBM / 4)) * 4;
      S[(m4 + 0) * SLD + k] = r[i].x;
      S[(m4 + 1) * SLD + k] = r[i].y;
      S[(m4 + 2) * SLD + k] = r[i].z;
      S[(m4 + 3) * SLD + k] = r[i].w;
    }
  }
}

template <bool TA, bool TB, bool VECA, bool VECB>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_small_kernel(int64_t M, int64_t N, int64_t K, float alpha,
                                                                  const float* __restrict__ A, int64_t lda,
                                                                  const float* __restrict__ B, int64_t ldb, float beta,
                                                                  float* __restrict__ C, int64_t ldc, GemmEpi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem_small[];  // [2 buffers][A tile | B tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = static_cast<int64_t>(blockIdx.y) * SBM, n0 = static_cast<int64_t>(blockIdx.x) * SBN;
  const int nk = static_cast<int>((K + SBK - 1) / SBK);
  constexpr int TILE = SBM * SLD;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float4 ra[SPT], rb[SPT];
  small_load<!TA, VECA>(ra, A, lda, m0, 0, M, K, tid);
  small_load<TB, VECB>(rb, B, ldb, n0, 0, N, K, tid);
  small_store<!TA>(smem_small, ra, tid);
  small_store<TB>(smem_small + TILE, rb, tid);
  __syncthreads();
  const int r16 = lane & 15, kq = lane >> 4;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {  // next slab into registers while this one is multiplied
      small_load<!TA, VECA>(ra, A, lda, m0, static_cast<int64_t>(kt + 1) * SBK, M, K, tid);
      small_load<TB, VECB>(rb, B, ldb, n0, static_cast<int64_t>(kt + 1) * SBK, N, K, tid);
    }
    const float* ap = smem_small + cur * 2 * TILE + (wm * 16 + r16) * SLD + kq * 8;
    const float* bp = smem_small + cur * 2 * TILE + TILE + (wn * 16 + r16) * SLD + kq * 8;
    const int groups = static_cast<int>((K - static_cast<int64_t>(kt) * SBK + 31) / 32);  // 32-deep groups with real k in them
#pragma unroll
    for (int g = 0; g < SBK / 32; ++g) {
      if (g >= groups) break;  // wave-uniform: the zero-filled tail of the last slab is not multiplied
      // contraction index of MFMA step (h, c) in lane quarter kq: k = 32 g + 16 h + 4 kq + c (the same for A and B)
      const float4 a0 = *reinterpret_cast<const float4*>(ap + 32 * g), a1 = *reinterpret_cast<const float4*>(ap + 32 * g + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(bp + 32 * g), b1 = *reinterpret_cast<const float4*>(bp + 32 * g + 4);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc, 0, 0, 0);
    }
    if (kt + 1 < nk) {
      small_store<!TA>(smem_small + (cur ^ 1) * 2 * TILE, ra, tid);
      small_store<TB>(smem_small + (cur ^ 1) * 2 * TILE + TILE, rb, tid);
    }
    __syncthreads();
    cur ^= 1;
  }
  // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
  const int64_t col = n0 + wn * 16 + r16;
  if (col >= N) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = m0 + wm * 16 + 4 * kq + r;
    if (row >= M) continue;
    float v = alpha * acc[r];
    if (beta != 0.f) v += beta * C[row * ldc + col];
    if (epi.rs != nullptr)
      v = fmaf(epi.rs[row], epi.cv[static_cast<int64_t>(static_cast<uint32_t>(row) / static_cast<uint32_t>(epi.L)) * epi.ldcv + col], v);
    if (epi.bias != nullptr) v = fmaxf(v + epi.bias[col], 0.f);
    C[row * ldc + col] = v;
  }
}

// The same tile for 16-byte-aligned operands (every call site of the hot path), rebuilt around what the step's small GEMMs
// are bound by -- the latency of a slab's loads, four to nine times per workgroup -- and around the fact that VALU work
// is not hidden by the fp32 MFMAs:
//   * TWO slabs of loads in flight: slab kt+2 is requested before slab kt is multiplied, slab kt+1 (requested an
//     iteration earlier) is written to LDS after it; the two register sets and the two LDS buffers alternate with literal
//     indices (two slabs per loop iteration);
//   * buffer loads with a constant 32-bit lane offset and a scalar slab offset: no per-slab address arithmetic;
//   * only the partial last slab carries k masks.
// T = 32: 256 threads, the tile of gemm_small_kernel.  (T = 64, a 64 x 64 tile of sixteen waves with half the operand
// re-streaming, was measured and is not instantiated: the 14 DocVec shapes took 224 us instead of 125 us.)
template <bool KC, int T>
__device__ __forceinline__ void small_store_t(float* __restrict__ S, const float4 (&r)[SBK / T], int tid) {
  constexpr int THREADS = T * T / 4;
#pragma unroll
  for (int i = 0; i < SBK / T; ++i) {
    const int v = tid + i * THREADS;
    // Column swizzle: column c of row r lives at c ^ 4 ((r >> 2) & 7) (whole float4s move), row stride 136 floats.  With the
    // reader's k mapping below this is free of bank conflicts for all three access forms (tools/lds/bank_sim.py, the lane groups
    // and bank moduli of MI355X_MICROARCH.md): the transposed scalar stores of a wave go to rows 4 (v % 8) + j, columns v / 8 +
    // const -- unswizzled with stride 132 that was banks 16 (v % 8) + 4 j + v / 8 mod 32, four lanes per bank (1.1 M / 1.9 M
    // conflict cycles per launch in profiles/r03_pmc) -- and the b128 operand reads were 2-way in every lane group.
    if (KC) {
      const int row = v / (SBK / 4);
      *reinterpret_cast<float4*>(&S[row * SLV + (((v % (SBK / 4)) * 4) ^ (((row >> 2) & 7) << 2))]) = r[i];
    } else {  // k = v / (T/4), rows 4*(v % (T/4))..+3: transposed scalar stores
      const int m4 = (v % (T / 4)) * 4;
      const int k = (v / (T / 4)) ^ (((m4 >> 2) & 7) << 2);
      S[(m4 + 0) * SLV + k] = r[i].x;
      S[(m4 + 1) * SLV + k] = r[i].y;
      S[(m4 + 2) * SLV + k] = r[i].z;
      S[(m4 + 3) * SLV + k] = r[i].w;
    }
  }
}

// FIN (ebn_dvn_finale_f32): Keras-form Adam on every element this tile produces (C and the column sums), see ebn_tn_finale.h.
template <bool TA, bool TB, int T, bool FIN = false>
__device__ __forceinline__ void small_vec_body(int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A,
                                               int64_t lda, const float* __restrict__ B, int64_t ldb, float beta,
                                               float* __restrict__ C, int64_t ldc, GemmEpi epi, int64_t tile_x, int64_t tile_y,
                                               const EbnTnFinale* fin = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float smem_small[];  // [2 buffers][A tile | B tile]
  constexpr int THREADS = T * T / 4, SPT = SBK / T, WPR = T / 16;  // float4 per thread, operand and slab; waves per tile row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WPR, wn = wave % WPR;
  const int64_t m0 = tile_y * T, n0 = tile_x * T;
  const int nk = static_cast<int>((K + SBK - 1) / SBK), nk_full = static_cast<int>(K / SBK);
  constexpr int TILE = T * SLV;
  constexpr bool A_KC = !TA, B_KC = TB;
  const i32x4n arsrc = make_rsrc(A_KC ? A + m0 * lda : A + m0), brsrc = make_rsrc(B_KC ? B + n0 * ldb : B + n0);
  const uint32_t step_a = static_cast<uint32_t>((A_KC ? SBK : SBK * lda) * 4), step_b = static_cast<uint32_t>((B_KC ? SBK : SBK * ldb) * 4);
  uint32_t oa[SPT], ob[SPT];  // lane byte offsets of the SPT float4 a thread owns per operand and slab
#pragma unroll
  for (int i = 0; i < SPT; ++i) {
    const int v = tid + i * THREADS;
    if (A_KC) {
      int64_t r = m0 + v / (SBK / 4);
      r = r < M ? r : M - 1;
      oa[i] = static_cast<uint32_t>(((r - m0) * lda + (v % (SBK / 4)) * 4) * 4);
    } else {
      int64_t c = m0 + (v % (T / 4)) * 4;
      c = c + 3 < M ? c : M - 4;
      oa[i] = static_cast<uint32_t>(((v / (T / 4)) * lda + (c - m0)) * 4);
    }
    if (B_KC) {
      int64_t r = n0 + v / (SBK / 4);
      r = r < N ? r : N - 1;
      ob[i] = static_cast<uint32_t>(((r - n0) * ldb + (v % (SBK / 4)) * 4) * 4);
    } else {
      int64_t c = n0 + (v % (T / 4)) * 4;
      c = c + 3 < N ? c : N - 4;
      ob[i] = static_cast<uint32_t>(((v / (T / 4)) * ldb + (c - n0)) * 4);
    }
  }
  // FOUR accumulators, fed round-robin: a 16x16x4 MFMA that depends on the previous one issues only when that one has
  // retired, and with one or two workgroups per CU there is no other wave to fill the gap -- a single chain of 32
  // dependent MFMAs per slab ran at less than half the matrix rate.  (Deterministic: the order of the sums is fixed.)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  float4 ra[2][SPT], rb[2][SPT];
  // column sums of a [K][N] B operand (the bias gradient next to a Dense kernel gradient): the first row of tiles adds up the
  // pieces it stages anyway -- thread t holds columns 4 (t % (T/4)) .. + 3 of the k rows t / (T/4) + i THREADS / (T/4)
  const bool do_colsum = !B_KC && epi.colsum != nullptr && tile_y == 0;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
  // slab KT -> register set SET; a piece at k >= K (partial last slab only) reads offset 0 and is replaced by zero
#define EBN_SM_LOAD(SET, KT)                                                                              \
  do {                                                                                                    \
    const uint32_t sa__ = static_cast<uint32_t>(KT) * step_a, sb__ = static_cast<uint32_t>(KT) * step_b;  \
    if ((KT) < nk_full) {                                                                                 \
      _Pragma("unroll") for (int i = 0; i < SPT; ++i) ra[SET][i] = bload4(arsrc, oa[i], sa__);            \
      _Pragma("unroll") for (int i = 0; i < SPT; ++i) rb[SET][i] = bload4(brsrc, ob[i], sb__);            \
    } else {                                                                                              \
      const int krem__ = static_cast<int>(K - static_cast<int64_t>(KT) * SBK);                            \
      _Pragma("unroll") for (int i = 0; i < SPT; ++i) {                                                   \
        const int v__ = tid + i * THREADS;                                                           \
        const bool oka__ = (A_KC ? (v__ % (SBK / 4)) * 4 : v__ / (T / 4)) < krem__;                     \
        const bool okb__ = (B_KC ? (v__ % (SBK / 4)) * 4 : v__ / (T / 4)) < krem__;                     \
        const float4 ta__ = bload4(arsrc, oka__ ? oa[i] + sa__ : 0u, 0u);                                 \
        const float4 tb__ = bload4(brsrc, okb__ ? ob[i] + sb__ : 0u, 0u);                                 \
        ra[SET][i] = make_float4(oka__ ? ta__.x : 0.f, oka__ ? ta__.y : 0.f, oka__ ? ta__.z : 0.f, oka__ ? ta__.w : 0.f); \
        rb[SET][i] = make_float4(okb__ ? tb__.x : 0.f, okb__ ? tb__.y : 0.f, okb__ ? tb__.z : 0.f, okb__ ? tb__.w : 0.f); \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)
#define EBN_SM_STORE(SET, BUF)                                            \
  do {                                                                    \
    small_store_t<A_KC, T>(smem_small + (BUF) * 2 * TILE, ra[SET], tid);       \
    small_store_t<B_KC, T>(smem_small + (BUF) * 2 * TILE + TILE, rb[SET], tid); \
    if (do_colsum) {                                                      \
      _Pragma("unroll") for (int i = 0; i < SPT; ++i) {                   \
        csum.x += rb[SET][i].x;                                           \
        csum.y += rb[SET][i].y;                                           \
        csum.z += rb[SET][i].z;                                           \
        csum.w += rb[SET][i].w;                                           \
      }                                                                   \
    }                                                                     \
  } while (0)
  const int r16 = lane & 15, kq = lane >> 4;
  // swizzled columns of a lane's two float4 per 32-deep group (ca: row wm * 16 + r16 of A's image, cb: row wn * 16 + r16 of B's)
  const int ca0 = (4 * kq) ^ ((((wm * 16 + r16) >> 2) & 7) << 2), ca1 = ca0 ^ 16;
  const int cb0 = (4 * kq) ^ ((((wn * 16 + r16) >> 2) & 7) << 2), cb1 = cb0 ^ 16;
  // A full slab multiplies its four 32-deep groups with the operand reads of group g + 1 issued BEFORE the MFMAs of group g
  // (explicit double buffer: left to itself the compiler reuses one register set and waits for every group's LDS reads
  // in front of its MFMAs -- with one wave per SIMD nothing else covers that latency).  The partial last slab stops at its
  // last non-empty group.
#define EBN_SM_READ(SET, g)                                                     \
  do {                                                                          \
    fa__[SET][0] = *reinterpret_cast<const float4*>(ap__ + 32 * (g) + ca0);     \
    fa__[SET][1] = *reinterpret_cast<const float4*>(ap__ + 32 * (g) + ca1);     \
    fb__[SET][0] = *reinterpret_cast<const float4*>(bp__ + 32 * (g) + cb0);     \
    fb__[SET][1] = *reinterpret_cast<const float4*>(bp__ + 32 * (g) + cb1);     \
  } while (0)
#define EBN_SM_MUL(SET)                                                                             \
  do {                                                                                              \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].x, fb__[SET][0].x, acc0, 0, 0, 0);     \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].y, fb__[SET][0].y, acc1, 0, 0, 0);     \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].z, fb__[SET][0].z, acc2, 0, 0, 0);     \
    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][0].w, fb__[SET][0].w, acc3, 0, 0, 0);     \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].x, fb__[SET][1].x, acc0, 0, 0, 0);     \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].y, fb__[SET][1].y, acc1, 0, 0, 0);     \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].z, fb__[SET][1].z, acc2, 0, 0, 0);     \
    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa__[SET][1].w, fb__[SET][1].w, acc3, 0, 0, 0);     \
  } while (0)
  // contraction index of MFMA step s in lane quarter kq: k = 32 g + 8 kq + s (the same for A and B)
#define EBN_SM_MMA(BUF, KT)                                                                                         \
  do {                                                                                                              \
    const float* ap__ = smem_small + (BUF) * 2 * TILE + (wm * 16 + r16) * SLV;                                      \
    const float* bp__ = smem_small + (BUF) * 2 * TILE + TILE + (wn * 16 + r16) * SLV;                               \
    float4 fa__[2][2], fb__[2][2];                                                                                  \
    if ((KT) < nk_full) {                                                                                           \
      EBN_SM_READ(0, 0);                                                                                            \
      EBN_SM_READ(1, 1);                                                                                            \
      __builtin_amdgcn_sched_barrier(0); /* keep the reads in front: the scheduler sinks them to their uses */     \
      EBN_SM_MUL(0);                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      EBN_SM_READ(0, 2);                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      EBN_SM_MUL(1);                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      EBN_SM_READ(1, 3);                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      EBN_SM_MUL(0);                                                                                                \
      EBN_SM_MUL(1);                                                                                                \
    } else {                                                                                                        \
      const int groups__ = static_cast<int>((K - static_cast<int64_t>(KT) * SBK + 31) / 32);                        \
      for (int g = 0; g < groups__; ++g) {                                                                          \
        EBN_SM_READ(0, g);                                                                                          \
        EBN_SM_MUL(0);                                                                                              \
      }                                                                                                             \
    }                                                                                                               \
  } while (0)
  EBN_SM_LOAD(0, 0);
  if (nk > 1) EBN_SM_LOAD(1, 1);
  EBN_SM_STORE(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // slab kt is in LDS buffer 0, register set 1 holds slab kt + 1, set 0 is free
    if (kt + 2 < nk) EBN_SM_LOAD(0, kt + 2);
    EBN_SM_MMA(0, kt);
    if (kt + 1 < nk) EBN_SM_STORE(1, 1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    // slab kt + 1 is in LDS buffer 1, register set 0 holds slab kt + 2, set 1 is free
    if (kt + 3 < nk) EBN_SM_LOAD(1, kt + 3);
    EBN_SM_MMA(1, kt + 1);
    if (kt + 2 < nk) EBN_SM_STORE(0, 0);
    __syncthreads();
  }
#undef EBN_SM_LOAD
#undef EBN_SM_STORE
#undef EBN_SM_MMA
#undef EBN_SM_READ
#undef EBN_SM_MUL
  const f32x4 acc = (acc0 + acc1) + (acc2 + acc3);
  if (do_colsum) {  // (block-uniform) fixed-order sum over the THREADS / (T/4) threads that share a column group; the tile
    constexpr int CG = T / 4, KR = THREADS / CG;  // buffers are free: the slab loop ends behind a barrier
    float* sc = smem_small;
    *reinterpret_cast<float4*>(&sc[(tid / CG) * (T + 4) + (tid % CG) * 4]) = csum;
    __syncthreads();
    if (tid < T && n0 + tid < N) {
      float t = 0.f;
      for (int r = 0; r < KR; ++r) t += sc[r * (T + 4) + tid];
      epi.colsum[n0 + tid] = t;
      if (FIN) ebn_adam_flat_apply(fin->adam, fin->adam.st->adam_alpha, (epi.colsum + n0 + tid) - fin->adam.grad, t);
    }
  }
  // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
  const int64_t col = n0 + wn * 16 + r16;
  if (col < N) {  // (no early return: the finale's workgroups go on to their share of the step's closing work)
    const float al = FIN ? fin->adam.st->adam_alpha : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = m0 + wm * 16 + 4 * kq + r;
      if (row >= M) continue;
      float v = alpha * acc[r];
      if (beta != 0.f) v += beta * C[row * ldc + col];
      if (epi.rs != nullptr)
        v = fmaf(epi.rs[row], epi.cv[static_cast<int64_t>(static_cast<uint32_t>(row) / static_cast<uint32_t>(epi.L)) * epi.ldcv + col], v);
      if (epi.bias != nullptr) v = fmaxf(v + epi.bias[col], 0.f);
      if (epi.l2w != nullptr) v = fmaf(epi.two_lambda, epi.l2w[row * ldc + col], v);
      C[row * ldc + col] = v;
      if (FIN) ebn_adam_flat_apply(fin->adam, al, (C + row * ldc + col) - fin->adam.grad, v);
    }
  }
}

template <bool TA, bool TB, int T>
__global__ __launch_bounds__(T * T / 4) void gemm_small_vec_kernel(int64_t M, int64_t N, int64_t K, float alpha,
                                                                      const float* __restrict__ A, int64_t lda,
                                                                      const float* __restrict__ B, int64_t ldb, float beta,
                                                                      float* __restrict__ C, int64_t ldc, GemmEpi epi) {
  small_vec_body<TA, TB, T>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, epi, blockIdx.x, blockIdx.y);
}

// TWO independent small-output GEMMs in one launch: the weight gradient dW = X^T.dY (TN) and the input gradient
// dX = dY.W^T (NT) of a Dense layer (or of the projections of the user encoder) read the same dY and do not depend on each
// other.  Each fills half of the chip on its own (a few hundred 32x32 tiles for 256 CUs x 2 resident workgroups) and
// costs a launch of the step's dependent chain; together they are one launch that fills it.
struct SmallProblem {
  int64_t M, N, K;
  float alpha;
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  float beta;
  float* C;
  int64_t ldc;
  int32_t tiles_x, tiles;  // 32x32 tiles per row of tiles, and in total
};

__global__ __launch_bounds__(GEMM_THREADS) void gemm_small_pair_kernel(SmallProblem p0, SmallProblem p1) {
  const GemmEpi none{nullptr, nullptr, 0, 1, nullptr};
  const int b = blockIdx.x;  // block-uniform branch: the workgroup runs one of the two bodies
  if (b < p0.tiles)
    small_vec_body<true, false, 32>(p0.M, p0.N, p0.K, p0.alpha, p0.A, p0.lda, p0.B, p0.ldb, p0.beta, p0.C, p0.ldc, none,
                                    b % p0.tiles_x, b / p0.tiles_x);
  else
    small_vec_body<false, true, 32>(p1.M, p1.N, p1.K, p1.alpha, p1.A, p1.lda, p1.B, p1.ldb, p1.beta, p1.C, p1.ldc, none,
                                    (b - p0.tiles) % p1.tiles_x, (b - p0.tiles) / p1.tiles_x);
}

// Up to EBN_TN_GROUP_MAX weight-gradient products C_i = A_i^T . B_i in one launch (ebn_gemm_tn_group_f32): every workgroup finds
// its problem from the running tile counts (scalar loop) and runs the TN body; optional bias-gradient column sums and L2 term.
struct SmallGroup {
  SmallProblem p[EBN_TN_GROUP_MAX];
  float* colsum[EBN_TN_GROUP_MAX];
  const float* l2w[EBN_TN_GROUP_MAX];
  float two_lambda[EBN_TN_GROUP_MAX];
  int32_t first[EBN_TN_GROUP_MAX + 1];  // first workgroup of problem i; first[n] = total
  int32_t n;
};

// T = 32: 256 threads per 32 x 32 tile; T = 64: 1024 threads (sixteen waves, one 16 x 16 block each) per 64 x 64 tile -- half the
// operand re-streaming (a group of Dense kernel gradients with K = 800 reads 210 MB through L2 on 32 x 32 tiles, 105 MB on 64 x 64),
// taken when the group's 64 x 64 tiles alone fill the chip.
// Workgroup -> tile order of the grouped launches: the hardware deals consecutive workgroups round-robin to the 8 XCDs (private L2s);
// remapped so that each XCD walks a CONTIGUOUS run of tiles -- the tiles of a run share their A column panel (K x T floats, read by
// tiles_x neighbours) and walk the B panels of one problem, so an operand slab is fetched into one L2 instead of eight (the group of
// c3's four Dense gradients moved 67 MB of fabric traffic for 17 MB of operands with the dispatch order as the tile order).  Bijective.
__device__ __forceinline__ int xcd_chunked_tile(int v, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = v & 7, idx = v >> 3;
  return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int T>
__global__ __launch_bounds__(T * T / 4) void gemm_small_tn_group_kernel(SmallGroup g) {
  const int b = xcd_chunked_tile(blockIdx.x, gridDim.x);
  int i = 0;
  while (i + 1 < g.n && b >= g.first[i + 1]) ++i;
  const SmallProblem& p = g.p[i];
  GemmEpi epi{nullptr, nullptr, 0, 1, nullptr};
  epi.colsum = g.colsum[i];
  epi.l2w = g.l2w[i];
  epi.two_lambda = g.two_lambda[i];
  const int t = b - g.first[i];
  small_vec_body<true, false, T>(p.M, p.N, p.K, p.alpha, p.A, p.lda, p.B, p.ldb, p.beta, p.C, p.ldc, epi, t % p.tiles_x, t / p.tiles_x);
}

// The finale of a one-rank NRMSDocVec step: tn_group's tiles with Adam in their epilogue, then -- dealt over the same workgroups -- the
// element-wise Adam over the ranges no tile owns and the user head's finishing sums with the batch loss.  See ebn_tn_finale.h.
static_assert(EBN_TN_FINALE_MAX_REST == EBN_DVN_FINALE_MAX_REST, "ebn_tn_finale.h / ebnerd_hip.h");
struct SmallGroupFin {
  SmallGroup g;
  EbnTnFinale f;
};

template <int T>
__global__ __launch_bounds__(T * T / 4) void gemm_small_tn_finale_kernel(SmallGroupFin gf) {
  extern __shared__ __attribute__((aligned(16))) float smem_small[];
  const SmallGroup& g = gf.g;
  const EbnTnFinale& f = gf.f;
  const int nwg = gridDim.x, tid = threadIdx.x, b = xcd_chunked_tile(blockIdx.x, nwg);
  constexpr int NT = T * T / 4;
  int i = 0;
  while (i + 1 < g.n && b >= g.first[i + 1]) ++i;
  const SmallProblem& p = g.p[i];
  GemmEpi epi{nullptr, nullptr, 0, 1, nullptr};
  epi.colsum = g.colsum[i];
  epi.l2w = g.l2w[i];
  epi.two_lambda = g.two_lambda[i];
  const int t = b - g.first[i];
  small_vec_body<true, false, T, true>(p.M, p.N, p.K, p.alpha, p.A, p.lda, p.B, p.ldb, p.beta, p.C, p.ldc, epi, t % p.tiles_x, t / p.tiles_x, &f);
  const float al = f.adam.st->adam_alpha;
  // element-wise Adam over the remaining ranges: the concatenated index space, one contiguous share per workgroup
  {
    const int64_t per = (f.rest_total + nwg - 1) / nwg, e0 = static_cast<int64_t>(b) * per;
    const int64_t e1 = e0 + per < f.rest_total ? e0 + per : f.rest_total;
    for (int64_t e = e0 + tid; e < e1; e += NT) {
      int64_t off = e;
      int r = 0;
      while (r + 1 < f.n_rest && off >= f.rest_len[r]) off -= f.rest_len[r++];
      off += f.rest_off[r];
      ebn_adam_flat_apply(f.adam, al, off, f.adam.grad[off]);
    }
  }
  // the user head's d(q) / d(b) sums over impressions and the batch loss (+ the L2 term), then Adam on d(q) / d(b)
  const int nh = (2 * f.A + 255) / 256 + 1;
  if (f.head_partials != nullptr && b < nh) {  // block-uniform
    __syncthreads();                            // the tile buffers are free (colsum's scratch included)
    ebn_user_head_finish_body(smem_small, b, nh, f.head_partials, f.B, f.A, f.dq, f.db, f.loss_rows, f.loss_out);
    if (b < nh - 1) {
      const int idx = b * 256 + tid;
      if (tid < 256 && idx < 2 * f.A) {
        const int sidx = idx / f.A, k = idx - sidx * f.A;
        float* gp = (sidx == 0 ? f.dq : f.db) + k;
        ebn_adam_flat_apply(f.adam, al, gp - f.adam.grad, *gp);  // (written by this thread a moment ago)
      }
    } else if (tid == 0 && f.n_l2 > 0) {  // loss += l2 * sum W^2, in dvn_dbn_apply_kernel's order
      float tot = 0.f;
      for (int l = 0; l < f.n_l2; ++l) {
        float tl = 0.f;
        for (int j = 0; j < f.l2_tiles[l]; ++j) tl += f.l2_part[l * f.l2_slots + j];
        tot += tl;
      }
      f.loss_out[0] += f.l2 * tot;
    }
  }
}

int launch_gemm_small(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                      const float* B, int64_t ldb, float beta, float* C, int64_t ldc, int vecA, int vecB, hipStream_t s,
                      GemmEpi epi) {
  const dim3 grid(static_cast<unsigned>(ebn_ceil_div(N, SBN)), static_cast<unsigned>(ebn_ceil_div(M, SBM))), block(GEMM_THREADS);
  constexpr size_t lds = static_cast<size_t>(2) * 2 * SBM * SLD * sizeof(float);  // 67.6 KB: above the 64 KB default limit
#define EBN_SMALL_ONE(TA, TB, VA, VB)                                                                                     \
  do {                                                                                                                     \
    static const hipError_t attr__ = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_small_kernel<TA, TB, VA, VB>), \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
    if (attr__ != hipSuccess) return static_cast<int>(attr__);                                                            \
    EBN_LAUNCH((gemm_small_kernel<TA, TB, VA, VB>), grid, block, lds, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, epi); \
  } while (0)
#define EBN_SMALL_VEC_T(TA, TB, T)                                                                                         \
  do {                                                                                                                     \
    constexpr size_t lds_t = static_cast<size_t>(2) * 2 * (T) * SLV * sizeof(float);                                       \
    static const hipError_t attr__ = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_small_vec_kernel<TA, TB, T>), \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_t)); \
    if (attr__ != hipSuccess) return static_cast<int>(attr__);                                                            \
    const dim3 grid_t(static_cast<unsigned>(ebn_ceil_div(N, T)), static_cast<unsigned>(ebn_ceil_div(M, T)));               \
    EBN_LAUNCH((gemm_small_vec_kernel<TA, TB, T>), grid_t, dim3((T) * (T) / 4), lds_t, s, M, N, K, alpha, A, lda, B, ldb, \
                       beta, C, ldc, epi);                                                                                 \
  } while (0)
#define EBN_SMALL_VEC(TA, TB) EBN_SMALL_VEC_T(TA, TB, 32)
  // 32-bit byte offsets inside a tile's resource: 32 rows (or K rows) x ld x 4 bytes
  const bool fits32 = (K + SBK) * (transA ? lda : 1) * 4 + 64 * lda * 4 < (int64_t{1} << 31) &&
                      (K + SBK) * (transB ? 1 : ldb) * 4 + 64 * ldb * 4 < (int64_t{1} << 31);
#define EBN_SMALL(TA, TB)                              \
  do {                                                 \
    if (vecA && vecB && fits32) EBN_SMALL_VEC(TA, TB); \
    else EBN_SMALL_ONE(TA, TB, false, false);          \
  } while (0)
  if (!transA && !transB) EBN_SMALL(false, false);
  else if (!transA && transB) EBN_SMALL(false, true);
  else if (transA && !transB) EBN_SMALL(true, false);
  else EBN_SMALL(true, true);
#undef EBN_SMALL
#undef EBN_SMALL_VEC
#undef EBN_SMALL_VEC_T
#undef EBN_SMALL_ONE
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

}  // namespace

// ---- planner ------------------------------------------------------------------------------------------------
// Picks the block tile and the split-K factor that minimise a small time model (microseconds):
//     t = slabs_per_wg * max(W * ts_full, ceil(W / R) * ts_lone) + t0   [+ t_reduce when splits > 1]
// * workgroups are dealt round-robin to 256 CUs and all cost the same: the busiest CU gets W = ceil(n_wg / 256) of
//   them (padding of M / N to the tile and the idle part of the last "turn" are what the tile choice trades);
// * ts_full = time of one 16-deep K slab of one tile on a CU at the kernel's steady-state MFMA rate (128x128:
//   590 us / (8 x 64 slabs) on the 24000x1200x1024 projection; 256x64: 514 us / (7 x 64); 64x64 tiles reach ~3/4 of
//   that rate); ts_lone = the same slab when the workgroup has the CU to itself and nothing hides its latencies;
//   R = workgroups a CU holds at once;
// * a split-K reduce costs a launch (~4.5 us) plus (splits + 1) passes over the output at ~2.5 TB/s.
// Calibrated against tools/gemm_probe.py on the 14 GEMM shapes of a c2 training step (profiles/r01_gemm_tuning.md).
struct GemmPlan {
  int bm, bn;
  int splits;
  int64_t kps;
  double cost;
};

static GemmPlan gemm_plan(int64_t M, int64_t N, int64_t K, int64_t ws_floats) {
  int64_t max_split = K / (4 * BK);  // keep >= 4 slabs per split
  if (max_split > 64) max_split = 64;
  const int64_t max_by_ws = (M * N > 0) ? ws_floats / (M * N) : 0;
  if (max_split > max_by_ws) max_split = max_by_ws;
  if (max_split < 1 || M * N >= (static_cast<int64_t>(1) << 31)) max_split = 1;  // the reduce kernel indexes in 32 bits
  static const struct {
    int bm, bn;
    double ts_full, ts_lone;
    int resident;
    int family;  // EBN_GEMM_FORCE_TILE value
  } kTiles[4] = {{128, 128, 1.15, 1.5, 3, 128}, {256, 64, 1.17, 1.5, 3, 256}, {64, 64, 0.32, 0.48, 4, 64}, {128, 64, 0.60, 0.8, 3, 96}};
  const double out_mb = static_cast<double>(M) * static_cast<double>(N) * 4e-6;
  GemmPlan best{128, 128, 1, ebn_ceil_div(K > 0 ? K : 1, BK) * BK, 1e300};
  {
    // 32x32 tiles, 128-deep slabs, never split (gemm_small_kernel): `W` workgroups on the busiest CU, 2 of them resident
    // (67 KB of LDS each); calibrated on tools/gemm_shapes_probe.py c2 / c3 / c4: 800x512x768 14.5 us, 640x400x1200 20.9 us,
    // 640x256x200 4.9 us, 1600x400x1200 39.7 us.
    // Only for outputs that leave the big tiles under-filled and K ranges a single workgroup can walk (no split-K here).
    const int64_t wgs = ebn_ceil_div(M, SBM) * ebn_ceil_div(N, SBN);
    const int64_t tiles64 = ebn_ceil_div(M, 64) * ebn_ceil_div(N, 64);
    if (tiles64 <= 256 && K <= 4096) {
      const int64_t W = ebn_ceil_div(wgs, 256);
      // two workgroups fit a CU (LDS): a pair shares it at ~1.6 us per 128-deep slab, a lone one takes ~1.1 us; from the
      // third workgroup of a CU on (a second round) the measured cost grows faster than that (1600x400x1200: 3.7 us)
      double per_slab = static_cast<double>(W / 2) * 1.6 + static_cast<double>(W % 2) * 1.1;
      if (W >= 3) per_slab *= 1.3;
      const double cost = static_cast<double>(ebn_ceil_div(K > 0 ? K : 1, SBK)) * per_slab + 3.0;
      // Taken outright inside the MEASURED envelope (K <= 1536, at most three workgroups on the busiest CU: the user-encoder
      // and DocVec shapes of c2 / c3 / c4): inside a training step (operands last touched a step ago, not re-read back to
      // back as in the shape probe) the under-filled big tiles -- one workgroup on some CUs, one slab of loads in flight --
      // lose to it even where the probe has them level (640x1200x400: 23.5 us in the step for 64x64, 13.8 us in the probe;
      // c2 step 1.355 -> 1.346 ms).  OUTSIDE the envelope (K up to 4096, or a fourth workgroup per CU) nothing was measured:
      // there it competes with the big tiles on modelled cost, with the same in-step bias (x 0.8) in its favour.
      // EBN_GEMM_FORCE_TILE = 32 | 64 | 128 | 256 still overrides.
      const bool measured = K <= 1536 && W <= 3;
      best = GemmPlan{32, 32, 1, ebn_ceil_div(K > 0 ? K : 1, SBK) * SBK, measured ? cost : 0.8 * cost};
      if (measured) return best;
    }
  }
  for (int t = 0; t < 4; ++t) {
    if (kTiles[t].bm == 256 && M < 256) continue;
    const int64_t tiles = ebn_ceil_div(M, kTiles[t].bm) * ebn_ceil_div(N, kTiles[t].bn);
    int64_t prev_s = 0;
    for (int64_t c = 1; c <= max_split; ++c) {
      const int64_t kps = ebn_ceil_div(ebn_ceil_div(K > 0 ? K : 1, c), BK) * BK;
      const int64_t sp = ebn_ceil_div(K > 0 ? K : 1, kps);  // effective split count for this K range
      if (sp == prev_s) continue;
      prev_s = sp;
      const int64_t W = ebn_ceil_div(tiles * sp, 256);
      const double full = static_cast<double>(W) * kTiles[t].ts_full;
      const double lone = static_cast<double>(ebn_ceil_div(W, kTiles[t].resident)) * kTiles[t].ts_lone;
      double cost = static_cast<double>(kps / BK) * (full > lone ? full : lone) + 3.0;
      if (sp > 1) cost += 4.5 + static_cast<double>(sp + 1) * out_mb / 2.5;
      // 64x64 tiles of a long split-K range: up to six co-resident workgroups per CU hide each other's per-slab fetch
      // latency (300x1200x24000: split 16 instead of 8, 168 -> 154 us; at K = 52800 342 -> 334 us); short ranges do not
      // repay the extra partials (400x200x24000: split 18 stays the best up to 54)
      if (kTiles[t].bm == 64 && sp > 1 && kps / BK >= 64 && W <= 6) cost *= 1.0 - 0.01 * static_cast<double>(W);
      if (cost < best.cost) best = GemmPlan{kTiles[t].bm, kTiles[t].bn, static_cast<int>(sp), kps, cost};
    }
  }
  return best;
}

extern "C" int64_t ebn_gemm_workspace_floats(int64_t M, int64_t N, int64_t K) {
  if (!ebn_dim_ok(M, N, K)) return 0;
  const GemmPlan p = gemm_plan(M, N, K, INT64_MAX / 4);
  int64_t slices = p.splits > 1 ? p.splits : 0;
  const int64_t tn = ebn_gemm_direct_tn_slices(M, N, K);  // (either operand layout may ask: the size covers the TN form's chunks)
  if (tn > slices) slices = tn;
  return ebn_sat_mul(slices * M, N);
}

extern "C" int ebn_gemm_plan(int64_t M, int64_t N, int64_t K, int64_t workspace_floats, int32_t* bm, int32_t* bn,
                             int32_t* splits) {
  EBN_REQUIRE(M >= 0 && N >= 0 && K >= 0 && bm && bn && splits, EBN_ERR_BAD_ARG);
  const GemmPlan p = gemm_plan(M, N, K, workspace_floats);
  *bm = p.bm;
  *bn = p.bn;
  *splits = p.splits;
  return EBN_OK;
}

static int gemm_dispatch(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                         int64_t lda, const float* B, int64_t ldb, float beta, float* C, int64_t ldc, float* workspace,
                         int64_t workspace_floats, int32_t site, GemmEpi epi, hipStream_t s, int32_t* defer_parts = nullptr) {
  // defer_parts != NULL (ebn_gemm_f32_partials): the product is left as *defer_parts dense [M][N] slices in `workspace` -- the
  // split-K partials without their combining launch, or the product itself as one slice -- for ebn_grad_finish_f32 to sum
  if (defer_parts != nullptr) {
    C = workspace;
    ldc = N;
    beta = 0.f;
    *defer_parts = 1;
  }
  // contiguous-axis extent must be a multiple of 4 too (K for k-contiguous operands, M/N otherwise)
  // ... and the fast tile fetch addresses an operand tile with 32-bit BYTE offsets from the tile's origin: 256 tile rows
  // (or the K range, for operands stored [K][mn]) times the leading dimension must stay below 4 GB
  const bool spanA = (transA ? (K + 16) * lda + 256 : 256 * lda + K + 16) * 4 < (int64_t{1} << 32);
  const bool spanB = (transB ? 256 * ldb + K + 16 : (K + 16) * ldb + 256) * 4 < (int64_t{1} << 32);
  const int vecA = ((lda % 4) == 0 && ebn_aligned16(A) && ((transA ? M : K) % 4) == 0 && spanA) ? 1 : 0;
  const int vecB = ((ldb % 4) == 0 && ebn_aligned16(B) && ((transB ? K : N) % 4) == 0 && spanB) ? 1 : 0;
  if ((epi.bias != nullptr || epi.rs != nullptr) && !(vecA && vecB)) return EBN_ERR_UNSUPPORTED;  // epilogue kernels are VEC only
  // tall output x small second operand: the LDS-free 16 x 16-block kernel, operand fragments straight from global memory
  // (ebn_gemm_direct.hip): A [M][K] by float4 (vecA), B [N][K] by float4 (vecB) or B [K][N] by dwords (no alignment needed)
  if (epi.bias == nullptr && epi.rs == nullptr && beta == 0.f && vecA && (vecB || !transB) &&
      ebn_gemm_direct_wanted(transA, transB, M, N, K)) {
    const int rc_dir = ebn_gemm_direct_launch(transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, s);
    if (rc_dir != EBN_ERR_UNSUPPORTED) return rc_dir;
  }
  // weight gradient of a tall product with a small, awkward output (AttLayer2 dW = Y^T.dpre): 16 x 16 blocks, K chunks across
  // workgroups, dense slices like any split-K product (ebn_gemm_direct.hip) -- combined below or left to ebn_grad_finish_f32
  if (transA && !transB && epi.bias == nullptr && epi.rs == nullptr && workspace != nullptr &&
      M * N < (int64_t{1} << 31)) {
    const int z = ebn_gemm_direct_tn_slices(M, N, K);
    if (z >= 2 && workspace_floats >= static_cast<int64_t>(z) * M * N) {
      const int rc_tn = ebn_gemm_direct_tn_launch(M, N, K, alpha, A, lda, B, ldb, workspace, s);
      if (rc_tn == EBN_OK) {
        if (defer_parts != nullptr) {
          *defer_parts = z;
          return EBN_OK;
        }
        int64_t grid = ebn_ceil_div(M * N, 256);
        if (grid > 4096) grid = 4096;
        EBN_LAUNCH(splitk_reduce_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, s, workspace, z, M, N, beta, C, ldc, epi);
        EBN_CHECK_LAUNCH();
        return EBN_OK;
      }
      if (rc_tn != EBN_ERR_UNSUPPORTED) return rc_tn;
    }
  }
  const GemmPlan plan = gemm_plan(M, N, K, workspace ? workspace_floats : 0);
  const int splits = plan.splits;
  const int64_t kps = plan.kps;
  int rc;
  if (plan.bm == 32)
    return launch_gemm_small(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, vecA, vecB, s, epi);
  if (!(vecA && vecB))  // unaligned operands (no call site of the hot path): the scalar-load kernels are built for the 64 x 64 tile only
    rc = launch_gemm<64, 64, 2>(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, vecA, vecB, splits, kps, workspace, s, site, epi);
  else if (plan.bm == 256)
    rc = launch_gemm<256, 64, 4>(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, vecA, vecB, splits, kps,
                                 workspace, s, site, epi);
  else if (plan.bm == 128 && plan.bn == 64)
    rc = launch_gemm<128, 64, 2>(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, vecA, vecB, splits, kps,
                                 workspace, s, site, epi);
  else if (plan.bm == 128)
    rc = launch_gemm<128, 128, 2>(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, vecA, vecB, splits, kps,
                                  workspace, s, site, epi);
  else
    rc = launch_gemm<64, 64, 2>(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, vecA, vecB, splits, kps,
                                workspace, s, site, epi);
  if (rc != EBN_OK) return rc;
  if (splits > 1 && defer_parts != nullptr) {
    *defer_parts = splits;
    return EBN_OK;
  }
  if (splits > 1) {
    int64_t grid = ebn_ceil_div(M * N, 256);
    if (grid > 4096) grid = 4096;
    EBN_LAUNCH(splitk_reduce_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, s, workspace,
                       splits, M, N, beta, C, ldc, epi);
    EBN_CHECK_LAUNCH();
  }
  return EBN_OK;
}

extern "C" int64_t ebn_gemm_partials_workspace_floats(int64_t M, int64_t N, int64_t K) {
  if (!ebn_dim_ok(M, N, K)) return 0;
  const GemmPlan p = gemm_plan(M, N, K, INT64_MAX / 4);
  int64_t slices = p.splits > 1 ? p.splits : 1;
  const int64_t tn = ebn_gemm_direct_tn_slices(M, N, K);
  if (tn > slices) slices = tn;
  return ebn_sat_mul(slices * M, N);
}

extern "C" int ebn_gemm_f32_partials(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                                     int64_t lda, const float* B, int64_t ldb, float* workspace, int64_t workspace_floats,
                                     int32_t* n_parts, ebn_stream_t stream) {
  EBN_REQUIRE(M > 0 && N > 0 && K >= 0 && n_parts, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(A && B && workspace && workspace_floats >= M * N, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N), EBN_ERR_BAD_ARG);
  return gemm_dispatch(transA, transB, M, N, K, alpha, A, lda, B, ldb, 0.0f, workspace, N, workspace, workspace_floats, 0,
                       GemmEpi{nullptr, nullptr, 0, 1, nullptr}, ebn_stream(stream), n_parts);
}

extern "C" int ebn_gemm_f32_site(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha,
                                 const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                                 int64_t ldc, float* workspace, int64_t workspace_floats, int32_t site,
                                 ebn_stream_t stream) {
  EBN_REQUIRE(M >= 0 && N >= 0 && K >= 0, EBN_ERR_BAD_ARG);
  if (M == 0 || N == 0) return EBN_OK;
  EBN_REQUIRE(A && B && C, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, EBN_ERR_BAD_ARG);
  return gemm_dispatch(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, workspace, workspace_floats, site,
                       GemmEpi{nullptr, nullptr, 0, 1, nullptr}, ebn_stream(stream));
}

namespace {
__global__ __launch_bounds__(256) void rank1_fill_kernel(float* __restrict__ C, int64_t ldc, int64_t M, int64_t N,
                                                         GemmEpi epi) {
  const int64_t total = M * N;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t row = i / N, col = i - row * N;
    C[row * ldc + col] = epi.rs[row] * epi.cv[(row / epi.L) * epi.ldcv + col];
  }
}
}  // namespace

extern "C" int ebn_gemm_f32_rank1(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                                  const float* B, int64_t ldb, float* C, int64_t ldc, const float* row_scale,
                                  const float* seq_rows, int64_t ld_seq, int32_t L, float* workspace,
                                  int64_t workspace_floats, ebn_stream_t stream) {
  EBN_REQUIRE(M >= 0 && N >= 0 && K >= 0 && L > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(M < (static_cast<int64_t>(1) << 31), EBN_ERR_UNSUPPORTED);
  if (M == 0 || N == 0) return EBN_OK;
  EBN_REQUIRE(A && B && C && row_scale && seq_rows, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(lda >= K && ldb >= K && ldc >= N && ld_seq >= N, EBN_ERR_BAD_ARG);
  const GemmEpi epi{row_scale, seq_rows, ld_seq, L, nullptr};
  hipStream_t s = ebn_stream(stream);
  const bool vec = (lda % 4) == 0 && ebn_aligned16(A) && (ldb % 4) == 0 && ebn_aligned16(B) && (K % 4) == 0;
  if (vec)
    return gemm_dispatch(0, 1, M, N, K, alpha, A, lda, B, ldb, 0.0f, C, ldc, workspace, workspace_floats, 0, epi, s);
  // unaligned operands: write the rank-1 term, then accumulate the product onto it with the scalar-load kernels
  int64_t grid = ebn_ceil_div(M * N, 256);
  if (grid > 4096) grid = 4096;
  EBN_LAUNCH(rank1_fill_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, s, C, ldc, M, N, epi);
  EBN_CHECK_LAUNCH();
  return gemm_dispatch(0, 1, M, N, K, alpha, A, lda, B, ldb, 1.0f, C, ldc, workspace, workspace_floats, 0,
                       GemmEpi{nullptr, nullptr, 0, 1, nullptr}, s);
}

extern "C" int ebn_dense_relu_fwd_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                                      int64_t ldb, const float* bias, float* C, int64_t ldc, float* workspace,
                                      int64_t workspace_floats, ebn_stream_t stream) {
  EBN_REQUIRE(M >= 0 && N >= 0 && K >= 0, EBN_ERR_BAD_ARG);
  if (M == 0 || N == 0) return EBN_OK;
  EBN_REQUIRE(A && B && C && bias, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(lda >= K && ldb >= N && ldc >= N, EBN_ERR_BAD_ARG);
  hipStream_t s = ebn_stream(stream);
  const bool vec = (lda % 4) == 0 && ebn_aligned16(A) && (K % 4) == 0 && (ldb % 4) == 0 && ebn_aligned16(B) && (N % 4) == 0;
  if (vec)
    return gemm_dispatch(0, 0, M, N, K, 1.0f, A, lda, B, ldb, 0.0f, C, ldc, workspace, workspace_floats, 0,
                         GemmEpi{nullptr, nullptr, 0, 1, bias}, s);
  // unaligned operands: plain product with the scalar-load kernels, then bias + ReLU in place
  const int rc = gemm_dispatch(0, 0, M, N, K, 1.0f, A, lda, B, ldb, 0.0f, C, ldc, workspace, workspace_floats, 0,
                               GemmEpi{nullptr, nullptr, 0, 1, nullptr}, s);
  if (rc != EBN_OK) return rc;
  for (int64_t r = 0; ldc != N && r < M; ++r) {  // strided C: row by row (rare)
    const int rr = ebn_bias_relu_f32(C + r * ldc, bias, C + r * ldc, 1, static_cast<int32_t>(N), stream);
    if (rr != EBN_OK) return rr;
  }
  return (ldc == N) ? ebn_bias_relu_f32(C, bias, C, M, static_cast<int32_t>(N), stream) : EBN_OK;
}

extern "C" int ebn_gemm_f32_ws(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha,
                               const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
                               float* C, int64_t ldc, float* workspace, int64_t workspace_floats,
                               ebn_stream_t stream) {
  return ebn_gemm_f32_site(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, workspace, workspace_floats, 0,
                           stream);
}

extern "C" int ebn_dense_bwd_pair_f32(int64_t R, int64_t K_in, int64_t N_out, const float* X, int64_t ldx, const float* dY,
                                      int64_t lddy, const float* W, int64_t ldw, float beta_w, float* dW, int64_t lddw,
                                      float* dX, int64_t lddx, float* workspace, int64_t workspace_floats,
                                      ebn_stream_t stream) {
  EBN_REQUIRE(R >= 0 && K_in >= 0 && N_out >= 0, EBN_ERR_BAD_ARG);
  if (K_in == 0 || N_out == 0) return EBN_OK;
  EBN_REQUIRE(X && dY && W && dW && dX, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(ldx >= K_in && lddy >= N_out && ldw >= N_out && lddw >= N_out && lddx >= K_in, EBN_ERR_BAD_ARG);
  hipStream_t s = ebn_stream(stream);
  // one launch when both GEMMs are small-output shapes with 16-byte-aligned operands; two launches otherwise
  const bool aligned = (ldx % 4) == 0 && (lddy % 4) == 0 && (ldw % 4) == 0 && ebn_aligned16(X) && ebn_aligned16(dY) &&
                       ebn_aligned16(W) && (K_in % 4) == 0 && (N_out % 4) == 0;
  const int64_t span = (R + SBK) * (ldx > lddy ? ldx : lddy) * 4 + 64 * ((ldx > lddy ? ldx : lddy) > ldw ? (ldx > lddy ? ldx : lddy) : ldw) * 4;
  const bool small = R > 0 && gemm_plan(K_in, N_out, R, workspace ? workspace_floats : 0).bm == 32 &&
                     gemm_plan(R, K_in, N_out, workspace ? workspace_floats : 0).bm == 32;
  if (aligned && small && span < (int64_t{1} << 31)) {
    SmallProblem p0{K_in, N_out, R, 1.0f, X, ldx, dY, lddy, beta_w, dW, lddw, static_cast<int32_t>(ebn_ceil_div(N_out, SBN)), 0};
    p0.tiles = p0.tiles_x * static_cast<int32_t>(ebn_ceil_div(K_in, SBM));
    SmallProblem p1{R, K_in, N_out, 1.0f, dY, lddy, W, ldw, 0.0f, dX, lddx, static_cast<int32_t>(ebn_ceil_div(K_in, SBN)), 0};
    p1.tiles = p1.tiles_x * static_cast<int32_t>(ebn_ceil_div(R, SBM));
    constexpr size_t lds = static_cast<size_t>(2) * 2 * SBM * SLV * sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_small_pair_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (attr != hipSuccess) return static_cast<int>(attr);
    EBN_LAUNCH(gemm_small_pair_kernel, dim3(static_cast<unsigned>(p0.tiles + p1.tiles)), dim3(GEMM_THREADS), lds, s, p0, p1);
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  const GemmEpi none{nullptr, nullptr, 0, 1, nullptr};
  if (R == 0) {  // empty contraction: dW <- beta_w * dW, nothing to write for dX
    return gemm_dispatch(1, 0, K_in, N_out, 0, 1.0f, X, ldx, dY, lddy, beta_w, dW, lddw, workspace, workspace_floats, 0, none, s);
  }
  const int rc = gemm_dispatch(1, 0, K_in, N_out, R, 1.0f, X, ldx, dY, lddy, beta_w, dW, lddw, workspace, workspace_floats, 0, none, s);
  if (rc != EBN_OK) return rc;
  return gemm_dispatch(0, 1, R, K_in, N_out, 1.0f, dY, lddy, W, ldw, 0.0f, dX, lddx, workspace, workspace_floats, 0, none, s);
}

extern "C" int ebn_gemm_f32_rowmap(const int32_t* ids, int64_t table_rows, int64_t M, int64_t N, int64_t K, const float* table,
                                   int64_t ldt, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t* oob_flag, ebn_stream_t stream) {
  EBN_REQUIRE(ebn_dim_ok(M, N, K) && table_rows > 0, EBN_ERR_BAD_ARG);
  if (M == 0 || N == 0) return EBN_OK;
  EBN_REQUIRE(ids && table && B && C && ldt >= K && ldb >= N && ldc >= N && K > 0, EBN_ERR_BAD_ARG);
  // the 256 x 64 tile with both operands fetched straight into LDS: aligned operands, whole tiles of rows, table offsets in 32 bits
  const bool ok = M >= 256 && (ldt % 4) == 0 && (ldb % 4) == 0 && (K % 4) == 0 && (N % 4) == 0 && ebn_aligned16(table) && ebn_aligned16(B) &&
                  (table_rows * ldt + K + 16) * 4 < (int64_t{1} << 32) && ((K + 16) * ldb + 256) * 4 < (int64_t{1} << 32);
  if (!ok) return EBN_ERR_UNSUPPORTED;
  GemmEpi epi{nullptr, nullptr, 0, 1, nullptr};
  epi.rowmap = ids;
  epi.rowmap_rows = table_rows;
  epi.oob = oob_flag;
  return launch_gemm<256, 64, 4>(0, 0, M, N, K, 1.0f, table, ldt, B, ldb, 0.0f, C, ldc, 1, 1, 1, ebn_ceil_div(K, BK) * BK, nullptr,
                                 ebn_stream(stream), 0, epi);
}

extern "C" int ebn_gemm_tn_group_f32(const ebn_tn_problem* problems, int32_t n, ebn_stream_t stream) {
  EBN_REQUIRE(problems != nullptr && n >= 1 && n <= EBN_TN_GROUP_MAX, EBN_ERR_BAD_ARG);
  int64_t tiles64 = 0;
  for (int i = 0; i < n; ++i) {
    const ebn_tn_problem& q = problems[i];
    EBN_REQUIRE(ebn_dim_ok(q.M, q.N, q.K) && q.K >= 1, EBN_ERR_BAD_ARG);
    if (q.M == 0 || q.N == 0) continue;
    EBN_REQUIRE(q.A && q.B && q.C, EBN_ERR_BAD_ARG);
    EBN_REQUIRE(q.lda >= q.M && q.ldb >= q.N && q.ldc >= q.N, EBN_ERR_BAD_ARG);
    EBN_REQUIRE((q.lda % 4) == 0 && (q.ldb % 4) == 0 && (q.M % 4) == 0 && (q.N % 4) == 0 && ebn_aligned16(q.A) && ebn_aligned16(q.B),
                EBN_ERR_UNSUPPORTED);
    EBN_REQUIRE((q.K + SBK) * (q.lda > q.ldb ? q.lda : q.ldb) * 4 + 64 * 4 < (int64_t{1} << 31), EBN_ERR_UNSUPPORTED);
    tiles64 += ebn_ceil_div(q.M, 64) * ebn_ceil_div(q.N, 64);
  }
  const int T = tiles64 >= 192 ? 64 : 32;  // 64 x 64 tiles once they occupy most of the 256 CUs on their own
  SmallGroup g{};
  int64_t total = 0;
  g.n = 0;
  for (int i = 0; i < n; ++i) {
    const ebn_tn_problem& q = problems[i];
    if (q.M == 0 || q.N == 0) continue;
    const int j = g.n++;
    g.p[j] = SmallProblem{q.M, q.N, q.K, 1.0f, q.A, q.lda, q.B, q.ldb, 0.0f, q.C, q.ldc, static_cast<int32_t>(ebn_ceil_div(q.N, T)), 0};
    g.p[j].tiles = g.p[j].tiles_x * static_cast<int32_t>(ebn_ceil_div(q.M, T));
    g.colsum[j] = q.colsum;
    g.l2w[j] = q.l2_W;
    g.two_lambda[j] = q.two_lambda;
    g.first[j] = static_cast<int32_t>(total);
    total += g.p[j].tiles;
    EBN_REQUIRE(total < (int64_t{1} << 30), EBN_ERR_UNSUPPORTED);
  }
  if (g.n == 0) return EBN_OK;
  g.first[g.n] = static_cast<int32_t>(total);
#define EBN_TN_GROUP(TT)                                                                                                          \
  do {                                                                                                                            \
    constexpr size_t lds = static_cast<size_t>(2) * 2 * (TT) * SLV * sizeof(float);                                               \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_small_tn_group_kernel<TT>),            \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));        \
    if (attr != hipSuccess) return static_cast<int>(attr);                                                                        \
    EBN_LAUNCH(gemm_small_tn_group_kernel<TT>, dim3(static_cast<unsigned>(total)), dim3((TT) * (TT) / 4), lds, ebn_stream(stream), g); \
  } while (0)
  if (T == 64) EBN_TN_GROUP(64);
  else EBN_TN_GROUP(32);
#undef EBN_TN_GROUP
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

int ebn_tn_group_finale_launch(const ebn_tn_problem* problems, int32_t n, const EbnTnFinale& fin, hipStream_t s) {
  EBN_REQUIRE(problems != nullptr && n >= 1 && n <= EBN_TN_GROUP_MAX, EBN_ERR_BAD_ARG);
  int64_t tiles64 = 0;
  for (int i = 0; i < n; ++i) {
    const ebn_tn_problem& q = problems[i];
    EBN_REQUIRE(ebn_dim_ok(q.M, q.N, q.K) && q.K >= 1 && q.M > 0 && q.N > 0, EBN_ERR_BAD_ARG);
    EBN_REQUIRE(q.A && q.B && q.C, EBN_ERR_BAD_ARG);
    EBN_REQUIRE(q.lda >= q.M && q.ldb >= q.N && q.ldc >= q.N, EBN_ERR_BAD_ARG);
    EBN_REQUIRE((q.lda % 4) == 0 && (q.ldb % 4) == 0 && (q.M % 4) == 0 && (q.N % 4) == 0 && ebn_aligned16(q.A) && ebn_aligned16(q.B),
                EBN_ERR_UNSUPPORTED);
    EBN_REQUIRE((q.K + SBK) * (q.lda > q.ldb ? q.lda : q.ldb) * 4 + 64 * 4 < (int64_t{1} << 31), EBN_ERR_UNSUPPORTED);
    tiles64 += ebn_ceil_div(q.M, 64) * ebn_ceil_div(q.N, 64);
  }
  const int T = tiles64 >= 192 ? 64 : 32;
  SmallGroupFin gf{};
  gf.f = fin;
  SmallGroup& g = gf.g;
  int64_t total = 0;
  g.n = n;
  for (int i = 0; i < n; ++i) {
    const ebn_tn_problem& q = problems[i];
    g.p[i] = SmallProblem{q.M, q.N, q.K, 1.0f, q.A, q.lda, q.B, q.ldb, 0.0f, q.C, q.ldc, static_cast<int32_t>(ebn_ceil_div(q.N, T)), 0};
    g.p[i].tiles = g.p[i].tiles_x * static_cast<int32_t>(ebn_ceil_div(q.M, T));
    g.colsum[i] = q.colsum;
    g.l2w[i] = q.l2_W;
    g.two_lambda[i] = q.two_lambda;
    g.first[i] = static_cast<int32_t>(total);
    total += g.p[i].tiles;
    EBN_REQUIRE(total < (int64_t{1} << 30), EBN_ERR_UNSUPPORTED);
  }
  g.first[n] = static_cast<int32_t>(total);
  EBN_REQUIRE(total >= (2 * fin.A + 255) / 256 + 1, EBN_ERR_UNSUPPORTED);  // the head's finishing blocks ride on the first workgroups
#define EBN_TN_FINALE(TT)                                                                                                         \
  do {                                                                                                                            \
    constexpr size_t lds = static_cast<size_t>(2) * 2 * (TT) * SLV * sizeof(float);                                               \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_small_tn_finale_kernel<TT>),           \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));        \
    if (attr != hipSuccess) return static_cast<int>(attr);                                                                        \
    EBN_LAUNCH(gemm_small_tn_finale_kernel<TT>, dim3(static_cast<unsigned>(total)), dim3((TT) * (TT) / 4), lds, s, gf);           \
  } while (0)
  if (T == 64) EBN_TN_FINALE(64);
  else EBN_TN_FINALE(32);
#undef EBN_TN_FINALE
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_gemm_f32(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha,
                            const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                            int64_t ldc, ebn_stream_t stream) {
  return ebn_gemm_f32_ws(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, nullptr, 0, stream);
}
