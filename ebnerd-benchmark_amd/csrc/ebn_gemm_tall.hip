// Exact-fp32 MFMA GEMM for TALL outputs with a narrow, awkward N: the two AttLayer2 products that run over every title token,
//   U  = Y . W          24000 x 200 x 400   (layers.py:65-66, K.dot(x, W))          -- NN
//   dY = dpre . W^T     24000 x 400 x 200   (its input gradient)                    -- NT
// The 32x32 tiles of ebn_gemm.hip run N = 200 as 256 columns and 400 as 448: 22-28 % of their MFMA work is padding, whatever the
// block tile (profiles/r03_tuning_notes.md: 46-56 us over four tile families for 28 us of MFMA work).  Here the output is tiled
// in 16 x 16 blocks of v_mfma_f32_16x16x4_f32 (the same 64 FLOP / clock / SIMD, bitwise an fp32 fma chain as well): N = 200 runs
// as 208, 400 as 400.
//
// Workgroup = 32 rows x up to 208 columns (2 x 13 blocks), 4 waves as 2 (row blocks) x 2 (7 + 6 column blocks); 750 workgroups
// for 24000 rows (2.93 per CU, all resident).  16-deep K slabs, double-buffered in LDS, fetched through registers (buffer loads:
// constant lane offsets + one scalar slab offset per operand, no vector-ALU work in the slab loop -- the fp32 MFMA does not
// overlap VALU instructions on gfx950).  LDS images: A (always [M][K]) and a [N][K] B as float4 (row, kq ^ ((row >> 2) & 3)) -- one conflict-free ds_read_b128
// feeds the four MFMA steps of a slab (k = 4 kq + step); a [K][N] B as rows of 212 floats (the four kq groups of a b32 read land
// in four different 16-bank windows).  Results leave through LDS as 448-byte row runs.
#include <stdlib.h>

#include "ebn_common.h"

typedef float ebn_tall_f32x4 __attribute__((ext_vector_type(4)));
typedef int ebn_tall_i32x4 __attribute__((ext_vector_type(4)));
__device__ ebn_tall_f32x4 ebn_tall_buffer_load_x4(ebn_tall_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TR = 32;                 // rows per workgroup
constexpr int TCT = 13;                // 16-column blocks per workgroup
constexpr int TCW = 7;                 // ... per wave (wave column 0: 7, wave column 1: 6)
constexpr int TBN = TCT * 16;          // 208
constexpr int TKS = 16;                // slab depth
constexpr int LDB_NN = TBN + 4;        // 212: 4 * 212 = 16 (mod 64)
constexpr int A_FLOATS = TR * TKS;     // 512
constexpr int B_FLOATS = TKS * LDB_NN; // 3392 >= 208 * 16
constexpr int BUF_FLOATS = A_FLOATS + B_FLOATS;
constexpr int B_VECS = TBN * TKS / 4;  // 832 float4 per B slab
constexpr int B_ROUNDS = (B_VECS + 255) / 256;  // 4
constexpr int OUT_LD = TCW * 16 + 4;   // 116: staging row of a wave's 16 x 112 block

__device__ __forceinline__ ebn_tall_i32x4 tall_rsrc(const float* base) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  ebn_tall_i32x4 r;
  r.x = static_cast<int>(static_cast<uint32_t>(a));
  r.y = static_cast<int>(static_cast<uint32_t>(a >> 32) & 0xFFFFu);
  r.z = -1;
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ float4 tall_load(ebn_tall_i32x4 r, uint32_t lane_bytes, uint32_t slab_bytes) {
  const ebn_tall_f32x4 t = ebn_tall_buffer_load_x4(r, static_cast<int>(lane_bytes), static_cast<int>(slab_bytes), 0);
  return make_float4(t.x, t.y, t.z, t.w);
}

struct TallRegs {
  float4 a;
  float4 b[B_ROUNDS];
};

struct TallPlan {        // per-thread constants of the slab fetch
  ebn_tall_i32x4 arsrc, brsrc;
  uint32_t oa, ob[B_ROUNDS];  // lane byte offsets into the resources
  uint32_t da, db[B_ROUNDS];  // LDS float offsets inside a buffer
  uint32_t ka, kb[B_ROUNDS];  // k of the piece inside a slab (tail slab: pieces at k >= krem are zero)
};

// full slab at scalar byte offsets (sa, sb): bare loads
__device__ __forceinline__ void tall_fetch(TallRegs& rg, const TallPlan& p, uint32_t sa, uint32_t sb) {
  rg.a = tall_load(p.arsrc, p.oa, sa);
#pragma unroll
  for (int i = 0; i < B_ROUNDS; ++i) rg.b[i] = tall_load(p.brsrc, p.ob[i], sb);
}
// last, partial slab: a float4 is all-in or all-out (K % 4 == 0); an out-of-range piece reads offset 0 and is replaced by zero
__device__ __forceinline__ void tall_fetch_tail(TallRegs& rg, const TallPlan& p, uint32_t sa, uint32_t sb, uint32_t krem) {
  {
    const bool ok = p.ka < krem;
    const float4 t = tall_load(p.arsrc, ok ? p.oa + sa : 0u, 0u);
    rg.a = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
  }
#pragma unroll
  for (int i = 0; i < B_ROUNDS; ++i) {
    const bool ok = p.kb[i] < krem;
    const float4 t = tall_load(p.brsrc, ok ? p.ob[i] + sb : 0u, 0u);
    rg.b[i] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
  }
}
// every thread stores every piece it holds: threads beyond the slab's piece count hold DUPLICATES of the first pieces (same
// address, same value), so the slab loop has no per-lane branches
template <int BUF>
__device__ __forceinline__ void tall_store(float* smem, const TallRegs& rg, const TallPlan& p) {
  float* d = smem + BUF * BUF_FLOATS;
  *reinterpret_cast<float4*>(d + p.da) = rg.a;
#pragma unroll
  for (int i = 0; i < B_ROUNDS; ++i) *reinterpret_cast<float4*>(d + p.db[i]) = rg.b[i];
}

// The MFMAs of one slab for a wave owning NC column blocks from block C0: step s contracts k = 4 kq + s.
template <int BUF, int NC, int C0, bool B_KC>
__device__ __forceinline__ void tall_mma(const float* smem, f32x4 (&acc)[NC], int wr, int ln, int kq) {
  const int sw = kq ^ ((ln >> 2) & 3);  // slot of float4 (row, kq) in the swizzled k-contiguous images (rows are 16-aligned here)
  const float* as = smem + BUF * BUF_FLOATS + ((16 * wr + ln) * 4 + sw) * 4;
  const float a[4] = {as[0], as[1], as[2], as[3]};
  float b[NC][4];
  if (B_KC) {
    const float* bs = smem + BUF * BUF_FLOATS + A_FLOATS + ((16 * C0 + ln) * 4 + sw) * 4;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      b[c][0] = bs[c * 256 + 0];
      b[c][1] = bs[c * 256 + 1];
      b[c][2] = bs[c * 256 + 2];
      b[c][3] = bs[c * 256 + 3];
    }
  } else {
    const float* bs = smem + BUF * BUF_FLOATS + A_FLOATS + (4 * kq) * LDB_NN + 16 * C0 + ln;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) b[c][s] = bs[s * LDB_NN + 16 * c];
  }
  // every operand read of the slab is issued before the first MFMA (the MFMAs then wait for them in order, lgkmcnt(n)): left to
  // itself the scheduler puts each read right in front of its two MFMAs and a full LDS round trip between every pair
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[c][s], acc[c], 0, 0, 0);
}

// The slab pipeline and the result store of a wave that owns NC column blocks from block C0 (the two wave columns run two
// instantiations: one scalar branch per workgroup, not per slab, and no accumulator copies where the paths would join).
template <int NC, int C0, bool B_KC>
__device__ __forceinline__ void tall_run(float* smem, const TallPlan& p, int64_t M, int64_t N, int64_t K, float alpha, float* __restrict__ C,
                                         int64_t ldc, int64_t ldb, int64_t m0, int64_t n0, int wave, int wr, int lane, int ln, int kq) {
  const uint32_t step_a = TKS * 4;
  const uint32_t step_b = static_cast<uint32_t>((B_KC ? TKS : TKS * ldb) * 4);
  f32x4 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_full = static_cast<int>(K / TKS);
  const uint32_t krem = static_cast<uint32_t>(K - static_cast<int64_t>(nk_full) * TKS);  // 0, 4, 8 or 12
  const int nk = nk_full + (krem ? 1 : 0);

  // Pipeline, TWO slabs deep: while slab t is multiplied out of LDS buffer t & 1, slab t + 1 is in flight into register stage
  // (t + 1) & 1 (requested a slab ago, written to LDS after this slab's MFMAs) and slab t + 2 is requested into stage t & 1.
  // One slab of MFMAs (2700 cycles per CU with three workgroups resident) does not cover the fetch latency, two do.
  TallRegs st0, st1;
#define EBN_TALL_FETCH_ANY(RG, J)                                                                    \
  do {                                                                                               \
    if ((J) < nk_full) tall_fetch(RG, p, static_cast<uint32_t>(J) * step_a, static_cast<uint32_t>(J) * step_b); \
    else if ((J) < nk) tall_fetch_tail(RG, p, static_cast<uint32_t>(J) * step_a, static_cast<uint32_t>(J) * step_b, krem); \
  } while (0)
  EBN_TALL_FETCH_ANY(st0, 0);
  EBN_TALL_FETCH_ANY(st1, 1);
  tall_store<0>(smem, st0, p);
  __syncthreads();
  int t = 0;
  uint32_t sa = 2 * step_a, sb = 2 * step_b;  // scalar offsets of slab t + 2
  for (; t + 3 < nk_full; t += 2) {            // slabs t + 2 and t + 3 are full slabs: bare loads, literal buffers
    tall_fetch(st0, p, sa, sb);
    tall_mma<0, NC, C0, B_KC>(smem, acc, wr, ln, kq);
    tall_store<1>(smem, st1, p);
    __syncthreads();
    tall_fetch(st1, p, sa + step_a, sb + step_b);
    tall_mma<1, NC, C0, B_KC>(smem, acc, wr, ln, kq);
    tall_store<0>(smem, st0, p);
    __syncthreads();
    sa += 2 * step_a;
    sb += 2 * step_b;
  }
  for (; t < nk; ++t) {  // the last slabs: nothing, a full slab or the partial slab left to request
    if ((t & 1) == 0) {
      EBN_TALL_FETCH_ANY(st0, t + 2);
      tall_mma<0, NC, C0, B_KC>(smem, acc, wr, ln, kq);
      if (t + 1 < nk) tall_store<1>(smem, st1, p);
    } else {
      EBN_TALL_FETCH_ANY(st1, t + 2);
      tall_mma<1, NC, C0, B_KC>(smem, acc, wr, ln, kq);
      if (t + 1 < nk) tall_store<0>(smem, st0, p);
    }
    __syncthreads();
  }
#undef EBN_TALL_FETCH_ANY

  // ---- results: C/D layout of the 16x16 MFMA is lane (col = lane & 15, row group = lane >> 4), register r <-> row 4 rg + r.
  // Each wave stages its 16 x 112 block in LDS (the slab buffers are free: every wave passed the last barrier) and stores it
  // as float4 rows.
  float* stg = smem + wave * (16 * OUT_LD);
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) stg[(4 * kq + r) * OUT_LD + 16 * c + ln] = alpha * acc[c][r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr int vpr = NC * 4;  // float4 per row of this wave's block
  const int64_t col0 = n0 + 16 * C0;
  for (int idx = lane; idx < 16 * vpr; idx += 64) {
    const int r = idx / vpr, f4 = idx - r * vpr;
    const int64_t row = m0 + 16 * wr + r, col = col0 + 4 * f4;
    if (row < M && col < N) *reinterpret_cast<float4*>(C + row * ldc + col) = *reinterpret_cast<const float4*>(stg + r * OUT_LD + 4 * f4);
  }
}

// B_KC: B stored [N][K] (k contiguous); else [K][N].
template <bool B_KC>
__global__ __launch_bounds__(256, 3) void gemm_tall16_kernel(int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A,
                                                             int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                             float* __restrict__ C, int64_t ldc) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF_FLOATS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the 7-block wave column alternates between the even and the odd SIMD pair from one workgroup to the next
  const int wr = wave >> 1, wc = (wave ^ static_cast<int>(blockIdx.x)) & 1;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * TR;
  const int64_t n0 = static_cast<int64_t>(blockIdx.y) * TBN;
  const int ln = lane & 15, kq = lane >> 4;

  TallPlan p;
  p.arsrc = tall_rsrc(A + m0 * lda);
  p.brsrc = tall_rsrc(B_KC ? B + n0 * ldb : B + n0);
  {
    const int m = (tid >> 2) & (TR - 1), q = tid & 3;  // threads 128.. duplicate threads 0..127
    int64_t row = m0 + m;
    row = row < M ? row : M - 1;
    p.oa = static_cast<uint32_t>(((row - m0) * lda + q * 4) * 4);
    p.da = (m * 4 + (q ^ ((m >> 2) & 3))) * 4;
    p.ka = q * 4;
  }
#pragma unroll
  for (int i = 0; i < B_ROUNDS; ++i) {
    int idx = tid + 256 * i;
    idx = idx < B_VECS ? idx : idx - B_VECS;  // the last round's spare threads duplicate the first pieces
    if (B_KC) {
      const int n = idx >> 2, q = idx & 3;
      int64_t row = n0 + n;
      row = row < N ? row : N - 1;
      p.ob[i] = static_cast<uint32_t>(((row - n0) * ldb + q * 4) * 4);
      p.db[i] = A_FLOATS + (n * 4 + (q ^ ((n >> 2) & 3))) * 4;
      p.kb[i] = q * 4;
    } else {
      const int k = idx / (TBN / 4), n4 = idx - k * (TBN / 4);
      int64_t col = n0 + n4 * 4;
      col = col < N ? col : N - 4;
      p.ob[i] = static_cast<uint32_t>((k * ldb + (col - n0)) * 4);
      p.db[i] = A_FLOATS + k * LDB_NN + n4 * 4;
      p.kb[i] = k;
    }
  }
  if (wc == 0) tall_run<TCW, 0, B_KC>(smem, p, M, N, K, alpha, C, ldc, ldb, m0, n0, wave, wr, lane, ln, kq);
  else tall_run<TCT - TCW, TCW, B_KC>(smem, p, M, N, K, alpha, C, ldc, ldb, m0, n0, wave, wr, lane, ln, kq);
}

int tall_mode() {  // EBN_GEMM_TALL = 0: never (validation / tuning); 2: also the [N][K] layouts; default 1: see ebn_gemm_tall_wanted
  static const int mode = [] { const char* e = getenv("EBN_GEMM_TALL"); return e ? atoi(e) : 1; }();
  return mode;
}

}  // namespace

// Shapes this kernel takes (the caller has checked 16-byte alignment of A, B and lda % 4 == ldb % 4 == 0): a tall output whose
// width the 64-wide tiles would pad by more than a tenth.  beta == 0 and no epilogue: the caller checks that too.
// Measured (24000 rows, MI355X): U = Y.W 44.8 us against 46.3-46.9 us on the 32x32 tiles, dY = dpre.W^T 51.6-52.5 against 49.8 us:
// with three waves per SIMD and a workgroup barrier every 900 MFMA cycles the kernel keeps the matrix pipe 61 % busy, the padded
// 32x32 kernels 69 % on 25 % more work.  So B stored [K][N] takes it, B stored [N][K] only on request (EBN_GEMM_TALL=2).
bool ebn_gemm_tall_wanted(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, const float* C, int64_t ldc) {
  if (transA || tall_mode() == 0 || (transB && tall_mode() != 2)) return false;
  if (M < 4096 || N < 64 || N > 2 * TBN || K < 64 || K > 2048) return false;
  if ((N % 4) || (K % 4) || (ldc % 4) || !ebn_aligned16(C)) return false;
  const int64_t pad64 = ebn_ceil_div(N, 64) * 64;
  return pad64 * 10 >= N * 11;
}

int ebn_gemm_tall_launch(int32_t transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B,
                         int64_t ldb, float* C, int64_t ldc, hipStream_t s) {
  // 32-bit byte offsets inside a workgroup's operand panels
  if ((TR * lda + K) * 4 >= (int64_t{1} << 31) || ((transB ? TBN : K) * ldb + (transB ? K : TBN)) * 4 >= (int64_t{1} << 31)) return EBN_ERR_UNSUPPORTED;
  if (ebn_ceil_div(M, TR) >= (int64_t{1} << 31)) return EBN_ERR_UNSUPPORTED;
  const dim3 grid(static_cast<unsigned>(ebn_ceil_div(M, TR)), static_cast<unsigned>(ebn_ceil_div(N, TBN)));
  if (transB) hipLaunchKernelGGL((gemm_tall16_kernel<true>), grid, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, C, ldc);
  else hipLaunchKernelGGL((gemm_tall16_kernel<false>), grid, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, C, ldc);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
