// fp32-accurate GEMM on the bf16 matrix pipe ("bf16x6 split"): an OPT-IN second precision of the projection matmuls
// (K.dot at layers.py:214-226 and its weight gradient).  The default path stays the exact-fp32 MFMA kernel of ebn_gemm.hip.
//
// Every fp32 operand element is split into three bf16 values a = a0 + a1 + a2 (8 + 8 + 8 significand bits: the split is
// exact), and the product keeps the six leading cross terms
//     a.b ~= a0.b0 + (a0.b1 + a1.b0) + (a1.b1 + a0.b2 + a2.b0)          (dropped: a1.b2 + a2.b1 + a2.b2 <= 2^-23 |a.b|)
// each of them a v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- products of bf16 pairs are exact in fp32, so the result
// carries the rounding of an fp32 accumulation chain plus a per-product error below one fp32 ulp.  The bf16 pipe runs 16x
// the exact-fp32 MFMA rate: six products cost 6/16 of the fp32 time before the split passes.
//
// Two kernels:
//   split_planes_kernel   fp32 matrix -> three bf16 planes in the tile-friendly layout [plane][K/8][rows][8]: for each group
//                         of 8 consecutive contraction indices, all rows x 16 bytes.  A 16-deep K slab of a row panel is then
//                         two CONTIGUOUS runs (one per octet) -- the tile fetch is perfectly coalesced 1 KB wave instructions
//                         straight into LDS (buffer_load ... lds), and the LDS image [octet][row][8] is read back as MFMA
//                         fragments (lane l: row l & 31, octet l >> 5) with conflict-free ds_read_b128.
//   gemm_bf16x6_kernel    C[M,N] = A . B^T on those planes, block tile 256 x 128 x 16, 4 waves as 2 x 2 (wave tile 128 x 64 =
//                         4 x 2 MFMA tiles, 48 MFMAs per slab), double-buffered LDS (2 x 36 KB: two workgroups per CU), one
//                         barrier per slab, XCD-aware tile order, deterministic split-K through slab partials.
#include <stdlib.h>

#include "ebn_common.h"

typedef float ebn_f32x4s __attribute__((ext_vector_type(4)));
typedef int ebn_i32x4s __attribute__((ext_vector_type(4)));
__device__ void ebn_raw_buffer_load_lds_s(ebn_i32x4s rsrc, __attribute__((address_space(3))) void* lds, int size, int voffset,
                                          int soffset, int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int SP_BM = 256, SP_BN = 128, SP_BK = 16;
constexpr int SP_THREADS = 256;

// round-to-nearest-even bf16 of an fp32: the hardware conversion of gfx950 (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t bf16_rne_bits(float x) {
  const __bf16 h = static_cast<__bf16>(x);
  return static_cast<uint32_t>(__builtin_bit_cast(unsigned short, h));
}
__device__ __forceinline__ void split3(float x, uint16_t& p0, uint16_t& p1, uint16_t& p2) {
  const uint32_t b0 = bf16_rne_bits(x);
  const float r1 = x - __uint_as_float(b0 << 16);  // exact
  const uint32_t b1 = bf16_rne_bits(r1);
  const float r2 = r1 - __uint_as_float(b1 << 16);  // exact
  const uint32_t b2 = bf16_rne_bits(r2);
  p0 = static_cast<uint16_t>(b0);
  p1 = static_cast<uint16_t>(b1);
  p2 = static_cast<uint16_t>(b2);
}

// src: fp32 [R_src][C_src] row-major (ld).  TRANS = false: operand rows = source rows, contraction = source columns;
// TRANS = true: operand rows = source columns, contraction = source rows.  out: [3][Kp/8][rows_p][8] bf16, rows_p / Kp the
// padded extents (zero-filled beyond the matrix: the GEMM fetches whole tiles without edge handling).
// One workgroup per 64 (rows) x 64 (k) output tile; the source tile goes through LDS so that both the global reads (along the
// source's contiguous axis) and the global writes (64 rows x 16 B per octet and plane = 1 KB per wave) are coalesced.
template <bool TRANS>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int64_t K,
                                                           uint16_t* __restrict__ out, int64_t rows_p, int64_t Kp) {
  __shared__ float tile[64][65];  // !TRANS: [row][k]; TRANS: [k][row]
  const int tid = threadIdx.x;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 64, k0 = static_cast<int64_t>(blockIdx.y) * 64;
  // load: 64 x 64 floats, 16 per thread; lanes run along the source's contiguous axis
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int a = it * 4 + (tid >> 6), c = tid & 63;  // a: slow source index inside the tile, c: fast (contiguous)
    const int64_t srow = TRANS ? (k0 + a) : (r0 + a), scol = TRANS ? (r0 + c) : (k0 + c);
    const bool ok = TRANS ? (srow < K && scol < rows) : (srow < rows && scol < K);
    const float v = src[ok ? srow * ld + scol : 0];
    tile[a][c] = ok ? v : 0.f;
  }
  __syncthreads();
  const int64_t plane = (Kp / 8) * rows_p * 8;
  // store: item (octet o, row r): 8 consecutive k of one row -> 16 bytes per plane; a wave = one octet x 64 rows = 1 KB
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int o = it * 4 + (tid >> 6), r = tid & 63;
    if (r0 + r >= rows_p || k0 + o * 8 >= Kp) continue;
    u16x8 v0, v1, v2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = TRANS ? tile[o * 8 + e][r] : tile[r][o * 8 + e];
      uint16_t p0, p1, p2;
      split3(x, p0, p1, p2);
      v0[e] = p0;
      v1[e] = p1;
      v2[e] = p2;
    }
    const int64_t off = (((k0 >> 3) + o) * rows_p + r0 + r) * 8;
    *reinterpret_cast<u16x8*>(out + off) = v0;
    *reinterpret_cast<u16x8*>(out + plane + off) = v1;
    *reinterpret_cast<u16x8*>(out + 2 * plane + off) = v2;
  }
}

__device__ __forceinline__ ebn_i32x4s make_rsrc_s(const void* base) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  ebn_i32x4s r;
  r.x = static_cast<int>(static_cast<uint32_t>(a));
  r.y = static_cast<int>(static_cast<uint32_t>(a >> 32) & 0xFFFFu);
  r.z = -1;
  r.w = 0x00020000;
  return r;
}

// C[M,N] (+ split-K partials) = A . B^T, A planes [3][Kp/8][a_rows][8], B planes [3][Kp/8][b_rows][8] (a_rows, b_rows multiples of
// 256, Kp a multiple of 16; each plane set below 4 GB: 32-bit byte offsets inside its buffer resource).
__global__ __launch_bounds__(SP_THREADS, 2) void gemm_bf16x6_kernel(const uint16_t* __restrict__ Apl, const uint16_t* __restrict__ Bpl,
                                                                    uint32_t a_rows, uint32_t b_rows, uint32_t Kp, int64_t M,
                                                                    int64_t N, float* __restrict__ C, int64_t ldc,
                                                                    int32_t slabs_per_split, float* __restrict__ Cpart) {
  constexpr int BM = SP_BM, BN = SP_BN;
  constexpr int TM = BM / 2 / 32, TN = BN / 2 / 32;  // MFMA tiles per wave: 4 x 2
  // LDS image of a slab (bf16 elements): A [plane][octet][BM][8] | B [plane][octet][BN][8]
  constexpr int A_ELEMS = 3 * 2 * BM * 8, B_ELEMS = 3 * 2 * BN * 8, BUF = A_ELEMS + B_ELEMS;  // 18432 elements = 36 KB
  __shared__ __attribute__((aligned(16))) uint16_t smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (as ebn_gemm.hip): each XCD walks a contiguous run of tiles of one K split
  int64_t tile_id = static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x;
  int64_t zsplit = blockIdx.z;
  {
    const int64_t per_z = static_cast<int64_t>(gridDim.x) * gridDim.y, nwg = per_z * gridDim.z;
    const int64_t orig = tile_id + per_z * blockIdx.z;
    const int64_t q = nwg / 8, r = nwg % 8, xcd = orig % 8, idx = orig / 8;
    const int64_t lin = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    zsplit = lin / per_z;
    tile_id = lin - zsplit * per_z;
  }
  const int64_t m0 = (tile_id / gridDim.x) * BM, n0 = (tile_id % gridDim.x) * BN;
  const int n_slabs_total = static_cast<int>(Kp / SP_BK);
  const int s_beg = static_cast<int>(zsplit) * slabs_per_split;
  const int s_end = (s_beg + slabs_per_split < n_slabs_total) ? s_beg + slabs_per_split : n_slabs_total;
  const int nk = s_end - s_beg;

  // ---- tile fetch: per slab 24 KB of A (3 planes x 2 octets x 256 rows x 16 B) and 12 KB of B, as 36 wave instructions of 1 KB
  // (64 rows x 16 B, contiguous in memory AND in LDS); wave w issues A pieces 6w..6w+5 and B pieces 3w..3w+2.
  // Byte offset of a piece inside the buffer resource = operand offset + (plane * plane_bytes) + ((2 s + octet) * rows + row0) * 16.
  const ebn_i32x4s arsrc = make_rsrc_s(Apl), brsrc = make_rsrc_s(Bpl);
  const uint32_t a_plane_b = (Kp / 8) * a_rows * 16u, b_plane_b = (Kp / 8) * b_rows * 16u;
  const uint32_t a_slab_b = 2u * a_rows * 16u, b_slab_b = 2u * b_rows * 16u;  // bytes per 16-deep slab (two octets)
  uint32_t a_src[6], b_src[3];  // scalar: piece base at slab 0 of this split
  int a_dst[6], b_dst[3];       // LDS element offsets inside a buffer
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int q = wave * 6 + j, p = q >> 3, oct = (q >> 2) & 1, chunk = q & 3;
    a_src[j] = static_cast<uint32_t>(p) * a_plane_b + (static_cast<uint32_t>(oct) * a_rows + static_cast<uint32_t>(m0) + chunk * 64u) * 16u +
               static_cast<uint32_t>(s_beg) * a_slab_b;
    a_dst[j] = ((p * 2 + oct) * BM + chunk * 64) * 8;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int q = wave * 3 + j, p = q >> 2, oct = (q >> 1) & 1, chunk = q & 1;
    b_src[j] = static_cast<uint32_t>(p) * b_plane_b + (static_cast<uint32_t>(oct) * b_rows + static_cast<uint32_t>(n0) + chunk * 64u) * 16u +
               static_cast<uint32_t>(s_beg) * b_slab_b;
    b_dst[j] = A_ELEMS + ((p * 2 + oct) * BN + chunk * 64) * 8;
  }
  const int lane_b = lane * 16;
  uint32_t sa = 0, sb = 0;  // scalar slab offsets
#define SP_FETCH(BUFI)                                                                                                   \
  do {                                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 6; ++j)                                                                        \
      ebn_raw_buffer_load_lds_s(arsrc, (__attribute__((address_space(3))) void*)(smem + (BUFI) * BUF + a_dst[j]), 16, lane_b, \
                                static_cast<int>(a_src[j] + sa), 0, 0);                                                  \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                        \
      ebn_raw_buffer_load_lds_s(brsrc, (__attribute__((address_space(3))) void*)(smem + (BUFI) * BUF + b_dst[j]), 16, lane_b, \
                                static_cast<int>(b_src[j] + sb), 0, 0);                                                  \
    sa += a_slab_b;                                                                                                      \
    sb += b_slab_b;                                                                                                      \
  } while (0)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // MFMA operand of lane l: row l & 31 of the 32-row tile, octet l >> 5: element offset of the lane inside an operand image
  const int frag = ((lane >> 5) * BM + (lane & 31)) * 8, fragb = ((lane >> 5) * BN + (lane & 31)) * 8;
#define SP_MMA(BUFI)                                                                                                     \
  {                                                                                                                      \
    const uint16_t* as = smem + (BUFI) * BUF + frag + wm * (BM / 2) * 8;                                                 \
    const uint16_t* bs = smem + (BUFI) * BUF + A_ELEMS + fragb + wn * (BN / 2) * 8;                                      \
    bf16x8 b[3][TN];                                                                                                     \
    _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                                        \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                     \
        b[p][j] = *reinterpret_cast<const bf16x8*>(bs + p * 2 * BN * 8 + j * 32 * 8);                                    \
    /* two row tiles at a time, product-major: consecutive MFMAs write four different accumulators, so none waits for */   \
    /* the result of the one in front of it (six products into ONE accumulator back to back is a dependent chain)     */   \
    _Pragma("unroll") for (int i = 0; i < TM; i += 2) {                                                                  \
      bf16x8 a[3][2];                                                                                                    \
      _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                                      \
        _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                                 \
          a[p][ii] = *reinterpret_cast<const bf16x8*>(as + p * 2 * BM * 8 + (i + ii) * 32 * 8);                          \
      /* small terms first: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0) */                                                       \
      SP_PROD(2, 0) SP_PROD(0, 2) SP_PROD(1, 1) SP_PROD(1, 0) SP_PROD(0, 1) SP_PROD(0, 0)                                \
    }                                                                                                                    \
  }
#define SP_PROD(PA, PB)                                                                                                  \
  _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                                       \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                       \
      acc[i + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][ii], b[PB][j], acc[i + ii][j], 0, 0, 0);

  if (nk > 0) {
    SP_FETCH(0);
    __syncthreads();
  }
#define SP_SLAB(CUR, KT)                 \
  {                                      \
    if ((KT) + 1 < nk) SP_FETCH((CUR) ^ 1); \
    SP_MMA(CUR)                          \
    __syncthreads();                     \
  }
  for (int kt = 0; kt < nk; kt += 2) {
    SP_SLAB(0, kt);
    if (kt + 1 < nk) SP_SLAB(1, kt + 1);
  }
#undef SP_SLAB
#undef SP_MMA
#undef SP_PROD
#undef SP_FETCH

  // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const bool split = gridDim.z > 1;
  float* out = split ? (Cpart + zsplit * M * N) : C;
  const int64_t ldo = split ? N : ldc;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        out[row * ldo + col] = acc[i][j][r];
      }
    }
  }
}

// C = alpha * sum_z part[z] + beta * C   (fixed order: deterministic)
__global__ __launch_bounds__(256) void split_reduce_kernel(const float* __restrict__ part, int splits, int64_t M, int64_t N, float alpha,
                                                          float beta, float* __restrict__ C, int64_t ldc) {
  const int64_t total = M * N;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * 256) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += part[static_cast<int64_t>(z) * total + i];
    const int64_t row = i / N, col = i - row * N;
    float* c = C + row * ldc + col;
    *c = (beta != 0.f) ? fmaf(beta, *c, alpha * s) : alpha * s;
  }
}

__global__ __launch_bounds__(256) void scale_inplace_kernel(float* __restrict__ C, int64_t ldc, int64_t M, int64_t N, float alpha) {
  const int64_t total = M * N;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t row = i / N, col = i - row * N;
    C[row * ldc + col] *= alpha;
  }
}

// Gather + dropout + split in one pass (the training step's embedding lookup in split precision): token r's row is
// table[ids[r]] with the inverted-dropout mask of flat element index r*D + c (the stream of ebn_gather_rows_f32, bit for
// bit), and the result goes out as bf16 planes in BOTH orientations the two projection GEMMs contract over --
//   outN: rows = tokens, contraction = embedding columns   (A operand of Q|K|V = X.Wqkv)
//   outT: rows = embedding columns, contraction = tokens   (A operand of dWqkv = X^T.dQKV)
// -- instead of the fp32 X nothing else reads.  One workgroup per 64 tokens x 64 columns; every table-row piece is a
// contiguous 256 bytes.
__global__ __launch_bounds__(256) void gather_split_planes_kernel(const int32_t* __restrict__ ids, const float* __restrict__ table,
                                                                  int64_t R, int64_t D, int64_t V, const uint32_t* __restrict__ key_ptr,
                                                                  uint32_t thresh, float scale, int32_t* __restrict__ oob_flag,
                                                                  uint16_t* __restrict__ outN, int64_t n_rows_p, int64_t n_Kp,
                                                                  uint16_t* __restrict__ outT, int64_t t_rows_p, int64_t t_Kp) {
  __shared__ float tile[64][65];  // [token][column]
  const int tid = threadIdx.x;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 64, c0 = static_cast<int64_t>(blockIdx.y) * 64;
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  int64_t id[16];
  bool bad = false;
#pragma unroll
  for (int it = 0; it < 16; ++it) {  // ids first, then the rows: all loads unconditional with clamped addresses
    const int64_t r = r0 + it * 4 + (tid >> 6);
    id[it] = ids[r < R ? r : 0];
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int a = it * 4 + (tid >> 6), c = tid & 63;
    const int64_t r = r0 + a, col = c0 + c;
    const bool in_range = id[it] >= 0 && id[it] < V;
    const bool ok = r < R && col < D;
    bad |= (r < R) && !in_range;
    float v = table[(ok && in_range) ? id[it] * D + col : 0];
    v = (ok && in_range) ? v : 0.f;
    if (do_drop) v = ebn_dropout_keep(key, static_cast<uint64_t>(r) * static_cast<uint64_t>(D) + static_cast<uint64_t>(col), thresh) ? v * scale : 0.f;
    tile[a][c] = v;
  }
  if (bad && oob_flag != nullptr) *oob_flag = 1;
  __syncthreads();
  const int64_t n_plane = (n_Kp / 8) * n_rows_p * 8, t_plane = (t_Kp / 8) * t_rows_p * 8;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int o = it * 4 + (tid >> 6), l = tid & 63;
    // N orientation: item (column octet o, token l)
    if (r0 + l < n_rows_p && c0 + o * 8 < n_Kp) {
      u16x8 v0, v1, v2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        uint16_t p0, p1, p2;
        split3(tile[l][o * 8 + e], p0, p1, p2);
        v0[e] = p0;
        v1[e] = p1;
        v2[e] = p2;
      }
      const int64_t off = (((c0 >> 3) + o) * n_rows_p + r0 + l) * 8;
      *reinterpret_cast<u16x8*>(outN + off) = v0;
      *reinterpret_cast<u16x8*>(outN + n_plane + off) = v1;
      *reinterpret_cast<u16x8*>(outN + 2 * n_plane + off) = v2;
    }
    // T orientation: item (token octet o, column l)
    if (c0 + l < t_rows_p && r0 + o * 8 < t_Kp) {
      u16x8 v0, v1, v2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        uint16_t p0, p1, p2;
        split3(tile[o * 8 + e][l], p0, p1, p2);
        v0[e] = p0;
        v1[e] = p1;
        v2[e] = p2;
      }
      const int64_t off = (((r0 >> 3) + o) * t_rows_p + c0 + l) * 8;
      *reinterpret_cast<u16x8*>(outT + off) = v0;
      *reinterpret_cast<u16x8*>(outT + t_plane + off) = v1;
      *reinterpret_cast<u16x8*>(outT + 2 * t_plane + off) = v2;
    }
  }
}

// TRANS split without a transpose through LDS: the source [K][rows] is row-contiguous along `rows`, and the plane layout
// [K/8][rows][8] wants, per octet of 8 source rows, every column's 8 values as one 16-byte chunk -- consecutive columns are
// consecutive chunks.  One workgroup per octet: thread c reads its column's 8 values (8 loads in flight, each wave-coalesced
// along the row) and writes three 16-byte chunks; a wave writes 1 KB contiguous per plane.  Whole source rows are read in
// order, so rows that are not 128-byte multiples apart (ld = 1200) cost nothing extra -- the 64 x 64-tile kernel read them as
// 256-byte pieces that straddle three lines instead of two (4.0 instead of 6.0 TB/s).
__global__ __launch_bounds__(256) void split_planes_t_rows_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int64_t K,
                                                                  uint16_t* __restrict__ out, int64_t rows_p, int64_t Kp) {
  const int64_t k0 = static_cast<int64_t>(blockIdx.x) * 8;
  const int64_t plane = (Kp / 8) * rows_p * 8;
  for (int64_t c = threadIdx.x; c < rows_p; c += 256) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool ok = c < rows && k0 + e < K;
      const float v = src[ok ? (k0 + e) * ld + c : 0];
      x[e] = ok ? v : 0.f;
    }
    u16x8 v0, v1, v2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      uint16_t p0, p1, p2;
      split3(x[e], p0, p1, p2);
      v0[e] = p0;
      v1[e] = p1;
      v2[e] = p2;
    }
    const int64_t off = ((k0 >> 3) * rows_p + c) * 8;
    *reinterpret_cast<u16x8*>(out + off) = v0;
    *reinterpret_cast<u16x8*>(out + plane + off) = v1;
    *reinterpret_cast<u16x8*>(out + 2 * plane + off) = v2;
  }
}

// Gather + dropout + split, whole table rows at a time (D % 4 == 0, 16-byte aligned table): one workgroup per 16 tokens x up
// to GS_CHUNK columns.  Loads are the fp32 gather's: 16-byte vectors, consecutive lanes on consecutive vectors of a row (a
// wave reads 1 KB contiguous of one table row, all of a thread's loads in flight).  The 64 x 64-tile kernel above reads every
// 4 KB table row as sixteen 256-byte pieces from sixteen workgroups at different times -- random 256-byte reads run at half
// the HBM rate of random 4 KB reads.  LDS tile [16][GS_CHUNK + 4] fp32 (the +4 makes both read patterns conflict-free).
constexpr int GS_TOK = 16, GS_CHUNK = 1024, GS_PITCH = GS_CHUNK + 4;  // (32 x 512 and 64 x 256 measured slower)
__global__ __launch_bounds__(256) void gather_split_rows_kernel(const int32_t* __restrict__ ids, const float4* __restrict__ table,
                                                                int64_t R, int64_t D, int64_t V, const uint32_t* __restrict__ key_ptr,
                                                                uint32_t thresh, float scale, int32_t* __restrict__ oob_flag,
                                                                uint16_t* __restrict__ outN, int64_t n_rows_p, int64_t n_Kp,
                                                                uint16_t* __restrict__ outT, int64_t t_rows_p, int64_t t_Kp) {
  extern __shared__ __attribute__((aligned(16))) float gs_tile[];  // [GS_TOK][GS_PITCH]
  const int tid = threadIdx.x;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * GS_TOK, c0 = static_cast<int64_t>(blockIdx.y) * GS_CHUNK;
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  const int64_t vpr = D >> 2;  // float4 per table row
  // ---- gather: item (token t, vector v4 of the chunk): GS_TOK * GS_CHUNK / 4 = 4096 items, 16 per thread
  constexpr int ITEMS = GS_TOK * GS_CHUNK / 4 / 256;
  int64_t id[ITEMS];
  bool bad = false;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int t = (it * 256 + tid) / (GS_CHUNK / 4);
    id[it] = ids[r0 + t < R ? r0 + t : 0];
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int item = it * 256 + tid, t = item / (GS_CHUNK / 4), v4 = item % (GS_CHUNK / 4);
    const int64_t r = r0 + t, col = c0 + v4 * 4;
    const bool in_range = id[it] >= 0 && id[it] < V;
    const bool ok = r < R && col < D;
    bad |= (r < R) && !in_range;
    float4 v = table[(ok && in_range) ? id[it] * vpr + (col >> 2) : 0];
    if (!(ok && in_range)) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (do_drop) {
      const uint64_t pair0 = (static_cast<uint64_t>(r) * static_cast<uint64_t>(D) + static_cast<uint64_t>(col)) >> 1;  // D % 4 == 0: even
      const uint32_t h0 = ebn_dropout_pair_hash(key, pair0), h1 = ebn_dropout_pair_hash(key, pair0 + 1);
      v.x = ((h0 & 0xFFFFu) >= thresh) ? v.x * scale : 0.f;
      v.y = ((h0 >> 16) >= thresh) ? v.y * scale : 0.f;
      v.z = ((h1 & 0xFFFFu) >= thresh) ? v.z * scale : 0.f;
      v.w = ((h1 >> 16) >= thresh) ? v.w * scale : 0.f;
    }
    *reinterpret_cast<float4*>(gs_tile + t * GS_PITCH + v4 * 4) = v;
  }
  if (bad && oob_flag != nullptr) *oob_flag = 1;
  __syncthreads();
  const int64_t n_plane = (n_Kp / 8) * n_rows_p * 8, t_plane = (t_Kp / 8) * t_rows_p * 8;
  // ---- T orientation (rows = columns, contraction = tokens): item (token octet o of 2, column c): a wave writes 1 KB contiguous
#pragma unroll
  for (int it = 0; it < (GS_TOK / 8) * GS_CHUNK / 256; ++it) {
    const int item = it * 256 + tid, o = item / GS_CHUNK, c = item % GS_CHUNK;
    if (c0 + c >= t_rows_p || r0 + o * 8 >= t_Kp) continue;
    u16x8 v0, v1, v2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      uint16_t p0, p1, p2;
      split3(gs_tile[(o * 8 + e) * GS_PITCH + c], p0, p1, p2);
      v0[e] = p0;
      v1[e] = p1;
      v2[e] = p2;
    }
    const int64_t off = (((r0 >> 3) + o) * t_rows_p + c0 + c) * 8;
    *reinterpret_cast<u16x8*>(outT + off) = v0;
    *reinterpret_cast<u16x8*>(outT + t_plane + off) = v1;
    *reinterpret_cast<u16x8*>(outT + 2 * t_plane + off) = v2;
  }
  // ---- N orientation (rows = tokens, contraction = columns): item (column octet oc, token t): 16 tokens x 16 B = 256 B runs
#pragma unroll
  for (int it = 0; it < GS_TOK * (GS_CHUNK / 8) / 256; ++it) {
    const int item = it * 256 + tid, t = item % GS_TOK, oc = item / GS_TOK;
    if (r0 + t >= n_rows_p || c0 + oc * 8 >= n_Kp) continue;
    const float4 lo = *reinterpret_cast<const float4*>(gs_tile + t * GS_PITCH + oc * 8);
    const float4 hi = *reinterpret_cast<const float4*>(gs_tile + t * GS_PITCH + oc * 8 + 4);
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    u16x8 v0, v1, v2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      uint16_t p0, p1, p2;
      split3(x[e], p0, p1, p2);
      v0[e] = p0;
      v1[e] = p1;
      v2[e] = p2;
    }
    const int64_t off = (((c0 >> 3) + oc) * n_rows_p + r0 + t) * 8;
    *reinterpret_cast<u16x8*>(outN + off) = v0;
    *reinterpret_cast<u16x8*>(outN + n_plane + off) = v1;
    *reinterpret_cast<u16x8*>(outN + 2 * n_plane + off) = v2;
  }
}

constexpr int64_t SP_ROW_PAD = 256;  // plane sets pad their rows to the larger tile edge, whichever operand they become
int64_t pad_rows(int64_t rows) { return ebn_ceil_div(rows > 0 ? rows : 1, SP_ROW_PAD) * SP_ROW_PAD; }
int64_t pad_k(int64_t K) { return ebn_ceil_div(K > 0 ? K : 1, SP_BK) * SP_BK; }

struct SplitPlan {
  int splits;
  int slabs_per_split;
};

SplitPlan split_plan(int64_t M, int64_t N, int64_t K) {
  SplitPlan p;
  const int64_t tiles = (pad_rows(M) / SP_BM) * ebn_ceil_div(N, SP_BN), slabs = pad_k(K) / SP_BK;
  // fill the chip (512 resident workgroups: two per CU) when the output alone has too few tiles, keeping >= 32 slabs per split
  int64_t splits = 1;
  if (tiles < 384) {
    splits = 512 / tiles;  // never one workgroup more than the chip holds at once: a 513th would run a round of its own
    if (splits > slabs / 32) splits = slabs / 32;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
  }
  const int64_t per = ebn_ceil_div(slabs, splits);  // (64-bit until the end: a K beyond 2^35 used to truncate to 0 slabs per split)
  p.slabs_per_split = static_cast<int>(per > INT32_MAX ? INT32_MAX : per);
  p.splits = static_cast<int>(ebn_ceil_div(slabs, per));
  return p;
}

}  // namespace

extern "C" int64_t ebn_planes_bytes(int64_t rows, int64_t K) { return ebn_dim_ok(rows, K) ? ebn_sat_mul(6 * pad_rows(rows), pad_k(K)) : 0; }

extern "C" int ebn_split_planes_f32(const float* src, int64_t ld, int64_t rows, int64_t K, int32_t trans, void* planes,
                                    ebn_stream_t stream) {
  EBN_REQUIRE(src && planes && rows > 0 && K > 0 && ld >= (trans ? rows : K), EBN_ERR_BAD_ARG);
  EBN_REQUIRE(ebn_aligned16(planes), EBN_ERR_ALIGN);
  const int64_t rows_p = pad_rows(rows), Kp = pad_k(K);
  const dim3 grid(static_cast<unsigned>(rows_p / 64), static_cast<unsigned>(ebn_ceil_div(Kp, 64)));
  if (!trans) EBN_LAUNCH((split_planes_kernel<false>), grid, dim3(256), 0, ebn_stream(stream), src, ld, rows, K,
                                 static_cast<uint16_t*>(planes), rows_p, Kp);
  else  // transposing source: whole rows in order (the 64 x 64-tile transpose through LDS straddled three lines per 256-byte tile row)
    EBN_LAUNCH(split_planes_t_rows_kernel, dim3(static_cast<unsigned>(Kp / 8)), dim3(256), 0, ebn_stream(stream), src, ld, rows, K,
                       static_cast<uint16_t*>(planes), rows_p, Kp);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_gather_split_planes_f32(const int32_t* ids, const float* table, int64_t n_rows, int32_t D, int64_t V,
                                           const ebn_step_state* st, int32_t site, float drop_p, int32_t* oob_flag, void* planes_n,
                                           void* planes_t, ebn_stream_t stream) {
  EBN_REQUIRE(ids && table && planes_n && planes_t && n_rows > 0 && D > 0 && V > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(ebn_aligned16(planes_n) && ebn_aligned16(planes_t), EBN_ERR_ALIGN);
  const EbnDrop d = ebn_make_drop(st, site, drop_p);
  const int64_t n_rows_p = pad_rows(n_rows), n_Kp = pad_k(D), t_rows_p = pad_rows(D), t_Kp = pad_k(n_rows);
  const int64_t tok_ext = n_rows_p > t_Kp ? n_rows_p : t_Kp, col_ext = t_rows_p > n_Kp ? t_rows_p : n_Kp;
  if ((D % 4) == 0 && ebn_aligned16(table)) {  // whole table rows at a time
    constexpr size_t lds = static_cast<size_t>(GS_TOK) * GS_PITCH * sizeof(float);  // 65,792 bytes: above the 64 KB default limit
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_split_rows_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (attr != hipSuccess) return static_cast<int>(attr);
    const dim3 grid_r(static_cast<unsigned>(ebn_ceil_div(tok_ext, GS_TOK)), static_cast<unsigned>(ebn_ceil_div(col_ext, GS_CHUNK)));
    EBN_LAUNCH(gather_split_rows_kernel, grid_r, dim3(256), lds, ebn_stream(stream), ids, reinterpret_cast<const float4*>(table), n_rows,
                       static_cast<int64_t>(D), V, d.key_ptr, d.thresh, d.scale, oob_flag, static_cast<uint16_t*>(planes_n), n_rows_p, n_Kp,
                       static_cast<uint16_t*>(planes_t), t_rows_p, t_Kp);
    EBN_CHECK_LAUNCH();
    return EBN_OK;
  }
  const dim3 grid(static_cast<unsigned>(ebn_ceil_div(tok_ext, 64)), static_cast<unsigned>(ebn_ceil_div(col_ext, 64)));
  EBN_LAUNCH(gather_split_planes_kernel, grid, dim3(256), 0, ebn_stream(stream), ids, table, n_rows, static_cast<int64_t>(D), V,
                     d.key_ptr, d.thresh, d.scale, oob_flag, static_cast<uint16_t*>(planes_n), n_rows_p, n_Kp,
                     static_cast<uint16_t*>(planes_t), t_rows_p, t_Kp);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int64_t ebn_gemm_planes_workspace_floats(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || !ebn_dim_ok(M, N, K)) return 0;
  return ebn_sat_mul(static_cast<int64_t>(split_plan(M, N, K).splits) * M, N);  // (one slice when K is not split: the beta != 0 combine reads it)
}

extern "C" int ebn_gemm_planes_f32(const void* a_planes, int64_t M, const void* b_planes, int64_t N, int64_t K, float alpha, float beta,
                                   float* C, int64_t ldc, float* workspace, int64_t workspace_floats, ebn_stream_t stream) {
  EBN_REQUIRE(M >= 0 && N >= 0 && K >= 0, EBN_ERR_BAD_ARG);
  if (M == 0 || N == 0) return EBN_OK;
  EBN_REQUIRE(a_planes && b_planes && C && ldc >= N, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(ebn_aligned16(a_planes) && ebn_aligned16(b_planes), EBN_ERR_ALIGN);
  EBN_REQUIRE(ebn_planes_bytes(M, K) < (int64_t{1} << 32) && ebn_planes_bytes(N, K) < (int64_t{1} << 32), EBN_ERR_UNSUPPORTED);
  const SplitPlan p = split_plan(M, N, K);
  hipStream_t s = ebn_stream(stream);
  const uint16_t* Ap = static_cast<const uint16_t*>(a_planes);
  const uint16_t* Bp = static_cast<const uint16_t*>(b_planes);
  const uint32_t a_rows = static_cast<uint32_t>(pad_rows(M)), b_rows = static_cast<uint32_t>(pad_rows(N)), Kp = static_cast<uint32_t>(pad_k(K));
  const bool direct = p.splits == 1 && beta == 0.f;  // the GEMM writes C itself; alpha applied afterwards when != 1
  EBN_REQUIRE(direct || (workspace && workspace_floats >= static_cast<int64_t>(p.splits) * M * N), EBN_ERR_BAD_ARG);
  const dim3 grid(static_cast<unsigned>(ebn_ceil_div(N, SP_BN)), static_cast<unsigned>(a_rows / SP_BM), static_cast<unsigned>(direct ? 1 : p.splits));
  if (direct) {
    EBN_LAUNCH(gemm_bf16x6_kernel, grid, dim3(SP_THREADS), 0, s, Ap, Bp, a_rows, b_rows, Kp, M, N, C, ldc, p.slabs_per_split, nullptr);
    EBN_CHECK_LAUNCH();
    if (alpha != 1.0f) {
      int64_t g = ebn_ceil_div(M * N, 256);
      if (g > 4096) g = 4096;
      EBN_LAUNCH(scale_inplace_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, s, C, ldc, M, N, alpha);
      EBN_CHECK_LAUNCH();
    }
    return EBN_OK;
  }
  // split-K, or beta != 0: every z-slice writes a partial, a fixed-order combine finishes
  if (p.splits == 1)  // the kernel treats gridDim.z == 1 as "write C": C is pointed at the partial buffer (ld = N)
    EBN_LAUNCH(gemm_bf16x6_kernel, grid, dim3(SP_THREADS), 0, s, Ap, Bp, a_rows, b_rows, Kp, M, N, workspace, N, p.slabs_per_split, nullptr);
  else
    EBN_LAUNCH(gemm_bf16x6_kernel, grid, dim3(SP_THREADS), 0, s, Ap, Bp, a_rows, b_rows, Kp, M, N, C, ldc, p.slabs_per_split, workspace);
  EBN_CHECK_LAUNCH();
  int64_t g = ebn_ceil_div(M * N, 256);
  if (g > 4096) g = 4096;
  EBN_LAUNCH(split_reduce_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, s, workspace, p.splits, M, N, alpha, beta, C, ldc);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

// bytes of workspace ebn_gemm_f32_split needs for (M, N, K): both operands' bf16 planes and the split-K partials
extern "C" int64_t ebn_gemm_split_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || !ebn_dim_ok(M, N, K)) return 0;
  return ebn_sat_add(ebn_sat_add(ebn_planes_bytes(M, K), ebn_planes_bytes(N, K)),
                     ebn_sat_add(ebn_sat_mul(ebn_gemm_planes_workspace_floats(M, N, K), 4), 256));
}

extern "C" int ebn_gemm_f32_split(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                                  int64_t lda, const float* B, int64_t ldb, float beta, float* C, int64_t ldc, void* workspace,
                                  int64_t workspace_bytes, ebn_stream_t stream) {
  EBN_REQUIRE(M >= 0 && N >= 0 && K >= 0, EBN_ERR_BAD_ARG);
  if (M == 0 || N == 0) return EBN_OK;
  EBN_REQUIRE(A && B && C && workspace, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(ebn_aligned16(workspace), EBN_ERR_ALIGN);
  EBN_REQUIRE(workspace_bytes >= ebn_gemm_split_workspace_bytes(M, N, K), EBN_ERR_BAD_ARG);
  if (K == 0) {  // empty contraction: C <- beta * C (the plane kernels need K > 0)
    return ebn_gemm_f32_ws(transA, transB, M, N, 0, alpha, A, lda, B, ldb, beta, C, ldc, nullptr, 0, stream);
  }
  char* base = static_cast<char*>(workspace);
  const int64_t a_bytes = ebn_planes_bytes(M, K), b_bytes = ebn_planes_bytes(N, K);
  void* Ap = base;
  void* Bp = base + a_bytes;
  float* part = reinterpret_cast<float*>(base + ((a_bytes + b_bytes + 255) / 256) * 256);
  // operand A as rows = m, contraction = k: the source is [M][K] (transA = 0) or [K][M] (transA = 1: transposed on the way);
  // operand B as rows = n: the source is [N][K] (transB = 1) or [K][N] (transB = 0: transposed on the way)
  int rc = ebn_split_planes_f32(A, lda, M, K, transA ? 1 : 0, Ap, stream);
  if (rc != EBN_OK) return rc;
  rc = ebn_split_planes_f32(B, ldb, N, K, transB ? 0 : 1, Bp, stream);
  if (rc != EBN_OK) return rc;
  return ebn_gemm_planes_f32(Ap, M, Bp, N, K, alpha, beta, C, ldc, part, ebn_gemm_planes_workspace_floats(M, N, K), stream);
}

// One entry point for both precisions of the projection GEMM: precision 0 = the exact-fp32 MFMA kernels (bitwise an fp32 fma
// chain; `workspace` holds split-K partials), precision 1 = bf16x6 split (fp32-accurate, fp32 accumulate; `workspace` holds the
// bf16 planes and the partials: ebn_gemm_prec_workspace_bytes).
extern "C" int64_t ebn_gemm_prec_workspace_bytes(int64_t M, int64_t N, int64_t K, int32_t precision) {
  if (precision == 1) return ebn_gemm_split_workspace_bytes(M, N, K);
  return ebn_sat_mul(ebn_gemm_workspace_floats(M, N, K), 4);
}

extern "C" int ebn_gemm_f32_prec(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                                 int64_t lda, const float* B, int64_t ldb, float beta, float* C, int64_t ldc, void* workspace,
                                 int64_t workspace_bytes, int32_t precision, ebn_stream_t stream) {
  EBN_REQUIRE(precision == 0 || precision == 1, EBN_ERR_BAD_ARG);
  if (precision == 1) return ebn_gemm_f32_split(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, workspace, workspace_bytes, stream);
  return ebn_gemm_f32_ws(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, static_cast<float*>(workspace), workspace_bytes / 4,
                         stream);
}
