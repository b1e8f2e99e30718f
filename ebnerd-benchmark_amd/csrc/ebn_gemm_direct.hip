// Exact-fp32 MFMA GEMM for TALL outputs times a SMALL second operand -- the two AttLayer2 products over every title token
//   U  = Y . W          24000 x 200 x 400   (layers.py:65-66, K.dot(x, W))          -- NN
//   dY = dpre . W^T     24000 x 400 x 200   (its input gradient)                    -- NT
// -- as an LDS-FREE, BARRIER-FREE kernel: every wave owns R x C output blocks of 16 x 16 (v_mfma_f32_16x16x4_f32: the same
// 64 FLOP / clock / SIMD as the 32 x 32 form, bitwise an fp32 fma chain) and fetches the MFMA operand fragments of its blocks
// straight from global memory into registers:
//   * A (always [M][K], k contiguous): lane (row i = lane & 15, quarter kq = lane >> 4) loads ONE float4 = A[i][16 g + 4 kq .. + 3]
//     per row block and 16-deep k group g; component s feeds MFMA step s (k = 16 g + 4 kq + s, the same assignment for B);
//   * B stored [N][K]: the same float4 per column block;  B stored [K][N]: four dword loads (64 contiguous bytes per 16 lanes).
// B is a weight matrix of a few hundred KB that every wave re-reads: it lives in L1 / L2.  A is read once per column group.
// Why this shape of kernel: 24000 x 200 has only 750 x 7 tiles of 32 x 32 -- every LDS-staged tiling either pads N = 200 to 256
// (28 % of the MFMA work) or ends with under-filled workgroups, and with 3 waves per SIMD and a workgroup barrier per 16-deep slab
// the matrix pipe sat at 51-69 % (profiles/r03_tuning_notes.md).  Here nothing synchronises: a wave issues the loads of k group
// g + 1 (two register sets, literal indices), multiplies group g -- R*C*4 MFMAs on R*C independent accumulators, 1300-2700 cycles,
// several L2 round trips -- and waits once per group.  N = 200 runs as 13 blocks (208), 400 as 25 (no padding); the (R, C) split
// is chosen so that the wave tasks fill the 1024 SIMDs in whole rounds (24000 x 200: 500 row groups x {7, 6} column blocks = 1000
// tasks = 0.98 of one round).
// The slab loop carries no vector-ALU work (buffer loads: constant lane offsets + scalar group offsets): the fp32 MFMA does not
// overlap VALU instructions on gfx950 (tools/microbench/mfma_valu_overlap.hip).
#include <stdlib.h>

#include "ebn_common.h"

typedef float ebn_dir_f32x4 __attribute__((ext_vector_type(4)));
typedef int ebn_dir_i32x4 __attribute__((ext_vector_type(4)));
__device__ ebn_dir_f32x4 ebn_dir_buffer_load_x4(ebn_dir_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float ebn_dir_buffer_load_x1(ebn_dir_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");

namespace {

constexpr bool INTERLEAVE = true;  // the requests of k group g + 1 between the MFMAs of group g (all of them in front: measured slower)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ ebn_dir_i32x4 dir_rsrc(const float* base) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  ebn_dir_i32x4 r;
  r.x = static_cast<int>(static_cast<uint32_t>(a));
  r.y = static_cast<int>(static_cast<uint32_t>(a >> 32) & 0xFFFFu);  // stride 0: raw buffer
  r.z = -1;                                                            // 4 GB of records: no bounds use
  r.w = 0x00020000;
  return r;
}

// operand fragments of one 16-deep k group: a[r] = float4 of row block r; b[c][s] = value of column block c for step s
template <int R, int NC>
struct DirFrags {
  f32x4 a[R];
  f32x4 b[NC];
};

// Everything a wave needs to fetch a k group: resources, constant lane offsets, scalar strides.
template <int R, int NC, bool B_KC>
struct DirPlan {
  ebn_dir_i32x4 arsrc, brsrc;
  uint32_t oa[R];   // lane byte offset of (row block r, row ln, quarter kq) inside the row group's A panel
  uint32_t ob[NC];  // B_KC: (column block c row ln, quarter kq) of B[N][K];  else: (k row 4 kq, column of block c) of B[K][N]
  uint32_t ldb4;    // !B_KC: bytes per k row of B
};

template <int R, int NC, bool B_KC>
__device__ __forceinline__ void dir_fetch(DirFrags<R, NC>& f, const DirPlan<R, NC, B_KC>& p, uint32_t sa, uint32_t sb) {
#pragma unroll
  for (int r = 0; r < R; ++r) f.a[r] = ebn_dir_buffer_load_x4(p.arsrc, static_cast<int>(p.oa[r]), static_cast<int>(sa), 0);
  if (B_KC) {
#pragma unroll
    for (int c = 0; c < NC; ++c) f.b[c] = ebn_dir_buffer_load_x4(p.brsrc, static_cast<int>(p.ob[c]), static_cast<int>(sb), 0);
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int c = 0; c < NC; ++c)
        f.b[c][s] = ebn_dir_buffer_load_x1(p.brsrc, static_cast<int>(p.ob[c]), static_cast<int>(sb + static_cast<uint32_t>(s) * p.ldb4), 0);
  }
}

// The partial last k group (K % 16 in {4, 8, 12}): quarters with 4 kq >= krem contribute zeros; their loads read offset 0 of the
// resource (always valid) and are replaced.  A handful of VALU selects, once per wave.
template <int R, int NC, bool B_KC>
__device__ __forceinline__ void dir_fetch_tail(DirFrags<R, NC>& f, const DirPlan<R, NC, B_KC>& p, uint32_t sa, uint32_t sb, bool ok) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const f32x4 t = ebn_dir_buffer_load_x4(p.arsrc, static_cast<int>(ok ? p.oa[r] + sa : 0u), 0, 0);
    f.a[r] = ok ? t : zero;
  }
  if (B_KC) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const f32x4 t = ebn_dir_buffer_load_x4(p.brsrc, static_cast<int>(ok ? p.ob[c] + sb : 0u), 0, 0);
      f.b[c] = ok ? t : zero;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float t = ebn_dir_buffer_load_x1(p.brsrc, static_cast<int>(ok ? p.ob[c] + sb + static_cast<uint32_t>(s) * p.ldb4 : 0u), 0, 0);
        f.b[c][s] = ok ? t : 0.f;
      }
  }
}

// R*NC*4 MFMAs on R*NC independent accumulators, step-major: consecutive MFMAs never write the same accumulator
template <int R, int NC>
__device__ __forceinline__ void dir_mma(f32x4 (&acc)[R][NC], const DirFrags<R, NC>& f) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[r][s], f.b[c][s], acc[r][c], 0, 0, 0);
}

// One wave task: row blocks [rb0, rb0 + R) x column blocks [cb0, cb0 + NC) of C = alpha * A . op(B).
template <int R, int NC, bool B_KC, int DEPTH>
__device__ __forceinline__ void dir_task(int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A, int64_t lda,
                                         const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc, int64_t rb0,
                                         int cb0, int lane) {
  const int ln = lane & 15, kq = lane >> 4;
  const int64_t row0 = rb0 * 16;
  DirPlan<R, NC, B_KC> p;
  p.arsrc = dir_rsrc(A + row0 * lda);
  p.brsrc = dir_rsrc(B);
  p.ldb4 = static_cast<uint32_t>(ldb * 4);
#pragma unroll
  for (int r = 0; r < R; ++r) {  // rows past M are CLAMPED: they only feed output rows that are never stored
    int64_t row = row0 + 16 * r + ln;
    row = row < M ? row : M - 1;
    p.oa[r] = static_cast<uint32_t>(((row - row0) * lda + 4 * kq) * 4);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {  // columns past N likewise
    int64_t col = static_cast<int64_t>(cb0 + c) * 16 + ln;
    col = col < N ? col : N - 1;
    p.ob[c] = B_KC ? static_cast<uint32_t>((col * ldb + 4 * kq) * 4) : static_cast<uint32_t>((4 * kq * ldb + col) * 4);
  }
  const uint32_t step_a = 64u, step_b = B_KC ? 64u : static_cast<uint32_t>(16 * ldb * 4);  // bytes per 16-deep k group

  f32x4 acc[R][NC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkf = static_cast<int>(K / 16);         // full k groups
  const int krem = static_cast<int>(K - 16 * static_cast<int64_t>(nkf));  // 0, 4, 8 or 12 (K % 4 == 0)
  DirFrags<R, NC> f0, f1;
  uint32_t sa = 0, sb = 0;
  // A wave issues in order and a VMEM instruction holds its issue slot for tens of cycles (address + data path hand-off): a clump
  // of 31 loads in front of 84 MFMAs leaves the matrix pipe idle for the length of the clump -- with one wave per SIMD nobody
  // else fills it (measured: 52 % pipe busy, 72 % of the wave cycles in SQ_WAIT_INST_ANY, although tools/microbench/mfma16_issue
  // sustains 32.6 cycles per MFMA from one wave).  So the requests of group g + 1 are INTERLEAVED with the MFMAs of group g: one
  // VMEM read after every PER_LOAD MFMAs (sched_group_barrier pattern), each hidden behind the 32 pipe cycles of the MFMA before it.
  constexpr int N_LOADS = R + (B_KC ? NC : 4 * NC), N_MFMA = 4 * R * NC, PER_LOAD = (N_MFMA / N_LOADS) > 0 ? (N_MFMA / N_LOADS) : 1;
#define EBN_DIR_STEP(FL, FM)                                                   \
  do {                                                                         \
    dir_fetch(FL, p, sa, sb);                                                  \
    sa += step_a;                                                              \
    sb += step_b;                                                              \
    dir_mma(acc, FM);                                                          \
    if (INTERLEAVE) {                                                          \
      _Pragma("unroll") for (int i__ = 0; i__ < N_LOADS; ++i__) {              \
        __builtin_amdgcn_sched_group_barrier(0x008, PER_LOAD, 0); /* MFMA */   \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        /* VMEM read */ \
      }                                                                        \
      __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA - PER_LOAD * N_LOADS, 0); \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#define EBN_DIR_FETCH(F)               \
  do {                                 \
    dir_fetch(F, p, sa, sb);           \
    sa += step_a;                      \
    sb += step_b;                      \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)
#define EBN_DIR_MMA(F)                 \
  do {                                 \
    dir_mma(acc, F);                   \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)
  // The sched_barriers keep the requests in front of the multiplies (the scheduler sinks loads to their uses).  The loops hold
  // ONLY unconditional fetches: a conditional one merging into them makes the compiler wait vmcnt(0), i.e. for the group it has
  // just requested.  Register sets alternate with literal names.
  {  // one group ahead (a third register set, two groups ahead, measured slower: load latency is not what is left)
    if (nkf > 0) {
      EBN_DIR_FETCH(f0);
      int g = 0;
      for (; g + 2 < nkf; g += 2) {
        EBN_DIR_STEP(f1, f0);
        EBN_DIR_STEP(f0, f1);
      }
      if (g + 1 < nkf) {  // two groups left: g is in f0
        EBN_DIR_STEP(f1, f0);
        EBN_DIR_MMA(f1);
      } else {
        EBN_DIR_MMA(f0);
      }
    }
  }
#undef EBN_DIR_FETCH
#undef EBN_DIR_MMA
#undef EBN_DIR_STEP
  if (krem > 0) {
    dir_fetch_tail(f0, p, sa, sb, 4 * kq < krem);
    dir_mma(acc, f0);
  }

  // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + i
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int64_t col = static_cast<int64_t>(cb0 + c) * 16 + ln;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + 16 * r + 4 * kq + i;
        if (row < M && col < N) C[row * ldc + col] = alpha * acc[r][c][i];
      }
    }
  }
}

// Workgroup = 4 independent waves (no LDS, no barrier); wave task t = (row group t / G, column group t % G): the column groups of
// one row group sit in one workgroup, so the A panel they share is fetched into one CU's L1.  Column group j covers the column
// blocks of its share: the first n_wide groups hold CW of them, the others CW - 1 (the two block counts an instantiation carries;
// the host plan only offers splits of that form).
template <int R, int CW, bool B_KC, int DEPTH>
__global__ __launch_bounds__(256, 2) void gemm_direct16_kernel(int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A,
                                                               int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                               float* __restrict__ C, int64_t ldc, int32_t G, int32_t n_wide, int64_t n_tasks) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  if (t >= n_tasks) return;  // wave-uniform; nothing synchronises
  const int64_t rg = t / G;
  const int cg = static_cast<int>(t - rg * G);
  // the first n_wide groups hold CW blocks, the others CW - 1
  const int cb0 = cg < n_wide ? cg * CW : n_wide * CW + (cg - n_wide) * (CW - 1);
  if (cg < n_wide) dir_task<R, CW, B_KC, DEPTH>(M, N, K, alpha, A, lda, B, ldb, C, ldc, rg * R, cb0, lane);
  else dir_task<R, CW - 1, B_KC, DEPTH>(M, N, K, alpha, A, lda, B, ldb, C, ldc, rg * R, cb0, lane);
}

// ---- TN: the weight gradient of a tall product, dW[M][N] = A^T . B with A [K][M], B [K][N], K in the tens of thousands and an
// output of a few hundred columns (AttLayer2: dW = Y^T . dpre, 400 x 200 x 24000).  64 x 64 tiles pad that output to 448 x 256
// (43 % of the MFMA work), so it is tiled in 16 x 16 blocks as well: a workgroup owns R x NC blocks of the output and ONE chunk
// of the K range; its four waves walk the chunk's 16-deep groups interleaved (wave w: groups w, w + 4, ...), fetching both
// fragments by dword rows (k rows are contiguous along m / n: 64 bytes per 16 lanes), and are summed through LDS in the fixed
// order ((w0 + w1) + w2) + w3.  The chunks' results are the dense slices of a deterministic split-K product: the caller's
// combining pass (or ebn_grad_finish_f32) sums them like any other split-K GEMM's.
template <int R, int NC>
struct TnPlan {
  ebn_dir_i32x4 arsrc, brsrc;
  uint32_t oa[R], ob[NC];
  uint32_t lda4, ldb4;
};

template <int R, int NC>
__device__ __forceinline__ void tn_fetch(DirFrags<R, NC>& f, const TnPlan<R, NC>& p, uint32_t sa, uint32_t sb) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      f.a[r][s] = ebn_dir_buffer_load_x1(p.arsrc, static_cast<int>(p.oa[r]), static_cast<int>(sa + static_cast<uint32_t>(s) * p.lda4), 0);
#pragma unroll
    for (int c = 0; c < NC; ++c)
      f.b[c][s] = ebn_dir_buffer_load_x1(p.brsrc, static_cast<int>(p.ob[c]), static_cast<int>(sb + static_cast<uint32_t>(s) * p.ldb4), 0);
  }
}

// the partial last group of the K range: row 4 kq + s of the group exists iff it is < krem
template <int R, int NC>
__device__ __forceinline__ void tn_fetch_tail(DirFrags<R, NC>& f, const TnPlan<R, NC>& p, uint32_t sa, uint32_t sb, int kq, int krem) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const bool ok = 4 * kq + s < krem;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float t = ebn_dir_buffer_load_x1(p.arsrc, static_cast<int>(ok ? p.oa[r] + sa + static_cast<uint32_t>(s) * p.lda4 : 0u), 0, 0);
      f.a[r][s] = ok ? t : 0.f;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float t = ebn_dir_buffer_load_x1(p.brsrc, static_cast<int>(ok ? p.ob[c] + sb + static_cast<uint32_t>(s) * p.ldb4 : 0u), 0, 0);
      f.b[c][s] = ok ? t : 0.f;
    }
  }
}

template <int R, int NC>
__device__ __forceinline__ void tn_task(float* red, int64_t M, int64_t N, float alpha, const float* __restrict__ A, int64_t lda,
                                        const float* __restrict__ B, int64_t ldb, float* __restrict__ out, int64_t k0, int64_t klen,
                                        int64_t rb0, int cb0, int lane, int wave) {
  const int ln = lane & 15, kq = lane >> 4;
  TnPlan<R, NC> p;
  p.arsrc = dir_rsrc(A + k0 * lda);
  p.brsrc = dir_rsrc(B + k0 * ldb);
  p.lda4 = static_cast<uint32_t>(lda * 4);
  p.ldb4 = static_cast<uint32_t>(ldb * 4);
#pragma unroll
  for (int r = 0; r < R; ++r) {  // columns past M / N are CLAMPED: they only feed outputs that are never stored
    int64_t m = (rb0 + r) * 16 + ln;
    m = m < M ? m : M - 1;
    p.oa[r] = static_cast<uint32_t>((4 * kq * lda + m) * 4);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int64_t n = static_cast<int64_t>(cb0 + c) * 16 + ln;
    n = n < N ? n : N - 1;
    p.ob[c] = static_cast<uint32_t>((4 * kq * ldb + n) * 4);
  }
  f32x4 acc[R][NC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave's groups: wave, wave + 4, ... of the chunk's ceil(klen / 16); only the chunk's very last group can be partial
  const int ng = static_cast<int>((klen + 15) / 16), krem = static_cast<int>(klen - 16 * (klen / 16));
  const int mine = ng > wave ? (ng - wave + 3) / 4 : 0;
  const bool tail_mine = krem > 0 && mine > 0 && ((ng - 1) & 3) == wave;
  const int nkf = mine - (tail_mine ? 1 : 0);  // full groups of this wave
  const uint32_t step_a = 64u * p.lda4, step_b = 64u * p.ldb4;  // 4 groups of 16 rows
  uint32_t sa = static_cast<uint32_t>(wave) * 16u * p.lda4, sb = static_cast<uint32_t>(wave) * 16u * p.ldb4;
  DirFrags<R, NC> f0, f1;
  constexpr int N_LOADS = 4 * (R + NC), N_MFMA = 4 * R * NC, PER_LOAD = (N_MFMA / N_LOADS) > 0 ? (N_MFMA / N_LOADS) : 1;
#define EBN_TN_FETCH(F)                \
  do {                                 \
    tn_fetch(F, p, sa, sb);            \
    sa += step_a;                      \
    sb += step_b;                      \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)
#define EBN_TN_STEP(FL, FM)                                                    \
  do {                                                                         \
    tn_fetch(FL, p, sa, sb);                                                   \
    sa += step_a;                                                              \
    sb += step_b;                                                              \
    dir_mma(acc, FM);                                                          \
    if (INTERLEAVE) {                                                          \
      _Pragma("unroll") for (int i__ = 0; i__ < N_LOADS; ++i__) {              \
        __builtin_amdgcn_sched_group_barrier(0x008, PER_LOAD, 0);              \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                     \
      }                                                                        \
      __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA - PER_LOAD * N_LOADS, 0); \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
  if (nkf > 0) {
    EBN_TN_FETCH(f0);
    int g = 0;
    for (; g + 2 < nkf; g += 2) {
      EBN_TN_STEP(f1, f0);
      EBN_TN_STEP(f0, f1);
    }
    if (g + 1 < nkf) {
      EBN_TN_STEP(f1, f0);
      dir_mma(acc, f1);
    } else {
      dir_mma(acc, f0);
    }
  }
#undef EBN_TN_FETCH
#undef EBN_TN_STEP
  if (tail_mine) {
    tn_fetch_tail(f0, p, sa, sb, kq, krem);
    dir_mma(acc, f0);
  }

  // ((w0 + w1) + w2) + w3 through one LDS image of a wave's accumulators
#pragma unroll 1
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) *reinterpret_cast<f32x4*>(red + ((r * NC + c) * 64 + lane) * 4) = acc[r][c];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[r][c] += *reinterpret_cast<const f32x4*>(red + ((r * NC + c) * 64 + lane) * 4);
    }
    __syncthreads();
  }
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int64_t col = static_cast<int64_t>(cb0 + c) * 16 + ln;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = (rb0 + r) * 16 + 4 * kq + i;
        if (row < M && col < N) out[row * N + col] = alpha * acc[r][c][i];
      }
    }
  }
}

// grid = tiles x chunks workgroups in XCD-contiguous order (the tiles of one chunk read the same k rows: one L2); slice z of
// `part` ([Z][M][N] dense) receives alpha * A[k in chunk z]^T . B[k in chunk z]
template <int R, int CW>
__global__ __launch_bounds__(256, (R * CW >= 20) ? 1 : 2) void gemm_direct16_tn_kernel(int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A,
                                                                  int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                                  float* __restrict__ part, int32_t G, int32_t n_wide, int32_t tiles,
                                                                  int64_t kps) {
  __shared__ __attribute__((aligned(16))) float red[R * CW * 256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t lin = blockIdx.x;
  {
    const int64_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = lin % 8, idx = lin / 8;
    lin = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int64_t z = lin / tiles;
  const int tile = static_cast<int>(lin - z * tiles);
  const int rg = tile / G, cg = tile - rg * G;
  const int cb0 = cg < n_wide ? cg * CW : n_wide * CW + (cg - n_wide) * (CW - 1);
  const int64_t k0 = z * kps;
  const int64_t klen = (k0 + kps < K) ? kps : K - k0;
  float* out = part + z * M * N;
  if (cg < n_wide) tn_task<R, CW>(red, M, N, alpha, A, lda, B, ldb, out, k0, klen, static_cast<int64_t>(rg) * R, cb0, lane, wave);
  else tn_task<R, CW - 1>(red, M, N, alpha, A, lda, B, ldb, out, k0, klen, static_cast<int64_t>(rg) * R, cb0, lane, wave);
}

struct TnDirectPlan {
  int R, CW, G, n_wide, tiles, Z;
  int64_t kps;
};

// Output tiling: R x CW blocks per workgroup with the fewest wasted blocks; K chunks so that tiles x chunks fill the 256 CUs once
// (kps a multiple of 64 rows: every wave of a workgroup gets whole groups).
TnDirectPlan tn_direct_plan(int64_t M, int64_t N, int64_t K) {
  const int64_t MB = ebn_ceil_div(M, 16), NB = ebn_ceil_div(N, 16);
  TnDirectPlan best{0, 0, 0, 0, 0, 0, 0};
  double best_cost = 1e300;
  static const int kR[3] = {5, 4, 3}, kC[2] = {5, 4};
  constexpr int wgs = 256;  // tiles x chunks fill the CUs once
  for (int ci = 0; ci < 2; ++ci) {
    const int64_t cw = kC[ci], G = ebn_ceil_div(NB, cw), n_wide = NB - G * (cw - 1);
    if (n_wide < 0 || n_wide > G) continue;
    for (int ri = 0; ri < 3; ++ri) {
      const int64_t tiles = ebn_ceil_div(MB, kR[ri]) * G;
      if (tiles > wgs) continue;
      int64_t Z = wgs / tiles;
      if (Z < 2) continue;
      if (Z > 64) Z = 64;
      const int64_t kps = ebn_ceil_div(ebn_ceil_div(K, Z), 64) * 64;
      Z = ebn_ceil_div(K, kps);
      if (Z < 2 || kps < 256) continue;
      // cost ~ blocks per workgroup (incl. padding blocks) x groups per wave
      const double cost = static_cast<double>(kR[ri] * (n_wide > 0 ? cw : cw - 1)) * static_cast<double>(kps / 64) * (tiles * Z > wgs ? 2.0 : 1.0);
      if (cost < best_cost) {
        best_cost = cost;
        best = TnDirectPlan{kR[ri], static_cast<int>(cw), static_cast<int>(G), static_cast<int>(n_wide), static_cast<int>(tiles), static_cast<int>(Z), kps};
      }
    }
  }
  return best;
}

struct DirectPlan {
  int R, CW, G, n_wide;
  int64_t tasks;
  double cost;
};

// (R, CW) that fill the 1024 SIMDs in the fewest, fullest rounds: cost = rounds x blocks per wave (x the K / 4 MFMAs of a block,
// common to all candidates).  Column groups: the NB column blocks over G = ceil(NB / CW) groups, n_wide of them CW blocks wide and
// the rest CW - 1 (the two block counts a (R, CW) instantiation carries): valid when 0 <= n_wide = NB - G (CW - 1) <= G.
DirectPlan direct_plan(int64_t M, int64_t N, bool b_kc) {
  const int64_t MB = ebn_ceil_div(M, 16), NB = ebn_ceil_div(N, 16);
  DirectPlan best{0, 0, 0, 0, 0, 1e300};
  static const int kR[4] = {4, 3, 2, 1}, kC[3] = {7, 5, 4};
  for (int ci = 0; ci < 3; ++ci) {
    const int64_t cw = kC[ci], G = ebn_ceil_div(NB, cw), n_wide = NB - G * (cw - 1);
    if (n_wide < 0 || n_wide > G) continue;
    const int64_t width = n_wide > 0 ? cw : cw - 1;  // blocks of the widest task
    for (int ri = 0; ri < 4; ++ri) {
      // one row block per wave with a [N][K] B: 6 float4 fragments (6 KB) per 20 MFMAs -- the fetch path, not the matrix pipe, sets
      // the pace (52800 x 400 x 200: 114 us against 90 us for two or three row blocks); with a [K][N] B the dword rows come out of
      // L1 and one row block per wave is the best-filling split (52800 x 200 x 400: 82 us)
      if (b_kc && kR[ri] == 1) continue;
      const int64_t tasks = ebn_ceil_div(MB, kR[ri]) * G;
      const double rounds = static_cast<double>(ebn_ceil_div(tasks, 1024));
      // a task's issue time ~ R * width blocks; fewer, fatter tasks fetch less per MFMA (R + width fragments for R * width blocks):
      // a small tie-break in their favour
      const double cost = rounds * kR[ri] * static_cast<double>(width) * (1.0 + 0.02 * static_cast<double>(kR[ri] + width) / static_cast<double>(kR[ri] * width));
      if (cost < best.cost) best = DirectPlan{kR[ri], static_cast<int>(cw), static_cast<int>(G), static_cast<int>(n_wide), tasks, cost};
    }
  }
  return best;
}

template <int R, int CW>
int direct_launch_rc(bool b_kc, const DirectPlan& pl, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B,
                     int64_t ldb, float* C, int64_t ldc, hipStream_t s) {
  const dim3 grid(static_cast<unsigned>(ebn_ceil_div(pl.tasks, 4))), block(256);
#define EBN_DIR_GO(KC, DP) \
  EBN_LAUNCH((gemm_direct16_kernel<R, CW, KC, DP>), grid, block, 0, s, M, N, K, alpha, A, lda, B, ldb, C, ldc, pl.G, pl.n_wide, pl.tasks)
  if constexpr (R == 1) {  // (the planner never gives one row block per wave to a [N][K] B: the fetch path sets the pace there)
    if (b_kc) return EBN_ERR_UNSUPPORTED;
    EBN_DIR_GO(false, 2);
  } else if (b_kc) EBN_DIR_GO(true, 2);  // two register sets of fragments (a third, two groups ahead, measured slower: load latency is not what is left)
  else EBN_DIR_GO(false, 2);
#undef EBN_DIR_GO
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

}  // namespace

// Shapes this kernel takes (the caller has checked: A not transposed, 16-byte aligned A with lda % 4 == 0 and K % 4 == 0; for B
// stored [N][K] the same of B; beta == 0, no epilogue): a tall output (enough 16-row blocks to fill the chip) times a second
// operand small enough to stay cache-resident while every wave re-reads it, over a contraction short enough that the A panel of a
// row group is walked once.
bool ebn_gemm_direct_wanted(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K) {
  if (transA) return false;
  if (M < 4096 || N < 48 || N > 512 || K < 32 || K > 2048 || (K % 4)) return false;
  constexpr int64_t max_b = int64_t{2} << 20;
  if (N * K * 4 > max_b) return false;  // a larger B streams from L2 / MALL for every wave
  const DirectPlan pl = direct_plan(M, N, transB != 0);
  if (pl.R == 0) return false;
  // B of 1-2 MB (the input-gradient product dQKV . Wqkv^T of a trainable table, N = 300, K = 1200): measured 139.6 against 150.3 us
  // for the LDS-staged tiles at 24000 rows, where A (115 MB, just written by the attention backward) still sits in the 256 MB
  // memory-side cache, but 343 against 325 us at 52800 rows (A = 253 MB from HBM; every (R, CW) plan): the 16-row fragment fetches
  // of the wave tasks stream worse from DRAM than the tile loads do -- taken only while A is at most half of that cache
  if (N * K * 4 > (int64_t{1} << 20) && M * K * 4 > (int64_t{128} << 20)) return false;
  return true;
}

int ebn_gemm_direct_launch(int32_t transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B,
                           int64_t ldb, float* C, int64_t ldc, hipStream_t s) {
  // 32-bit byte offsets: a row group's A panel (64 rows) and the whole of B
  if ((64 * lda + K) * 4 >= (int64_t{1} << 31) || ((transB ? N : K) * ldb + (transB ? K : N) + 16 * ldb) * 4 >= (int64_t{1} << 31)) return EBN_ERR_UNSUPPORTED;
  const DirectPlan pl = direct_plan(M, N, transB != 0);
  if (pl.R == 0 || ebn_ceil_div(pl.tasks, 4) >= (int64_t{1} << 31)) return EBN_ERR_UNSUPPORTED;
  const bool kc = transB != 0;
#define EBN_DIR_CASE(RR, CC) \
  if (pl.R == RR && pl.CW == CC) return direct_launch_rc<RR, CC>(kc, pl, M, N, K, alpha, A, lda, B, ldb, C, ldc, s)
  EBN_DIR_CASE(4, 7); EBN_DIR_CASE(3, 7); EBN_DIR_CASE(2, 7); EBN_DIR_CASE(1, 7);
  EBN_DIR_CASE(4, 5); EBN_DIR_CASE(3, 5); EBN_DIR_CASE(2, 5); EBN_DIR_CASE(1, 5);
  EBN_DIR_CASE(4, 4); EBN_DIR_CASE(3, 4); EBN_DIR_CASE(2, 4); EBN_DIR_CASE(1, 4);
#undef EBN_DIR_CASE
  return EBN_ERR_UNSUPPORTED;
}

// The TN form (weight gradient of a tall product).  Returns the number of K chunks (= dense [Z][M][N] slices the launch writes into
// `part`), 0 when the shape is not this kernel's: an output of at most 512 rows (32 row blocks) and 1280 columns under a long
// contraction.  Measured envelope (MI355X, GEMM + combining pass, against the 64 x 64 split-K tiles): 400 x 200 x 24000 47.6 vs 55.0 us,
// x 52800 82 vs 105 us (43 % tile padding removed); 300 x 1200 x 24000 152.6 vs 170.7 us, x 52800 308.5 vs 345.0 us (only 8 %
// padding there: the gain is the barrier-free, wave-private pipeline).
int ebn_gemm_direct_tn_slices(int64_t M, int64_t N, int64_t K) {
  if (M < 48 || N < 48 || M > 512 || N > 1280 || K < 4096) return 0;
  return tn_direct_plan(M, N, K).Z;
}

int ebn_gemm_direct_tn_launch(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B, int64_t ldb,
                              float* part, hipStream_t s) {
  const TnDirectPlan pl = tn_direct_plan(M, N, K);
  if (pl.Z < 2) return EBN_ERR_UNSUPPORTED;
  if ((pl.kps + 16) * (lda > ldb ? lda : ldb) * 4 >= (int64_t{1} << 31)) return EBN_ERR_UNSUPPORTED;  // 32-bit offsets inside a chunk
  const dim3 grid(static_cast<unsigned>(pl.tiles * pl.Z)), block(256);
#define EBN_TN_CASE(RR, CC)                                                                                                        \
  if (pl.R == RR && pl.CW == CC) {                                                                                                 \
    EBN_LAUNCH((gemm_direct16_tn_kernel<RR, CC>), grid, block, 0, s, M, N, K, alpha, A, lda, B, ldb, part, pl.G, pl.n_wide, pl.tiles, \
                       pl.kps);                                                                                                    \
    EBN_CHECK_LAUNCH();                                                                                                            \
    return EBN_OK;                                                                                                                 \
  }
  EBN_TN_CASE(5, 5) EBN_TN_CASE(4, 5) EBN_TN_CASE(3, 5) EBN_TN_CASE(5, 4) EBN_TN_CASE(4, 4) EBN_TN_CASE(3, 4)
#undef EBN_TN_CASE
  return EBN_ERR_UNSUPPORTED;
}
