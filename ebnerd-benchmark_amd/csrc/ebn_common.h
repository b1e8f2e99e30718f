// Shared device helpers for the gfx950 NRMS kernels. Wave width is 64 throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ebnerd_hip.h"

#define EBN_WAVE 64

#define EBN_CHECK_LAUNCH()                   \
  do {                                       \
    hipError_t e__ = hipGetLastError();      \
    if (e__ != hipSuccess) return (int)e__;  \
  } while (0)

// Every kernel launch of the library goes through EBN_LAUNCH: a process-wide count (ebn_launch_count) lets the host layer REPORT how
// many launches a training step makes (bench.py's `launches_per_step`) instead of stating it.  One relaxed atomic add on the host.
__attribute__((visibility("hidden"))) void ebnx_note_launch(void);
#define EBN_LAUNCH(...)                 \
  do {                                  \
    ebnx_note_launch();                 \
    hipLaunchKernelGGL(__VA_ARGS__);    \
  } while (0)

#define EBN_REQUIRE(cond, code) \
  do {                          \
    if (!(cond)) return (code); \
  } while (0)

// Largest extent any single dimension of a problem may have: row counts, widths and contraction lengths are multiplied into 64-bit
// element counts and narrowed into 32-bit launch geometry; beyond this bound the host-side planners would overflow (found by the
// ASAN / fuzz pass of tests/test_abi_asan.py).  2^31 - 1 rows of 4-byte columns is already past the 288 GB of the device.
#define EBN_DIM_MAX ((int64_t)INT32_MAX)

// size queries answer 0 ("nothing to allocate") for extents outside [0, EBN_DIM_MAX]; the entry points themselves reject them
static inline bool ebn_dim_ok(int64_t a, int64_t b = 0, int64_t c = 0, int64_t d = 0) {
  return a >= 0 && b >= 0 && c >= 0 && d >= 0 && a <= EBN_DIM_MAX && b <= EBN_DIM_MAX && c <= EBN_DIM_MAX && d <= EBN_DIM_MAX;
}

// saturating int64 arithmetic for the size queries: an impossible problem answers INT64_MAX (an allocation that fails), never a
// wrapped, small or negative size
static inline int64_t ebn_sat_mul(int64_t a, int64_t b) {
  int64_t r;
  return (a < 0 || b < 0) ? 0 : (__builtin_mul_overflow(a, b, &r) ? INT64_MAX : r);
}
static inline int64_t ebn_sat_add(int64_t a, int64_t b) {
  int64_t r;
  return __builtin_add_overflow(a, b, &r) ? INT64_MAX : r;
}

static inline hipStream_t ebn_stream(ebn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline bool ebn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- counter-based dropout stream (mirrored in oracle/nrms_numpy.py) -------------
__host__ __device__ __forceinline__ uint32_t ebn_lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

__host__ __device__ __forceinline__ uint32_t ebn_dropout_key(uint32_t seed, uint32_t step, uint32_t site) {
  uint32_t k = ebn_lowbias32(seed ^ 0x9E3779B9u);
  k = k + step * 0x85EBCA6Bu + site * 0xC2B2AE35u;
  return ebn_lowbias32(k);
}

// One 32-bit hash decides TWO consecutive elements (16 bits each): element idx uses the low (even idx) or high
// (odd idx) half of lowbias32(pair ^ key'), pair = idx >> 1, key' = key ^ (bits 32.. of pair) * 0x27D4EB2F.
// `thresh` is a 16-bit threshold: keep iff half >= thresh (drop probability = thresh / 65536).
__host__ __device__ __forceinline__ uint32_t ebn_dropout_pair_hash(uint32_t key, uint64_t pair) {
  return ebn_lowbias32(static_cast<uint32_t>(pair) ^ key ^ (static_cast<uint32_t>(pair >> 32) * 0x27D4EB2Fu));
}

__host__ __device__ __forceinline__ bool ebn_dropout_keep(uint32_t key, uint64_t idx, uint32_t thresh) {
  const uint32_t h = ebn_dropout_pair_hash(key, idx >> 1);
  const uint32_t half = (idx & 1u) ? (h >> 16) : (h & 0xFFFFu);
  return half >= thresh;
}

static inline uint32_t ebn_dropout_threshold(float p) {
  double t = static_cast<double>(p) * 65536.0;
  if (t < 0) t = 0;
  if (t > 65535.0) t = 65535.0;
  return static_cast<uint32_t>(t);
}

// Dropout parameters resolved on the host; `key_ptr` is read on the device so that a
// captured graph sees the key of the current step.
struct EbnDrop {
  const uint32_t* key_ptr;  // nullptr -> dropout disabled
  uint32_t thresh;
  float scale;
};

static inline EbnDrop ebn_make_drop(const ebn_step_state* st, int32_t site, float p) {
  EbnDrop d;
  if (st == nullptr || p <= 0.0f || site < 0 || site >= EBN_N_SITES) {
    d.key_ptr = nullptr;
    d.thresh = 0;
    d.scale = 1.0f;
  } else {
    d.key_ptr = &st->drop_key[site];
    d.thresh = ebn_dropout_threshold(p);
    d.scale = 1.0f / (1.0f - p);
  }
  return d;
}

__device__ __forceinline__ float ebn_drop_mult(uint32_t key, uint64_t idx, uint32_t thresh, float scale) {
  return ebn_dropout_keep(key, idx, thresh) ? scale : 0.0f;
}

// ---- wave-level reductions (64 lanes) ------------------------------------------
__device__ __forceinline__ float ebn_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float ebn_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// reductions over aligned groups of GW lanes (GW = 32 or 64)
template <int GW>
__device__ __forceinline__ float ebn_group_sum(float v) {
#pragma unroll
  for (int off = GW / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

static inline int64_t ebn_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
