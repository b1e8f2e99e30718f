// Issue rate of v_mfma_f32_16x16x4_f32 (and 32x32x2 for scale) from ONE wave per SIMD and from two / three, with NACC independent
// accumulators fed round-robin.  Question behind it (round 4): the LDS-free AttLayer2 GEMM issues 84 independent 16x16x4 MFMAs per
// k group from one wave per SIMD and reaches 52 % matrix-pipe utilisation with 72 % of its wave cycles in SQ_WAIT_INST_ANY.
// build: hipcc -O3 --offload-arch=gfx950 mfma16_issue.hip -o mfma16_issue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITERS = 1000;

template <int NACC, bool DISTINCT_AB>
__global__ __launch_bounds__(768) void k16(float* out, float seed) {
  f32x4 acc[NACC];
  float a[NACC], b[NACC];
  for (int i = 0; i < NACC; ++i) {
    acc[i] = f32x4{seed, seed, seed, seed};
    a[i] = seed + i + threadIdx.x;
    b[i] = seed - i;
  }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(DISTINCT_AB ? a[i] : a[0], DISTINCT_AB ? b[i] : b[0], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(768) void k32(float* out, float seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = seed + i;
  const float a = seed + threadIdx.x, b = seed + 2.f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
float timeit(K kern, int threads, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 1.0f);
  hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 10 * 1e3f;
}

#define ROW16(NACC, DIST)                                                                                          \
  for (int waves : {1, 2, 3}) {                                                                                    \
    const float us = timeit(k16<NACC, DIST>, 256 * waves, out);                                                    \
    const double per = us * 2400.0 / (double(ITERS) * NACC * waves);                                               \
    printf("16x16x4 f32  nacc %2d  %s  %d wave(s)/SIMD: %8.1f us  = %5.1f cycles per MFMA per SIMD at 2.4 GHz (32 = peak)\n", NACC, \
           DIST ? "distinct A/B" : "shared A/B  ", waves, us, per);                                                \
  }
#define ROW32(NACC)                                                                                                \
  for (int waves : {1, 2, 3}) {                                                                                    \
    const float us = timeit(k32<NACC>, 256 * waves, out);                                                          \
    const double per = us * 2400.0 / (double(ITERS) * NACC * waves);                                               \
    printf("32x32x2 f32  nacc %2d                %d wave(s)/SIMD: %8.1f us  = %5.1f cycles per MFMA per SIMD at 2.4 GHz (64 = peak)\n", NACC, waves, us, per); \
  }

int main() {
  float* out;
  hipMalloc(&out, 256 * 768 * sizeof(float));
  ROW16(1, false) ROW16(2, false) ROW16(4, false) ROW16(8, false) ROW16(21, false) ROW16(21, true) ROW16(32, true)
  ROW32(1) ROW32(2) ROW32(4) ROW32(8)
  return 0;
}
