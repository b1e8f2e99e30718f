// Does VALU work hide under fp32 MFMAs on gfx950?  Times, per SIMD:
//   M: a loop of independent v_mfma_f32_32x32x2_f32          (matrix pipe only)
//   V: a loop of independent v_fma_f32                        (vector ALU only)
//   S: both in ONE wave, interleaved                          (same-wave overlap)
//   X: two waves per SIMD, one runs M and the other V         (cross-wave overlap)
// build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITERS = 2000;
constexpr int MFMA_PER_ITER = 4;    // 4 independent accumulators
constexpr int FMA_PER_ITER = 64;    // 64 cycles of MFMA = 16 VALU issue slots per MFMA -> 4 MFMAs ~ 64 fmas

template <bool DO_M, bool DO_V>
__device__ __forceinline__ float body(float seed) {
  f32x16 acc[MFMA_PER_ITER];
  for (int i = 0; i < MFMA_PER_ITER; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = seed + i;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed * (i + 1);
  const float a = seed + 1.f, b = seed + 2.f;
  for (int it = 0; it < ITERS; ++it) {
    if (DO_M) {
#pragma unroll
      for (int i = 0; i < MFMA_PER_ITER; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    if (DO_V) {
#pragma unroll
      for (int k = 0; k < FMA_PER_ITER / 16; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], a, b);
    }
  }
  float s = 0.f;
  for (int i = 0; i < MFMA_PER_ITER; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  return s;
}

// D: ONE accumulator -- every MFMA depends on the previous one (the shape of the attention kernels' 10- and 16-step chains)
__global__ __launch_bounds__(512) void kdep(float* out, float seed) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = seed;
  const float a = seed + 1.f, b = seed + 2.f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < MFMA_PER_ITER; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, float seed) {
  float s;
  if (MODE == 0) s = body<true, false>(seed);
  else if (MODE == 1) s = body<false, true>(seed);
  else if (MODE == 2) s = body<true, true>(seed);
  else {  // 8 waves per workgroup = 2 per SIMD: waves 0-3 run M, waves 4-7 run V
    if ((threadIdx.x >> 6) < 4) s = body<true, false>(seed);
    else s = body<false, true>(seed);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 1.0f);
  hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 10 * 1e3f;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const float m = run<0>(out, 256), v = run<1>(out, 256), s = run<2>(out, 256), x = run<3>(out, 512);
  const float m2 = run<0>(out, 512), v2 = run<1>(out, 512);
  float d1, d2;
  {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int threads : {256, 512}) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kdep, dim3(256), dim3(threads), 0, 0, out, 1.0f);
      hipEventRecord(e0);
      for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kdep, dim3(256), dim3(threads), 0, 0, out, 1.0f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      (threads == 256 ? d1 : d2) = ms / 10 * 1e3f;
    }
  }
  printf("dependent chain (one accumulator): one wave per SIMD %.1f us, two waves per SIMD %.1f us\n", d1, d2);
  printf("one wave per SIMD:  M %.1f us   V %.1f us   same-wave M+V %.1f us  (sum %.1f, max %.1f)\n", m, v, s, m + v, m > v ? m : v);
  printf("two waves per SIMD: M|V split %.1f us   (M alone on 2 waves %.1f, V alone on 2 waves %.1f)\n", x, m2, v2);
  printf("expected M: %d MFMAs x 64 cycles = %.1f us at 2.4 GHz\n", ITERS * MFMA_PER_ITER, ITERS * MFMA_PER_ITER * 64 / 2400.0);
  return 0;
}
