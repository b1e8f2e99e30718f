// How much of the attention kernels' memory time is the SHAPE of their accesses?  The backward group kernel moves, per (title, group of 4
// heads): 30 rows x {Q, K, V pieces of 320 B out of a 4800-byte QKV row, a 320-B piece of a 1600-byte dO row} in, 30 rows x 3 x 320 B
// of dQKV out -- 268 MB per c2 step in 66 us (4.1 TB/s) even with the arithmetic removed (profiles/r03_tuning_notes.md).  This probe
// moves the SAME bytes with the same workgroup geometry (256 threads, one (title, group) per workgroup, registers -> registers, no LDS,
// no arithmetic) in two layouts of the 1200-column matrices:
//   split:       columns [Q | K | V] x [head] x [d]      -> three 320-byte pieces per row (what Keras' weight order gives)
//   interleaved: columns [head] x [Q | K | V] x [d]      -> one 960-byte piece per row
// and reports GB/s for reads only, writes only, and both.   build: hipcc -O3 --offload-arch=gfx950 row_piece_copy.hip -o row_piece_copy
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int L = 30, H = 20, D = 20, E = H * D, G = 4, GD = G * D;  // 80 floats = 320 B per piece

template <bool INTERLEAVED, bool DO_READ, bool DO_WRITE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ qkv, const float4* __restrict__ dout, float4* __restrict__ out, int n_titles) {
  const int wg = blockIdx.x, title = wg / (H / G), g = wg % (H / G);
  const int64_t row0 = static_cast<int64_t>(title) * L;
  // per row: 3 * 20 float4 of QKV + 20 float4 of dO = 80 float4; 30 rows -> 2400 float4 per workgroup, ~9.4 per thread
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    const int idx = threadIdx.x + 256 * t;
    const int idc = idx < L * 80 ? idx : 0;
    const int r = idc / 80, c = idc % 80;  // c < 60: QKV piece float4 c; else dO piece
    int64_t off;
    if (c < 60) {
      const int part = c / 20, c4 = c % 20;
      off = (row0 + r) * (3 * E / 4) + (INTERLEAVED ? g * 60 + c : part * (E / 4) + g * 20 + c4);
      v[t] = DO_READ ? qkv[off] : make_float4(1.f, 2.f, 3.f, 4.f);
    } else {
      off = (row0 + r) * (E / 4) + g * 20 + (c - 60);
      v[t] = DO_READ ? dout[off] : make_float4(1.f, 2.f, 3.f, 4.f);
    }
  }
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    const int idx = threadIdx.x + 256 * t;
    if (idx >= L * 80) continue;
    const int r = idx / 80, c = idx % 80;
    if (c < 60) {
      const int part = c / 20, c4 = c % 20;
      const int64_t off = (row0 + r) * (3 * E / 4) + (INTERLEAVED ? g * 60 + c : part * (E / 4) + g * 20 + c4);
      if (DO_WRITE) out[off] = v[t];
      else { acc.x += v[t].x; acc.y += v[t].y; acc.z += v[t].z; acc.w += v[t].w; }
    } else if (!DO_WRITE) {
      acc.x += v[t].x;
    }
  }
  if (!DO_WRITE && acc.x == 12345.678f) out[0] = acc;  // keep the loads
}

template <class K>
float timeit(K kern, int grid, const float4* a, const float4* b, float4* c, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, b, c, n);
  hipEventRecord(e0);
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, b, c, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1e3f;
}

int main() {
  for (int n_titles : {800, 1760, 3200}) {  // c2 (24000 rows), c4 (52800), c5h50-ish: QKV 115 / 253 / 461 MB
    const size_t R = static_cast<size_t>(n_titles) * L;
    float4 *qkv, *dout, *out;
    hipMalloc(&qkv, R * 3 * E * 4);
    hipMalloc(&dout, R * E * 4);
    hipMalloc(&out, R * 3 * E * 4);
    hipMemset(qkv, 0, R * 3 * E * 4);
    hipMemset(dout, 0, R * E * 4);
    const int grid = n_titles * (H / G);
    const double rd = double(R) * 4 * E * 4, wr = double(R) * 3 * E * 4;
    const float a0 = timeit(k<false, true, false>, grid, qkv, dout, out, n_titles), a1 = timeit(k<false, false, true>, grid, qkv, dout, out, n_titles),
                a2 = timeit(k<false, true, true>, grid, qkv, dout, out, n_titles);
    const float b0 = timeit(k<true, true, false>, grid, qkv, dout, out, n_titles), b1 = timeit(k<true, false, true>, grid, qkv, dout, out, n_titles),
                b2 = timeit(k<true, true, true>, grid, qkv, dout, out, n_titles);
    printf("%5d titles (%6.0f MB in, %6.0f MB out)  split [Q|K|V][head]: read %6.1f us %5.0f GB/s | write %6.1f us %5.0f GB/s | both %6.1f us %5.0f GB/s\n", n_titles,
           rd / 1e6, wr / 1e6, a0, rd / a0 / 1e3, a1, wr / a1 / 1e3, a2, (rd + wr) / a2 / 1e3);
    printf("%5d titles                                interleaved [head][Q|K|V]: read %6.1f us %5.0f GB/s | write %6.1f us %5.0f GB/s | both %6.1f us %5.0f GB/s\n", n_titles,
           b0, rd / b0 / 1e3, b1, wr / b1 / 1e3, b2, (rd + wr) / b2 / 1e3);
    hipFree(qkv);
    hipFree(dout);
    hipFree(out);
  }
  return 0;
}
