// a1: title-token embedding gather (+ fused inverted dropout) and its backward scatter.
// Replaces tf.keras.layers.Embedding at nrms.py:125-134 and Dropout at nrms.py:136.
//
// HBM-bound: per token the kernel reads 4 B of id + D*4 B of table row and writes D*4 B.
// Work items are 16-byte vectors; consecutive lanes take consecutive vectors of a row so a
// wave covers whole 128-B lines of the (V x D) table; each thread keeps GATHER_UNROLL independent
// row loads in flight before the first store (random rows -> latency-bound otherwise).
#include "ebn_common.h"

namespace {

constexpr int GATHER_THREADS = 256;
constexpr int GATHER_UNROLL = 2;  // measured 1..8 on MI355X: 2 is best with the fused dropout (0.70 of the HBM peak at c2; 4 -> 0.66, 8 -> 0.65)

// IT: index type of a work item (uint32_t whenever n_items < 2^32: a 64-bit division by the runtime row length costs
// ~100 VALU instructions per item and was a third of this kernel's time); SHIFT >= 0: vpr == 1 << SHIFT (D = 1024
// -> 8), the row/column split is a shift and a mask.
template <typename IT, int SHIFT>
__global__ __launch_bounds__(GATHER_THREADS) void gather_rows_vec4_kernel(
    const int32_t* __restrict__ ids, const float4* __restrict__ table, float4* __restrict__ out,
    int64_t n_items, int32_t vpr, int64_t V, const uint32_t* __restrict__ key_ptr, uint32_t thresh,
    float scale, int32_t* __restrict__ oob_flag) {
  const IT base = (static_cast<IT>(blockIdx.x) * GATHER_UNROLL) * GATHER_THREADS + threadIdx.x;
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  float4 v[GATHER_UNROLL];
  IT item[GATHER_UNROLL];
  bool ok[GATHER_UNROLL];
  int32_t col[GATHER_UNROLL];
  int64_t id[GATHER_UNROLL];
  // ids first, then the rows: every load is unconditional with a clamped address (a load under a per-item guard is
  // serialised behind the previous item's), validity is applied with selects afterwards
#pragma unroll
  for (int u = 0; u < GATHER_UNROLL; ++u) {
    item[u] = base + static_cast<IT>(u) * GATHER_THREADS;
    ok[u] = static_cast<int64_t>(item[u]) < n_items;
    const IT it = ok[u] ? item[u] : 0;
    const IT row = (SHIFT >= 0) ? (it >> (SHIFT >= 0 ? SHIFT : 0)) : it / static_cast<IT>(vpr);
    col[u] = static_cast<int32_t>(it - row * static_cast<IT>(vpr));
    id[u] = ids[row];
  }
  bool bad = false;
#pragma unroll
  for (int u = 0; u < GATHER_UNROLL; ++u) {
    const bool in_range = id[u] >= 0 && id[u] < V;
    bad |= ok[u] && !in_range;
    const float4 t = table[(in_range ? id[u] : 0) * vpr + col[u]];
    v[u] = (ok[u] && in_range) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (bad && oob_flag != nullptr) *oob_flag = 1;
#pragma unroll
  for (int u = 0; u < GATHER_UNROLL; ++u) {
    if (!ok[u]) continue;
    if (do_drop) {
      const uint64_t pair0 = static_cast<uint64_t>(item[u]) * 2u;  // elements 4*item .. 4*item+3 = pairs 2*item, 2*item+1
      const uint32_t h0 = ebn_dropout_pair_hash(key, pair0), h1 = ebn_dropout_pair_hash(key, pair0 + 1);
      v[u].x = ((h0 & 0xFFFFu) >= thresh) ? v[u].x * scale : 0.f;
      v[u].y = ((h0 >> 16) >= thresh) ? v[u].y * scale : 0.f;
      v[u].z = ((h1 & 0xFFFFu) >= thresh) ? v[u].z * scale : 0.f;
      v[u].w = ((h1 >> 16) >= thresh) ? v[u].w * scale : 0.f;
    }
    out[item[u]] = v[u];
  }
}

// D not a multiple of 4 (or unaligned base): one float per work item.
__global__ __launch_bounds__(GATHER_THREADS) void gather_rows_scalar_kernel(
    const int32_t* __restrict__ ids, const float* __restrict__ table, float* __restrict__ out,
    int64_t n_items, int32_t D, int64_t V, const uint32_t* __restrict__ key_ptr, uint32_t thresh,
    float scale, int32_t* __restrict__ oob_flag) {
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  for (int64_t it = static_cast<int64_t>(blockIdx.x) * GATHER_THREADS + threadIdx.x; it < n_items;
       it += static_cast<int64_t>(gridDim.x) * GATHER_THREADS) {
    const int64_t row = it / D;
    const int32_t col = static_cast<int32_t>(it - row * D);
    const int64_t id = ids[row];
    float x = 0.f;
    if (id >= 0 && id < V) {
      x = table[id * D + col];
    } else if (oob_flag != nullptr) {
      *oob_flag = 1;
    }
    if (do_drop) x *= ebn_drop_mult(key, static_cast<uint64_t>(it), thresh, scale);
    out[it] = x;
  }
}

// Backward: dense dTable[ids[r],:] += dX[r,:]*mult. Hardware fp32 atomics (global_atomic_add_f32);
// hot rows (token 0 of padded history) serialise in L2, not in HBM.
template <typename IT>
__global__ __launch_bounds__(GATHER_THREADS) void scatter_add_rows_kernel(
    const int32_t* __restrict__ ids, const float* __restrict__ dX, float* __restrict__ dTable,
    int64_t n_items, int32_t D, int64_t V, const uint32_t* __restrict__ key_ptr, uint32_t thresh,
    float scale) {
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  for (IT it = static_cast<IT>(blockIdx.x) * GATHER_THREADS + threadIdx.x; static_cast<int64_t>(it) < n_items;
       it += static_cast<IT>(gridDim.x) * GATHER_THREADS) {
    const IT row = it / static_cast<IT>(D);  // 32-bit when the item count allows: see gather_rows_vec4_kernel
    const int32_t col = static_cast<int32_t>(it - row * static_cast<IT>(D));
    const int64_t id = ids[row];
    if (id < 0 || id >= V) continue;
    float g = dX[it];
    if (do_drop) g *= ebn_drop_mult(key, static_cast<uint64_t>(it), thresh, scale);
    if (g != 0.f) unsafeAtomicAdd(&dTable[id * D + col], g);
  }
}

constexpr double FIXED_SCALE = 1099511627776.0;  // 2^40
constexpr float FIXED_TERM_MAX = 2097152.0f;      // 2^21: any single gradient term this large raises the range flag
constexpr long long FIXED_SUM_MAX = 1ll << 62;    // |accumulated sum| >= 2^22 (as 2^40-scaled int64): range flag

// Deterministic backward: 64-bit integer atomics on a fixed-point accumulator (order-independent sums).
template <typename IT>
__global__ __launch_bounds__(GATHER_THREADS) void scatter_add_rows_fixed_kernel(
    const int32_t* __restrict__ ids, const float* __restrict__ dX, long long* __restrict__ acc, int64_t n_items,
    int32_t D, int64_t V, const uint32_t* __restrict__ key_ptr, uint32_t thresh, float scale,
    int32_t* __restrict__ range_flag) {
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  for (IT it = static_cast<IT>(blockIdx.x) * GATHER_THREADS + threadIdx.x; static_cast<int64_t>(it) < n_items;
       it += static_cast<IT>(gridDim.x) * GATHER_THREADS) {
    const IT row = it / static_cast<IT>(D);  // 32-bit when the item count allows: see gather_rows_vec4_kernel
    const int32_t col = static_cast<int32_t>(it - row * static_cast<IT>(D));
    const int64_t id = ids[row];
    if (id < 0 || id >= V) continue;
    float g = dX[it];
    if (do_drop) g *= ebn_drop_mult(key, static_cast<uint64_t>(it), thresh, scale);
    if (g != 0.f) {
      // one term at or beyond 2^21 (or a NaN): the 2^40-scaled sum could leave the int64 range unnoticed -> flag it
      if (!(fabsf(g) < FIXED_TERM_MAX) && range_flag != nullptr) *range_flag = 1;
      const long long q = __double2ll_rn(static_cast<double>(g) * FIXED_SCALE);
      atomicAdd(reinterpret_cast<unsigned long long*>(&acc[id * D + col]), static_cast<unsigned long long>(q));
    }
  }
}

__global__ __launch_bounds__(256) void fixed_to_f32_kernel(long long* __restrict__ acc, float* __restrict__ out,
                                                           int64_t n, int32_t* __restrict__ range_flag) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const long long q = acc[i];
    out[i] = static_cast<float>(static_cast<double>(q) * (1.0 / FIXED_SCALE));
    if (q != 0) {
      acc[i] = 0;
      if ((q >= FIXED_SUM_MAX || q <= -FIXED_SUM_MAX) && range_flag != nullptr) *range_flag = 1;
    }
  }
}

// Keras-form Adam (ebn_score_optim.hip: adam_keras_kernel) reading the gradient straight from the fixed-point
// accumulator: convert, update theta / m / v and re-zero the accumulator in ONE pass over the table -- 4 reads + 4 writes
// per element where fixed_to_f32 + Adam made 6 reads + 5 writes (the table sweep is what a trainable-table step pays on
// top of the frozen-table one).  Same arithmetic as the two kernels in sequence.
__global__ __launch_bounds__(256) void adam_keras_fixed_kernel(float* __restrict__ theta, long long* __restrict__ acc,
                                                               float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                               const ebn_step_state* __restrict__ st, float omb1,
                                                               float omb2, float eps, float gscale,
                                                               int32_t* __restrict__ range_flag) {
  const float alpha = st->adam_alpha;
  bool bad = false;
#define EBN_ADAMQ(T, Q, M, V)                                                                    \
  {                                                                                              \
    const float gg = static_cast<float>(static_cast<double>(Q) * (1.0 / FIXED_SCALE)) * gscale; \
    (M) = (M) + (gg - (M)) * omb1;                                                               \
    (V) = (V) + (gg * gg - (V)) * omb2;                                                          \
    (T) = (T) - alpha * (M) / (sqrtf(V) + eps);                                                  \
    bad |= (Q) >= FIXED_SUM_MAX || (Q) <= -FIXED_SUM_MAX;                                        \
  }
  const int64_t n4 = n / 4;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<int64_t>(gridDim.x) * 256) {
    const longlong2 q0 = reinterpret_cast<const longlong2*>(acc)[2 * i], q1 = reinterpret_cast<const longlong2*>(acc)[2 * i + 1];
    float4 t = reinterpret_cast<const float4*>(theta)[i], mm = reinterpret_cast<const float4*>(m)[i],
           vv = reinterpret_cast<const float4*>(v)[i];
    EBN_ADAMQ(t.x, q0.x, mm.x, vv.x)
    EBN_ADAMQ(t.y, q0.y, mm.y, vv.y)
    EBN_ADAMQ(t.z, q1.x, mm.z, vv.z)
    EBN_ADAMQ(t.w, q1.y, mm.w, vv.w)
    reinterpret_cast<float4*>(theta)[i] = t;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if ((q0.x | q0.y) != 0) reinterpret_cast<longlong2*>(acc)[2 * i] = make_longlong2(0, 0);
    if ((q1.x | q1.y) != 0) reinterpret_cast<longlong2*>(acc)[2 * i + 1] = make_longlong2(0, 0);
  }
  for (int64_t i = n4 * 4 + static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) {
    const long long q = acc[i];
    float t = theta[i], mm = m[i], vv = v[i];
    EBN_ADAMQ(t, q, mm, vv)
    theta[i] = t;
    m[i] = mm;
    v[i] = vv;
    if (q != 0) acc[i] = 0;
  }
#undef EBN_ADAMQ
  if (bad && range_flag != nullptr) *range_flag = 1;
}

// ---- the same deterministic accumulation with the duplicates of a workgroup's tokens combined BEFORE they reach the atomics.
// Real EB-NeRD batches hammer a few table rows: a left-padded history slot is a title of 30 x token 0 (_behaviors.py:647-654),
// unknown articles map to the all-zero title too (dataloader.py:43), and token frequencies are Zipfian -- under SURVEY 8(d)'s Z
// inputs ~20 % of a step's gradient rows land on row 0 and the one-atomic-per-element kernel above runs 2.8-3.1 x slower than on
// uniform ids (c1 49 -> 139 us, c4 102 -> 318 us: thousands of 64-bit atomics serialising on each of row 0's D addresses).
// Here a workgroup owns RUN_CHUNK consecutive tokens: wave 0 sorts their (id, position) pairs (bitonic network over the 64
// lanes, shuffles only), then thread = column walks the tokens in id order, sums every run of equal ids in a register and
// issues ONE atomic per (distinct id, column).  The sums are 2^40-scaled integers (wrapping int64 addition is associative), so
// the accumulator is bit-identical to the plain kernel's whatever the grouping -- uniform ids pay one shuffle sort per 64 tokens
// and issue the same atomics as before; Zipf ids issue fewer atomics than uniform ones.
// (Round 3 built a global counting sort + segmented reduction instead: four launches, 204 us at c1 -- measured slower than
// the atomics on uniform AND on Zipf ids (round 4: 1.127 / 1.244 ms per c1 step against 0.968 / 1.060), and removed.)
constexpr int RUN_CHUNK = 64;
__global__ __launch_bounds__(1024) void scatter_add_rows_fixed_runs_kernel(
    const int32_t* __restrict__ ids, const float* __restrict__ dX, long long* __restrict__ acc, int64_t n_tok, int32_t D, int64_t V,
    const uint32_t* __restrict__ key_ptr, uint32_t thresh, float scale, int32_t* __restrict__ range_flag) {
  __shared__ int32_t s_id[RUN_CHUNK + 1], s_pos[RUN_CHUNK];
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * RUN_CHUNK;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int64_t t = t0 + lane;
    const int64_t id = t < n_tok ? static_cast<int64_t>(ids[t]) : -1;
    // key = (id, position): unique, so the sort is a total order; ids outside [0, V) (and the tail of a short chunk) sort last
    long long k = (id >= 0 && id < V) ? ((id << 6) | lane) : 0x7FFFFFFFFFFFFFFFll;
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
      for (int j = size >> 1; j > 0; j >>= 1) {
        const long long o = __shfl_xor(k, j, 64);
        const bool asc = (lane & size) == 0, lower = (lane & j) == 0;
        k = (lower == asc) ? (k < o ? k : o) : (k > o ? k : o);
      }
    }
    const bool ok = k != 0x7FFFFFFFFFFFFFFFll;
    s_id[lane] = ok ? static_cast<int32_t>(k >> 6) : -1;
    s_pos[lane] = ok ? static_cast<int32_t>(k & 63) : 0;
    if (lane == 0) s_id[RUN_CHUNK] = -1;
  }
  __syncthreads();
  const bool do_drop = key_ptr != nullptr;
  const uint32_t key = do_drop ? *key_ptr : 0u;
  bool bad = false;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {  // consecutive lanes on consecutive columns: dX rows are read in 256-byte pieces
    unsigned long long run = 0;
    for (int j0 = 0; j0 < RUN_CHUNK; j0 += 8) {
      if (s_id[j0] < 0) break;  // (block-uniform) nothing but skipped tokens from here on
      float g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = dX[(t0 + s_pos[j0 + j]) * D + c];  // all 8 loads first; skipped slots re-read position 0 of the chunk
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int32_t id = s_id[j0 + j];
        if (id < 0) break;
        float x = g[j];
        if (do_drop) x *= ebn_drop_mult(key, static_cast<uint64_t>(t0 + s_pos[j0 + j]) * static_cast<uint64_t>(D) + static_cast<uint64_t>(c), thresh, scale);
        if (x != 0.f) {
          bad |= !(fabsf(x) < FIXED_TERM_MAX);
          run += static_cast<unsigned long long>(__double2ll_rn(static_cast<double>(x) * FIXED_SCALE));
        }
        if (s_id[j0 + j + 1] != id) {  // (block-uniform) last token of this id in the chunk
          if (run != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[static_cast<int64_t>(id) * D + c]), run);
          run = 0;
        }
      }
    }
  }
  if (bad && range_flag != nullptr) *range_flag = 1;
}

__global__ __launch_bounds__(GATHER_THREADS) void expand_titles_kernel(const int32_t* __restrict__ art_idx,
                                                                       const int32_t* __restrict__ token_matrix,
                                                                       int32_t* __restrict__ ids_out, int64_t n_items,
                                                                       int32_t T, int64_t n_rows,
                                                                       int32_t* __restrict__ oob_flag) {
  for (int64_t it = static_cast<int64_t>(blockIdx.x) * GATHER_THREADS + threadIdx.x; it < n_items;
       it += static_cast<int64_t>(gridDim.x) * GATHER_THREADS) {
    const int64_t r = it / T;
    const int32_t t = static_cast<int32_t>(it - r * T);
    const int64_t a = art_idx[r];
    int32_t v = 0;
    if (a >= 0 && a < n_rows) {
      v = token_matrix[a * T + t];
    } else if (oob_flag != nullptr) {
      *oob_flag = 1;
    }
    ids_out[it] = v;
  }
}

}  // namespace

namespace {
int scatter_fixed_atomic_launch(const int32_t* ids, const float* dX, int64_t* acc, int64_t n_tok, int32_t D, int64_t V, const EbnDrop& dr,
                                int32_t* range_flag, hipStream_t s) {
  const int64_t n_items = n_tok * D;
  int64_t grid = ebn_ceil_div(n_items, GATHER_THREADS);
  if (grid > 256 * 32) grid = 256 * 32;
  if (n_items + grid * GATHER_THREADS < (static_cast<int64_t>(1) << 32))
    EBN_LAUNCH(scatter_add_rows_fixed_kernel<uint32_t>, dim3(static_cast<unsigned>(grid)), dim3(GATHER_THREADS), 0, s, ids, dX,
                       reinterpret_cast<long long*>(acc), n_items, D, V, dr.key_ptr, dr.thresh, dr.scale, range_flag);
  else
    EBN_LAUNCH(scatter_add_rows_fixed_kernel<int64_t>, dim3(static_cast<unsigned>(grid)), dim3(GATHER_THREADS), 0, s, ids, dX,
                       reinterpret_cast<long long*>(acc), n_items, D, V, dr.key_ptr, dr.thresh, dr.scale, range_flag);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
}  // namespace

extern "C" int ebn_embedding_grad_scatter_fixed(const int32_t* ids, const float* dX, int64_t* acc, int64_t n_tok,
                                                int32_t D, int64_t V, const ebn_step_state* st, int32_t site,
                                                float drop_p, int32_t* range_flag, ebn_stream_t stream) {
  EBN_REQUIRE(ids && dX && acc, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_tok >= 0 && D > 0 && V > 0, EBN_ERR_BAD_ARG);
  if (n_tok == 0) return EBN_OK;
  const EbnDrop dr = ebn_make_drop(st, site, drop_p);
  // (id << 6) | position must fit the sort key, and a chunk index the grid: both hold for every table this path can address
  if (V >= (int64_t{1} << 56) || ebn_ceil_div(n_tok, RUN_CHUNK) >= (int64_t{1} << 31))
    return scatter_fixed_atomic_launch(ids, dX, acc, n_tok, D, V, dr, range_flag, ebn_stream(stream));
  int threads = static_cast<int>(ebn_ceil_div(D, 64) * 64);
  if (threads > 1024) threads = 1024;
  EBN_LAUNCH(scatter_add_rows_fixed_runs_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(n_tok, RUN_CHUNK))), dim3(threads), 0,
                     ebn_stream(stream), ids, dX, reinterpret_cast<long long*>(acc), n_tok, D, V, dr.key_ptr, dr.thresh, dr.scale, range_flag);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_embedding_grad_scatter_fixed_atomic(const int32_t* ids, const float* dX, int64_t* acc, int64_t n_tok,
                                                       int32_t D, int64_t V, const ebn_step_state* st, int32_t site,
                                                       float drop_p, int32_t* range_flag, ebn_stream_t stream) {
  EBN_REQUIRE(ids && dX && acc, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_tok >= 0 && D > 0 && V > 0, EBN_ERR_BAD_ARG);
  if (n_tok == 0) return EBN_OK;
  return scatter_fixed_atomic_launch(ids, dX, acc, n_tok, D, V, ebn_make_drop(st, site, drop_p), range_flag, ebn_stream(stream));
}

extern "C" int ebn_fixed_to_f32(int64_t* acc, float* out, int64_t n, int32_t* range_flag, ebn_stream_t stream) {
  EBN_REQUIRE(acc && out && n >= 0, EBN_ERR_BAD_ARG);
  if (n == 0) return EBN_OK;
  int64_t grid = ebn_ceil_div(n, 256);
  if (grid > 256 * 16) grid = 256 * 16;
  EBN_LAUNCH(fixed_to_f32_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, ebn_stream(stream),
                     reinterpret_cast<long long*>(acc), out, n, range_flag);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_adam_keras_step_fixed_f32(float* theta, int64_t* acc, float* m, float* v, int64_t n,
                                             const ebn_step_state* st, double beta1, double beta2, double eps_d,
                                             float grad_scale, int32_t* range_flag, ebn_stream_t stream) {
  EBN_REQUIRE(theta && acc && m && v && st && n >= 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(ebn_aligned16(acc) && ebn_aligned16(theta) && ebn_aligned16(m) && ebn_aligned16(v), EBN_ERR_ALIGN);
  if (n == 0) return EBN_OK;
  int64_t grid = ebn_ceil_div(ebn_ceil_div(n, 4), 256);
  if (grid > 256 * 16) grid = 256 * 16;
  EBN_LAUNCH(adam_keras_fixed_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, ebn_stream(stream), theta,
                     reinterpret_cast<long long*>(acc), m, v, n, st, static_cast<float>(1.0 - beta1),
                     static_cast<float>(1.0 - beta2), static_cast<float>(eps_d), grad_scale, range_flag);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_expand_titles_i32(const int32_t* art_idx, const int32_t* token_matrix, int32_t* ids_out,
                                     int64_t n_titles, int32_t T, int64_t n_rows, int32_t* oob_flag,
                                     ebn_stream_t stream) {
  EBN_REQUIRE(art_idx && token_matrix && ids_out, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_titles >= 0 && T > 0 && n_rows > 0, EBN_ERR_BAD_ARG);
  if (n_titles == 0) return EBN_OK;
  const int64_t n_items = n_titles * T;
  int64_t grid = ebn_ceil_div(n_items, GATHER_THREADS);
  if (grid > 256 * 32) grid = 256 * 32;
  EBN_LAUNCH(expand_titles_kernel, dim3(static_cast<unsigned>(grid)), dim3(GATHER_THREADS), 0,
                     ebn_stream(stream), art_idx, token_matrix, ids_out, n_items, T, n_rows, oob_flag);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_gather_rows_f32(const int32_t* ids, const float* table, float* out, int64_t n_tok,
                                   int32_t D, int64_t V, const ebn_step_state* st, int32_t site,
                                   float drop_p, int32_t* oob_flag, ebn_stream_t stream) {
  EBN_REQUIRE(ids && table && out, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_tok >= 0 && D > 0 && V > 0, EBN_ERR_BAD_ARG);
  if (n_tok == 0) return EBN_OK;
  const EbnDrop dr = ebn_make_drop(st, site, drop_p);
  if ((D % 4) == 0 && ebn_aligned16(table) && ebn_aligned16(out)) {
    const int32_t vpr = D / 4;
    const int64_t n_items = n_tok * vpr;
    const int64_t per_block = static_cast<int64_t>(GATHER_THREADS) * GATHER_UNROLL;
    const int64_t grid = ebn_ceil_div(n_items, per_block);
    EBN_REQUIRE(grid <= 0x7FFFFFFF, EBN_ERR_UNSUPPORTED);
    const bool small = n_items + per_block < (static_cast<int64_t>(1) << 32);  // every item index fits 32 bits
#define EBN_GATHER_LAUNCH(IT, SH)                                                                                     \
  EBN_LAUNCH((gather_rows_vec4_kernel<IT, SH>), dim3(static_cast<unsigned>(grid)), dim3(GATHER_THREADS), 0,   \
                     ebn_stream(stream), ids, reinterpret_cast<const float4*>(table), reinterpret_cast<float4*>(out), \
                     n_items, vpr, V, dr.key_ptr, dr.thresh, dr.scale, oob_flag)
    if (small && vpr == 256) EBN_GATHER_LAUNCH(uint32_t, 8);
    else if (small && vpr == 64) EBN_GATHER_LAUNCH(uint32_t, 6);
    else if (small) EBN_GATHER_LAUNCH(uint32_t, -1);
    else EBN_GATHER_LAUNCH(int64_t, -1);
#undef EBN_GATHER_LAUNCH
  } else {
    const int64_t n_items = n_tok * D;
    int64_t grid = ebn_ceil_div(n_items, GATHER_THREADS);
    if (grid > 256 * 32) grid = 256 * 32;
    EBN_LAUNCH(gather_rows_scalar_kernel, dim3(static_cast<unsigned>(grid)), dim3(GATHER_THREADS),
                       0, ebn_stream(stream), ids, table, out, n_items, D, V, dr.key_ptr, dr.thresh,
                       dr.scale, oob_flag);
  }
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_embedding_grad_scatter_f32(const int32_t* ids, const float* dX, float* dTable,
                                              int64_t n_tok, int32_t D, int64_t V,
                                              const ebn_step_state* st, int32_t site, float drop_p,
                                              ebn_stream_t stream) {
  EBN_REQUIRE(ids && dX && dTable, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_tok >= 0 && D > 0 && V > 0, EBN_ERR_BAD_ARG);
  if (n_tok == 0) return EBN_OK;
  const EbnDrop dr = ebn_make_drop(st, site, drop_p);
  const int64_t n_items = n_tok * D;
  int64_t grid = ebn_ceil_div(n_items, GATHER_THREADS);
  if (grid > 256 * 32) grid = 256 * 32;
  if (n_items + grid * GATHER_THREADS < (static_cast<int64_t>(1) << 32))
    EBN_LAUNCH(scatter_add_rows_kernel<uint32_t>, dim3(static_cast<unsigned>(grid)), dim3(GATHER_THREADS), 0,
                       ebn_stream(stream), ids, dX, dTable, n_items, D, V, dr.key_ptr, dr.thresh, dr.scale);
  else
    EBN_LAUNCH(scatter_add_rows_kernel<int64_t>, dim3(static_cast<unsigned>(grid)), dim3(GATHER_THREADS), 0,
                       ebn_stream(stream), ids, dX, dTable, n_items, D, V, dr.key_ptr, dr.thresh, dr.scale);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
