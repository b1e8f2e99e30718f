// Row-sharded word-embedding table (BASELINE.json configs[4]): device-side lookup plan.
//
// The reference has one Embedding on one device (nrms.py:125-134).  With the (V x D) table split by rows over W ranks a
// lookup is: distinct token ids of the local batch -> route each to its owner -> owners gather -> rows come back ->
// expand to token order.  Everything data-dependent about that (which ids are distinct, who owns them, where each row
// lands in the exchange buffers) is decided HERE, on the device, into FIXED-CAPACITY buffers, so the step has no host
// sync and the collectives are equal-split all-to-alls whose sizes the host knows up front:
//
//   slot_rows[o*cap + j] = owner-local row number of the j-th distinct id this rank needs from owner o (ascending), -1 pad
//   inv[t]               = o*cap + j of token t                      (row of the received (W*cap, D) buffer)
//   counts[o]            = number of distinct ids requested from owner o; counts[W] = 1 if any exceeded cap (overflow:
//                          the excess ids were dropped -- the host must treat the step as failed); counts[W+1] = 1 if an
//                          id was outside [0, V).  The two flags are STICKY: a plan call only ever raises them, the
//                          caller clears them when it reads them (once per epoch, not per step)
//
// Dedup is sort-free: V is small next to HBM (250 002 rows -> a 1 MB int32 presence map), so tokens mark their id in a
// direct-address map laid out by (owner, local row), a segmented prefix sum over the map numbers the present ids per
// owner, and tokens read their slot back.  All integer work, HBM-bound on ~(2*V + 3*n_tok)*4 bytes.
#include "ebn_common.h"

namespace {

constexpr int PLAN_THREADS = 256;
constexpr int PLAN_PER_THREAD = 8;
constexpr int PLAN_CHUNK = PLAN_THREADS * PLAN_PER_THREAD;  // keys of the map scanned by one workgroup

struct ShardGeom {
  int64_t V;
  int32_t world;
  int32_t cyclic;  // 0: rank o owns the block [o*per, (o+1)*per); 1: rank o owns ids = o (mod world)
  int64_t per;     // rows per rank = ceil(V / world)
  int64_t cap;
  int32_t chunks_per_owner;
};

__device__ __forceinline__ int64_t shard_key(const ShardGeom& g, int64_t id, int32_t* owner) {
  const int64_t o = g.cyclic ? id % g.world : id / g.per;
  const int64_t local = g.cyclic ? id / g.world : id - o * g.per;
  *owner = static_cast<int32_t>(o);
  return o * g.per + local;
}

// mark = 0, slot_rows = -1 (padding), counts[0..world) = 0 (the two flag words behind them are left alone: sticky)
__global__ __launch_bounds__(PLAN_THREADS) void shard_init_kernel(int32_t* __restrict__ mark, int64_t n_mark,
                                                                  int32_t* __restrict__ slot_rows, int64_t n_slot,
                                                                  int32_t* __restrict__ counts, int32_t n_counts) {
  const int64_t total = n_mark + n_slot + n_counts;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * PLAN_THREADS + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * PLAN_THREADS) {
    if (i < n_mark) mark[i] = 0;
    else if (i < n_mark + n_slot) slot_rows[i - n_mark] = -1;
    else counts[i - n_mark - n_slot] = 0;
  }
}

__global__ __launch_bounds__(PLAN_THREADS) void shard_mark_kernel(const int32_t* __restrict__ ids, int64_t n_tok,
                                                                  ShardGeom g, int32_t* __restrict__ mark,
                                                                  int32_t* __restrict__ counts) {
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * PLAN_THREADS + threadIdx.x; t < n_tok;
       t += static_cast<int64_t>(gridDim.x) * PLAN_THREADS) {
    const int64_t id = ids[t];
    if (id < 0 || id >= g.V) {
      counts[g.world + 1] = 1;
      continue;
    }
    int32_t o;
    mark[shard_key(g, id, &o)] = 1;  // racing writers store the same value
  }
}

// block c = (owner, k-th chunk of that owner's `per` keys): number of marked keys in the chunk
__global__ __launch_bounds__(PLAN_THREADS) void shard_chunk_count_kernel(const int32_t* __restrict__ mark, ShardGeom g,
                                                                         int32_t* __restrict__ chunk_sum) {
  const int32_t o = blockIdx.x / g.chunks_per_owner, k = blockIdx.x % g.chunks_per_owner;
  const int64_t lo = static_cast<int64_t>(k) * PLAN_CHUNK;
  int32_t n = 0;
#pragma unroll
  for (int u = 0; u < PLAN_PER_THREAD; ++u) {
    const int64_t local = lo + static_cast<int64_t>(u) * PLAN_THREADS + threadIdx.x;  // coalesced
    n += (local < g.per && mark[o * g.per + local] != 0) ? 1 : 0;
  }
  __shared__ int32_t red[PLAN_THREADS / EBN_WAVE];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t s = 0;
    for (int w = 0; w < PLAN_THREADS / EBN_WAVE; ++w) s += red[w];
    chunk_sum[blockIdx.x] = s;
  }
}

// one thread per owner: exclusive prefix over that owner's chunk counts (a few to a few thousand entries)
__global__ void shard_chunk_scan_kernel(ShardGeom g, int32_t* __restrict__ chunk_sum, int32_t* __restrict__ counts) {
  const int32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= g.world) return;
  int32_t run = 0;
  for (int32_t k = 0; k < g.chunks_per_owner; ++k) {
    const int32_t c = chunk_sum[o * g.chunks_per_owner + k];
    chunk_sum[o * g.chunks_per_owner + k] = run;
    run += c;
  }
  counts[o] = run;
  if (run > g.cap) counts[g.world] = 1;
}

// number the marked keys of each chunk (ascending local row), write the request list and turn the map into key -> slot+1
__global__ __launch_bounds__(PLAN_THREADS) void shard_assign_kernel(int32_t* __restrict__ mark, ShardGeom g,
                                                                    const int32_t* __restrict__ chunk_off,
                                                                    int32_t* __restrict__ slot_rows) {
  const int32_t o = blockIdx.x / g.chunks_per_owner, k = blockIdx.x % g.chunks_per_owner;
  // thread t owns PLAN_PER_THREAD CONSECUTIVE keys so that slots ascend with the local row
  const int64_t first = static_cast<int64_t>(k) * PLAN_CHUNK + static_cast<int64_t>(threadIdx.x) * PLAN_PER_THREAD;
  int32_t present[PLAN_PER_THREAD];
  int32_t n = 0;
#pragma unroll
  for (int u = 0; u < PLAN_PER_THREAD; ++u) {
    const int64_t local = first + u;
    present[u] = (local < g.per && mark[o * g.per + local] != 0) ? 1 : 0;
    n += present[u];
  }
  // exclusive scan of n over the workgroup: in-wave shuffles, then the 4 wave totals
  int32_t incl = n;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t v = __shfl_up(incl, off, 64);
    if ((threadIdx.x & 63) >= off) incl += v;
  }
  __shared__ int32_t wave_tot[PLAN_THREADS / EBN_WAVE];
  if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
  __syncthreads();
  int32_t base = chunk_off[blockIdx.x];
  for (int w = 0; w < static_cast<int>(threadIdx.x >> 6); ++w) base += wave_tot[w];
  int32_t j = base + incl - n;
#pragma unroll
  for (int u = 0; u < PLAN_PER_THREAD; ++u) {
    if (!present[u]) continue;
    const int64_t local = first + u;
    if (j < g.cap) {
      slot_rows[static_cast<int64_t>(o) * g.cap + j] = static_cast<int32_t>(local);
      mark[o * g.per + local] = static_cast<int32_t>(static_cast<int64_t>(o) * g.cap + j) + 1;
    } else {
      mark[o * g.per + local] = 0;  // over capacity: dropped (counts[world] is already set)
    }
    ++j;
  }
}

__global__ __launch_bounds__(PLAN_THREADS) void shard_inverse_kernel(const int32_t* __restrict__ ids, int64_t n_tok,
                                                                     ShardGeom g, const int32_t* __restrict__ mark,
                                                                     int32_t* __restrict__ inv) {
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * PLAN_THREADS + threadIdx.x; t < n_tok;
       t += static_cast<int64_t>(gridDim.x) * PLAN_THREADS) {
    const int64_t id = ids[t];
    int32_t slot = -1;
    if (id >= 0 && id < g.V) {
      int32_t o;
      slot = mark[shard_key(g, id, &o)] - 1;
    }
    inv[t] = slot;  // -1 (out-of-range id or dropped): the expanding gather writes a zero row and raises its flag
  }
}

ShardGeom make_geom(int64_t V, int32_t world, int32_t cyclic, int64_t cap) {
  ShardGeom g;
  g.V = V;
  g.world = world;
  g.cyclic = cyclic ? 1 : 0;
  g.per = ebn_ceil_div(V, world);
  g.cap = cap;
  g.chunks_per_owner = static_cast<int32_t>(ebn_ceil_div(g.per, PLAN_CHUNK));
  return g;
}

}  // namespace

extern "C" int64_t ebn_shard_plan_workspace_ints(int64_t V, int32_t world) {
  if (V <= 0 || world <= 0 || !ebn_dim_ok(V, world)) return 0;
  const ShardGeom g = make_geom(V, world, 0, 1);
  return static_cast<int64_t>(world) * g.per + static_cast<int64_t>(world) * g.chunks_per_owner;
}

extern "C" int ebn_shard_plan_i32(const int32_t* ids, int64_t n_tok, int64_t V, int32_t world, int32_t cyclic,
                                  int64_t cap, int32_t* workspace, int32_t* slot_rows, int32_t* inv, int32_t* counts,
                                  ebn_stream_t stream) {
  EBN_REQUIRE(workspace && slot_rows && counts && (ids || n_tok == 0) && (inv || n_tok == 0), EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_tok >= 0 && V > 0 && world > 0 && cap > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(static_cast<int64_t>(world) * cap < 0x7FFFFFFF && V < 0x7FFFFFFF, EBN_ERR_UNSUPPORTED);
  const ShardGeom g = make_geom(V, world, cyclic, cap);
  EBN_REQUIRE(static_cast<int64_t>(world) * g.chunks_per_owner < 0x7FFFFFFF, EBN_ERR_UNSUPPORTED);
  hipStream_t s = ebn_stream(stream);
  int32_t* mark = workspace;
  int32_t* chunk = workspace + static_cast<int64_t>(world) * g.per;
  // (a kernel, not hipMemsetAsync: memset nodes of a captured hipGraph were seen to leave the tail of `counts` holding
  // garbage on later replays of the graph -- tests/test_multi_rank_gpu.py::test_two_rank_fit_keeps_ranks_in_lock_step)
  {
    const int64_t n_init = static_cast<int64_t>(world) * g.per + static_cast<int64_t>(world) * cap + world;
    const unsigned init_grid = static_cast<unsigned>(ebn_ceil_div(n_init, PLAN_THREADS) < 2048 ? ebn_ceil_div(n_init, PLAN_THREADS) : 2048);
    EBN_LAUNCH(shard_init_kernel, dim3(init_grid), dim3(PLAN_THREADS), 0, s, mark, static_cast<int64_t>(world) * g.per,
                       slot_rows, static_cast<int64_t>(world) * cap, counts, world);
    EBN_CHECK_LAUNCH();
  }
  const unsigned tok_grid = static_cast<unsigned>(n_tok > 0 ? (ebn_ceil_div(n_tok, PLAN_THREADS) < 4096 ? ebn_ceil_div(n_tok, PLAN_THREADS) : 4096) : 1);
  const unsigned n_chunks = static_cast<unsigned>(world) * static_cast<unsigned>(g.chunks_per_owner);
  if (n_tok > 0) {
    EBN_LAUNCH(shard_mark_kernel, dim3(tok_grid), dim3(PLAN_THREADS), 0, s, ids, n_tok, g, mark, counts);
    EBN_CHECK_LAUNCH();
  }
  EBN_LAUNCH(shard_chunk_count_kernel, dim3(n_chunks), dim3(PLAN_THREADS), 0, s, mark, g, chunk);
  EBN_CHECK_LAUNCH();
  EBN_LAUNCH(shard_chunk_scan_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(world, 64))), dim3(64), 0, s, g, chunk, counts);
  EBN_CHECK_LAUNCH();
  EBN_LAUNCH(shard_assign_kernel, dim3(n_chunks), dim3(PLAN_THREADS), 0, s, mark, g, chunk, slot_rows);
  EBN_CHECK_LAUNCH();
  if (n_tok > 0) {
    EBN_LAUNCH(shard_inverse_kernel, dim3(tok_grid), dim3(PLAN_THREADS), 0, s, ids, n_tok, g, mark, inv);
    EBN_CHECK_LAUNCH();
  }
  return EBN_OK;
}
