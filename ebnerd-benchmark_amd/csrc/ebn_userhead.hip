// The "head" of a training step in ONE launch: everything between the user encoder's two GEMM groups.
//
//   user AttLayer2 after its x.W matmul (layers.py:65-81)  ->  Dot + softmax (nrms.py:201-202)  ->  compiled loss
//   (nrms.py:56-67)  ->  d(scores)  ->  d(cand), d(user)  ->  AttLayer2 backward up to d(pre-tanh), d(q), d(b)
//
// All of it is local to one impression: 20 x 200 pre-activations, 20 x 400 attention outputs, 5 x 400 candidate vectors --
// about 60 KB.  As separate launches (attpool_fwd, score_loss_train, sum, attpool_bwd_pool, attpool_bwd_dpre, reduce) these
// are six links of the step's dependent chain at 5-9 us each, every one re-reading from HBM what the previous one wrote;
// here one workgroup per impression stages its rows in LDS once and walks the whole chain out of LDS.  A second, tiny
// launch sums the per-impression partials of d(q), d(b) and the batch loss in a fixed order (deterministic).
//
// Same formulas as ebn_attpool.hip / ebn_score_optim.hip (un-stabilised exp with +1e-7, softmax scorer, the three loss
// kinds); summation orders are this kernel's own and fixed.
#include "ebn_common.h"
#include "ebn_finish.h"

namespace {

constexpr int HEAD_THREADS = 1024;  // 16 waves: every phase is short and latency-bound, the workgroup has its CU to itself
constexpr float HEAD_KERAS_EPS = 1e-7f;  // K.epsilon(), layers.py:75-77

struct HeadArgs {
  float* U;             // [B*L, A] in: x.W of the user AttLayer2; out: d(pre-tanh)
  const float* b;       // [A]
  const float* q;       // [A]
  const float* X;       // [B*L, E] AttLayer2 input (user-level self-attention output)
  const float* cand;    // [B*C, E]
  const float* labels;  // [B*C]
  float* w;             // [B*L]
  float* user;          // [B, E]
  float* scores;        // [B*C]
  float* probs;         // [B*C]
  float* loss_rows;     // [B]
  float* dcand;         // [B*C, E]
  float* duser;         // [B, E]
  float* de;            // [B*L]
  float* partials;      // [B][2][A]: d(q), d(b) of impression b
  int L, C, E, A, loss_kind;
  float inv_batch;
};

__device__ __forceinline__ float head_bce_probs(float p, float y, float* dldp) {  // = bce_probs of ebn_score_optim.hip
  const float lo = HEAD_KERAS_EPS, hi = 1.0f - HEAD_KERAS_EPS;
  const bool in_range = p >= lo && p <= hi;
  const float pc = fminf(fmaxf(p, lo), hi);
  const float a = pc + HEAD_KERAS_EPS, bb = 1.0f - pc + HEAD_KERAS_EPS;
  *dldp = in_range ? (-(y / a) + (1.0f - y) / bb) : 0.f;
  return -(y * logf(a) + (1.0f - y) * logf(bb));
}

// LDS layout (floats): T[L][A] tanh(U+b) | Xs[L][E] | Cs[C][E] | us[E] user | du[E] | e[L] (-> w) | dw[L] (-> de) | sc[C] | ds[C]
__global__ __launch_bounds__(HEAD_THREADS) void user_head_train_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int L = a.L, C = a.C, E = a.E, A = a.A;
  float* T = sm;
  float* Xs = T + L * A;
  float* Cs = Xs + L * E;
  float* us = Cs + C * E;
  float* du = us + E;
  float* ev = du + E;
  float* dwv = ev + L;
  float* sc = dwv + L;
  float* dsv = sc + C;
  const int64_t bi = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A4 = A >> 2, E4 = E >> 2;
  // ---- stage this impression's rows: 16-byte loads, eight per thread in flight, issued unconditionally with a clamped index
  // BEFORE the first LDS store of a batch (a load under a guard, or behind a store, waits for its own round trip)
  {
    const float4* Ug = reinterpret_cast<const float4*>(a.U + bi * L * A);
    const float4* Xg = reinterpret_cast<const float4*>(a.X + bi * L * E);
    const float4* Cg = reinterpret_cast<const float4*>(a.cand + bi * C * E);
    const float4* bg = reinterpret_cast<const float4*>(a.b);
    constexpr int NB = 2;  // float4 loads per thread in flight per batch (1024 threads: 32 KB per batch)
    for (int base = 0; base < L * E4; base += HEAD_THREADS * NB) {
      float4 v[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = base + j * HEAD_THREADS + tid;
        v[j] = Xg[i < L * E4 ? i : L * E4 - 1];
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = base + j * HEAD_THREADS + tid;
        if (i < L * E4) reinterpret_cast<float4*>(Xs)[i] = v[j];
      }
    }
    for (int base = 0; base < C * E4; base += HEAD_THREADS * NB) {
      float4 v[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = base + j * HEAD_THREADS + tid;
        v[j] = Cg[i < C * E4 ? i : C * E4 - 1];
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = base + j * HEAD_THREADS + tid;
        if (i < C * E4) reinterpret_cast<float4*>(Cs)[i] = v[j];
      }
    }
    for (int base = 0; base < L * A4; base += HEAD_THREADS * NB) {
      float4 v[NB], bb[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = base + j * HEAD_THREADS + tid;
        const int ic = i < L * A4 ? i : L * A4 - 1;
        v[j] = Ug[ic];
        bb[j] = bg[ic % A4];
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = base + j * HEAD_THREADS + tid;
        float4 t;
        t.x = tanhf(v[j].x + bb[j].x);
        t.y = tanhf(v[j].y + bb[j].y);
        t.z = tanhf(v[j].z + bb[j].z);
        t.w = tanhf(v[j].w + bb[j].w);
        if (i < L * A4) reinterpret_cast<float4*>(T)[i] = t;
      }
    }
  }
  __syncthreads();
  // ---- e[l] = tanh(.)[l,:] . q      (layers.py:65-68): one wave per row
  for (int l = wave; l < L; l += HEAD_THREADS / 64) {
    float part = 0.f;
    for (int k4 = lane; k4 < A4; k4 += 64) {
      const float4 t = reinterpret_cast<const float4*>(T + l * A)[k4];
      const float4 qq = reinterpret_cast<const float4*>(a.q)[k4];
      part = fmaf(t.w, qq.w, fmaf(t.z, qq.z, fmaf(t.y, qq.y, fmaf(t.x, qq.x, part))));
    }
    part = ebn_wave_sum(part);
    if (lane == 0) ev[l] = part;
  }
  __syncthreads();
  // ---- w = exp(e) / (sum exp(e) + 1e-7)      (layers.py:71-77, no max-subtraction)
  if (wave == 0) {
    float s = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float ex = expf(ev[l]);
      ev[l] = ex;
      s += ex;
    }
    s = ebn_wave_sum(s) + HEAD_KERAS_EPS;
    for (int l = lane; l < L; l += 64) {
      const float wl = ev[l] / s;
      ev[l] = wl;
      a.w[bi * L + l] = wl;
    }
  }
  __syncthreads();
  // ---- user = sum_l w_l x_l      (layers.py:79-81)
  for (int c4 = tid; c4 < E4; c4 += HEAD_THREADS) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < L; ++l) {
      const float wl = ev[l];
      const float4 x = reinterpret_cast<const float4*>(Xs + l * E)[c4];
      acc.x = fmaf(wl, x.x, acc.x);
      acc.y = fmaf(wl, x.y, acc.y);
      acc.z = fmaf(wl, x.z, acc.z);
      acc.w = fmaf(wl, x.w, acc.w);
    }
    reinterpret_cast<float4*>(us)[c4] = acc;
    reinterpret_cast<float4*>(a.user + bi * E)[c4] = acc;
  }
  __syncthreads();
  // ---- scores[c] = cand[c] . user      (nrms.py:201)
  for (int c = wave; c < C; c += HEAD_THREADS / 64) {
    float part = 0.f;
    for (int e4 = lane; e4 < E4; e4 += 64) {
      const float4 x = reinterpret_cast<const float4*>(Cs + c * E)[e4];
      const float4 u = reinterpret_cast<const float4*>(us)[e4];
      part = fmaf(x.w, u.w, fmaf(x.z, u.z, fmaf(x.y, u.y, fmaf(x.x, u.x, part))));
    }
    part = ebn_wave_sum(part);
    if (lane == 0) sc[c] = part;
  }
  __syncthreads();
  // ---- softmax, compiled loss, d(scores)      (nrms.py:202, 56-67)
  if (wave == 0) {
    const float* y = a.labels + bi * C;
    float mx = -INFINITY, ysum = 0.f;
    for (int c = lane; c < C; c += 64) {
      mx = fmaxf(mx, sc[c]);
      ysum += y[c];
    }
    mx = ebn_wave_max(mx);
    ysum = ebn_wave_sum(ysum);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(sc[c] - mx);
    se = ebn_wave_sum(se);
    const float lse = mx + logf(se);
    const float invbc = a.inv_batch / static_cast<float>(C);
    float loss = 0.f, dot = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float s = sc[c];
      a.scores[bi * C + c] = s;
      const float p = expf(s - mx) / se;
      a.probs[bi * C + c] = p;
      float ds;
      if (a.loss_kind == 0) {
        const float logp = s - lse;
        loss -= y[c] * logp;
        ds = (expf(logp) * ysum - y[c]) * a.inv_batch;
      } else if (a.loss_kind == 1) {
        loss += fmaxf(s, 0.f) - s * y[c] + log1pf(expf(-fabsf(s)));
        ds = (1.0f / (1.0f + expf(-s)) - y[c]) * invbc;
      } else {
        loss += head_bce_probs(p, y[c], &ds);
        dot = fmaf(p, ds, dot);
      }
      dsv[c] = ds;
    }
    if (a.loss_kind == 2) {
      dot = ebn_wave_sum(dot);
      for (int c = lane; c < C; c += 64) dsv[c] = (expf(sc[c] - mx) / se) * (dsv[c] - dot) * invbc;
    }
    loss = ebn_wave_sum(loss) * (a.loss_kind == 0 ? a.inv_batch : invbc);
    if (lane == 0) a.loss_rows[bi] = loss;
  }
  __syncthreads();
  // ---- d(cand)[c,:] = ds[c] * user ; d(user) = sum_c ds[c] * cand[c,:]
  for (int e4 = tid; e4 < E4; e4 += HEAD_THREADS) {
    const float4 u = reinterpret_cast<const float4*>(us)[e4];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < C; ++c) {
      const float ds = dsv[c];
      const float4 x = reinterpret_cast<const float4*>(Cs + c * E)[e4];
      acc.x = fmaf(ds, x.x, acc.x);
      acc.y = fmaf(ds, x.y, acc.y);
      acc.z = fmaf(ds, x.z, acc.z);
      acc.w = fmaf(ds, x.w, acc.w);
      reinterpret_cast<float4*>(a.dcand + (bi * C + c) * E)[e4] = make_float4(ds * u.x, ds * u.y, ds * u.z, ds * u.w);
    }
    reinterpret_cast<float4*>(du)[e4] = acc;
    reinterpret_cast<float4*>(a.duser + bi * E)[e4] = acc;
  }
  __syncthreads();
  // ---- AttLayer2 backward: dw[l] = d(user) . x_l ; de = w * (dw - sum w dw)
  for (int l = wave; l < L; l += HEAD_THREADS / 64) {
    float part = 0.f;
    for (int e4 = lane; e4 < E4; e4 += 64) {
      const float4 x = reinterpret_cast<const float4*>(Xs + l * E)[e4];
      const float4 g = reinterpret_cast<const float4*>(du)[e4];
      part = fmaf(g.w, x.w, fmaf(g.z, x.z, fmaf(g.y, x.y, fmaf(g.x, x.x, part))));
    }
    part = ebn_wave_sum(part);
    if (lane == 0) dwv[l] = part;
  }
  __syncthreads();
  if (wave == 0) {
    float s = 0.f;
    for (int l = lane; l < L; l += 64) s = fmaf(ev[l], dwv[l], s);
    s = ebn_wave_sum(s);
    for (int l = lane; l < L; l += 64) {
      const float d = ev[l] * (dwv[l] - s);
      dwv[l] = d;
      a.de[bi * L + l] = d;
    }
  }
  __syncthreads();
  // ---- d(pre-tanh)[l,k] = de[l] q[k] (1 - t^2) -> U ; this impression's d(q)[k] = sum_l de[l] t[l,k], d(b)[k] = sum_l d(pre)[l,k]
  for (int k = tid; k < A; k += HEAD_THREADS) {
    const float qk = a.q[k];
    float dq = 0.f, db = 0.f;
    for (int l = 0; l < L; ++l) {
      const float t = T[l * A + k], d = dwv[l];
      dq = fmaf(d, t, dq);
      const float dp = d * qk * (1.0f - t * t);
      a.U[(bi * L + l) * A + k] = dp;
      db += dp;
    }
    a.partials[(bi * 2 + 0) * A + k] = dq;
    a.partials[(bi * 2 + 1) * A + k] = db;
  }
}

__global__ __launch_bounds__(256) void user_head_finish_kernel(const float* __restrict__ partials, int64_t B, int A,
                                                               float* __restrict__ dq, float* __restrict__ db,
                                                               const float* __restrict__ loss_rows,
                                                               float* __restrict__ loss_out) {
  __shared__ float sw[4];
  ebn_user_head_finish_body(sw, blockIdx.x, gridDim.x, partials, B, A, dq, db, loss_rows, loss_out);
}

size_t head_lds_bytes(int L, int C, int E, int A) {
  return sizeof(float) * (static_cast<size_t>(L) * A + static_cast<size_t>(L) * E + static_cast<size_t>(C) * E + 2 * static_cast<size_t>(E) +
                          2 * static_cast<size_t>(L) + 2 * static_cast<size_t>(C));
}

}  // namespace

extern "C" int ebn_user_head_supported(int32_t L, int32_t C, int32_t E, int32_t A) {
  if (L <= 0 || C <= 0 || E <= 0 || A <= 0 || (E % 4) != 0 || (A % 4) != 0) return 0;
  return head_lds_bytes(L, C, E, A) <= 150 * 1024 ? 1 : 0;  // one workgroup's LDS (160 KB per CU)
}

extern "C" int64_t ebn_user_head_partials_len(int64_t B, int32_t A) { return ebn_dim_ok(B, A) ? B * 2 * static_cast<int64_t>(A) : 0; }

extern "C" int ebn_user_head_train_f32(float* U, const float* b, const float* q, const float* X, const float* cand,
                                       const float* labels, float* w, float* user, float* scores, float* probs,
                                       float* loss_rows, float* loss_out, float* dcand, float* duser, float* de, float* dq,
                                       float* db, float* partials, int64_t B, int32_t L, int32_t C, int32_t E, int32_t A,
                                       int32_t loss_kind, float inv_batch, ebn_stream_t stream) {
  EBN_REQUIRE(U && b && q && X && cand && labels && w && user && scores && probs && loss_rows && loss_out && dcand && duser &&
                  de && partials && ((dq != nullptr) == (db != nullptr)),
              EBN_ERR_BAD_ARG);
  EBN_REQUIRE(B >= 0 && L > 0 && C > 0 && E > 0 && A > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(loss_kind >= 0 && loss_kind <= 2 && ebn_user_head_supported(L, C, E, A), EBN_ERR_UNSUPPORTED);
  EBN_REQUIRE(ebn_aligned16(U) && ebn_aligned16(b) && ebn_aligned16(q) && ebn_aligned16(X) && ebn_aligned16(cand) &&
                  ebn_aligned16(user) && ebn_aligned16(dcand) && ebn_aligned16(duser),
              EBN_ERR_ALIGN);
  if (B == 0) return EBN_OK;
  hipStream_t s = ebn_stream(stream);
  const size_t lds = head_lds_bytes(L, C, E, A);
  // the dynamic-LDS limit is raised once, to the most ebn_user_head_supported admits (above the 64 KB default: long histories)
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&user_head_train_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  if (attr != hipSuccess) return static_cast<int>(attr);
  HeadArgs a{U, b, q, X, cand, labels, w, user, scores, probs, loss_rows, dcand, duser, de, partials, L, C, E, A, loss_kind, inv_batch};
  EBN_LAUNCH(user_head_train_kernel, dim3(static_cast<unsigned>(B)), dim3(HEAD_THREADS), lds, s, a);
  EBN_CHECK_LAUNCH();
  if (dq == nullptr) return EBN_OK;  // d(q), d(b) and the batch loss are left to ebn_grad_finish_f32 (EBN_FINISH_HEAD job)
  EBN_LAUNCH(user_head_finish_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(2 * A, 256) + 1)), dim3(256), 0, s, partials, B, A,
                     dq, db, loss_rows, loss_out);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
