// a8/a9: click scorer (Dot(axes=-1) + softmax / sigmoid, nrms.py:201-205), the compiled loss
// (nrms.py:56-67) with its backward into the news/user representations, the ragged pair scorer
// of the evaluation path, and a10: the Keras-form Adam step (nrms.py:69-80) with the per-step
// device state.  All latency- or HBM-bound elementwise/row work.
#include <atomic>

#include "ebn_common.h"

namespace {

// loss_kind 2: binary cross-entropy on the CLIPPED SOFTMAX OUTPUTS (SURVEY.md A.5's reading of nrms.py:54,61-62 -- what
// Keras' binary_crossentropy(from_logits=False) does when the tensor carries no cached logits) [KERAS-SEMANTICS]:
//   p^ = clip(p, eps, 1 - eps);  l = -(y log(p^ + eps) + (1 - y) log(1 - p^ + eps)),  eps = K.epsilon() = 1e-7
// Returns l and d(l)/d(p) (zero where the clip is active: clip_by_value passes no gradient outside its range).
constexpr float EBN_KERAS_EPS = 1e-7f;
__device__ __forceinline__ float bce_probs(float p, float y, float* dldp) {
  const float lo = EBN_KERAS_EPS, hi = 1.0f - EBN_KERAS_EPS;
  const bool in_range = p >= lo && p <= hi;
  const float pc = fminf(fmaxf(p, lo), hi);
  const float a = pc + EBN_KERAS_EPS, b = 1.0f - pc + EBN_KERAS_EPS;
  *dldp = in_range ? (-(y / a) + (1.0f - y) / b) : 0.f;
  return -(y * logf(a) + (1.0f - y) * logf(b));
}

// one 4-wave workgroup per impression b: the waves take candidates round-robin, wave 0 finishes the activation
__global__ __launch_bounds__(256) void score_fwd_kernel(const float* __restrict__ cand,
                                                       const float* __restrict__ user,
                                                       float* __restrict__ scores, float* __restrict__ probs,
                                                       int C, int E, int mode) {
  extern __shared__ float sm[];  // C scores
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* u = user + b * E;
  for (int c = wave; c < C; c += 4) {
    const float* x = cand + (b * C + c) * E;
    float part = 0.f;
    for (int e = lane; e < E; e += 64) part = fmaf(x[e], u[e], part);
    part = ebn_wave_sum(part);
    if (lane == 0) sm[c] = part;
  }
  __syncthreads();
  if (wave != 0) return;
  if (mode == 1) {  // sigmoid (scorer model, nrms.py:205)
    for (int c = lane; c < C; c += 64) {
      const float s = sm[c];
      if (scores) scores[b * C + c] = s;
      probs[b * C + c] = 1.0f / (1.0f + expf(-s));
    }
    return;
  }
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, sm[c]);
  mx = ebn_wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) sum += expf(sm[c] - mx);
  sum = ebn_wave_sum(sum);
  for (int c = lane; c < C; c += 64) {
    const float s = sm[c];
    if (scores) scores[b * C + c] = s;
    probs[b * C + c] = expf(s - mx) / sum;
  }
}

__global__ __launch_bounds__(64) void score_loss_bwd_kernel(const float* __restrict__ cand,
                                                            const float* __restrict__ user,
                                                            const float* __restrict__ scores,
                                                            const float* __restrict__ labels,
                                                            float* __restrict__ loss_rows,
                                                            float* __restrict__ dcand, float* __restrict__ duser,
                                                            int C, int E, int loss_kind, float inv_batch) {
  extern __shared__ float sm[];  // ds[C]
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const float* s = scores + b * C;
  const float* y = labels + b * C;
  float loss = 0.f;
  if (loss_kind == 0) {
    // categorical CE on the logits: -sum_c y log_softmax(s); ds = (softmax*sum(y) - y)/B
    float mx = -INFINITY, ysum = 0.f;
    for (int c = lane; c < C; c += 64) {
      mx = fmaxf(mx, s[c]);
      ysum += y[c];
    }
    mx = ebn_wave_max(mx);
    ysum = ebn_wave_sum(ysum);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(s[c] - mx);
    se = ebn_wave_sum(se);
    const float lse = mx + logf(se);
    for (int c = lane; c < C; c += 64) {
      const float logp = s[c] - lse;
      loss -= y[c] * logp;
      sm[c] = (expf(logp) * ysum - y[c]) * inv_batch;
    }
    loss = ebn_wave_sum(loss) * inv_batch;
  } else if (loss_kind == 1) {
    // sigmoid CE on the logits, mean over (b,c): max(s,0) - s*y + log1p(exp(-|s|))
    const float invbc = inv_batch / static_cast<float>(C);
    for (int c = lane; c < C; c += 64) {
      const float x = s[c];
      loss += fmaxf(x, 0.f) - x * y[c] + log1pf(expf(-fabsf(x)));
      sm[c] = (1.0f / (1.0f + expf(-x)) - y[c]) * invbc;
    }
    loss = ebn_wave_sum(loss) * invbc;
  } else {
    // BCE on the clipped softmax outputs, mean over (b,c); ds = p * (dl/dp - sum_k p_k dl/dp_k)  (softmax backward)
    const float invbc = inv_batch / static_cast<float>(C);
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, s[c]);
    mx = ebn_wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(s[c] - mx);
    se = ebn_wave_sum(se);
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float p = expf(s[c] - mx) / se;
      float dldp;
      loss += bce_probs(p, y[c], &dldp);
      sm[c] = dldp;
      dot = fmaf(p, dldp, dot);
    }
    dot = ebn_wave_sum(dot);
    for (int c = lane; c < C; c += 64) sm[c] = (expf(s[c] - mx) / se) * (sm[c] - dot) * invbc;
    loss = ebn_wave_sum(loss) * invbc;
  }
  if (lane == 0) loss_rows[b] = loss;
  __syncthreads();
  const float* u = user + b * E;
  for (int e = lane; e < E; e += 64) {
    const float ue = u[e];
    float du = 0.f;
    for (int c = 0; c < C; ++c) {
      const float ds = sm[c];
      const int64_t off = (b * C + c) * E + e;
      du = fmaf(ds, cand[off], du);
      dcand[off] = ds * ue;
    }
    duser[b * E + e] = du;
  }
}

// Training step: scorer forward (softmax model, nrms.py:201-202), compiled loss and its backward into the representations
// in ONE launch, one 4-wave workgroup per impression -- score_fwd_kernel (mode 0) + score_loss_bwd_kernel, operation for
// operation (each sits at the ~5 us launch floor of a chain of dependent launches when run separately).
__global__ __launch_bounds__(256) void score_loss_train_kernel(
    const float* __restrict__ cand, const float* __restrict__ user, const float* __restrict__ labels,
    float* __restrict__ scores, float* __restrict__ probs, float* __restrict__ loss_rows, float* __restrict__ dcand,
    float* __restrict__ duser, int C, int E, int loss_kind, float inv_batch) {
  extern __shared__ float sm[];  // C scores, then C ds
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* u = user + b * E;
  for (int c = wave; c < C; c += 4) {
    const float* x = cand + (b * C + c) * E;
    float part = 0.f;
    for (int e = lane; e < E; e += 64) part = fmaf(x[e], u[e], part);
    part = ebn_wave_sum(part);
    if (lane == 0) sm[c] = part;
  }
  __syncthreads();
  if (wave == 0) {
    const float* y = labels + b * C;
    float mx = -INFINITY, ysum = 0.f;
    for (int c = lane; c < C; c += 64) {
      mx = fmaxf(mx, sm[c]);
      ysum += y[c];
    }
    mx = ebn_wave_max(mx);
    ysum = ebn_wave_sum(ysum);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(sm[c] - mx);
    se = ebn_wave_sum(se);
    const float lse = mx + logf(se);
    float loss = 0.f, dot = 0.f;
    const float invbc = inv_batch / static_cast<float>(C);
    for (int c = lane; c < C; c += 64) {
      const float sc = sm[c];
      scores[b * C + c] = sc;
      const float p = expf(sc - mx) / se;
      probs[b * C + c] = p;
      float ds;
      if (loss_kind == 0) {
        const float logp = sc - lse;
        loss -= y[c] * logp;
        ds = (expf(logp) * ysum - y[c]) * inv_batch;
      } else if (loss_kind == 1) {
        loss += fmaxf(sc, 0.f) - sc * y[c] + log1pf(expf(-fabsf(sc)));
        ds = (1.0f / (1.0f + expf(-sc)) - y[c]) * invbc;
      } else {  // BCE on the clipped softmax outputs: ds needs sum_k p_k dl/dp_k first
        loss += bce_probs(p, y[c], &ds);
        dot = fmaf(p, ds, dot);
      }
      sm[C + c] = ds;
    }
    if (loss_kind == 2) {
      dot = ebn_wave_sum(dot);
      for (int c = lane; c < C; c += 64) sm[C + c] = (expf(sm[c] - mx) / se) * (sm[C + c] - dot) * invbc;
    }
    loss = ebn_wave_sum(loss) * (loss_kind == 0 ? inv_batch : invbc);
    if (lane == 0) loss_rows[b] = loss;
  }
  __syncthreads();
  const float* ds = sm + C;
  for (int e = threadIdx.x; e < E; e += 256) {
    const float ue = u[e];
    float du = 0.f;
    for (int c = 0; c < C; ++c) {
      const int64_t off = (b * C + c) * E + e;
      du = fmaf(ds[c], cand[off], du);
      dcand[off] = ds[c] * ue;
    }
    duser[b * E + e] = du;
  }
}

// Streaming AUC of compile(metrics=["AUC"]) (ebnerd_nrms.py:244-248; tf.keras.metrics.AUC: 200 thresholds): bucket every
// (label, prediction) pair of a batch into the positive / negative histograms -- bucket = number of thresholds strictly
// below the prediction (lower bound in the ascending float64 threshold list), integer atomics (order-independent).
__global__ __launch_bounds__(256) void auc_hist_kernel(const float* __restrict__ probs, const float* __restrict__ labels,
                                                       int64_t n, const double* __restrict__ thr, int n_thr,
                                                       unsigned long long* __restrict__ pos_hist,
                                                       unsigned long long* __restrict__ neg_hist) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) {
    const double p = static_cast<double>(probs[i]);
    int lo = 0, hi = n_thr;  // first index with thr[idx] >= p
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (thr[mid] < p) lo = mid + 1;
      else hi = mid;
    }
    atomicAdd((labels[i] > 0.f) ? &pos_hist[lo] : &neg_hist[lo], 1ull);
  }
}

// 4 pairs per 256-thread block, one wave per pair
__global__ __launch_bounds__(256) void pair_score_kernel(const float* __restrict__ user,
                                                         const float* __restrict__ news,
                                                         const int32_t* __restrict__ u_idx,
                                                         const int32_t* __restrict__ n_idx,
                                                         float* __restrict__ out, int64_t n_pairs, int E,
                                                         int mode) {
  const int lane = threadIdx.x & 63;
  const int64_t p = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (p >= n_pairs) return;
  const float* u = user + static_cast<int64_t>(u_idx[p]) * E;
  const float* x = news + static_cast<int64_t>(n_idx[p]) * E;
  float part = 0.f;
  for (int e = lane; e < E; e += 64) part = fmaf(u[e], x[e], part);
  part = ebn_wave_sum(part);
  if (lane == 0) out[p] = (mode == 1) ? 1.0f / (1.0f + expf(-part)) : part;
}

__global__ void step_advance_kernel(ebn_step_state* st, double beta1, double beta2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint32_t t = st->step + 1u;
  st->step = t;
  const double b1t = pow(beta1, static_cast<double>(t));
  const double b2t = pow(beta2, static_cast<double>(t));
  st->adam_alpha = static_cast<float>(static_cast<double>(st->lr) * sqrt(1.0 - b2t) / (1.0 - b1t));
  for (uint32_t s = 0; s < EBN_N_SITES; ++s) st->drop_key[s] = ebn_dropout_key(st->seed, t, s);
}

__global__ __launch_bounds__(256) void adam_keras_kernel(float* __restrict__ theta, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                         const ebn_step_state* __restrict__ st, float omb1,
                                                         float omb2, float eps, float gscale) {
  const float alpha = st->adam_alpha;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  const int64_t n4 = n / 4;
  float4* t4 = reinterpret_cast<float4*>(theta);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
#define EBN_ADAM1(T, G, M, V)                 \
  {                                           \
    const float gg = (G) * gscale;            \
    (M) = (M) + (gg - (M)) * omb1;            \
    (V) = (V) + (gg * gg - (V)) * omb2;       \
    (T) = (T) - alpha * (M) / (sqrtf(V) + eps); \
  }
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += stride) {
    float4 t = t4[i], gr = g4[i], mm = m4[i], vv = v4[i];
    EBN_ADAM1(t.x, gr.x, mm.x, vv.x)
    EBN_ADAM1(t.y, gr.y, mm.y, vv.y)
    EBN_ADAM1(t.z, gr.z, mm.z, vv.z)
    EBN_ADAM1(t.w, gr.w, mm.w, vv.w)
    t4[i] = t;
    m4[i] = mm;
    v4[i] = vv;
  }
  for (int64_t i = n4 * 4 + static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
    float t = theta[i], mm = m[i], vv = v[i];
    EBN_ADAM1(t, g[i], mm, vv)
    theta[i] = t;
    m[i] = mm;
    v[i] = vv;
  }
#undef EBN_ADAM1
}

__global__ __launch_bounds__(256) void adam_keras_scalar_kernel(float* __restrict__ theta,
                                                                const float* __restrict__ g,
                                                                float* __restrict__ m, float* __restrict__ v,
                                                                int64_t n, const ebn_step_state* __restrict__ st,
                                                                float omb1, float omb2, float eps, float gscale) {
  const float alpha = st->adam_alpha;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const float gg = g[i] * gscale;
    const float mm = m[i] + (gg - m[i]) * omb1;
    const float vv = v[i] + (gg * gg - v[i]) * omb2;
    m[i] = mm;
    v[i] = vv;
    theta[i] = theta[i] - alpha * mm / (sqrtf(vv) + eps);
  }
}

}  // namespace

extern "C" int ebn_abi_version(void) { return EBN_ABI_VERSION; }

static std::atomic<int64_t> g_launches{0};
void ebnx_note_launch(void) { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" int64_t ebn_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

extern "C" const char* ebn_error_string(int code) {
  switch (code) {
    case EBN_OK: return "ok";
    case EBN_ERR_BAD_ARG: return "ebnerd_hip: bad argument (null pointer, negative size or too-small leading dimension)";
    case EBN_ERR_UNSUPPORTED: return "ebnerd_hip: shape not supported by the gfx950 kernels";
    case EBN_ERR_ALIGN: return "ebnerd_hip: pointer / leading dimension must be 16-byte aligned";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "ebnerd_hip: unknown error";
}

extern "C" int ebn_step_advance(ebn_step_state* st, double beta1, double beta2, ebn_stream_t stream) {
  EBN_REQUIRE(st, EBN_ERR_BAD_ARG);
  EBN_LAUNCH(step_advance_kernel, dim3(1), dim3(64), 0, ebn_stream(stream), st, beta1, beta2);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_score_fwd_f32(const float* cand, const float* user, float* scores, float* probs, int64_t B,
                                 int32_t C, int32_t E, int32_t mode, ebn_stream_t stream) {
  EBN_REQUIRE(cand && user && probs, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(B >= 0 && C > 0 && E > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(C <= 8192, EBN_ERR_UNSUPPORTED);
  if (B == 0) return EBN_OK;
  EBN_LAUNCH(score_fwd_kernel, dim3(static_cast<unsigned>(B)), dim3(256), C * sizeof(float),
                     ebn_stream(stream), cand, user, scores, probs, C, E, mode);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_score_loss_bwd_f32(const float* cand, const float* user, const float* scores,
                                      const float* labels, float* loss_rows, float* dcand, float* duser,
                                      int64_t B, int32_t C, int32_t E, int32_t loss_kind, float inv_batch,
                                      ebn_stream_t stream) {
  EBN_REQUIRE(cand && user && scores && labels && loss_rows && dcand && duser, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(B >= 0 && C > 0 && E > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(C <= 8192 && loss_kind >= 0 && loss_kind <= 2, EBN_ERR_UNSUPPORTED);
  if (B == 0) return EBN_OK;
  EBN_LAUNCH(score_loss_bwd_kernel, dim3(static_cast<unsigned>(B)), dim3(64), C * sizeof(float),
                     ebn_stream(stream), cand, user, scores, labels, loss_rows, dcand, duser, C, E, loss_kind,
                     inv_batch);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_score_loss_train_f32(const float* cand, const float* user, const float* labels, float* scores,
                                        float* probs, float* loss_rows, float* loss_out, float* dcand, float* duser,
                                        int64_t B, int32_t C, int32_t E, int32_t loss_kind, float inv_batch,
                                        ebn_stream_t stream) {
  EBN_REQUIRE(cand && user && labels && scores && probs && loss_rows && loss_out && dcand && duser, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(B >= 0 && C > 0 && E > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(C <= 8192 && loss_kind >= 0 && loss_kind <= 2, EBN_ERR_UNSUPPORTED);
  if (B == 0) return EBN_OK;
  EBN_LAUNCH(score_loss_train_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 2 * C * sizeof(float),
                     ebn_stream(stream), cand, user, labels, scores, probs, loss_rows, dcand, duser, C, E, loss_kind,
                     inv_batch);
  EBN_CHECK_LAUNCH();
  return ebn_sum_f32(loss_rows, B, 1.0f, loss_out, 0, stream);  // fixed-order reduction: deterministic batch loss
}

extern "C" int ebn_auc_hist_f32(const float* probs, const float* labels, int64_t n, const double* thresholds,
                                int32_t n_thresholds, int64_t* pos_hist, int64_t* neg_hist, ebn_stream_t stream) {
  EBN_REQUIRE(probs && labels && thresholds && pos_hist && neg_hist, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n >= 0 && n_thresholds > 0, EBN_ERR_BAD_ARG);
  if (n == 0) return EBN_OK;
  int64_t grid = ebn_ceil_div(n, 256);
  if (grid > 1024) grid = 1024;
  EBN_LAUNCH(auc_hist_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, ebn_stream(stream), probs, labels, n,
                     thresholds, n_thresholds, reinterpret_cast<unsigned long long*>(pos_hist),
                     reinterpret_cast<unsigned long long*>(neg_hist));
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_pair_score_f32(const float* user, const float* news, const int32_t* u_idx,
                                  const int32_t* n_idx, float* out, int64_t n_pairs, int32_t E, int32_t mode,
                                  ebn_stream_t stream) {
  EBN_REQUIRE(user && news && u_idx && n_idx && out, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n_pairs >= 0 && E > 0, EBN_ERR_BAD_ARG);
  if (n_pairs == 0) return EBN_OK;
  EBN_LAUNCH(pair_score_kernel, dim3(static_cast<unsigned>(ebn_ceil_div(n_pairs, 4))), dim3(256), 0,
                     ebn_stream(stream), user, news, u_idx, n_idx, out, n_pairs, E, mode);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_adam_keras_step_f32(float* theta, const float* g, float* m, float* v, int64_t n,
                                       const ebn_step_state* st, double beta1, double beta2, double eps_d,
                                       float grad_scale, ebn_stream_t stream) {
  EBN_REQUIRE(theta && g && m && v && st, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(n >= 0, EBN_ERR_BAD_ARG);
  if (n == 0) return EBN_OK;
  const float omb1 = static_cast<float>(1.0 - beta1), omb2 = static_cast<float>(1.0 - beta2);
  const float eps = static_cast<float>(eps_d);
  const bool vec = ebn_aligned16(theta) && ebn_aligned16(g) && ebn_aligned16(m) && ebn_aligned16(v);
  int64_t grid = ebn_ceil_div(vec ? ebn_ceil_div(n, 4) : n, 256);
  if (grid > 256 * 16) grid = 256 * 16;
  if (grid < 1) grid = 1;
  if (vec)
    EBN_LAUNCH(adam_keras_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, ebn_stream(stream),
                       theta, g, m, v, n, st, omb1, omb2, eps, grad_scale);
  else
    EBN_LAUNCH(adam_keras_scalar_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0,
                       ebn_stream(stream), theta, g, m, v, n, st, omb1, omb2, eps, grad_scale);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
