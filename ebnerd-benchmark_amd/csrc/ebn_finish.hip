// ONE launch for the small finishing passes of a training step's backward (the gradients of
// layers.py:35-52's W, b, q and of the projection weights, nrms.py:56-67's batch loss).  Each of them is the second stage of a
// deterministic reduction -- the sum of the split-K slices of a weight-gradient GEMM, the sum over row blocks of the AttLayer2
// d(q) / d(b) column partials, the sum over impressions of the per-impression head partials and loss rows -- a few microseconds
// of work that cost a launch of the step's dependent chain each (4 of the 24 launches of a c2 step, 25 us of kernel time of
// which most is launch ramp and tail).  Their producers can leave them out (ebn_gemm_f32_partials; NULL dq / db in
// ebn_attpool_bwd_dpre_f32 and ebn_user_head_train_f32) and this pass runs them side by side: block ranges per job, the device
// bodies of the stand-alone kernels, hence the same bits.
#include "ebn_common.h"
#include "ebn_finish.h"
#include "ebn_reduce.h"

namespace {

struct FinishJobs {
  ebn_finish_job job[EBN_FINISH_MAX_JOBS];
  int32_t first_block[EBN_FINISH_MAX_JOBS + 1];  // job j owns blocks [first_block[j], first_block[j + 1])
  int32_t n;
};

// ebn_grad_finish_adam_f32: the same jobs with Adam applied to every gradient element they finish, and -- blocks behind the jobs' --
// Adam over the ranges of the flat buffers whose gradients earlier launches wrote (4096 elements per block)
struct FinishAdam {
  EbnAdamFlat adam;
  int32_t n_rest;
  int64_t rest_off[EBN_ADAM_FLAT_MAX_REST];
  int64_t rest_len[EBN_ADAM_FLAT_MAX_REST];
  int64_t rest_total;
};
constexpr int REST_PER_BLOCK = 4096;

template <bool ADAM>
__device__ __forceinline__ void grad_finish_body(const FinishJobs& js, const FinishAdam* fa) {
  __shared__ float sm[32][33];
  const EbnAdamFlat* ad = ADAM ? &fa->adam : nullptr;
  if (ADAM && static_cast<int>(blockIdx.x) >= js.first_block[js.n]) {  // element-wise Adam over the remaining ranges
    const float al = ad->st->adam_alpha;
    const int64_t e0 = static_cast<int64_t>(static_cast<int>(blockIdx.x) - js.first_block[js.n]) * REST_PER_BLOCK;
    for (int64_t e = e0 + threadIdx.x; e < e0 + REST_PER_BLOCK && e < fa->rest_total; e += 1024) {
      int64_t off = e;
      int r = 0;
      while (r + 1 < fa->n_rest && off >= fa->rest_len[r]) off -= fa->rest_len[r++];
      off += fa->rest_off[r];
      ebn_adam_flat_apply(*ad, al, off, ad->grad[off]);
    }
    return;
  }
  int j = 0;
  while (j + 1 < js.n && static_cast<int>(blockIdx.x) >= js.first_block[j + 1]) ++j;  // block-uniform
  const ebn_finish_job& q = js.job[j];
  const int blk = static_cast<int>(blockIdx.x) - js.first_block[j], nblk = js.first_block[j + 1] - js.first_block[j];
  if (q.kind == EBN_FINISH_SPLITK) {
    ebn_splitk_sum_body(static_cast<uint32_t>(blk) * 1024u + threadIdx.x, static_cast<uint32_t>(nblk) * 1024u, q.partials, q.n_parts,
                        static_cast<uint32_t>(q.rows * q.cols), static_cast<uint32_t>(q.cols), q.beta, q.out0, q.ld, nullptr, nullptr, 0, 1, nullptr, ad);
  } else if (q.kind == EBN_FINISH_COLRED) {
    ebn_reduce_partials_body(sm, blk, q.partials, q.n_parts, 2, static_cast<int>(q.cols), q.scale, q.out0, q.out1, q.beta != 0.f ? 1 : 0, nullptr,
                             nullptr, ad);
  } else {
    ebn_user_head_finish_body(&sm[0][0], blk, nblk, q.partials, q.rows, static_cast<int>(q.cols), q.out0, q.out1, q.loss_rows, q.loss_out);
    if (ADAM && blk < nblk - 1 && threadIdx.x < 256) {  // d(q) / d(b): written by this thread a moment ago
      const int idx = blk * 256 + static_cast<int>(threadIdx.x), A = static_cast<int>(q.cols);
      if (idx < 2 * A) {
        float* gp = (idx / A == 0 ? q.out0 : q.out1) + (idx % A);
        ebn_adam_flat_apply(*ad, ad->st->adam_alpha, gp - ad->grad, *gp);
      }
    }
  }
}

__global__ __launch_bounds__(1024) void grad_finish_kernel(FinishJobs js) { grad_finish_body<false>(js, nullptr); }
__global__ __launch_bounds__(1024) void grad_finish_adam_kernel(FinishJobs js, FinishAdam fa) { grad_finish_body<true>(js, &fa); }

}  // namespace

static int plan_jobs(const ebn_finish_job* jobs, int32_t n_jobs, FinishJobs& js) {
  EBN_REQUIRE(n_jobs >= 0 && n_jobs <= EBN_FINISH_MAX_JOBS && (jobs != nullptr || n_jobs == 0), EBN_ERR_BAD_ARG);
  js.n = 0;
  js.first_block[0] = 0;
  for (int32_t i = 0; i < n_jobs; ++i) {
    const ebn_finish_job& q = jobs[i];
    EBN_REQUIRE(q.kind >= EBN_FINISH_SPLITK && q.kind <= EBN_FINISH_HEAD && q.rows >= 0 && q.cols >= 0 && q.n_parts >= 0, EBN_ERR_BAD_ARG);
    if (q.rows == 0 || q.cols == 0) continue;
    EBN_REQUIRE(q.partials && q.out0, EBN_ERR_BAD_ARG);
    int64_t blocks;
    if (q.kind == EBN_FINISH_SPLITK) {
      EBN_REQUIRE(q.n_parts >= 1 && q.ld >= q.cols, EBN_ERR_BAD_ARG);
      EBN_REQUIRE(q.rows * q.cols < (int64_t{1} << 31), EBN_ERR_UNSUPPORTED);
      blocks = ebn_ceil_div(q.rows * q.cols, 1024);  // one element per thread (a 1024 x 1200 projection gradient: 1200 blocks); a cap
      if (blocks > 16384) blocks = 16384;            // that made some threads walk two elements doubled the pass (15.7 us for 7.7)
    } else if (q.kind == EBN_FINISH_COLRED) {
      EBN_REQUIRE(q.out1 && q.cols < (1 << 24), EBN_ERR_BAD_ARG);
      blocks = ebn_ceil_div(2 * q.cols, 32);
    } else {
      EBN_REQUIRE(q.out1 && q.loss_rows && q.loss_out && q.cols < (1 << 24), EBN_ERR_BAD_ARG);
      blocks = ebn_ceil_div(2 * q.cols, 256) + 1;
    }
    js.job[js.n] = q;
    js.first_block[js.n + 1] = js.first_block[js.n] + static_cast<int32_t>(blocks);
    ++js.n;
  }
  return EBN_OK;
}

extern "C" int ebn_grad_finish_f32(const ebn_finish_job* jobs, int32_t n_jobs, ebn_stream_t stream) {
  FinishJobs js;
  const int rc = plan_jobs(jobs, n_jobs, js);
  if (rc != EBN_OK) return rc;
  if (js.n == 0) return EBN_OK;
  EBN_LAUNCH(grad_finish_kernel, dim3(static_cast<unsigned>(js.first_block[js.n])), dim3(1024), 0, ebn_stream(stream), js);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}

extern "C" int ebn_grad_finish_adam_f32(const ebn_finish_job* jobs, int32_t n_jobs, const ebn_adam_flat* a, const ebn_step_state* st,
                                        ebn_stream_t stream) {
  EBN_REQUIRE(a != nullptr && st != nullptr && a->theta && a->grad && a->m && a->v && a->numel > 0, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(a->n_rest >= 0 && a->n_rest <= EBN_ADAM_FLAT_MAX_REST, EBN_ERR_BAD_ARG);
  FinishJobs js;
  const int rc = plan_jobs(jobs, n_jobs, js);
  if (rc != EBN_OK) return rc;
  auto inside = [&](const float* p, int64_t count) { return p >= a->grad && count >= 0 && (p - a->grad) + count <= a->numel; };
  for (int i = 0; i < js.n; ++i) {  // every gradient a job finishes lives in the flat gradient buffer
    const ebn_finish_job& q = js.job[i];
    if (q.kind == EBN_FINISH_SPLITK) EBN_REQUIRE(inside(q.out0, (q.rows - 1) * q.ld + q.cols), EBN_ERR_BAD_ARG);
    else EBN_REQUIRE(inside(q.out0, q.cols) && inside(q.out1, q.cols), EBN_ERR_BAD_ARG);
  }
  FinishAdam fa{};
  fa.adam = EbnAdamFlat{a->grad, a->theta, a->m, a->v, st, static_cast<float>(1.0 - a->beta1), static_cast<float>(1.0 - a->beta2),
                        static_cast<float>(a->eps), a->grad_scale};
  fa.n_rest = a->n_rest;
  for (int i = 0; i < a->n_rest; ++i) {
    EBN_REQUIRE(a->rest_off[i] >= 0 && a->rest_len[i] >= 0 && a->rest_off[i] + a->rest_len[i] <= a->numel, EBN_ERR_BAD_ARG);
    fa.rest_off[i] = a->rest_off[i];
    fa.rest_len[i] = a->rest_len[i];
    fa.rest_total += a->rest_len[i];
  }
  const int64_t blocks = js.first_block[js.n] + ebn_ceil_div(fa.rest_total, REST_PER_BLOCK);
  EBN_REQUIRE(blocks < (int64_t{1} << 30), EBN_ERR_UNSUPPORTED);
  if (blocks == 0) return EBN_OK;
  EBN_LAUNCH(grad_finish_adam_kernel, dim3(static_cast<unsigned>(blocks)), dim3(1024), 0, ebn_stream(stream), js, fa);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
