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

__global__ __launch_bounds__(1024) void grad_finish_kernel(FinishJobs js) {
  __shared__ float sm[32][33];
  int j = 0;
  while (j + 1 < js.n && static_cast<int>(blockIdx.x) >= js.first_block[j + 1]) ++j;  // block-uniform
  const ebn_finish_job& q = js.job[j];
  const int blk = static_cast<int>(blockIdx.x) - js.first_block[j], nblk = js.first_block[j + 1] - js.first_block[j];
  if (q.kind == EBN_FINISH_SPLITK) {
    ebn_splitk_sum_body(static_cast<uint32_t>(blk) * 1024u + threadIdx.x, static_cast<uint32_t>(nblk) * 1024u, q.partials, q.n_parts,
                        static_cast<uint32_t>(q.rows * q.cols), static_cast<uint32_t>(q.cols), q.beta, q.out0, q.ld, nullptr, nullptr, 0, 1, nullptr);
  } else if (q.kind == EBN_FINISH_COLRED) {
    ebn_reduce_partials_body(sm, blk, q.partials, q.n_parts, 2, static_cast<int>(q.cols), q.scale, q.out0, q.out1, q.beta != 0.f ? 1 : 0, nullptr,
                             nullptr);
  } else {
    ebn_user_head_finish_body(&sm[0][0], blk, nblk, q.partials, q.rows, static_cast<int>(q.cols), q.out0, q.out1, q.loss_rows, q.loss_out);
  }
}

}  // namespace

extern "C" int ebn_grad_finish_f32(const ebn_finish_job* jobs, int32_t n_jobs, ebn_stream_t stream) {
  EBN_REQUIRE(n_jobs >= 0 && n_jobs <= EBN_FINISH_MAX_JOBS && (jobs != nullptr || n_jobs == 0), EBN_ERR_BAD_ARG);
  FinishJobs js;
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
  if (js.n == 0) return EBN_OK;
  EBN_LAUNCH(grad_finish_kernel, dim3(static_cast<unsigned>(js.first_block[js.n])), dim3(1024), 0, ebn_stream(stream), js);
  EBN_CHECK_LAUNCH();
  return EBN_OK;
}
