// The LAST launch of a one-rank NRMSDocVec training step (ebn_dvn_finale_f32): the weight-gradient tiles of ebn_gemm_tn_group_f32 and,
// dealt over the same workgroups, everything that used to follow or precede them as launches of their own --
//   * the sums over impressions of the user head's d(q) / d(b) partials and loss rows (was user_head_finish_kernel) and the L2
//     term of the loss;
//   * Keras-form Adam (nrms.py:69-80) on every parameter: in the epilogue of the tile that has just produced its gradient, and as an
//     element-wise pass over the remaining ranges of the flat parameter buffer (was adam_keras_kernel).
// Shared between ebn_gemm.hip (the kernel) and ebn_docvec.hip (the entry point, which knows the layout of the step's scratch).
#pragma once
#include "ebn_adam_flat.h"
#include "ebn_common.h"

constexpr int EBN_TN_FINALE_MAX_REST = 12;

struct EbnTnFinale {
  EbnAdamFlat adam;
  // element-wise Adam over the ranges of the flat buffers no tile owns
  int32_t n_rest;
  int64_t rest_off[EBN_TN_FINALE_MAX_REST];
  int64_t rest_len[EBN_TN_FINALE_MAX_REST];
  int64_t rest_total;
  // the user head's finishing sums (ebn_user_head_finish_body) + the L2 term of the loss
  const float* head_partials;
  int64_t B;
  int32_t A;
  float* dq;
  float* db;
  const float* loss_rows;
  float* loss_out;
  const float* l2_part;  // [n_l2][l2_slots] sums of squares left by the forward launches
  int32_t n_l2, l2_slots;
  int32_t l2_tiles[EBN_DVN_MAX_LAYERS];
  float l2;
};

int ebn_tn_group_finale_launch(const ebn_tn_problem* problems, int32_t n, const EbnTnFinale& fin, hipStream_t s);
