// Stage-level composition: SelfAttention + AttLayer2 over a batch of sequences, forward and
// backward -- the news encoder after its gather (nrms.py:137-156) and the user encoder
// (nrms.py:108-111).  Host code only: enqueues the kernels of the other translation units on
// the caller's stream in dependency order; no allocation, no sync (hipGraph-capturable).
#include "ebn_common.h"

#define EBN_TRY(call)            \
  do {                           \
    int rc__ = (call);           \
    if (rc__ != EBN_OK) return rc__; \
  } while (0)

static int check_dims(const ebn_encoder_dims* d) {
  EBN_REQUIRE(d, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(d->n_seq >= 0 && d->L > 0 && d->Din > 0 && d->h > 0 && d->d > 0 && d->A > 0, EBN_ERR_BAD_ARG);
  return EBN_OK;
}

extern "C" int ebn_encoder_fwd_f32(const ebn_encoder_dims* dims, const ebn_encoder_params* p,
                                   const ebn_encoder_acts* a, const ebn_encoder_scratch* s,
                                   const ebn_step_state* st, ebn_stream_t stream) {
  EBN_TRY(check_dims(dims));
  EBN_REQUIRE(p && a && p->Wqkv && p->W && p->b && p->q, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(a->X && a->QKV && a->Y && a->U && a->w && a->out, EBN_ERR_BAD_ARG);
  const int64_t R = dims->n_seq * dims->L;
  const int E = dims->h * dims->d;
  if (R == 0) return EBN_OK;
  float* ws = s ? s->gemm_ws : nullptr;
  const int64_t ws_n = s ? s->gemm_ws_floats : 0;
  // Q|K|V = X.Wqkv   (layers.py:214,220,226)
  EBN_TRY(ebn_gemm_f32_site(0, 0, R, 3 * E, dims->Din, 1.0f, a->X, dims->Din, p->Wqkv, 3 * E, 0.0f, a->QKV, 3 * E, ws, ws_n,
                            1, stream));
  // Y = dropout(P^T V)   (layers.py:231-252, nrms.py:154)
  EBN_TRY(ebn_attn_fwd_f32(a->QKV, 3 * E, a->Y, E, dims->n_seq, dims->L, dims->h, dims->d, st, dims->drop_site,
                           dims->drop_p, stream));
  // U = Y.W ; AttLayer2 tail   (layers.py:65-81)
  EBN_TRY(ebn_gemm_f32_ws(0, 0, R, dims->A, E, 1.0f, a->Y, E, p->W, dims->A, 0.0f, a->U, dims->A, ws, ws_n, stream));
  EBN_TRY(ebn_attpool_fwd_f32(a->U, p->b, p->q, a->Y, a->out, a->w, dims->n_seq, dims->L, E, dims->A, stream));
  return EBN_OK;
}

extern "C" int ebn_encoder_fwd_gather_f32(const ebn_encoder_dims* dims, const ebn_encoder_params* p, const ebn_encoder_acts* a,
                                          const ebn_encoder_scratch* s, const int32_t* ids, const float* table, int64_t table_rows,
                                          int32_t* oob_flag, ebn_stream_t stream) {
  EBN_TRY(check_dims(dims));
  EBN_REQUIRE(p && a && ids && table && p->Wqkv && p->W && p->b && p->q, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(a->QKV && a->Y && a->U && a->w && a->out, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(dims->drop_p <= 0.0f, EBN_ERR_UNSUPPORTED);  // a dropout between gather and projection needs the rows materialised
  const int64_t R = dims->n_seq * dims->L;
  const int E = dims->h * dims->d;
  if (R == 0) return EBN_OK;
  float* ws = s ? s->gemm_ws : nullptr;
  const int64_t ws_n = s ? s->gemm_ws_floats : 0;
  // Q|K|V = table[ids].Wqkv: the rows go table -> LDS -> MFMA, X is never written (nrms.py:125-139 in inference mode)
  EBN_TRY(ebn_gemm_f32_rowmap(ids, table_rows, R, 3 * E, dims->Din, table, dims->Din, p->Wqkv, 3 * E, a->QKV, 3 * E, oob_flag, stream));
  EBN_TRY(ebn_attn_fwd_f32(a->QKV, 3 * E, a->Y, E, dims->n_seq, dims->L, dims->h, dims->d, nullptr, -1, 0.0f, stream));
  EBN_TRY(ebn_gemm_f32_ws(0, 0, R, dims->A, E, 1.0f, a->Y, E, p->W, dims->A, 0.0f, a->U, dims->A, ws, ws_n, stream));
  EBN_TRY(ebn_attpool_fwd_f32(a->U, p->b, p->q, a->Y, a->out, a->w, dims->n_seq, dims->L, E, dims->A, stream));
  return EBN_OK;
}

extern "C" int ebn_encoder_bwd_f32(const ebn_encoder_dims* dims, const ebn_encoder_params* p,
                                   const ebn_encoder_acts* a, const float* dout, const ebn_encoder_grads* g,
                                   const ebn_encoder_scratch* s, float* dX, int32_t accumulate,
                                   const ebn_step_state* st, ebn_stream_t stream) {
  EBN_TRY(check_dims(dims));
  EBN_REQUIRE(p && a && g && s && dout, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(p->Wqkv && p->W && p->q, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(a->X && a->QKV && a->Y && a->U && a->w, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(g->dWqkv && g->dW && g->db && g->dq, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(s->dY && s->dQKV && s->de && s->partials, EBN_ERR_BAD_ARG);
  const int64_t R = dims->n_seq * dims->L;
  const int E = dims->h * dims->d;
  const int A = dims->A;
  if (R == 0) return EBN_OK;
  const float beta = accumulate ? 1.0f : 0.0f;
  // AttLayer2 backward: de; then dq, db and U <- d(pre-tanh)
  EBN_TRY(ebn_attpool_bwd_pool_f32(a->Y, a->w, dout, nullptr, s->de, dims->n_seq, dims->L, E, stream));
  EBN_TRY(ebn_attpool_bwd_dpre_f32(a->U, p->q, s->de, g->dq, g->db, s->partials, R, A, accumulate, stream));
  // dW = Y^T . dpre ; dY = dpre . W^T + w (x) dout
  // (with the pooling term folded into the attention backward, dW and dY below are one call: ebn_dense_bwd_pair_f32)
  // The pooling term w (x) dout rides in the attention backward when its MFMA path takes the shape (it is added while dO
  // is staged: the GEMM needs no rank-1 epilogue); otherwise in the epilogue of the dpre.W^T GEMM.
  const bool fold = ebn_attn_bwd_pooled_supported(dims->L, dims->d) != 0 && (E % 4) == 0 && ebn_aligned16(dout) &&
                    ebn_aligned16(s->dY) && ebn_aligned16(a->QKV) && ebn_aligned16(s->dQKV);
  if (fold) {
    EBN_TRY(ebn_dense_bwd_pair_f32(R, E, A, a->Y, E, a->U, A, p->W, A, beta, g->dW, A, s->dY, E, s->gemm_ws, s->gemm_ws_floats,
                                   stream));
    EBN_TRY(ebn_attn_bwd_pooled_f32(a->QKV, 3 * E, s->dY, E, a->w, dout, E, s->dQKV, 3 * E, dims->n_seq, dims->L, dims->h,
                                    dims->d, st, dims->drop_site, dims->drop_p, stream));
  } else {
    EBN_TRY(ebn_gemm_f32_ws(1, 0, E, A, R, 1.0f, a->Y, E, a->U, A, beta, g->dW, A, s->gemm_ws, s->gemm_ws_floats, stream));
    EBN_TRY(ebn_gemm_f32_rank1(R, E, A, 1.0f, a->U, A, p->W, A, s->dY, E, a->w, dout, E, dims->L, s->gemm_ws,
                               s->gemm_ws_floats, stream));
    // self-attention core backward (re-derives the dropout mask of Y)
    EBN_TRY(ebn_attn_bwd_f32(a->QKV, 3 * E, s->dY, E, s->dQKV, 3 * E, dims->n_seq, dims->L, dims->h, dims->d, st, dims->drop_site,
                             dims->drop_p, stream));
  }
  // dWqkv = X^T . dQKV ; dX = dQKV . Wqkv^T  (one launch for the pair where both are small-output shapes: the user encoder)
  if (dX != nullptr)
    return ebn_dense_bwd_pair_f32(R, dims->Din, 3 * E, a->X, dims->Din, s->dQKV, 3 * E, p->Wqkv, 3 * E, beta, g->dWqkv, 3 * E, dX,
                                  dims->Din, s->gemm_ws, s->gemm_ws_floats, stream);
  EBN_TRY(ebn_gemm_f32_ws(1, 0, dims->Din, 3 * E, R, 1.0f, a->X, dims->Din, s->dQKV, 3 * E, beta, g->dWqkv, 3 * E,
                          s->gemm_ws, s->gemm_ws_floats, stream));
  return EBN_OK;
}

// The user encoder and the scorer of a TRAINING step, forward and backward: ebn_encoder_fwd_f32 (user level) +
// ebn_score_loss_train_f32 + ebn_encoder_bwd_f32 as one call.  When the per-impression head fits one workgroup's LDS
// (ebn_user_head_supported) and the attention backward takes the pooling term (ebn_attn_bwd_pooled_supported), the middle of
// it -- AttLayer2 after its matmul, scorer, loss, their backward up to d(pre-tanh) -- is ONE launch (ebn_user_head_train_f32)
// instead of six; otherwise the three stage calls run as they are.  Same results either way (tolerance of the summation order).
extern "C" int ebn_user_stage_train_f32(const ebn_encoder_dims* dims, const ebn_encoder_params* p, const ebn_encoder_acts* a,
                                        const float* cand, const float* labels, float* scores, float* probs, float* loss_rows,
                                        float* loss_out, float* dcand, float* duser, const ebn_encoder_grads* g,
                                        const ebn_encoder_scratch* s, float* head_partials, float* dX, int32_t C,
                                        int32_t loss_kind, float inv_batch, const ebn_step_state* st, ebn_stream_t stream) {
  EBN_TRY(check_dims(dims));
  EBN_REQUIRE(p && a && g && s && cand && labels && scores && probs && loss_rows && loss_out && dcand && duser && dX, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(p->Wqkv && p->W && p->b && p->q && a->X && a->QKV && a->Y && a->U && a->w && a->out, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(g->dWqkv && g->dW && g->db && g->dq && s->dY && s->dQKV && s->de && s->partials, EBN_ERR_BAD_ARG);
  EBN_REQUIRE(C > 0 && dims->drop_p <= 0.0f, EBN_ERR_BAD_ARG);  // the user encoder has no dropout (nrms.py:108-111)
  const int64_t B = dims->n_seq, R = dims->n_seq * dims->L;
  const int E = dims->h * dims->d, A = dims->A, L = dims->L;
  if (R == 0) return EBN_OK;
  const bool fused = head_partials != nullptr && ebn_user_head_supported(L, C, E, A) != 0 &&
                     ebn_attn_bwd_pooled_supported(L, dims->d) != 0 && ebn_aligned16(a->U) && ebn_aligned16(a->Y) &&
                     ebn_aligned16(cand) && ebn_aligned16(dcand) && ebn_aligned16(duser) && ebn_aligned16(a->out) &&
                     ebn_aligned16(p->b) && ebn_aligned16(p->q) && ebn_aligned16(s->dY) && ebn_aligned16(a->QKV) &&
                     ebn_aligned16(s->dQKV);
  if (!fused) {
    EBN_TRY(ebn_encoder_fwd_f32(dims, p, a, s, st, stream));
    EBN_TRY(ebn_score_loss_train_f32(cand, a->out, labels, scores, probs, loss_rows, loss_out, dcand, duser, B, C, E, loss_kind,
                                     inv_batch, stream));
    return ebn_encoder_bwd_f32(dims, p, a, duser, g, s, dX, 0, st, stream);
  }
  float* ws = s->gemm_ws;
  const int64_t ws_n = s->gemm_ws_floats;
  // forward up to the AttLayer2 matmul (layers.py:214-252, 65)
  EBN_TRY(ebn_gemm_f32_site(0, 0, R, 3 * E, dims->Din, 1.0f, a->X, dims->Din, p->Wqkv, 3 * E, 0.0f, a->QKV, 3 * E, ws, ws_n, 1, stream));
  EBN_TRY(ebn_attn_fwd_f32(a->QKV, 3 * E, a->Y, E, B, L, dims->h, dims->d, nullptr, -1, 0.0f, stream));
  EBN_TRY(ebn_gemm_f32_ws(0, 0, R, A, E, 1.0f, a->Y, E, p->W, A, 0.0f, a->U, A, ws, ws_n, stream));
  // the head: U becomes d(pre-tanh); w, user vector, scores, loss, d(cand), d(user), de, d(q), d(b)
  EBN_TRY(ebn_user_head_train_f32(a->U, p->b, p->q, a->Y, cand, labels, a->w, a->out, scores, probs, loss_rows, loss_out, dcand,
                                  duser, s->de, g->dq, g->db, head_partials, B, L, C, E, A, loss_kind, inv_batch, stream));
  // dW = Y^T.dpre | dY = dpre.W^T (one launch), attention backward with the pooling term w (x) d(user) folded in,
  // dWqkv = X^T.dQKV | dX = dQKV.Wqkv^T (one launch)
  EBN_TRY(ebn_dense_bwd_pair_f32(R, E, A, a->Y, E, a->U, A, p->W, A, 0.0f, g->dW, A, s->dY, E, ws, ws_n, stream));
  EBN_TRY(ebn_attn_bwd_pooled_f32(a->QKV, 3 * E, s->dY, E, a->w, duser, E, s->dQKV, 3 * E, B, L, dims->h, dims->d, nullptr, -1, 0.0f,
                                  stream));
  return ebn_dense_bwd_pair_f32(R, dims->Din, 3 * E, a->X, dims->Din, s->dQKV, 3 * E, p->Wqkv, 3 * E, 0.0f, g->dWqkv, 3 * E, dX,
                                dims->Din, ws, ws_n, stream);
}
