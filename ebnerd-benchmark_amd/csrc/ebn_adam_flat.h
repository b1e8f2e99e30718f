// Keras-form Adam (nrms.py:69-80) on ONE element of the flat parameter buffers, for the kernels that apply the optimizer where a gradient
// element is produced (ebn_dvn_finale_f32, ebn_grad_finish_adam_f32) instead of in a pass of its own (adam_keras_kernel).
#pragma once
#include "ebn_common.h"

// Adam on the flat parameter buffers: an element is addressed by the ADDRESS of its gradient (grad + offset).
struct EbnAdamFlat {
  const float* grad;  // base of the flat gradient buffer
  float* theta;
  float* m;
  float* v;
  const ebn_step_state* st;
  float omb1, omb2, eps, gscale;
};

// the update of adam_keras_kernel (ebn_score_optim.hip), one element
#define EBN_ADAM_ELEMENT(T, G, M, V, ALPHA, OMB1, OMB2, EPS, GSCALE) \
  {                                                                  \
    const float gg__ = (G) * (GSCALE);                               \
    (M) = (M) + (gg__ - (M)) * (OMB1);                               \
    (V) = (V) + (gg__ * gg__ - (V)) * (OMB2);                        \
    (T) = (T) - (ALPHA) * (M) / (sqrtf(V) + (EPS));                  \
  }

static __device__ __forceinline__ void ebn_adam_flat_apply(const EbnAdamFlat& ad, float alpha, int64_t off, float g) {
  float t = ad.theta[off], mm = ad.m[off], vv = ad.v[off];
  EBN_ADAM_ELEMENT(t, g, mm, vv, alpha, ad.omb1, ad.omb2, ad.eps, ad.gscale)
  ad.theta[off] = t;
  ad.m[off] = mm;
  ad.v[off] = vv;
}

