// BCE-with-logits of one element (and its gradient), shared by glue.hip (bce_fwd_kernel, pred_bce_kernel) and conv_wino.hip (the predictor + criterion inside the
// last decoder convolution's epilogue).  models/trainer.py:206, utils/train_epoch.py:94,106:
//   l = (1 - t) * x - log_sigmoid(x),  log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|));  dl/dx = (sigmoid(x) - t) * g / n
// e = exp(-|x|) is shared by both:
//   log1p(e) = log(w) * e / (w - 1), w = fl(1 + e)   (w - 1 is exact; the ratio undoes the rounding of 1 + e)
//   sigmoid(x) = x >= 0 ? 1 / w' : e / w'            (w' = 1 + e)
// exp and the two reciprocals are the hardware approximations (<= 1 ulp each; the element's absolute error stays below 1e-7, and the mean is accumulated in
// fp64) -- with libm's expf / log1pf and IEEE divisions the kernels are VALU bound at half the HBM rate.
#pragma once
#include <hip/hip_runtime.h>

// FASTLOG: log(w), w in [1, 2], through the hardware's log2 (v_log_f32, 1 ulp) times ln 2 instead of libm's logf, and the rounding of 1 + e undone by an added term
// (e - (w - 1)) / w that reuses sigmoid's reciprocal instead of a second one -- the fused convolution epilogue (conv_wino.hip), where the criterion's vector instructions are
// not hidden behind memory time; <= 1e-7 absolute per element like the rest.
template <bool GRAD, bool FASTLOG = false>
__device__ __forceinline__ float bce_element(float x, float t, float gs, float& d) {
    const float e = __expf(-fabsf(x));
    const float w = 1.f + e;
    const float r = __frcp_rn(w);
    const float wm1 = w - 1.f;
    float l1p;
    if (FASTLOG) l1p = __builtin_fmaf(e - wm1, r, __log2f(w) * 0.693147180559945309f);      // log(1 + e) = log(w) + log((1 + e) / w), the second term = (e - (w - 1)) / w to O(1e-15)
    else l1p = wm1 == 0.f ? e : logf(w) * (e * __frcp_rn(wm1));
    if (GRAD) d = ((x >= 0.f ? r : e * r) - t) * gs;
    return (1.f - t) * x - (fminf(x, 0.f) - l1p);
}
