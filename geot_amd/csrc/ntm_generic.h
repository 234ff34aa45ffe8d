// ntm_generic.h -- launchers of the runtime-class-count NTM kernels (ntm_generic.hip), called by the C ABI entry
// points in ntm.hip whenever c != 17.
#pragma once
#include <hip/hip_runtime.h>

namespace geot {

constexpr int GEN_MAXC = 32;   // a half-wave per matrix row

hipError_t gen_sig_t_mean(bool backward, int b, int n, int c, const float *p, const float *W, const float *cm,
                          const float *grad_out, float *out, hipStream_t s);
hipError_t gen_correct_fwd(int b, int n, int c, float lam, const float *logits, const float *insT, const float *E,
                           float *out, hipStream_t s);
hipError_t gen_correct_bwd(int b, int n, int c, float lam, const float *logits, const float *insT, const float *E,
                           const float *grad_out, float *grad_logits, float *grad_insT, float *grad_E, hipStream_t s);

} // namespace geot
