/* C ABI of the retired fused MLP forward (was part of include/ppf_hip.h, ABI 9) */
#pragma once
#include "ppf_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* ---- fused MLP forward (csrc/mlpfwd.hip, round 5): timm Mlp + residual + the following LayerNorm in one launch -------------------
 * `x = x + drop_path(mlp(norm2(x)))` then the next norm (deit:76-81; cait:153-157 with colscale = gamma_2):
 *   h = gelu(A W1^T + b1)  bf16 [M][hid]  and  dgelu = gelu'(A W1^T + b1) as 8-bit codes [M][hid]   (both written for backward)
 *   xout = res + rowscale[m / rows_per_group] * colscale[n] * (h W2^T + b2)   (fp32, may alias res; aux_out optional = bf16(h W2^T + b2))
 *   ln_out = bf16(LN(xout) * ln_w + ln_b), ln_mean, ln_rstd                   (ln_out == NULL: no LayerNorm)
 * A [M][D] bf16 = the LayerNorm output feeding fc1, W1 [hid][D], W2 [D][hid] bf16 (nn.Linear layout), D in {192, 384}, hid % 64 == 0,
 * tiles of rows_per_tile <= 112 rows (one workgroup each; the activation tile stays in LDS, the hidden layer is never read back). */
int ppf_mlp_fwd_supported(int D, int hid, int rows_per_tile);
int ppf_mlp_fwd(const void* A, const void* W1, const float* b1, const void* W2, const float* b2, int M, int D, int hid, int rows_per_tile,
                void* h_out, void* dgelu_out, const float* res, float* xout, const float* rowscale, int rows_per_group, const float* colscale,
                void* aux_out, const float* ln_w, const float* ln_b, void* ln_out, float* ln_mean, float* ln_rstd, float eps, ppf_stream_t stream);

#ifdef __cplusplus
}
#endif
