// fp32 VERIFICATION path (PPNet.precise = True / PPF_PRECISE=1): kernels (forward; since round 5 a DeiT backward too) that keep every operand and every
// intermediate in fp32 (exact erf GELU, fp32 FMA contractions through ppf_sgemm), so the whole forward / loss can be held to the
// north-star tolerance (1e-3 rel) against the reference-generated fixtures, separating "bf16 operand rounding" from "kernel bug".
// Written for clarity, not speed: it is never on the measured path.
#include "ppf_common.h"
#include <math.h>

namespace {

// img [B][C][H][W] fp32 -> cols fp32 [B*gh*gw][C*p*p], column order (c, py, px) = Conv2d weight.reshape(D,-1) order (deit:174)
__global__ __launch_bounds__(256) void im2col_f32_kernel(const float* __restrict__ img, float* __restrict__ cols, int B, int C, int H, int W, int p,
                                                         int64_t total) {
    const int gh = H / p, gw = W / p;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t t = i;
        const int px = t % p; t /= p;
        const int py = t % p; t /= p;
        const int c = t % C; t /= C;
        const int gx = t % gw; t /= gw;
        const int gy = t % gh; t /= gh;
        const int b = (int)t;
        cols[i] = img[(((size_t)b * C + c) * H + (gy * p + py)) * W + gx * p + px];
    }
}

// one wave per output row: y = (x - mean) * rsqrt(var + eps) * w + b   (two-pass variance, as nn.LayerNorm)
__global__ __launch_bounds__(256) void ln_f32_kernel(const float* __restrict__ x, const int* __restrict__ row_map, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* src = x + (size_t)(row_map ? row_map[r] : r) * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += src[c];
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = src[c] - mean; v += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)D + eps);
    for (int c = lane; c < D; c += 64) y[(size_t)r * D + c] = (src[c] - mean) * rstd * w[c] + b[c];
}

// in-place epilogue of an fp32 product C[M][N]:  kind 0: +bias | 1: gelu_erf(+bias) | 2: sigmoid(+bias)
//                                                 3: res + rowscale[m / rows_per_group] * colscale[n] * (C + bias)
__global__ __launch_bounds__(256) void epilogue_f32_kernel(float* __restrict__ C, const float* __restrict__ bias, int kind, const float* __restrict__ res,
                                                           const float* __restrict__ rowscale, int rows_per_group, const float* __restrict__ colscale,
                                                           int M, int N) {
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i % N), m = (int)(i / N);
        float v = C[i] + (bias ? bias[n] : 0.f);
        if (kind == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        else if (kind == 2) v = 1.0f / (1.0f + expf(-v));
        else if (kind == 3) v = res[i] + (rowscale ? rowscale[m / rows_per_group] : 1.0f) * (colscale ? colscale[n] : 1.0f) * v;
        else if (kind == 4) v = fmaxf(v, 0.f);
        C[i] = v;
    }
}

// Attention.forward with softmax_with_policy (deit:29-60) in fp32.  One workgroup per sample, thread q owns query row q (N <= 256);
// heads in sequence, K_h / V_h staged in LDS (broadcast reads).  hm (optional) [B][N][NP] receives the head-mean probabilities
// (deit:104): the same thread adds every head's row, so the order is fixed.
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, const float* __restrict__ policy,
                                                       float* __restrict__ hm, int NP, int H, int N, int D, int self_keep, int eps_n) {
    extern __shared__ float lds[];
    const int hd = D / H, b = blockIdx.x, q = threadIdx.x;
    float* tK = lds;
    float* tV = lds + (size_t)N * hd;
    const float scale = 1.0f / sqrtf((float)hd), eps = 1e-6f, epsn = eps / (float)(eps_n > 0 ? eps_n : N);
    const float* base = qkv + (size_t)b * N * 3 * D;
    const float* pol = policy ? policy + (size_t)b * N : nullptr;
    for (int h = 0; h < H; ++h) {
        __syncthreads();
        for (int i = threadIdx.x; i < N * hd; i += 256) {
            const int j = i / hd, d = i % hd;
            tK[i] = base[(size_t)j * 3 * D + D + h * hd + d];
            tV[i] = base[(size_t)j * 3 * D + 2 * D + h * hd + d];
        }
        __syncthreads();
        if (q >= N) continue;
        float qr[64], acc[64];
        for (int d = 0; d < hd; ++d) { qr[d] = base[(size_t)q * 3 * D + h * hd + d]; acc[d] = 0.f; }
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) {
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += qr[d] * tK[j * hd + d];
            mx = fmaxf(mx, s * scale);                                  // the max runs over ALL keys (deit:38), masked or not
        }
        float sum = 0.f;
        for (int j = 0; j < N; ++j) {
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += qr[d] * tK[j * hd + d];
            float keep = pol ? pol[j] : 1.0f;
            if (self_keep && j == q) keep = 1.0f;                       // keep + (1 - keep) * eye
            const float e = expf(s * scale - mx) * keep;
            sum += e;
        }
        const float zinv = 1.0f / (sum + eps);
        for (int j = 0; j < N; ++j) {
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += qr[d] * tK[j * hd + d];
            float keep = pol ? pol[j] : 1.0f;
            if (self_keep && j == q) keep = 1.0f;
            const float p = (expf(s * scale - mx) * keep + epsn) * zinv;
            for (int d = 0; d < hd; ++d) acc[d] += p * tV[j * hd + d];
            if (hm) {
                float* dst = hm + ((size_t)b * N + q) * NP + j;
                *dst = (h == 0 ? 0.f : *dst) + p / (float)H;
            }
        }
        for (int d = 0; d < hd; ++d) out[((size_t)b * N + q) * D + h * hd + d] = acc[d];
    }
}

// TalkingHeadAttn.forward (cait:115-132) in fp32: one workgroup per (sample, query); thread j owns key j (N <= 256).
// lds: q row [D] | S [H][N] raw scores | A [H][N] post-softmax mixed | red[256]
__global__ __launch_bounds__(256) void th_attn_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ wl, const float* __restrict__ bl,
                                                          const float* __restrict__ ww, const float* __restrict__ bw, float* __restrict__ out,
                                                          float* __restrict__ hm, int NP, int H, int N, int D) {
    extern __shared__ float lds[];
    const int hd = D / H, b = blockIdx.x / N, q = blockIdx.x % N, j = threadIdx.x;
    float* qrow = lds;
    float* S = qrow + D;
    float* A = S + (size_t)H * N;
    float* red = A + (size_t)H * N;
    const float scale = 1.0f / sqrtf((float)hd);
    const float* base = qkv + (size_t)b * N * 3 * D;
    for (int d = threadIdx.x; d < D; d += 256) qrow[d] = base[(size_t)q * 3 * D + d] * scale;      // q * scale first (cait:121)
    __syncthreads();
    if (j < N)
        for (int h = 0; h < H; ++h) {
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += qrow[h * hd + d] * base[(size_t)j * 3 * D + D + h * hd + d];
            S[h * N + j] = s;
        }
    __syncthreads();
    for (int g = 0; g < H; ++g) {                                       // proj_l across heads, then softmax over keys
        float m = -INFINITY;
        if (j < N) {
            m = bl[g];
            for (int h = 0; h < H; ++h) m += wl[g * H + h] * S[h * N + j];
        }
        red[threadIdx.x] = m;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
        const float mx = red[0];
        __syncthreads();
        const float e = j < N ? expf(m - mx) : 0.f;
        red[threadIdx.x] = e;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        const float sum = red[0];
        __syncthreads();
        if (j < N) A[g * N + j] = e / sum;                              // A temporarily holds P
    }
    __syncthreads();
    float mixed[16];
    if (j < N) {
        float mean = 0.f;
        for (int g = 0; g < H; ++g) {                                   // proj_w across heads (post-softmax)
            float a = bw[g];
            for (int h = 0; h < H; ++h) a += ww[g * H + h] * A[h * N + j];
            mixed[g] = a;
            mean += a;
        }
        if (hm) hm[((size_t)b * N + q) * NP + j] = mean / (float)H;     // attn.mean over heads of the RETURNED attention (cait:328)
    }
    __syncthreads();
    if (j < N)
        for (int g = 0; g < H; ++g) A[g * N + j] = mixed[g];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const int g = c / hd;
        float acc = 0.f;
        for (int k = 0; k < N; ++k) acc += A[g * N + k] * base[(size_t)k * 3 * D + 2 * D + c];
        out[((size_t)b * N + q) * D + c] = acc;
    }
}

// ClassAttn.forward + softmax_with_policy WITHOUT the identity term (cait:50-90): q [B][D] (cls rows, unscaled), k / v [B*N1][D].
// One workgroup per sample, thread j = key.  attn_mean [B][N1] = mean over heads of the probabilities (rollout row, cait:249-251).
__global__ __launch_bounds__(256) void class_attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             const float* __restrict__ policy, float* __restrict__ attn_mean, float* __restrict__ out,
                                                             int H, int N1, int D) {
    extern __shared__ float lds[];
    const int hd = D / H, b = blockIdx.x, j = threadIdx.x;
    float* P = lds;                     // [H][N1]
    float* red = P + (size_t)H * N1;
    const float scale = 1.0f / sqrtf((float)hd), eps = 1e-6f;
    float mean = 0.f;
    for (int h = 0; h < H; ++h) {
        float s = -INFINITY;
        if (j < N1) {
            s = 0.f;
            for (int d = 0; d < hd; ++d) s += q[(size_t)b * D + h * hd + d] * scale * k[((size_t)b * N1 + j) * D + h * hd + d];
        }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
        const float mx = red[0];
        __syncthreads();
        const float e = j < N1 ? expf(s - mx) * (policy ? policy[(size_t)b * N1 + j] : 1.0f) : 0.f;
        red[threadIdx.x] = e;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        const float sum = red[0];
        __syncthreads();
        if (j < N1) {
            const float p = (e + eps / (float)N1) / (sum + eps);
            P[h * N1 + j] = p;
            mean += p;
        }
    }
    if (j < N1 && attn_mean) attn_mean[(size_t)b * N1 + j] = mean / (float)H;
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const int h = c / hd;
        float acc = 0.f;
        for (int t = 0; t < N1; ++t) acc += P[h * N1 + t] * v[((size_t)b * N1 + t) * D + c];
        out[(size_t)b * D + c] = acc;
    }
}

inline int grid_for(int64_t work) {
    int64_t g = (work + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}


// ---- fp32 BACKWARD of the verification mode (round 5; DeiT): gradients of the micro fixtures held to 1e-3 of the reference's ----------------

// LayerNorm backward, one wave per row r of dy; x row = row_map ? row_map[r] : r (mean / rstd recomputed from x as the forward did):
//   dx_out[xrow] = (dres_in ? dres_in[xrow] : 0) + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w;  dw += dy * xhat, db += dy (atomics)
// dres_in and dx_out MAY be the same buffer (every element is read, then written, by the same lane): neither is __restrict__.  dw / db may be
// NULL (a frozen LayerNorm: requires_grad = False).
__global__ __launch_bounds__(256) void ln_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x, const int* __restrict__ row_map,
                                                         const float* __restrict__ w, const float* dres_in, float* dx_out,
                                                         float* __restrict__ dw, float* __restrict__ db, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const size_t xr = (size_t)(row_map ? row_map[r] : r);
    const float* src = x + xr * D;
    const float* g = dy + (size_t)r * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += src[c];
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = src[c] - mean; v += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)D + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < D; c += 64) { const float xh = (src[c] - mean) * rstd, gg = g[c] * w[c]; s1 += gg; s2 += gg * xh; }
    const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
    for (int c = lane; c < D; c += 64) {
        const float xh = (src[c] - mean) * rstd, gg = g[c] * w[c];
        dx_out[xr * D + c] = (dres_in ? dres_in[xr * D + c] : 0.f) + rstd * (gg - c1 - xh * c2);
        if (dw) atomicAdd(dw + c, g[c] * xh);
        if (db) atomicAdd(db + c, g[c]);
    }
}

// elementwise backward pieces on [M][N]: kind 0: out = a * gelu'(b) (exact erf form, b = pre-activation) | 1: out = a * b * (1 - b) (sigmoid, b = its
// output) | 2: out = a * rowscale[m / rows_per_group] (DropPath factor on a branch gradient; rowscale == NULL: copy) | 3: out = a * b |
// 4: out = a * rowscale[m / rows_per_group] * b[n] (b = a LayerScale vector [N]) | 5: out = b > 0 ? a : 0 (ReLU, b = its output)
__global__ __launch_bounds__(256) void ew_bwd_f32_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                         const float* __restrict__ rowscale, int rows_per_group, int M, int N) {
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float v = a[i];
        if (kind == 0) {
            const float xv = b[i];
            v *= 0.5f * (1.0f + erff(xv * 0.70710678118654752440f)) + xv * 0.39894228040143267794f * expf(-0.5f * xv * xv);
        } else if (kind == 1) v *= b[i] * (1.0f - b[i]);
        else if (kind == 2) v *= rowscale ? rowscale[(int)(i / N) / rows_per_group] : 1.0f;
        else if (kind == 3) v *= b[i];
        else if (kind == 5) v = b[i] > 0.f ? v : 0.f;
        else v *= (rowscale ? rowscale[(int)(i / N) / rows_per_group] : 1.0f) * b[i % N];
        out[i] = v;
    }
}

// out[n] += sum_m in[m][n]  (bias gradients).  A 1024-thread workgroup owns 32 columns: 32 row groups each add every 32nd row (128-byte
// segments, eight independent loads in flight per lane), the 32 group sums are added in a fixed order through LDS -- deterministic, and
// M / 32 sequential loads per lane instead of M (the bottleneck head's B k = 10 496 rows: advice r5).
__global__ __launch_bounds__(1024) void colsum_f32_kernel(const float* in, float* out, int M, int N) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5, n = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (n < N) {
#pragma unroll 8
        for (int m = g; m < M; m += 32) s += in[(size_t)m * N + n];
    }
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[i][cl];
        out[n] += t;
    }
}

// Backward of attn_f32_kernel: one workgroup per (sample, head).  Phase 1: thread q owns query row q -- probabilities p (the eps form),
// dP = dO V^T, delta = sum p dP, dS = scale * a * (dP - delta) with a = e keep / Z (d p_j / d s_i = a_i (delta_ij - p_j)), dQ = dS K; p and dS
// rows go to the scratch [B][H][2][N][N].  Phase 2: thread j owns key row j: dK_j = sum_i dS_ij Q_i, dV_j = sum_i p_ij dO_i.
__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ policy,
                                                           float* __restrict__ dqkv, float* __restrict__ scratch, int H, int N, int D, int self_keep, int eps_n) {
    extern __shared__ float lds[];
    const int hd = D / H, b = blockIdx.x / H, h = blockIdx.x % H, t = threadIdx.x;
    float* tA = lds;                              // K, later Q
    float* tB = lds + (size_t)N * hd;             // V, later dO
    const float scale = 1.0f / sqrtf((float)hd), eps = 1e-6f, epsn = eps / (float)(eps_n > 0 ? eps_n : N);
    const float* base = qkv + (size_t)b * N * 3 * D;
    const float* dob = dout + (size_t)b * N * D;
    float* gb = dqkv + (size_t)b * N * 3 * D;
    const float* pol = policy ? policy + (size_t)b * N : nullptr;
    float* P = scratch + ((size_t)blockIdx.x * 2) * N * N;
    float* dS = P + (size_t)N * N;
    for (int i = t; i < N * hd; i += 256) {
        const int j = i / hd, d = i % hd;
        tA[i] = base[(size_t)j * 3 * D + D + h * hd + d];
        tB[i] = base[(size_t)j * 3 * D + 2 * D + h * hd + d];
    }
    __syncthreads();
    if (t < N) {
        const int q = t;
        float qr[64], dor[64], dq[64];
        for (int d = 0; d < hd; ++d) { qr[d] = base[(size_t)q * 3 * D + h * hd + d]; dor[d] = dob[(size_t)q * D + h * hd + d]; dq[d] = 0.f; }
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) {
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += qr[d] * tA[j * hd + d];
            mx = fmaxf(mx, s * scale);
        }
        float sum = 0.f;
        for (int j = 0; j < N; ++j) {
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += qr[d] * tA[j * hd + d];
            float keep = pol ? pol[j] : 1.0f;
            if (self_keep && j == q) keep = 1.0f;
            sum += expf(s * scale - mx) * keep;
        }
        const float zinv = 1.0f / (sum + eps);
        float delta = 0.f;
        for (int j = 0; j < N; ++j) {
            float s = 0.f, dp = 0.f;
            for (int d = 0; d < hd; ++d) { s += qr[d] * tA[j * hd + d]; dp += dor[d] * tB[j * hd + d]; }
            float keep = pol ? pol[j] : 1.0f;
            if (self_keep && j == q) keep = 1.0f;
            const float p = (expf(s * scale - mx) * keep + epsn) * zinv;
            P[(size_t)q * N + j] = p;
            dS[(size_t)q * N + j] = dp;
            delta += p * dp;
        }
        for (int j = 0; j < N; ++j) {
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += qr[d] * tA[j * hd + d];
            float keep = pol ? pol[j] : 1.0f;
            if (self_keep && j == q) keep = 1.0f;
            const float a = expf(s * scale - mx) * keep * zinv;
            const float ds = scale * a * (dS[(size_t)q * N + j] - delta);
            dS[(size_t)q * N + j] = ds;
            for (int d = 0; d < hd; ++d) dq[d] += ds * tA[j * hd + d];
        }
        for (int d = 0; d < hd; ++d) gb[(size_t)q * 3 * D + h * hd + d] = dq[d];
    }
    __threadfence();
    __syncthreads();
    for (int i = t; i < N * hd; i += 256) {
        const int j = i / hd, d = i % hd;
        tA[i] = base[(size_t)j * 3 * D + h * hd + d];             // Q
        tB[i] = dob[(size_t)j * D + h * hd + d];                  // dO
    }
    __syncthreads();
    if (t < N) {
        const int j = t;
        float dk[64], dv[64];
        for (int d = 0; d < hd; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        for (int i = 0; i < N; ++i) {
            const float ds = dS[(size_t)i * N + j], p = P[(size_t)i * N + j];
            for (int d = 0; d < hd; ++d) { dk[d] += ds * tA[i * hd + d]; dv[d] += p * tB[i * hd + d]; }
        }
        for (int d = 0; d < hd; ++d) { gb[(size_t)j * 3 * D + D + h * hd + d] = dk[d]; gb[(size_t)j * 3 * D + 2 * D + h * hd + d] = dv[d]; }
    }
}

}  // namespace

extern "C" {

int ppf_im2col_patch_f32(const float* img, float* cols, int B, int C, int H, int W, int patch, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && C > 0 && patch > 0 && H % patch == 0 && W % patch == 0, PPF_ERR_SHAPE, "ppf_im2col_patch_f32: bad shape");
    const int64_t total = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(im2col_f32_kernel, dim3(grid_for(total)), dim3(256), 0, stream, img, cols, B, C, H, W, patch, total);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_layernorm_fwd_f32(const float* x, const int* row_map, const float* w, const float* b, float* y, int rows, int D, float eps,
                          hipStream_t stream) {
    PPF_CHECK_ARG(rows > 0 && D > 0, PPF_ERR_SHAPE, "ppf_layernorm_fwd_f32: bad shape");
    hipLaunchKernelGGL(ln_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, row_map, w, b, y, rows, D, eps);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_epilogue_f32(float* C, const float* bias, int kind, const float* res, const float* rowscale, int rows_per_group, const float* colscale,
                     int M, int N, hipStream_t stream) {
    PPF_CHECK_ARG(M > 0 && N > 0 && kind >= 0 && kind <= 4 && (kind != 3 || res), PPF_ERR_ARG, "ppf_epilogue_f32: bad arguments");
    hipLaunchKernelGGL(epilogue_f32_kernel, dim3(grid_for((int64_t)M * N)), dim3(256), 0, stream, C, bias, kind, res, rowscale,
                       rows_per_group > 0 ? rows_per_group : 1, colscale, M, N);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_attn_fwd_f32(const float* qkv, float* out, const float* policy, float* headmean, int NP, int B, int H, int N, int D, int self_keep,
                     int eps_n, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && H > 0 && N > 0 && N <= 256 && D % H == 0 && D / H <= 64 && (!headmean || NP >= N), PPF_ERR_SHAPE,
                  "ppf_attn_fwd_f32: bad shape B=%d H=%d N=%d D=%d", B, H, N, D);
    const size_t lds = (size_t)2 * N * (D / H) * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)attn_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { ppf_set_error("ppf_attn_fwd_f32: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(attn_f32_kernel, dim3(B), dim3(256), lds, stream, qkv, out, policy, headmean, NP, H, N, D, self_keep, eps_n);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_th_attn_fwd_f32(const float* qkv, const float* wl, const float* bl, const float* ww, const float* bw, float* out, float* headmean, int NP,
                        int B, int H, int N, int D, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && H > 0 && H <= 16 && N > 0 && N <= 256 && D % H == 0 && (!headmean || NP >= N), PPF_ERR_SHAPE, "ppf_th_attn_fwd_f32: bad shape");
    const size_t lds = ((size_t)D + 2 * (size_t)H * N + 256) * sizeof(float);
    hipLaunchKernelGGL(th_attn_f32_kernel, dim3(B * N), dim3(256), lds, stream, qkv, wl, bl, ww, bw, out, headmean, NP, H, N, D);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_class_attn_fwd_f32(const float* q, const float* k, const float* v, const float* policy, float* attn_mean, float* out, int B, int H, int N1,
                           int D, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && H > 0 && N1 > 0 && N1 <= 256 && D % H == 0, PPF_ERR_SHAPE, "ppf_class_attn_fwd_f32: bad shape");
    const size_t lds = ((size_t)H * N1 + 256) * sizeof(float);
    hipLaunchKernelGGL(class_attn_f32_kernel, dim3(B), dim3(256), lds, stream, q, k, v, policy, attn_mean, out, H, N1, D);
    PPF_LAUNCH_CHECK();
    return 0;
}


// block-wide sum over 256 threads through red[256] (every thread calls; returns the total to all)
__device__ inline float block_sum256(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float r = red[0];
    __syncthreads();
    return r;
}
__device__ inline float block_max256(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    const float r = red[0];
    __syncthreads();
    return r;
}

// Backward of th_attn_f32_kernel (cait:115-132): one workgroup per (sample, query), thread j = key.  With S = (q scale) K^T, M = wl S + bl,
// P = softmax_j M, A = ww P + bw, out = A V:   dA = dO V^T, dV += A^T dO, dww += dA P^T, dbw += sum dA, dP = ww^T dA,
// dM = P (dP - sum_j P dP), dwl += dM S^T, dbl += sum dM, dS = wl^T dM, dQ = scale dS K, dK += dS^T (q scale).
// dK / dV (dqkv zero-filled by the caller) and the four mixer gradients are accumulated with fp32 atomics (verification only).
// lds: qrow [D] | S [H][N] | P [H][N] | T [H][N] (dS) | red [256] | acc [2 H H + 2 H]
__global__ __launch_bounds__(256) void th_attn_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ wl,
                                                              const float* __restrict__ bl, const float* __restrict__ ww, const float* __restrict__ bw,
                                                              float* __restrict__ dqkv, float* __restrict__ dwl, float* __restrict__ dbl,
                                                              float* __restrict__ dww, float* __restrict__ dbw, int H, int N, int D) {
    extern __shared__ float lds[];
    const int hd = D / H, b = blockIdx.x / N, q = blockIdx.x % N, j = threadIdx.x;
    float* qrow = lds;
    float* S = qrow + D;
    float* P = S + (size_t)H * N;
    float* T = P + (size_t)H * N;
    float* red = T + (size_t)H * N;
    float* acc = red + 256;                       // dwl [H][H] | dww [H][H] | dbl [H] | dbw [H]
    const int nacc = 2 * H * H + 2 * H;
    const float scale = 1.0f / sqrtf((float)hd);
    const float* base = qkv + (size_t)b * N * 3 * D;
    const float* dor = dout + ((size_t)b * N + q) * D;
    float* gb = dqkv + (size_t)b * N * 3 * D;
    for (int d = threadIdx.x; d < D; d += 256) qrow[d] = base[(size_t)q * 3 * D + d] * scale;
    for (int i = threadIdx.x; i < nacc; i += 256) acc[i] = 0.f;
    __syncthreads();
    if (j < N)
        for (int h = 0; h < H; ++h) {
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += qrow[h * hd + d] * base[(size_t)j * 3 * D + D + h * hd + d];
            S[h * N + j] = s;
        }
    __syncthreads();
    for (int g = 0; g < H; ++g) {
        float m = -INFINITY;
        if (j < N) {
            m = bl[g];
            for (int h = 0; h < H; ++h) m += wl[g * H + h] * S[h * N + j];
        }
        const float mx = block_max256(m, red);
        const float e = j < N ? expf(m - mx) : 0.f;
        const float sum = block_sum256(e, red);
        if (j < N) P[g * N + j] = e / sum;
    }
    __syncthreads();
    float dA[16], dP[16], dM[16];
    for (int g = 0; g < H; ++g) { dA[g] = 0.f; dP[g] = 0.f; dM[g] = 0.f; }
    if (j < N) {
        for (int g = 0; g < H; ++g) {
            float a = bw[g], da = 0.f;
            for (int h = 0; h < H; ++h) a += ww[g * H + h] * P[h * N + j];
            for (int d = 0; d < hd; ++d) {
                const float o = dor[g * hd + d];
                da += o * base[(size_t)j * 3 * D + 2 * D + g * hd + d];
                atomicAdd(gb + (size_t)j * 3 * D + 2 * D + g * hd + d, a * o);                     // dV
            }
            dA[g] = da;
            atomicAdd(acc + 2 * H * H + H + g, da);                                                // dbw
            for (int h = 0; h < H; ++h) atomicAdd(acc + H * H + g * H + h, da * P[h * N + j]);     // dww
        }
        for (int h = 0; h < H; ++h) {
            float v = 0.f;
            for (int g = 0; g < H; ++g) v += ww[g * H + h] * dA[g];
            dP[h] = v;
        }
    }
    for (int g = 0; g < H; ++g) {
        const float dot = block_sum256(j < N ? P[g * N + j] * dP[g] : 0.f, red);
        if (j < N) dM[g] = P[g * N + j] * (dP[g] - dot);
    }
    if (j < N) {
        for (int g = 0; g < H; ++g) {
            atomicAdd(acc + 2 * H * H + g, dM[g]);                                                  // dbl
            for (int h = 0; h < H; ++h) atomicAdd(acc + g * H + h, dM[g] * S[h * N + j]);          // dwl
        }
        for (int h = 0; h < H; ++h) {
            float ds = 0.f;
            for (int g = 0; g < H; ++g) ds += wl[g * H + h] * dM[g];
            T[h * N + j] = ds;
            for (int d = 0; d < hd; ++d) atomicAdd(gb + (size_t)j * 3 * D + D + h * hd + d, ds * qrow[h * hd + d]);      // dK
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const int h = c / hd;
        float a = 0.f;
        for (int k = 0; k < N; ++k) a += T[h * N + k] * base[(size_t)k * 3 * D + D + c];
        gb[(size_t)q * 3 * D + c] = a * scale;                                                      // dQ (single writer)
    }
    for (int i = threadIdx.x; i < nacc; i += 256) {
        float* dst = i < H * H ? dwl + i : i < 2 * H * H ? dww + (i - H * H) : i < 2 * H * H + H ? dbl + (i - 2 * H * H) : dbw + (i - 2 * H * H - H);
        atomicAdd(dst, acc[i]);
    }
}

// Backward of class_attn_f32_kernel (cait:50-90): one workgroup per sample, thread j = key; dq [B][D], dk / dv [B*N1][D] (single writers).
// lds: dS [H][N1] | red [256]
__global__ __launch_bounds__(256) void class_attn_bwd_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                                 const float* __restrict__ policy, const float* __restrict__ dout,
                                                                 float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                                 int H, int N1, int D) {
    extern __shared__ float lds[];
    const int hd = D / H, b = blockIdx.x, j = threadIdx.x;
    float* dS = lds;
    float* red = dS + (size_t)H * N1;
    const float scale = 1.0f / sqrtf((float)hd), eps = 1e-6f;
    for (int h = 0; h < H; ++h) {
        float s = -INFINITY;
        if (j < N1) {
            s = 0.f;
            for (int d = 0; d < hd; ++d) s += q[(size_t)b * D + h * hd + d] * scale * k[((size_t)b * N1 + j) * D + h * hd + d];
        }
        const float mx = block_max256(s, red);
        const float e = j < N1 ? expf(s - mx) * (policy ? policy[(size_t)b * N1 + j] : 1.0f) : 0.f;
        const float sum = block_sum256(e, red);
        float p = 0.f, dp = 0.f;
        if (j < N1) {
            p = (e + eps / (float)N1) / (sum + eps);
            for (int d = 0; d < hd; ++d) {
                const float o = dout[(size_t)b * D + h * hd + d];
                dp += o * v[((size_t)b * N1 + j) * D + h * hd + d];
                dv[((size_t)b * N1 + j) * D + h * hd + d] = p * o;
            }
        }
        const float delta = block_sum256(p * dp, red);
        if (j < N1) {
            const float ds = e / (sum + eps) * (dp - delta);
            dS[h * N1 + j] = ds;
            for (int d = 0; d < hd; ++d) dk[((size_t)b * N1 + j) * D + h * hd + d] = ds * scale * q[(size_t)b * D + h * hd + d];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const int h = c / hd;
        float a = 0.f;
        for (int t = 0; t < N1; ++t) a += dS[h * N1 + t] * k[((size_t)b * N1 + t) * D + c];
        dq[(size_t)b * D + c] = a * scale;
    }
}


// ---- fp32 backward of the verification mode (DeiT; tests/test_gpu_precise.py holds grad/* of the reference fixtures to 1e-3 with it) ----
int ppf_layernorm_bwd_f32(const float* dy, const float* x, const int* row_map, const float* w, const float* dres_in, float* dx_out, float* dw, float* db,
                          int rows, int D, float eps, hipStream_t stream) {
    PPF_CHECK_ARG(dy && x && w && dx_out && rows > 0 && D > 0, PPF_ERR_ARG, "ppf_layernorm_bwd_f32: bad arguments");
    hipLaunchKernelGGL(ln_bwd_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, dy, x, row_map, w, dres_in, dx_out, dw, db, rows, D, eps);
    PPF_LAUNCH_CHECK();
    return 0;
}

// kind 0: out = a * gelu'(b) | 1: out = a * b * (1 - b) | 2: out = a * rowscale[m / rows_per_group] (rowscale NULL: copy) | 3: out = a * b |
// 4: out = a * rowscale[..] * b[n] (LayerScale column vector) | 5: out = b > 0 ? a : 0; out may alias a
int ppf_ew_bwd_f32(int kind, const float* a, const float* b, float* out, const float* rowscale, int rows_per_group, int M, int N, hipStream_t stream) {
    PPF_CHECK_ARG(a && out && M > 0 && N > 0 && kind >= 0 && kind <= 5 && (kind == 2 || b), PPF_ERR_ARG, "ppf_ew_bwd_f32: bad arguments");
    hipLaunchKernelGGL(ew_bwd_f32_kernel, dim3(grid_for((int64_t)M * N)), dim3(256), 0, stream, kind, a, b, out, rowscale, rows_per_group > 0 ? rows_per_group : 1, M, N);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_colsum_f32(const float* in, float* out, int M, int N, hipStream_t stream) {
    PPF_CHECK_ARG(in && out && M > 0 && N > 0, PPF_ERR_ARG, "ppf_colsum_f32: bad arguments");
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + 31) / 32), dim3(1024), 0, stream, in, out, M, N);
    PPF_LAUNCH_CHECK();
    return 0;
}

// dqkv fp32 [B*N][3D] from dout fp32 [B*N][D]; scratch: B*H*2*N*N floats
int ppf_attn_bwd_f32(const float* qkv, const float* dout, const float* policy, float* dqkv, float* scratch, int B, int H, int N, int D, int self_keep,
                     int eps_n, hipStream_t stream) {
    PPF_CHECK_ARG(qkv && dout && dqkv && scratch && B > 0 && H > 0 && N > 0 && N <= 256 && D % H == 0 && D / H <= 64, PPF_ERR_SHAPE,
                  "ppf_attn_bwd_f32: bad shape B=%d H=%d N=%d D=%d", B, H, N, D);
    const size_t lds = (size_t)2 * N * (D / H) * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { ppf_set_error("ppf_attn_bwd_f32: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(attn_bwd_f32_kernel, dim3(B * H), dim3(256), lds, stream, qkv, dout, policy, dqkv, scratch, H, N, D, self_keep, eps_n);
    PPF_LAUNCH_CHECK();
    return 0;
}


// dqkv [B*N][3D] must be ZERO-FILLED by the caller (dK / dV accumulate); dwl / dww [H][H], dbl / dbw [H] accumulate (fp32 atomics)
int ppf_th_attn_bwd_f32(const float* qkv, const float* dout, const float* wl, const float* bl, const float* ww, const float* bw, float* dqkv,
                        float* dwl, float* dbl, float* dww, float* dbw, int B, int H, int N, int D, hipStream_t stream) {
    PPF_CHECK_ARG(qkv && dout && wl && bl && ww && bw && dqkv && dwl && dbl && dww && dbw && B > 0 && H > 0 && H <= 16 && N > 0 && N <= 256 && D % H == 0,
                  PPF_ERR_SHAPE, "ppf_th_attn_bwd_f32: bad shape B=%d H=%d N=%d D=%d", B, H, N, D);
    const size_t lds = ((size_t)D + 3 * (size_t)H * N + 256 + 2 * H * H + 2 * H) * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)th_attn_bwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { ppf_set_error("ppf_th_attn_bwd_f32: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(th_attn_bwd_f32_kernel, dim3(B * N), dim3(256), lds, stream, qkv, dout, wl, bl, ww, bw, dqkv, dwl, dbl, dww, dbw, H, N, D);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_class_attn_bwd_f32(const float* q, const float* k, const float* v, const float* policy, const float* dout, float* dq, float* dk, float* dv,
                           int B, int H, int N1, int D, hipStream_t stream) {
    PPF_CHECK_ARG(q && k && v && dout && dq && dk && dv && B > 0 && H > 0 && N1 > 0 && N1 <= 256 && D % H == 0, PPF_ERR_SHAPE,
                  "ppf_class_attn_bwd_f32: bad shape");
    const size_t lds = ((size_t)H * N1 + 256) * sizeof(float);
    hipLaunchKernelGGL(class_attn_bwd_f32_kernel, dim3(B), dim3(256), lds, stream, q, k, v, policy, dout, dq, dk, dv, H, N1, D);
    PPF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
