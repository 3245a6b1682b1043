// CaiT-specific kernels (tools/cait_models_attn.py): talking-heads self-attention (cait:93-132) and class attention
// (cait:34-90).  H <= 4 heads, head_dim in {32, 48, 64}, N <= 224 tokens.
//
// Talking heads mixes the heads linearly BEFORE the softmax (proj_l) and AFTER it (proj_w), so the per-head flash structure of
// attention.hip does not apply.  The mixed logits / probabilities are materialised once per layer as (B,H,N,NP) tensors (79 MB
// fp32 at B=128 -- CaiT is a parity configuration, not the headline one) and the five per-(sample, head) matrix products run on
// the batched bf16 MFMA GEMM (ppf_gemm_bf16_batched).  This file holds what is not a plain GEMM:
//   th_scores       S'_g = sum_h Wl[g,h] * (scale q_h k_h^T) + bl[g]     (MFMA, all heads of a (query, key) tile in registers)
//                   and, in backward mode, dWl[g,h] = sum dS'_g * S_h      (same recomputation, fused reduction)
//   th_softmax_mix  P_g = softmax(S'_g);  A_g = sum_h Ww[g,h] P_h + bw[g] (bf16 MFMA operand) ; head-mean of A (rollout input)
//   th_softmax_bwd  dP, dWw, dbw, dS', dbl, dS_h = sum_g Wl[g,h] dS'_g
//   class_attn_fwd / _bwd   one query (the cls token) per (sample, head) with the policy softmax without identity term
#include "ppf_common.h"
#include <type_traits>

namespace {

constexpr float SOFTMAX_EPS = 1e-6f;

__device__ __forceinline__ int kswz(int row) {
    const int u = row >> 1;
    return ((u & 1) << 2) | (((u >> 2) & 1) << 1) | ((u >> 1) & 1);
}
__device__ __forceinline__ int row_off(int row, int c16) { return row * 128 + ((c16 ^ kswz(row)) << 4); }

struct ThParams {
    const bf16_t* qkv;      // [B*N][3D]
    int B, H, N, D, NP;
    const float* wl; const float* bl;      // proj_l [H][H], [H]
    float* sp;              // fwd: out S' [B][H][N][NP];  dwl mode: in dS' [B][H][N][NP]
    float* dwl;             // [H][H] (+=, atomics)
    float scale;
};

// grid (key blocks of 64, B), 8 waves: wave w owns queries 32w.. ; LDS: K rows of the 64 keys for all heads
template <int HD, int H, bool DWL>
__global__ __launch_bounds__(512, 2) void th_scores_kernel(const ThParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char tK[H * 64 * 128];
    __shared__ float red[8][H * H];
    constexpr int KS = HD / 16, CH = HD / 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const int b = blockIdx.y, key_begin = blockIdx.x * 64, N = p.N;
    const bf16_t* base = p.qkv + (size_t)b * N * 3 * p.D;
    for (int i = tid; i < H * 64 * CH; i += 512) {
        const int c = i % CH, r = (i / CH) % 64, h = i / (CH * 64);
        const int key = min(key_begin + r, N - 1);
        *reinterpret_cast<uint4*>(tK + h * 64 * 128 + row_off(r, c)) = *reinterpret_cast<const uint4*>(base + (size_t)key * 3 * p.D + p.D + h * HD + c * 8);
    }
    float wl[H][H], blv[H];
#pragma unroll
    for (int g = 0; g < H; ++g) {
        blv[g] = p.bl[g];
#pragma unroll
        for (int h = 0; h < H; ++h) wl[g][h] = p.wl[g * H + h];
    }
    __syncthreads();
    const int q0 = wave * 32;
    float part[H][H];
#pragma unroll
    for (int g = 0; g < H; ++g)
#pragma unroll
        for (int h = 0; h < H; ++h) part[g][h] = 0.f;
    if (q0 < N) {
        const int q = q0 + (lane & 31), qc = min(q, N - 1);
        bf16x8 qf[H][KS];
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[h][ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)qc * 3 * p.D + h * HD + ks * 16 + hh * 8);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 s[H];
#pragma unroll
            for (int h = 0; h < H; ++h) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[h][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(tK + h * 64 * 128 + row_off(t * 32 + (lane & 31), ks * 2 + hh));
                    s[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[h][ks], s[h], 0, 0, 0);       // [key][q]
                }
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int key = key_begin + t * 32 + 8 * g4 + 4 * hh;
                if (q < N && key < N) {                                   // NP is a multiple of 4: whole float4 groups
                    if constexpr (!DWL) {
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            float o[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                float a = blv[g];
#pragma unroll
                                for (int h = 0; h < H; ++h) a += wl[g][h] * (s[h][4 * g4 + i] * p.scale);
                                o[i] = a;
                            }
                            *reinterpret_cast<float4*>(p.sp + (((size_t)b * H + g) * N + q) * p.NP + key) = make_float4(o[0], o[1], o[2], o[3]);
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < H; ++g) {
                            const float4 d = *reinterpret_cast<const float4*>(p.sp + (((size_t)b * H + g) * N + q) * p.NP + key);
                            const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (key + i < N) {
#pragma unroll
                                    for (int h = 0; h < H; ++h) part[g][h] += dv[i] * (s[h][4 * g4 + i] * p.scale);
                                }
                        }
                    }
                }
            }
        }
    }
    if constexpr (DWL) {
#pragma unroll
        for (int g = 0; g < H; ++g)
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float v = wave_sum(part[g][h]);
                if (lane == 0) red[wave][g * H + h] = v;
            }
        __syncthreads();
        if (tid < H * H) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += red[w][tid];
            unsafeAtomicAdd(p.dwl + tid, a);
        }
    }
}

struct ThSmParams {
    float* sp;               // in S' / out P           [B][H][N][NP]
    bf16_t* a16;             // out A bf16              [B][H][N][NPK]
    float* hm;               // out head-mean of A      [B][N][NP]
    const float* ww; const float* bw;
    int B, H, N, NP, NPK;
    // backward
    const float* da;         // in dA fp32 [B][H][N][NP]  (overwritten with dS')
    float* ds_prime;         // == da
    bf16_t* ds16;            // out dS bf16 [B][H][N][NPK]
    const float* wl;
    float* dww; float* dbw; float* dbl;
};

// one wave per (b, q): lane owns keys 4*lane..4*lane+3
template <int H>
__global__ __launch_bounds__(256) void th_softmax_mix_kernel(const ThSmParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.B * p.N) return;
    const int b = row / p.N, q = row % p.N, k0 = 4 * lane;
    float pr[H][4];
#pragma unroll
    for (int g = 0; g < H; ++g) {
        float* src = p.sp + (((size_t)b * H + g) * p.N + q) * p.NP;
        float v[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (k0 < p.NP) {
            const float4 t = *reinterpret_cast<const float4*>(src + k0);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) { if (k0 + i >= p.N) v[i] = -INFINITY; mx = fmaxf(mx, v[i]); }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = (k0 + i < p.N) ? __expf(v[i] - mx) : 0.f; s += v[i]; }
        s = 1.0f / wave_sum(s);
#pragma unroll
        for (int i = 0; i < 4; ++i) pr[g][i] = v[i] * s;
        if (k0 < p.NP) *reinterpret_cast<float4*>(src + k0) = make_float4(pr[g][0], pr[g][1], pr[g][2], pr[g][3]);
    }
    float mean[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < H; ++g) {
        float a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc = p.bw[g];
#pragma unroll
            for (int h = 0; h < H; ++h) acc += p.ww[g * H + h] * pr[h][i];
            a[i] = (k0 + i < p.N) ? acc : 0.f;
            mean[i] += a[i];
        }
        if (k0 < p.NPK)
            *reinterpret_cast<uint2*>(p.a16 + (((size_t)b * H + g) * p.N + q) * p.NPK + k0) = make_uint2(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]));
    }
    if (k0 < p.NP) {
        const float ih = 1.0f / (float)H;
        *reinterpret_cast<float4*>(p.hm + ((size_t)b * p.N + q) * p.NP + k0) = make_float4(mean[0] * ih, mean[1] * ih, mean[2] * ih, mean[3] * ih);
    }
}

template <int H>
__global__ __launch_bounds__(256) void th_softmax_bwd_kernel(const ThSmParams p) {
    __shared__ float red[4][2 * H * H + 2 * H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    const bool active = row < p.B * p.N;
    const int b = active ? row / p.N : 0, q = active ? row % p.N : 0, k0 = 4 * lane;
    float pr[H][4], da[H][4];
#pragma unroll
    for (int g = 0; g < H; ++g) {
        const size_t o = (((size_t)b * H + g) * p.N + q) * p.NP + k0;
        float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), dv = pv;
        if (active && k0 < p.NP) { pv = *reinterpret_cast<const float4*>(p.sp + o); dv = *reinterpret_cast<const float4*>(p.da + o); }
        pr[g][0] = pv.x; pr[g][1] = pv.y; pr[g][2] = pv.z; pr[g][3] = pv.w;
        da[g][0] = dv.x; da[g][1] = dv.y; da[g][2] = dv.z; da[g][3] = dv.w;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (k0 + i >= p.N) { pr[g][i] = 0.f; da[g][i] = 0.f; }
    }
    // parameter-gradient partials of proj_w: dWw[g][h] = sum dA_g P_h, dbw[g] = sum dA_g
    float acc_ww[H][H], acc_bw[H], acc_bl[H];
#pragma unroll
    for (int g = 0; g < H; ++g) {
        float sb = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) sb += da[g][i];
        acc_bw[g] = sb;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) s += da[g][i] * pr[h][i];
            acc_ww[g][h] = s;
        }
    }
    // dP_h = sum_g Ww[g][h] dA_g ;  dS'_h = P_h (dP_h - sum_key dP_h P_h)
    float dsp[H][4];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float dp[4], dot = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a = 0.f;
#pragma unroll
            for (int g = 0; g < H; ++g) a += p.ww[g * H + h] * da[g][i];
            dp[i] = a;
            dot += a * pr[h][i];
        }
        dot = wave_sum(dot);
        float sb = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { dsp[h][i] = pr[h][i] * (dp[i] - dot); sb += dsp[h][i]; }
        acc_bl[h] = sb;
        if (active && k0 < p.NP)
            *reinterpret_cast<float4*>(p.ds_prime + (((size_t)b * H + h) * p.N + q) * p.NP + k0) = make_float4(dsp[h][0], dsp[h][1], dsp[h][2], dsp[h][3]);
    }
    // dS_h = sum_g Wl[g][h] dS'_g   (bf16 operand of the dQ / dK products)
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a = 0.f;
#pragma unroll
            for (int g = 0; g < H; ++g) a += p.wl[g * H + h] * dsp[g][i];
            o[i] = a;
        }
        if (active && k0 < p.NPK)
            *reinterpret_cast<uint2*>(p.ds16 + (((size_t)b * H + h) * p.N + q) * p.NPK + k0) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
    }
    // reductions: wave -> workgroup -> atomics
    int slot = 0;
#pragma unroll
    for (int g = 0; g < H; ++g)
#pragma unroll
        for (int h = 0; h < H; ++h) { const float v = wave_sum(acc_ww[g][h]); if (lane == 0) red[wave][slot] = v; ++slot; }
#pragma unroll
    for (int g = 0; g < H; ++g) { const float v = wave_sum(acc_bw[g]); if (lane == 0) red[wave][slot] = v; ++slot; }
#pragma unroll
    for (int g = 0; g < H; ++g) { const float v = wave_sum(acc_bl[g]); if (lane == 0) red[wave][slot] = v; ++slot; }
    __syncthreads();
    if (threadIdx.x < H * H + 2 * H) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        float* dst = threadIdx.x < H * H ? p.dww + threadIdx.x : (threadIdx.x < H * H + H ? p.dbw + (threadIdx.x - H * H) : p.dbl + (threadIdx.x - H * H - H));
        unsafeAtomicAdd(dst, v);
    }
}

// ------------------------------------------------------------------------------------------------ class attention
struct CaParams {
    const bf16_t* q;        // [B][D]     (unscaled q projection of the cls row)
    const bf16_t* k;        // [B*N1][D]
    const bf16_t* v;        // [B*N1][D]
    const float* policy;    // [B][N1] or null
    float* attn;            // [B][H][N1]  probabilities
    float* zinv;            // [B][H]
    float* rowmean;         // [B][N1] head-mean (rollout init row)
    bf16_t* out;            // [B][D]
    const bf16_t* dout;     // [B][D]
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    int B, H, N1, D;
    float scale;
};

// one workgroup per sample, one wave per head (H <= 4)
__global__ __launch_bounds__(256) void class_attn_fwd_kernel(const CaParams p) {
    __shared__ float prob[4][256];
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, b = blockIdx.x;
    const int hd = p.D / p.H, N1 = p.N1;
    if (h < p.H) {
        const bf16_t* qrow = p.q + (size_t)b * p.D + h * hd;
        float s[4], mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lane + 64 * i;
            float a = -INFINITY;
            if (j < N1) {
                const bf16_t* krow = p.k + ((size_t)b * N1 + j) * p.D + h * hd;
                a = 0.f;
                for (int d = 0; d < hd; d += 2) {
                    const float2 kk = unpack_bf16x2(*reinterpret_cast<const uint32_t*>(krow + d));
                    const float2 qq = unpack_bf16x2(*reinterpret_cast<const uint32_t*>(qrow + d));
                    a += kk.x * qq.x + kk.y * qq.y;
                }
                a *= p.scale;
            }
            s[i] = a;
            mx = fmaxf(mx, a);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lane + 64 * i;
            const float keep = (j < N1) ? (p.policy ? p.policy[(size_t)b * N1 + j] : 1.0f) : 0.f;
            s[i] = (j < N1) ? __expf(s[i] - mx) * keep : 0.f;
            sum += s[i];
        }
        sum = wave_sum(sum);
        const float zi = 1.0f / (sum + SOFTMAX_EPS), c = SOFTMAX_EPS / (float)N1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lane + 64 * i;
            if (j < N1) {
                const float a = (s[i] + c) * zi;
                prob[h][j] = a;
                p.attn[((size_t)b * p.H + h) * N1 + j] = a;
            }
        }
        if (lane == 0) p.zinv[(size_t)b * p.H + h] = zi;
    }
    __syncthreads();
    if (h < p.H && lane < hd) {
        float acc = 0.f;
        for (int j = 0; j < N1; ++j) acc += prob[h][j] * bf16_to_f32(p.v[((size_t)b * N1 + j) * p.D + h * hd + lane]);
        p.out[(size_t)b * p.D + h * hd + lane] = f32_to_bf16(acc);
    }
    for (int j = threadIdx.x; j < N1; j += 256) {
        float m = 0.f;
        for (int g = 0; g < p.H; ++g) m += prob[g][j];
        p.rowmean[(size_t)b * N1 + j] = m / (float)p.H;
    }
}

__global__ __launch_bounds__(256) void class_attn_bwd_kernel(const CaParams p) {
    __shared__ float dsv[4][256];
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, b = blockIdx.x;
    const int hd = p.D / p.H, N1 = p.N1;
    if (h < p.H) {
        const bf16_t* dorow = p.dout + (size_t)b * p.D + h * hd;
        const float zi = p.zinv[(size_t)b * p.H + h], c = SOFTMAX_EPS / (float)N1;
        float g[4], a[4], dot = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lane + 64 * i;
            g[i] = 0.f; a[i] = 0.f;
            if (j < N1) {
                const bf16_t* vrow = p.v + ((size_t)b * N1 + j) * p.D + h * hd;
                float acc = 0.f;
                for (int d = 0; d < hd; d += 2) {
                    const float2 vv = unpack_bf16x2(*reinterpret_cast<const uint32_t*>(vrow + d));
                    const float2 dd = unpack_bf16x2(*reinterpret_cast<const uint32_t*>(dorow + d));
                    acc += vv.x * dd.x + vv.y * dd.y;
                }
                g[i] = acc;
                a[i] = p.attn[((size_t)b * p.H + h) * N1 + j];
                dot += acc * a[i];
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lane + 64 * i;
            if (j < N1) {
                const float pt = a[i] - c * zi;                          // e/(sum+eps): the eps/N term carries no gradient
                const float ds = pt * (g[i] - dot) * p.scale;
                dsv[h][j] = ds;
                // dK_j = ds * q ; dV_j = a_j * dout
                const bf16_t* qrow = p.q + (size_t)b * p.D + h * hd;
                bf16_t* dkrow = p.dk + ((size_t)b * N1 + j) * p.D + h * hd;
                bf16_t* dvrow = p.dv + ((size_t)b * N1 + j) * p.D + h * hd;
                for (int d = 0; d < hd; d += 2) {
                    const float2 qq = unpack_bf16x2(*reinterpret_cast<const uint32_t*>(qrow + d));
                    const float2 dd = unpack_bf16x2(*reinterpret_cast<const uint32_t*>(dorow + d));
                    *reinterpret_cast<uint32_t*>(dkrow + d) = pack_bf16x2(ds * qq.x, ds * qq.y);
                    *reinterpret_cast<uint32_t*>(dvrow + d) = pack_bf16x2(a[i] * dd.x, a[i] * dd.y);
                }
            }
        }
    }
    __syncthreads();
    if (h < p.H && lane < hd) {
        float acc = 0.f;
        for (int j = 0; j < N1; ++j) acc += dsv[h][j] * bf16_to_f32(p.k[((size_t)b * N1 + j) * p.D + h * hd + lane]);
        p.dq[(size_t)b * p.D + h * hd + lane] = f32_to_bf16(acc);
    }
}

// dn16[r][c] = bf16(a[r][c] + b[r][c] + (r % N1 == 0 ? cq[r / N1][c] : 0)): merges the k-, v- and q-path input gradients
__global__ __launch_bounds__(256) void merge3_cast_kernel(const float* a, const float* b, const float* cq, bf16_t* out, int rows, int D, int N1) {
    const size_t total = (size_t)rows * D / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t e = i * 2, r = e / D, c = e % D;
        float2 v = *reinterpret_cast<const float2*>(a + e);
        const float2 w = *reinterpret_cast<const float2*>(b + e);
        v.x += w.x; v.y += w.y;
        if (r % N1 == 0) { const float2 z = *reinterpret_cast<const float2*>(cq + (r / N1) * D + c); v.x += z.x; v.y += z.y; }
        *reinterpret_cast<uint32_t*>(out + e) = pack_bf16x2(v.x, v.y);
    }
}

// ------------------------------------------------------------------------------------------------ fused talking heads
// One launch per direction instead of the materialised (B,H,N,N) fp32 logits / probabilities / gradients of the kernels above
// (th_scores -> th_softmax_mix, and dA GEMM -> th_softmax_bwd -> th_dwl).  A wave owns 16 queries; K (and V in backward) of ALL heads of
// the sample sit in LDS, and the score tiles of the H heads are recomputed on the MFMA pipe in every pass -- the pipe is idle otherwise.
// v_mfma_f32_16x16x32_bf16 with "swapped" operands (first = 16 keys, second = 16 queries): a lane owns ONE query (lane & 15) and four
// consecutive keys (4 * (lane >> 4) ..) of every 16-key tile, for all H heads at once -- 4 x H accumulator registers per tile -- which is what
// the two head mixes need (they combine the H heads of one (query, key) position), and the softmax reductions are in-lane plus two
// cross-group shuffles.  (The forward kernel processes two key tiles per loop trip, the backward kernel one: registers.)  head_dim = 48 runs as one full 32-wide contraction step and one half
// step (the lanes that would hold d >= 48 pass zeros; the LDS rows stay 96 bytes).
// A first version with 32x32x16 tiles (32 queries per wave, 16 registers per head and tile) needed > 256 VGPRs in backward (both the
// score and the dA tiles of all heads are live in the elementwise stage), spilled, and ran one wave per SIMD: 184 us; see
// profiles/r3_cait.txt.
// Saved for backward: row maximum and 1 / sum of the mixed logits per (b, g, q) and the bf16 mixed probabilities A (operand of the
// dV product); nothing else of size N x N.
struct ThFusedParams {
    const bf16_t* qkv;      // [B*N][3D]
    const bf16_t* dout;     // [B*N][D]            backward
    const float* wl; const float* bl; const float* ww; const float* bw;
    bf16_t* a16;            // fwd out [B][H][N][NPK]   A_g = sum_h Ww[g,h] P_h + bw[g]
    bf16_t* out;            // fwd out [B*N][D]         O_g = A_g V_g (cait:128), or null: the caller multiplies
    float* hm;              // fwd out [B][N][NP]       mean over heads of A
    float* rowmax;          // [B][H][N]  fwd out / bwd in
    float* zinv;            // [B][H][N]
    bf16_t* ds16;           // bwd out [B][H][N][NPK]   dS_h = sum_g Wl[g,h] dS'_g
    float* partial;         // bwd out [workgroups][2*H*H + 2*H]: dWw | dbw | dbl | dWl
    int B, N, D, NP, NPK;
    float scale;
};
constexpr float TH_LOG2E = 1.44269504088896340736f;
constexpr int TH_WAVES = 7, TH_NTHR = TH_WAVES * 64;      // 7 x 16 = 112 queries per workgroup: N = 196 -> 2 workgroups per sample

// [H][N][HD] bf16 image of one third (K or V) of the sample's packed qkv rows; global reads are row-contiguous
template <int HD, int H>
__device__ __forceinline__ void th_stage(unsigned char* img, const bf16_t* src, int N, int ld, int tid) {
    constexpr int CH = HD / 8, RC = H * CH;
    for (int i = tid; i < N * RC; i += TH_NTHR) {
        const int row = i / RC, c24 = i - row * RC, h = c24 / CH, c = c24 - h * CH;
        *reinterpret_cast<uint4*>(img + ((size_t)(h * N + row) * HD + c * 8) * 2) = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + c24 * 8);
    }
}
// first MFMA operand from an LDS image: row key_base + (lane & 15) (clamped to the last valid row: finite values, masked by the caller),
// contraction slots 32 ks + 8 (lane >> 4) .. + 7 (zeros beyond the head dimension)
template <int HD>
__device__ __forceinline__ bf16x8 th_frag(const unsigned char* img, int N, int h, int key_base, int ks, int lane) {
    const int row = min(key_base + (lane & 15), N - 1), k0 = 32 * ks + 8 * (lane >> 4);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (HD % 32 == 0 || k0 < HD) v = *reinterpret_cast<const uint4*>(img + ((size_t)(h * N + row) * HD + k0) * 2);
    return __builtin_bit_cast(bf16x8, v);
}
// second MFMA operand straight from global memory: row `row` of a [rows][ld] bf16 matrix, columns col0 + (same slots)
template <int HD>
__device__ __forceinline__ bf16x8 th_frag_global(const bf16_t* src, size_t row, int ld, int col0, int ks, int lane, bool live) {
    const int k0 = 32 * ks + 8 * (lane >> 4);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (live && (HD % 32 == 0 || k0 < HD)) v = *reinterpret_cast<const uint4*>(src + row * ld + col0 + k0);
    return __builtin_bit_cast(bf16x8, v);
}

// first operand of the A.V product, read transposed from the V image: output index d = dbase + (lane & 15), contraction slots
// 8 (lane >> 4) + j <-> keys kb + 4 (lane >> 4) + j (j < 4) and kb + 16 + 4 (lane >> 4) + (j - 4): the register order of two adjacent
// 16-key probability tiles.  Rows past N repeat the last valid one (their probabilities are zero).
template <int HD>
__device__ __forceinline__ bf16x8 th_frag_tr(const unsigned char* img, int N, int h, int kb, int dbase, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int s = lane & 15, grp = lane >> 4;
    const int d = dbase + 4 * (s & 3);
    const int r0 = min(kb + 4 * grp + (s >> 2), N - 1), r1 = min(kb + 16 + 4 * grp + (s >> 2), N - 1);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + ((size_t)(h * N + r0) * HD + d) * 2));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + ((size_t)(h * N + r1) * HD + d) * 2));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// grid (ceil(ceil(N / 16) / 7), B), 7 waves; LDS: K image of all heads (+ the V image when the A.V product runs here)
template <int HD, int H>
__global__ __launch_bounds__(TH_NTHR, 1) void th_fwd_kernel(const ThFusedParams p) {
    constexpr int KS = (HD + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char th_lds[];
    unsigned char* tK = th_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4;
    const int b = blockIdx.y, N = p.N, NT2 = (N + 31) / 32;
    const bf16_t* base = p.qkv + (size_t)b * N * 3 * p.D;
    const int q0 = (blockIdx.x * TH_WAVES + wave) * 16, q = q0 + (lane & 15), qc = min(q, N - 1);
    bf16x8 qf[H][KS];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[h][ks] = th_frag_global<HD>(base, (size_t)qc, 3 * p.D, h * HD, ks, lane, true);
    th_stage<HD, H>(tK, base + p.D, N, 3 * p.D, tid);
    const unsigned char* tV = th_lds + (size_t)H * N * HD * 2;
    if (p.out) th_stage<HD, H>(th_lds + (size_t)H * N * HD * 2, base + 2 * p.D, N, 3 * p.D, tid);
    float wl[H][H], ww[H][H], blv[H], bwv[H];
#pragma unroll
    for (int g = 0; g < H; ++g) {
        blv[g] = p.bl[g]; bwv[g] = p.bw[g];
#pragma unroll
        for (int h = 0; h < H; ++h) { wl[g][h] = p.wl[g * H + h] * p.scale; ww[g][h] = p.ww[g * H + h]; }
    }
    __syncthreads();
    if (q0 >= N) return;
    // pass 1: running maximum and sum of exp of the mixed logits, per head, over this lane's keys
    float m[H], l[H];
#pragma unroll
    for (int g = 0; g < H; ++g) { m[g] = -1e30f; l[g] = 0.f; }
#pragma unroll 1
    for (int t2 = 0; t2 < NT2; ++t2) {
        f32x4 s[2][H];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int h = 0; h < H; ++h) {
                s[u][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    s[u][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th_frag<HD>(tK, N, h, t2 * 32 + u * 16, ks, lane), qf[h][ks], s[u][h], 0, 0, 0);
            }
        float o[H][8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool valid = t2 * 32 + u * 16 + 4 * grp + i < N;
#pragma unroll
                for (int g = 0; g < H; ++g) {
                    float a = blv[g];
#pragma unroll
                    for (int h = 0; h < H; ++h) a += wl[g][h] * s[u][h][i];
                    o[g][u * 4 + i] = valid ? a : -INFINITY;
                }
            }
#pragma unroll
        for (int g = 0; g < H; ++g) {
            float mt = m[g];
#pragma unroll
            for (int r = 0; r < 8; ++r) mt = fmaxf(mt, o[g][r]);
            float acc = l[g] * __builtin_amdgcn_exp2f((m[g] - mt) * TH_LOG2E);
#pragma unroll
            for (int r = 0; r < 8; ++r) acc += __builtin_amdgcn_exp2f((o[g][r] - mt) * TH_LOG2E);
            l[g] = acc; m[g] = mt;
        }
    }
    float m2[H], zi[H];
#pragma unroll
    for (int g = 0; g < H; ++g) {
        float mm = m[g], sum = l[g];
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            const float mo = __shfl_xor(mm, sh, 64), lo = __shfl_xor(sum, sh, 64);
            const float mn = fmaxf(mm, mo);
            sum = sum * __builtin_amdgcn_exp2f((mm - mn) * TH_LOG2E) + lo * __builtin_amdgcn_exp2f((mo - mn) * TH_LOG2E);
            mm = mn;
        }
        m2[g] = mm * TH_LOG2E; zi[g] = 1.0f / sum;
        if (grp == 0 && q < N) {
            const size_t si = ((size_t)b * H + g) * N + q;
            p.rowmax[si] = mm; p.zinv[si] = zi[g];
        }
    }
    // pass 2: P_g = softmax(S'_g), A_g = sum_h Ww[g,h] P_h + bw[g]  ->  bf16 (saved: operand of the dV product), the head mean (rollout
    // input) and, with p.out, O_g = A_g V_g right here: the two tiles of a trip are exactly one 32-key contraction step
    constexpr int DB = HD / 16;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const float ih = 1.0f / (float)H;
    f32x4 o[H][DB];
#pragma unroll
    for (int g = 0; g < H; ++g)
#pragma unroll
        for (int db = 0; db < DB; ++db) o[g][db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int t2 = 0; t2 < NT2; ++t2) {
        f32x4 s[2][H];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int h = 0; h < H; ++h) {
                s[u][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    s[u][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th_frag<HD>(tK, N, h, t2 * 32 + u * 16, ks, lane), qf[h][ks], s[u][h], 0, 0, 0);
            }
        uint32_t pk[H][4];                                  // bf16 pairs of A_g: [u = 0: keys 0-1, 2-3 | u = 1: keys 0-1, 2-3]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int key0 = t2 * 32 + u * 16 + 4 * grp;
            float av[H][4], mean[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool valid = key0 + i < N;
                float pr[H];
#pragma unroll
                for (int g = 0; g < H; ++g) {
                    float a = blv[g];
#pragma unroll
                    for (int h = 0; h < H; ++h) a += wl[g][h] * s[u][h][i];
                    pr[g] = valid ? __builtin_amdgcn_exp2f(a * TH_LOG2E - m2[g]) * zi[g] : 0.f;
                }
                float mu = 0.f;
#pragma unroll
                for (int g = 0; g < H; ++g) {
                    float a = bwv[g];
#pragma unroll
                    for (int h = 0; h < H; ++h) a += ww[g][h] * pr[h];
                    a = valid ? a : 0.f;
                    av[g][i] = a; mu += a;
                }
                mean[i] = mu * ih;
            }
#pragma unroll
            for (int g = 0; g < H; ++g) { pk[g][2 * u] = pack_bf16x2(av[g][0], av[g][1]); pk[g][2 * u + 1] = pack_bf16x2(av[g][2], av[g][3]); }
            if (q < N) {
                if (key0 < p.NPK) {
#pragma unroll
                    for (int g = 0; g < H; ++g) *reinterpret_cast<uint2*>(p.a16 + (((size_t)b * H + g) * N + q) * p.NPK + key0) = make_uint2(pk[g][2 * u], pk[g][2 * u + 1]);
                }
                if (key0 < p.NP) *reinterpret_cast<float4*>(p.hm + ((size_t)b * N + q) * p.NP + key0) = make_float4(mean[0], mean[1], mean[2], mean[3]);
            }
        }
        if (p.out) {                                        // uniform
#pragma unroll
            for (int g = 0; g < H; ++g) {
                const u32x4 uu = {pk[g][0], pk[g][1], pk[g][2], pk[g][3]};
                const bf16x8 pf = __builtin_bit_cast(bf16x8, uu);
#pragma unroll
                for (int db = 0; db < DB; ++db)
                    o[g][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th_frag_tr<HD>(tV, N, g, t2 * 32, db * 16, lane), pf, o[g][db], 0, 0, 0);
            }
        }
    }
    if (p.out && q < N) {
        bf16_t* orow = p.out + ((size_t)b * N + q) * p.D;
#pragma unroll
        for (int g = 0; g < H; ++g)
#pragma unroll
            for (int db = 0; db < DB; ++db)
                *reinterpret_cast<uint2*>(orow + g * HD + db * 16 + 4 * grp) = make_uint2(pack_bf16x2(o[g][db][0], o[g][db][1]), pack_bf16x2(o[g][db][2], o[g][db][3]));
    }
}

// Backward of  A = proj_w(softmax(proj_l(scale q k^T)))  given dO: everything between the dA = dO V^T product and dS, in registers.
//   dA_g = dO_g V_g^T;  dP_h = sum_g Ww[g,h] dA_g;  dS'_h = P_h (dP_h - sum_key dP_h P_h);  dS_h = sum_g Wl[g,h] dS'_g   (written, bf16)
//   dWw[g,h] = sum dA_g P_h;  dbw[g] = sum dA_g;  dbl[h] = sum dS'_h;  dWl[g,h] = sum dS'_g (scale q_h k_h)           (per-workgroup partials)
// Pass 1 accumulates the softmax-backward row term, pass 2 recomputes the tiles and emits.  Same grid as the forward kernel; LDS: the K and
// V images of all heads (2 * H * N * HD * 2 bytes: 150.5 KB for CaiT-XXS) -> one workgroup of 7 waves per CU.
template <int HD, int H>
__global__ __launch_bounds__(TH_NTHR, 1) void th_bwd_kernel(const ThFusedParams p) {
    constexpr int KS = (HD + 31) / 32, PW = 2 * H * H + 2 * H;
    extern __shared__ __attribute__((aligned(16))) unsigned char th_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4;
    const int b = blockIdx.y, N = p.N, NT = (N + 15) / 16;
    unsigned char* tK = th_lds;
    unsigned char* tV = th_lds + (size_t)H * N * HD * 2;
    float* red = reinterpret_cast<float*>(th_lds + (size_t)2 * H * N * HD * 2);       // [TH_WAVES][PW]
    const bf16_t* base = p.qkv + (size_t)b * N * 3 * p.D;
    const int q0 = (blockIdx.x * TH_WAVES + wave) * 16, q = q0 + (lane & 15), qc = min(q, N - 1);
    const bool qvalid = q < N;
    bf16x8 qf[H][KS], dof[H][KS];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[h][ks] = th_frag_global<HD>(base, (size_t)qc, 3 * p.D, h * HD, ks, lane, true);
            dof[h][ks] = th_frag_global<HD>(p.dout, (size_t)b * N + qc, p.D, h * HD, ks, lane, qvalid);      // padded queries: dA = 0
        }
    th_stage<HD, H>(tK, base + p.D, N, 3 * p.D, tid);
    th_stage<HD, H>(tV, base + 2 * p.D, N, 3 * p.D, tid);
    float wl[H][H], ww[H][H], blv[H], m2[H], zi[H];
#pragma unroll
    for (int g = 0; g < H; ++g) {
        blv[g] = p.bl[g];
        const size_t si = ((size_t)b * H + g) * N + qc;
        m2[g] = p.rowmax[si] * TH_LOG2E;
        zi[g] = qvalid ? p.zinv[si] : 0.f;                                    // padded queries: P = 0
#pragma unroll
        for (int h = 0; h < H; ++h) { wl[g][h] = p.wl[g * H + h]; ww[g][h] = p.ww[g * H + h]; }
    }
    __syncthreads();
    float acc[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) acc[i] = 0.f;
    if (q0 < N) {
        float dot[H];
#pragma unroll
        for (int h = 0; h < H; ++h) dot[h] = 0.f;
        // pass 1: dot_h = sum_key dP_h P_h
#pragma unroll 1
        for (int t = 0; t < NT; ++t) {
            f32x4 s[1][H], da[1][H];
#pragma unroll
            for (int u = 0; u < 1; ++u)
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    s[u][h] = f32x4{0.f, 0.f, 0.f, 0.f}; da[u][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        s[u][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th_frag<HD>(tK, N, h, t * 16, ks, lane), qf[h][ks], s[u][h], 0, 0, 0);
                        da[u][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th_frag<HD>(tV, N, h, t * 16, ks, lane), dof[h][ks], da[u][h], 0, 0, 0);
                    }
                }
#pragma unroll
            for (int u = 0; u < 1; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool valid = t * 16 + 4 * grp + i < N;
                    float pr[H];
#pragma unroll
                    for (int g = 0; g < H; ++g) {
                        float a = blv[g];
#pragma unroll
                        for (int h = 0; h < H; ++h) a += wl[g][h] * (s[u][h][i] * p.scale);
                        pr[g] = valid ? __builtin_amdgcn_exp2f(a * TH_LOG2E - m2[g]) * zi[g] : 0.f;
                    }
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        float dp = 0.f;
#pragma unroll
                        for (int g = 0; g < H; ++g) dp += ww[g][h] * da[u][g][i];
                        dot[h] += dp * pr[h];
                    }
                }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            dot[h] += __shfl_xor(dot[h], 16, 64);
            dot[h] += __shfl_xor(dot[h], 32, 64);
        }
        // pass 2: recompute the tiles, emit dS and the parameter-gradient sums
#pragma unroll 1
        for (int t = 0; t < NT; ++t) {
            f32x4 s[1][H], da[1][H];
#pragma unroll
            for (int u = 0; u < 1; ++u)
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    s[u][h] = f32x4{0.f, 0.f, 0.f, 0.f}; da[u][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        s[u][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th_frag<HD>(tK, N, h, t * 16, ks, lane), qf[h][ks], s[u][h], 0, 0, 0);
                        da[u][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th_frag<HD>(tV, N, h, t * 16, ks, lane), dof[h][ks], da[u][h], 0, 0, 0);
                    }
                }
#pragma unroll
            for (int u = 0; u < 1; ++u) {
                const int key0 = t * 16 + 4 * grp;
                float dsv[H][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool valid = key0 + i < N;
                    float sr[H], pr[H], dav[H], dsp[H];
#pragma unroll
                    for (int h = 0; h < H; ++h) sr[h] = s[u][h][i] * p.scale;
#pragma unroll
                    for (int g = 0; g < H; ++g) {
                        float a = blv[g];
#pragma unroll
                        for (int h = 0; h < H; ++h) a += wl[g][h] * sr[h];
                        pr[g] = valid ? __builtin_amdgcn_exp2f(a * TH_LOG2E - m2[g]) * zi[g] : 0.f;
                        dav[g] = valid ? da[u][g][i] : 0.f;
                    }
#pragma unroll
                    for (int g = 0; g < H; ++g) {
                        acc[H * H + g] += dav[g];
#pragma unroll
                        for (int h = 0; h < H; ++h) acc[g * H + h] += dav[g] * pr[h];
                    }
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        float dp = 0.f;
#pragma unroll
                        for (int g = 0; g < H; ++g) dp += ww[g][h] * dav[g];
                        dsp[h] = pr[h] * (dp - dot[h]);
                        acc[H * H + H + h] += dsp[h];
                    }
#pragma unroll
                    for (int g = 0; g < H; ++g)
#pragma unroll
                        for (int h = 0; h < H; ++h) acc[H * H + 2 * H + g * H + h] += dsp[g] * sr[h];
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        float a = 0.f;
#pragma unroll
                        for (int g = 0; g < H; ++g) a += wl[g][h] * dsp[g];
                        dsv[h][i] = a;
                    }
                }
                if (qvalid && key0 < p.NPK) {
#pragma unroll
                    for (int h = 0; h < H; ++h)
                        *reinterpret_cast<uint2*>(p.ds16 + (((size_t)b * H + h) * N + q) * p.NPK + key0) =
                            make_uint2(pack_bf16x2(dsv[h][0], dsv[h][1]), pack_bf16x2(dsv[h][2], dsv[h][3]));
                }
            }
        }
    }
    // parameter-gradient partials: wave -> workgroup, one slot row per workgroup (summed in a fixed order by th_param_reduce_kernel)
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) red[wave * PW + i] = v;
    }
    __syncthreads();
    if (tid < PW) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < TH_WAVES; ++w) v += red[w * PW + tid];
        p.partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * PW + tid] = v;
    }
}

// The three per-head products behind th_bwd in ONE launch (they were three batched 128x128-tile GEMMs with 48 of 128 columns used):
//   dQ_h = scale dS_h K_h,   dK_h = scale dS_h^T Q_h,   dV_h = A_h^T dO_h        (written into the packed dqkv rows)
// One 8-wave workgroup per (sample, head).  K, Q and dO of the head sit in LDS as [token][head_dim] images (zero rows past N); dS and A
// stream through a double buffer of 32-query tiles by LDS-DMA (a tile of the [B][H][N][NPK] tensors is one contiguous block).  All
// operands that are contracted over their ROWS (K for dQ; Q, dO, dS, A for dK / dV) are read with ds_read_b64_tr_b16, every product is
// v_mfma_f32_16x16x32_bf16 with swapped operands (output index d in the rows): dK / dV accumulate in registers over the query tiles
// (wave w owns the key blocks w and w + 8), dQ of a tile is complete after one pass over the keys.
constexpr int TG_WAVES = 8, TG_NTHR = TG_WAVES * 64;
template <int HD>
__global__ __launch_bounds__(TG_NTHR, 1) void th_grads_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const bf16_t* __restrict__ ds16,
                                                              const bf16_t* __restrict__ a16, bf16_t* __restrict__ dqkv, int B, int H, int N, int D, int NPK,
                                                              float scale) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    constexpr int DB = HD / 16, IMG_ROWS = 224, IMG = IMG_ROWS * HD * 2, QT = 32, MAXKB = 2, CH = HD / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char gl[];
    unsigned char* iK = gl;
    unsigned char* iQ = gl + IMG;
    unsigned char* iO = gl + 2 * IMG;
    unsigned char* tiles = gl + 3 * IMG;                       // [2 buffers][dS tile | A tile], QT rows x NPK bf16 each
    const int tile_bytes = QT * NPK * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, l15 = lane & 15, s4 = l15 >> 2, c4 = 4 * (l15 & 3);
    const int h = blockIdx.x, b = blockIdx.y, ld = 3 * D;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(qkv + (size_t)b * N * ld + h * HD);
    const unsigned char* dob = reinterpret_cast<const unsigned char*>(dout + (size_t)b * N * D + h * HD);
    const unsigned char* gds = reinterpret_cast<const unsigned char*>(ds16 + ((size_t)b * H + h) * N * NPK);
    const unsigned char* ga = reinterpret_cast<const unsigned char*>(a16 + ((size_t)b * H + h) * N * NPK);
    const int nqt = (N + QT - 1) / QT, nkb = (N + 15) / 16, nks = (N + 31) / 32;
    // everything enters LDS by LDS-DMA (lane-linear destination, per-lane source).  A tile = QT rows of NPK bf16, contiguous in global
    // memory; rows past N repeat the last one (they meet zero rows of Q / dO).
    auto issue = [&](int qt, int buf) {
        const int chunks = tile_bytes / 16;                    // 16-byte pieces of one tile
        for (int c0 = wave * 64; c0 < chunks; c0 += TG_NTHR) {
            const int c = c0 + lane;
            if (c < chunks) {
                const int row = (c * 16) / (NPK * 2), colb = c * 16 - row * NPK * 2;
                const size_t src = (size_t)min(qt * QT + row, N - 1) * NPK * 2 + colb;
                // (the builtin, not lds_dma16_hidden: here the compiler's drain in front of the transposed reads measured FASTER in the step --
                //  cait_xxs24 10 092 vs 10 048 img/s same-box, round 6 -- the next tile pair is small and the wait keeps the eight waves together)
                __builtin_amdgcn_global_load_lds((gbl_void*)(gds + src), (lds_void*)(tiles + buf * 2 * tile_bytes + c0 * 16), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void*)(ga + src), (lds_void*)(tiles + buf * 2 * tile_bytes + tile_bytes + c0 * 16), 16, 0, 0);
            }
        }
    };
    // K, Q, dO images [IMG_ROWS][HD]: rows < N by DMA, rows past N zero-filled
    for (int c0 = wave * 64; c0 < N * CH; c0 += TG_NTHR) {
        const int c = c0 + lane;
        if (c < N * CH) {
            const int r = c / CH, cc = c - r * CH;
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + ((size_t)r * ld + D) * 2 + cc * 16), (lds_void*)(iK + c0 * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + (size_t)r * ld * 2 + cc * 16), (lds_void*)(iQ + c0 * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(dob + (size_t)r * D * 2 + cc * 16), (lds_void*)(iO + c0 * 16), 16, 0, 0);
        }
    }
    issue(0, 0);
    for (int i = N * CH + tid; i < IMG_ROWS * CH; i += TG_NTHR) {
        const uint4 z = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(iK + (size_t)i * 16) = z;
        *reinterpret_cast<uint4*>(iQ + (size_t)i * 16) = z;
        *reinterpret_cast<uint4*>(iO + (size_t)i * 16) = z;
    }
    // transposed fragment: 8 contraction slots = rows r0 + 4 grp + j (j < 4) and r0 + 16 + 4 grp + (j - 4), output index col0 + (lane & 15)
    auto tr = [&](const unsigned char* img, int pitch, int r0, int col0) -> bf16x8 {
        const unsigned char* a0 = img + (size_t)(r0 + 4 * grp + s4) * pitch + (col0 + c4) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0 + 16 * pitch));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    f32x4 dk[MAXKB][DB], dv[MAXKB][DB];
#pragma unroll
    for (int i = 0; i < MAXKB; ++i)
#pragma unroll
        for (int db = 0; db < DB; ++db) { dk[i][db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i][db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int qt = 0; qt < nqt; ++qt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // tile qt (and, first trip, the images) visible; every wave is past tile qt - 1
        if (qt + 1 < nqt) issue(qt + 1, (qt + 1) & 1);
        const unsigned char* tS = tiles + (qt & 1) * 2 * tile_bytes;
        const unsigned char* tA = tS + tile_bytes;
        // ---- dK, dV: contraction over the 32 queries of this tile; wave w owns the key blocks w and w + 8
        bf16x8 qT[DB], oT[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db) { qT[db] = tr(iQ, HD * 2, qt * QT, 16 * db); oT[db] = tr(iO, HD * 2, qt * QT, 16 * db); }
#pragma unroll
        for (int i = 0; i < MAXKB; ++i) {
            const int kb = wave + TG_WAVES * i;
            if (kb < nkb) {                                    // wave-uniform
                const bf16x8 sT = tr(tS, NPK * 2, 0, 16 * kb), aT = tr(tA, NPK * 2, 0, 16 * kb);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    dk[i][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT[db], sT, dk[i][db], 0, 0, 0);
                    dv[i][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oT[db], aT, dv[i][db], 0, 0, 0);
                }
            }
        }
        // ---- dQ of this tile: 2 query blocks x DB d-blocks output tiles, dealt to the waves from the top (waves 7, 6, ... : the waves with
        // one key block above take them first); contraction over all keys
        for (int ti = TG_WAVES - 1 - wave; ti < 2 * DB; ti += TG_WAVES) {
            const int qb = ti / DB, db = ti - qb * DB;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const unsigned char* srow = tS + (size_t)(16 * qb + l15) * NPK * 2;
            for (int ks = 0; ks < nks; ++ks) {
                const bf16x8 kT = tr(iK, HD * 2, 32 * ks, 16 * db);
                // second operand: this lane's query row, keys 32 ks + 4 grp + j and 32 ks + 16 + 4 grp + j (the slot order of tr)
                const int k0 = 32 * ks + 4 * grp;
                uint2 lo = make_uint2(0, 0), hi = lo;
                if (k0 < NPK) lo = *reinterpret_cast<const uint2*>(srow + k0 * 2);
                if (k0 + 16 < NPK) hi = *reinterpret_cast<const uint2*>(srow + (k0 + 16) * 2);
                typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
                const u32x4 u = {lo.x, lo.y, hi.x, hi.y};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT, __builtin_bit_cast(bf16x8, u), acc, 0, 0, 0);
            }
            const int q = qt * QT + 16 * qb + l15;
            if (q < N)
                *reinterpret_cast<uint2*>(dqkv + ((size_t)b * N + q) * ld + h * HD + 16 * db + 4 * grp) =
                    make_uint2(pack_bf16x2(acc[0] * scale, acc[1] * scale), pack_bf16x2(acc[2] * scale, acc[3] * scale));
        }
    }
#pragma unroll
    for (int i = 0; i < MAXKB; ++i) {
        const int key = 16 * (wave + TG_WAVES * i) + l15;
        if (wave + TG_WAVES * i < nkb && key < N) {
            bf16_t* row = dqkv + ((size_t)b * N + key) * ld + h * HD + 4 * grp;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                *reinterpret_cast<uint2*>(row + D + 16 * db) = make_uint2(pack_bf16x2(dk[i][db][0] * scale, dk[i][db][1] * scale), pack_bf16x2(dk[i][db][2] * scale, dk[i][db][3] * scale));
                *reinterpret_cast<uint2*>(row + 2 * D + 16 * db) = make_uint2(pack_bf16x2(dv[i][db][0], dv[i][db][1]), pack_bf16x2(dv[i][db][2], dv[i][db][3]));
            }
        }
    }
}

// out_j += sum over the workgroups' partial rows, fixed order (bit-reproducible): one workgroup per parameter-gradient element
__global__ __launch_bounds__(256) void th_param_reduce_kernel(const float* __restrict__ partial, int nparts, int PW, int H, float* dww, float* dbw,
                                                              float* dbl, float* dwl) {
    __shared__ float red[256];
    const int j = blockIdx.x, t = threadIdx.x;
    float a = 0.f;
    for (int i = t; i < nparts; i += 256) a += partial[(size_t)i * PW + j];
    red[t] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    if (t == 0) {
        float* dst = j < H * H ? dww + j : j < H * H + H ? dbw + (j - H * H) : j < H * H + 2 * H ? dbl + (j - H * H - H) : dwl + (j - H * H - 2 * H);
        *dst += red[0];
    }
}

// workgroups per sample of the fused kernels: 16 queries per wave, TH_WAVES waves
inline int th_groups(int N) { return ((N + 15) / 16 + TH_WAVES - 1) / TH_WAVES; }

template <typename F>
int dispatch_th(int hd, int H, const char* who, F&& f) {
#define PPF_TH_CASE(HDV, HV) if (hd == HDV && H == HV) return f(std::integral_constant<int, HDV>(), std::integral_constant<int, HV>());
    PPF_TH_CASE(48, 4) PPF_TH_CASE(48, 2) PPF_TH_CASE(64, 4) PPF_TH_CASE(64, 2) PPF_TH_CASE(32, 4) PPF_TH_CASE(32, 2)
#undef PPF_TH_CASE
    ppf_set_error("%s: unsupported (head_dim=%d, heads=%d): head_dim in {32,48,64}, heads in {2,4}", who, hd, H);
    return PPF_ERR_SHAPE;
}

}  // namespace

extern "C" {

// Mixed pre-softmax logits of the talking-heads attention (cait:119-123): sp[b][g][q][key] (NP = N rounded up to 4).
int ppf_th_scores(const void* qkv, const float* wl, const float* bl, float* sp, int B, int H, int N, int D, int NP, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && N > 0 && N <= 224 && D % H == 0 && NP % 4 == 0 && NP >= N, PPF_ERR_SHAPE, "ppf_th_scores: bad shape");
    ThParams p; p.qkv = (const bf16_t*)qkv; p.B = B; p.H = H; p.N = N; p.D = D; p.NP = NP; p.wl = wl; p.bl = bl; p.sp = sp; p.dwl = nullptr;
    p.scale = 1.0f / sqrtf((float)(D / H));
    return dispatch_th(D / H, H, "ppf_th_scores", [&](auto hd, auto hv) {
        hipLaunchKernelGGL((th_scores_kernel<decltype(hd)::value, decltype(hv)::value, false>), dim3((N + 63) / 64, B), dim3(512), 0, stream, p);
        PPF_LAUNCH_CHECK();
        return 0;
    });
}

// dWl[g][h] += sum_{b,q,key} dS'_g * (scale q_h.k_h)   (recomputes the raw scores on the MFMA)
int ppf_th_dwl(const void* qkv, const float* ds_prime, float* dwl, int B, int H, int N, int D, int NP, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && N > 0 && N <= 224 && D % H == 0 && NP % 4 == 0 && NP >= N, PPF_ERR_SHAPE, "ppf_th_dwl: bad shape");
    ThParams p; p.qkv = (const bf16_t*)qkv; p.B = B; p.H = H; p.N = N; p.D = D; p.NP = NP; p.wl = dwl; p.bl = dwl; p.sp = (float*)ds_prime; p.dwl = dwl;
    p.scale = 1.0f / sqrtf((float)(D / H));
    return dispatch_th(D / H, H, "ppf_th_dwl", [&](auto hd, auto hv) {
        hipLaunchKernelGGL((th_scores_kernel<decltype(hd)::value, decltype(hv)::value, true>), dim3((N + 63) / 64, B), dim3(512), 0, stream, p);
        PPF_LAUNCH_CHECK();
        return 0;
    });
}

// In place: sp <- softmax(sp) (P); a16 [B][H][N][NPK] bf16 = proj_w(P) (zero padded to NPK, a multiple of 8);
// hm [B][N][NP] = mean over heads of proj_w(P)  (cait:124-126, 228).
int ppf_th_softmax_mix(float* sp, void* a16, float* hm, const float* ww, const float* bw, int B, int H, int N, int NP, int NPK, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && N > 0 && NP % 4 == 0 && NPK % 8 == 0 && NPK >= NP && NP >= N && NPK <= 256, PPF_ERR_SHAPE, "ppf_th_softmax_mix: bad shape");
    ThSmParams p = ThSmParams(); p.sp = sp; p.a16 = (bf16_t*)a16; p.hm = hm; p.ww = ww; p.bw = bw; p.B = B; p.H = H; p.N = N; p.NP = NP; p.NPK = NPK;
    const int grid = (B * N + 3) / 4;
    if (H == 4) hipLaunchKernelGGL(th_softmax_mix_kernel<4>, dim3(grid), dim3(256), 0, stream, p);
    else if (H == 2) hipLaunchKernelGGL(th_softmax_mix_kernel<2>, dim3(grid), dim3(256), 0, stream, p);
    else { ppf_set_error("ppf_th_softmax_mix: heads must be 2 or 4"); return PPF_ERR_SHAPE; }
    PPF_LAUNCH_CHECK();
    return 0;
}

// Backward of the two head mixes and the softmax: da [B][H][N][NP] (in: dA, out: dS'), prob = P from the forward;
// ds16 = bf16(sum_g Wl[g][h] dS'_g) [B][H][N][NPK]; dww/dbw/dbl accumulate (+=).
int ppf_th_softmax_bwd(const float* prob, float* da, void* ds16, const float* ww, const float* wl, float* dww, float* dbw, float* dbl, int B,
                       int H, int N, int NP, int NPK, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && N > 0 && NP % 4 == 0 && NPK % 8 == 0 && NPK >= NP && NP >= N && NPK <= 256, PPF_ERR_SHAPE, "ppf_th_softmax_bwd: bad shape");
    ThSmParams p = ThSmParams(); p.sp = (float*)prob; p.da = da; p.ds_prime = da; p.ds16 = (bf16_t*)ds16; p.ww = ww; p.wl = wl; p.dww = dww; p.dbw = dbw;
    p.dbl = dbl; p.B = B; p.H = H; p.N = N; p.NP = NP; p.NPK = NPK;
    const int grid = (B * N + 3) / 4;
    if (H == 4) hipLaunchKernelGGL(th_softmax_bwd_kernel<4>, dim3(grid), dim3(256), 0, stream, p);
    else if (H == 2) hipLaunchKernelGGL(th_softmax_bwd_kernel<2>, dim3(grid), dim3(256), 0, stream, p);
    else { ppf_set_error("ppf_th_softmax_bwd: heads must be 2 or 4"); return PPF_ERR_SHAPE; }
    PPF_LAUNCH_CHECK();
    return 0;
}

// Fused talking-heads attention, forward part up to the mixed probabilities (cait:119-126): from packed qkv [B*N][3D] computes
// A = proj_w(softmax(proj_l(scale q k^T))) as bf16 a16 [B][H][N][NPK] (zero padded; the operand of the A.V product), its head mean
// hm [B][N][NP] (rollout input, cait:228), the softmax statistics rowmax / zinv [B][H][N] the backward kernel recomputes P from and, when
// out != NULL, the attention output O = A V as bf16 [B*N][D] (cait:128) from the same launch.
// Returns PPF_ERR_SHAPE for (head_dim, heads, N) combinations it does not cover (see ppf_th_fused_supported).
int ppf_th_fused_supported(int H, int N, int D) {
    if (H <= 0 || D % H != 0) return 0;
    const int hd = D / H;
    if (!((hd == 32 || hd == 48 || hd == 64) && (H == 2 || H == 4))) return 0;
    if (N <= 0 || N > 224) return 0;
    return (size_t)2 * H * N * hd * 2 + TH_WAVES * (2 * H * H + 2 * H) * sizeof(float) <= 160 * 1024 ? 1 : 0;
}
int ppf_th_fwd(const void* qkv, const float* wl, const float* bl, const float* ww, const float* bw, void* a16, float* hm, float* rowmax, float* zinv,
               void* out, int B, int H, int N, int D, int NP, int NPK, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && ppf_th_fused_supported(H, N, D), PPF_ERR_SHAPE, "ppf_th_fwd: unsupported shape (B=%d H=%d N=%d D=%d)", B, H, N, D);
    PPF_CHECK_ARG(NP % 4 == 0 && NPK % 8 == 0 && NPK >= NP && NP >= N && NPK <= 256, PPF_ERR_SHAPE, "ppf_th_fwd: bad padding");
    ThFusedParams p = ThFusedParams(); p.qkv = (const bf16_t*)qkv; p.wl = wl; p.bl = bl; p.ww = ww; p.bw = bw; p.a16 = (bf16_t*)a16; p.hm = hm;
    p.rowmax = rowmax; p.zinv = zinv; p.out = (bf16_t*)out; p.B = B; p.N = N; p.D = D; p.NP = NP; p.NPK = NPK; p.scale = 1.0f / sqrtf((float)(D / H));
    return dispatch_th(D / H, H, "ppf_th_fwd", [&](auto hd, auto hv) {
        constexpr int HD = decltype(hd)::value, HV = decltype(hv)::value;
        const size_t lds = (size_t)(out ? 2 : 1) * HV * N * HD * 2;
        auto k = th_fwd_kernel<HD, HV>;
        static bool attr_set = false;               // one per (HD, HV) instantiation of this generic lambda
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(talking heads): %s", hipGetErrorString(e)); return (int)e; }
            attr_set = true;
        }
        hipLaunchKernelGGL(k, dim3(th_groups(N), B), dim3(TH_NTHR), lds, stream, p);
        PPF_LAUNCH_CHECK();
        return 0;
    });
}
// Backward of the same: dout = dO bf16 [B*N][D]; writes ds16 = bf16 dS [B][H][N][NPK] (operand of the dQ / dK products) and one row of
// parameter-gradient partials per workgroup into `partial` (>= ppf_th_bwd_partial_floats floats), then adds their fixed-order sums to
// dww [H][H], dbw [H], dbl [H], dwl [H][H] (reduce_stream: the stream of that small reduction; it is ordered behind `stream` by the
// caller when they differ -- pass the same stream for plain in-order use).
size_t ppf_th_bwd_partial_floats(int B, int H, int N) { return (size_t)B * th_groups(N) * (2 * H * H + 2 * H); }
int ppf_th_bwd(const void* qkv, const void* dout, const float* wl, const float* bl, const float* ww, const float* rowmax, const float* zinv, void* ds16,
               float* partial, int B, int H, int N, int D, int NPK, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && ppf_th_fused_supported(H, N, D), PPF_ERR_SHAPE, "ppf_th_bwd: unsupported shape (B=%d H=%d N=%d D=%d)", B, H, N, D);
    PPF_CHECK_ARG(NPK % 8 == 0 && NPK >= N && NPK <= 256, PPF_ERR_SHAPE, "ppf_th_bwd: bad padding");
    ThFusedParams p = ThFusedParams(); p.qkv = (const bf16_t*)qkv; p.dout = (const bf16_t*)dout; p.wl = wl; p.bl = bl; p.ww = ww; p.rowmax = (float*)rowmax;
    p.zinv = (float*)zinv; p.ds16 = (bf16_t*)ds16; p.partial = partial; p.B = B; p.N = N; p.D = D; p.NPK = NPK; p.scale = 1.0f / sqrtf((float)(D / H));
    return dispatch_th(D / H, H, "ppf_th_bwd", [&](auto hd, auto hv) {
        constexpr int HD = decltype(hd)::value, HV = decltype(hv)::value;
        const size_t lds = (size_t)2 * HV * N * HD * 2 + TH_WAVES * (2 * HV * HV + 2 * HV) * sizeof(float);
        auto k = th_bwd_kernel<HD, HV>;
        static bool attr_set = false;               // one per (HD, HV) instantiation of this generic lambda
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(talking heads): %s", hipGetErrorString(e)); return (int)e; }
            attr_set = true;
        }
        hipLaunchKernelGGL(k, dim3(th_groups(N), B), dim3(TH_NTHR), lds, stream, p);
        PPF_LAUNCH_CHECK();
        return 0;
    });
}
// dqkv [B*N][3D] (bf16, every column written) from ds16 / a16 [B][H][N][NPK] (th_bwd's dS, th_fwd's A), packed qkv and dout = dO [B*N][D]:
// dQ_h = scale dS_h K_h, dK_h = scale dS_h^T Q_h, dV_h = A_h^T dO_h -- the three products behind ppf_th_bwd in one launch.
// Needs N <= 208 (13 key blocks on 8 waves x 2), head_dim in {32, 48, 64} and the images + tile buffers in 160 KiB (ppf_th_grads_supported).
int ppf_th_grads_supported(int H, int N, int D) {
    if (H <= 0 || D % H != 0) return 0;
    const int hd = D / H, NPK = (N + 7) / 8 * 8;
    if (!(hd == 32 || hd == 48 || hd == 64) || N <= 0 || N > 208) return 0;
    return (size_t)3 * 224 * hd * 2 + (size_t)4 * 32 * NPK * 2 + 64 <= 160 * 1024 ? 1 : 0;
}
int ppf_th_grads(const void* qkv, const void* dout, const void* ds16, const void* a16, void* dqkv, int B, int H, int N, int D, int NPK, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && ppf_th_grads_supported(H, N, D) && NPK == (N + 7) / 8 * 8, PPF_ERR_SHAPE, "ppf_th_grads: unsupported shape (B=%d H=%d N=%d D=%d NPK=%d)", B, H, N, D, NPK);
    PPF_CHECK_ARG(qkv && dout && ds16 && a16 && dqkv, PPF_ERR_ARG, "ppf_th_grads: null pointer");
    const int hd = D / H;
    const size_t lds = (size_t)3 * 224 * hd * 2 + (size_t)4 * 32 * NPK * 2 + 64;      // + slack: the last transposed read of a tile overhangs its row by 16 bytes
    const float scale = 1.0f / sqrtf((float)hd);
#define PPF_TH_GRADS(HDV)                                                                                                                     \
    {                                                                                                                                         \
        auto k = th_grads_kernel<HDV>;                                                                                                        \
        static bool attr_set = false;                                                                                                         \
        if (!attr_set) {                                                                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);    \
            if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(th_grads): %s", hipGetErrorString(e)); return (int)e; }                 \
            attr_set = true;                                                                                                                  \
        }                                                                                                                                     \
        hipLaunchKernelGGL(k, dim3(H, B), dim3(TG_NTHR), lds, stream, (const bf16_t*)qkv, (const bf16_t*)dout, (const bf16_t*)ds16,               \
                           (const bf16_t*)a16, (bf16_t*)dqkv, B, H, N, D, NPK, scale);                                                        \
    }
    if (hd == 48) PPF_TH_GRADS(48) else if (hd == 64) PPF_TH_GRADS(64) else PPF_TH_GRADS(32)
#undef PPF_TH_GRADS
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_th_param_reduce(const float* partial, int B, int H, int N, float* dww, float* dbw, float* dbl, float* dwl, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && (H == 2 || H == 4) && N > 0, PPF_ERR_SHAPE, "ppf_th_param_reduce: bad shape");
    const int PW = 2 * H * H + 2 * H;
    hipLaunchKernelGGL(th_param_reduce_kernel, dim3(PW), dim3(256), 0, stream, partial, B * th_groups(N), PW, H, dww, dbw, dbl, dwl);
    PPF_LAUNCH_CHECK();
    return 0;
}

// Class attention (cait:71-90, policy softmax cait:50-69): q [B][D] (cls rows), k/v [B*N1][D]; out [B][D],
// attn [B][H][N1], zinv [B][H], rowmean [B][N1] = mean over heads (the rollout's class-attention row).
int ppf_class_attn_fwd(const void* q, const void* k, const void* v, const float* policy, float* attn, float* zinv, float* rowmean, void* out,
                       int B, int H, int N1, int D, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && H >= 1 && H <= 4 && N1 <= 256 && D % H == 0 && (D / H) <= 64 && (D / H) % 2 == 0, PPF_ERR_SHAPE, "ppf_class_attn_fwd: bad shape");
    CaParams p = CaParams(); p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.policy = policy; p.attn = attn; p.zinv = zinv;
    p.rowmean = rowmean; p.out = (bf16_t*)out; p.B = B; p.H = H; p.N1 = N1; p.D = D; p.scale = 1.0f / sqrtf((float)(D / H));
    hipLaunchKernelGGL(class_attn_fwd_kernel, dim3(B), dim3(256), 0, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_class_attn_bwd(const void* q, const void* k, const void* v, const float* attn, const float* zinv, const void* dout, void* dq, void* dk,
                       void* dv, int B, int H, int N1, int D, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && H >= 1 && H <= 4 && N1 <= 256 && D % H == 0 && (D / H) <= 64 && (D / H) % 2 == 0, PPF_ERR_SHAPE, "ppf_class_attn_bwd: bad shape");
    CaParams p = CaParams(); p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.attn = (float*)attn; p.zinv = (float*)zinv;
    p.dout = (const bf16_t*)dout; p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.B = B; p.H = H; p.N1 = N1; p.D = D;
    p.scale = 1.0f / sqrtf((float)(D / H));
    hipLaunchKernelGGL(class_attn_bwd_kernel, dim3(B), dim3(256), 0, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

// out bf16 [rows][D] = a + b (+ cq[row / N1] on rows that are a multiple of N1)
int ppf_merge3_cast(const float* a, const float* b, const float* cq, void* out, int rows, int D, int N1, hipStream_t stream) {
    PPF_CHECK_ARG(rows > 0 && D % 2 == 0 && N1 > 0, PPF_ERR_SHAPE, "ppf_merge3_cast: bad shape");
    const size_t total = (size_t)rows * D / 2;
    hipLaunchKernelGGL(merge3_cast_kernel, dim3((int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256), 0, stream, a, b, cq, (bf16_t*)out, rows, D, N1);
    PPF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
