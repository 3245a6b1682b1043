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
