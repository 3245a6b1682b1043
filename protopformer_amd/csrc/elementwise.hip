// HBM-bound streaming kernels: fp32->bf16 parameter cast, patch im2col, token assembly (cls/pos) and its
// backward, fused AdamW(+EMA+bf16 recast).  All vectorised 16 B per lane, grid-stride.
#include "ppf_common.h"

namespace {

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(in)[2 * i], b = reinterpret_cast<const float4*>(in)[2 * i + 1];
        reinterpret_cast<uint4*>(out)[i] = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
    }
}

// the inverse widening (exact): the bf16 wire format of the gradient exchange back into the fp32 gradient buffer
__global__ __launch_bounds__(256) void widen_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const uint4 u = reinterpret_cast<const uint4*>(in)[i];
        const float2 a = unpack_bf16x2(u.x), b = unpack_bf16x2(u.y), c = unpack_bf16x2(u.z), d = unpack_bf16x2(u.w);
        reinterpret_cast<float4*>(out)[2 * i] = make_float4(a.x, a.y, b.x, b.y);
        reinterpret_cast<float4*>(out)[2 * i + 1] = make_float4(c.x, c.y, d.x, d.y);
    }
}

// img [B][C][H][W] fp32 -> cols [B*gh*gw][C*p*p] bf16, column order (c, py, px) = Conv2d weight.reshape(D,-1) order.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ cols, int B, int C,
                                                     int H, int W, int p, int64_t total8) {
    const int gh = H / p, gw = W / p, pp8 = p / 8, kdim = C * p * p;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
        int64_t t = i;
        const int px8 = t % pp8; t /= pp8;
        const int py = t % p; t /= p;
        const int c = t % C; t /= C;
        const int gx = t % gw; t /= gw;
        const int gy = t % gh; t /= gh;
        const int b = (int)t;
        const float* src = img + (((size_t)b * C + c) * H + (gy * p + py)) * W + gx * p + px8 * 8;
        const float4 a = reinterpret_cast<const float4*>(src)[0], bq = reinterpret_cast<const float4*>(src)[1];
        bf16_t* dst = cols + ((size_t)(b * gh + gy) * gw + gx) * kdim + (c * p + py) * p + px8 * 8;
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(bq.x, bq.y), pack_bf16x2(bq.z, bq.w));
    }
}

// x[b][t][:] = (t < lead ? cls : tok[b][t-lead]) + pos[t]      (lead = 1: DeiT cls row; lead = 0: CaiT)
__global__ __launch_bounds__(256) void assemble_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, float* __restrict__ x, int B, int Np,
                                                       int D, int lead) {
    const int T = Np + lead, d4 = D / 4;
    const int64_t total = (int64_t)B * T * d4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = i % d4;
        const int t = (i / d4) % T;
        const int b = i / ((int64_t)d4 * T);
        float4 v = (t < lead) ? reinterpret_cast<const float4*>(cls)[c]
                              : reinterpret_cast<const float4*>(tok + ((size_t)b * Np + (t - lead)) * D)[c];
        const float4 pe = reinterpret_cast<const float4*>(pos + (size_t)t * D)[c];
        reinterpret_cast<float4*>(x)[i] = make_float4(v.x + pe.x, v.y + pe.y, v.z + pe.z, v.w + pe.w);
    }
}

// dx [B][T][D] -> dtok bf16 [B*Np][D] (rows t >= lead), dpos[t] += sum_b dx[b][t], dcls += sum_b dx[b][0] (lead = 1).
// A workgroup owns 64 column pairs of one token; its four waves walk a quarter of the batch each (8 loads in flight per lane)
// and the quarters are added in a fixed order through LDS: a single writer per output, no float atomics -> bit-identical
// from run to run.
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const float* __restrict__ dx, bf16_t* __restrict__ dtok,
                                                           float* __restrict__ dpos, float* __restrict__ dcls, int B, int Np,
                                                           int D, int lead) {
    __shared__ float2 part[4][64];
    const int T = Np + lead, d2 = D / 2, cblocks = (d2 + 63) / 64;
    const int t = blockIdx.x / cblocks, cp = (blockIdx.x % cblocks) * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int c = cp * 2;
    const int bq = (B + 3) / 4, b0 = q * bq, b1 = min(B, b0 + bq);
    float2 acc = make_float2(0.f, 0.f);
    if (cp < d2) {
#pragma unroll 8
        for (int b = b0; b < b1; ++b) {
            const float2 v = *reinterpret_cast<const float2*>(dx + ((size_t)b * T + t) * D + c);
            acc.x += v.x; acc.y += v.y;
            if (t >= lead) *reinterpret_cast<uint32_t*>(dtok + ((size_t)b * Np + (t - lead)) * D + c) = pack_bf16x2(v.x, v.y);
        }
    }
    part[q][threadIdx.x & 63] = acc;
    __syncthreads();
    if (q == 0 && cp < d2) {
        const float2 a0 = part[0][threadIdx.x], a1 = part[1][threadIdx.x], a2 = part[2][threadIdx.x], a3 = part[3][threadIdx.x];
        const float sx = (a0.x + a1.x) + (a2.x + a3.x), sy = (a0.y + a1.y) + (a2.y + a3.y);
        float* dp = dpos + (size_t)t * D + c;
        dp[0] += sx; dp[1] += sy;
        if (t < lead && dcls) { dcls[c] += sx; dcls[c + 1] += sy; }
    }
}

// Fused AdamW (decoupled weight decay, torch.optim.AdamW semantics) over a flat parameter buffer split in
// segments with their own lr / weight decay; optionally updates an EMA copy and re-emits the bf16 weights.
struct Seg { int64_t begin, end; float lr, wd; };
constexpr int MAX_SEGS = 8;
struct AdamParams {
    float* p; const float* g; float* m; float* v; float* ema; bf16_t* p16;
    int64_t n; int nseg; Seg seg[MAX_SEGS];
    float beta1, beta2, eps, bc1, bc2_sqrt, ema_decay, grad_scale;
    const float* hyper;          // device-resident step state (PPF_HYPER_* layout) or NULL: then the scalars above are used
    const float* guard;          // the step's loss (device scalar) or NULL: a non-finite loss skips the whole update ...
    int* nonfinite;              // ... and raises this flag (engine_proto.py:66-70 exits BEFORE optimizer.step() on such a loss)
};
// hyper[0..7] lr per segment, [8..15] weight decay per segment, [16] 1-beta1^t, [17] sqrt(1-beta2^t), [18] gradient scale
// (1/world), [19] clip factor written by clip_finish_kernel (1 when clipping is off).  Reading them from memory lets a
// captured HIP graph of the step be replayed with a new step count / learning rate (the host refreshes hyper before a replay).
__global__ __launch_bounds__(256) void adamw_kernel(AdamParams a) {
    if (a.guard) {
        const float loss = *a.guard;
        if (!(fabsf(loss) <= 3.4028234e38f)) {              // NaN or +-inf: parameters, moments, EMA and the bf16 shadow stay as they are
            if (blockIdx.x == 0 && threadIdx.x == 0) *a.nonfinite = 1;
            return;
        }
    }
    if (a.hyper) {
#pragma unroll
        for (int s = 0; s < MAX_SEGS; ++s) { a.seg[s].lr = a.hyper[s]; a.seg[s].wd = a.hyper[8 + s]; }
        a.bc1 = a.hyper[16]; a.bc2_sqrt = a.hyper[17]; a.grad_scale = a.hyper[18] * a.hyper[19];
    }
    const int64_t n4 = a.n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t e0 = i * 4;
        float lr = 0.f, wd = 0.f;
#pragma unroll
        for (int s = 0; s < MAX_SEGS; ++s)
            if (s < a.nseg && e0 >= a.seg[s].begin && e0 < a.seg[s].end) { lr = a.seg[s].lr; wd = a.seg[s].wd; }
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        const float4 g4 = reinterpret_cast<const float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i], v = reinterpret_cast<float4*>(a.v)[i];
        float pv[4] = {p.x, p.y, p.z, p.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w}, mv[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g = gv[k] * a.grad_scale;
            pv[k] *= (1.0f - lr * wd);
            mv[k] = a.beta1 * mv[k] + (1.0f - a.beta1) * g;
            vv[k] = a.beta2 * vv[k] + (1.0f - a.beta2) * g * g;
            const float denom = sqrtf(vv[k]) / a.bc2_sqrt + a.eps;
            pv[k] -= (lr / a.bc1) * (mv[k] / denom);
        }
        reinterpret_cast<float4*>(a.p)[i] = make_float4(pv[0], pv[1], pv[2], pv[3]);
        reinterpret_cast<float4*>(a.m)[i] = make_float4(mv[0], mv[1], mv[2], mv[3]);
        reinterpret_cast<float4*>(a.v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (a.ema) {
            float4 e = reinterpret_cast<float4*>(a.ema)[i];
            const float d = a.ema_decay, o = 1.0f - a.ema_decay;
            reinterpret_cast<float4*>(a.ema)[i] = make_float4(e.x * d + o * pv[0], e.y * d + o * pv[1], e.z * d + o * pv[2], e.w * d + o * pv[3]);
        }
        if (a.p16) reinterpret_cast<uint2*>(a.p16)[i] = make_uint2(pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]));
    }
}

// dz = bf16(df * f * (1 - f)) for f = sigmoid(z); dbias[c] += column sums of the rounded dz (bias grad of the add-on layer).
// partial != NULL: each workgroup writes its column sums to partial[block][cols] and sigmoid_bias_reduce_kernel adds them in a
// fixed order (deterministic); partial == NULL: fp32 atomics.
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ df, const float* __restrict__ f, bf16_t* __restrict__ dz,
                                                          float* __restrict__ dbias, float* __restrict__ partial, int rows, int cols,
                                                          int rows_per_block) {
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int c = threadIdx.x; c < cols; c += 256) {
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) {
            const size_t o = (size_t)r * cols + c;
            const float fv = f[o];
            const bf16_t h = f32_to_bf16(df[o] * fv * (1.0f - fv));
            dz[o] = h;
            acc += bf16_to_f32(h);
        }
        if (partial) partial[(size_t)blockIdx.x * cols + c] = acc;
        else if (dbias) unsafeAtomicAdd(dbias + c, acc);
    }
}
// 64 columns per workgroup, 16 groups of lanes each summing every 16th partial (independent loads, in flight together), then the 16
// group sums in a fixed order through LDS
__global__ __launch_bounds__(1024) void sigmoid_bias_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dbias, int nblk, int cols) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < cols) {
#pragma unroll 8
        for (int b = g; b < nblk; b += 16) acc += partial[(size_t)b * cols + c];
    }
    red[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        dbias[c] += t;
    }
}

// ---- global gradient-norm clipping (timm NativeScaler / dispatch_clip_grad, engine_proto.py:74-76): two deterministic passes
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n4, float* __restrict__ partial) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// hyper[19] = min(1, max_norm / (pre_scale * ||g|| + 1e-6))   (torch.nn.utils.clip_grad_norm_'s coefficient)
__global__ __launch_bounds__(256) void clip_finish_kernel(const float* __restrict__ partial, int nblk, float max_norm, float pre_scale,
                                                          float* __restrict__ hyper, float* __restrict__ norm_out) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float nrm = pre_scale * sqrtf(red[0]);
        hyper[19] = fminf(1.0f, max_norm / (nrm + 1e-6f));
        if (norm_out) *norm_out = nrm;
    }
}

// ---- DropPath factors floor(keep + U) / keep (timm DropPath, deit:71,79-80) for every (slot, sample) of a step.
// U comes from Philox4x32-10 keyed by `seed`, counter = (state[0] = step number, element index): one single-workgroup launch
// per step that also advances the step number in device memory, so a captured graph replays fresh draws.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        c = make_uint4((uint32_t)(p1 >> 32) ^ c.y ^ k.x, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k.y, (uint32_t)p0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}
__global__ __launch_bounds__(256) void droppath_kernel(float* __restrict__ out, const float* __restrict__ keep, int nslot, int B,
                                                       uint64_t seed, unsigned long long* __restrict__ state) {
    const unsigned long long step = state[0];
    const int total = nslot * B;
    for (int i4 = threadIdx.x; i4 * 4 < total; i4 += 256) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)i4, 0u, (uint32_t)step, (uint32_t)(step >> 32)),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const uint32_t rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i4 * 4 + j;
            if (i < total) {
                const float kp = keep[i / B];
                const float u = (float)(rv[j] >> 8) * (1.0f / 16777216.0f);          // [0, 1) on a 24-bit grid
                out[i] = floorf(kp + u) / kp;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) state[0] = step + 1;
}

// ---- row gather / scatter of the token reservation (integer index plumbing of protopformer.py:156-162)
// rows[b*(1+k)] = b*N ; rows[b*(1+k)+1+j] = b*N + 1 + idx[b][j]
__global__ __launch_bounds__(256) void rows_map_kernel(const int* __restrict__ idx, int* __restrict__ rows, int B, int k, int N) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * (k + 1)) return;
    const int b = i / (k + 1), j = i % (k + 1);
    rows[i] = b * N + (j == 0 ? 0 : 1 + idx[b * k + j - 1]);
}
// dst[r] = src[rows[r]] (gather) or dst[rows[r]] = src[r] (scatter); rows of `vec` 16-byte words
__global__ __launch_bounds__(256) void move_rows_kernel(const uint4* __restrict__ src, const int* __restrict__ rows, uint4* __restrict__ dst,
                                                        int nrows, int vec, int scatter) {
    const int64_t total = (int64_t)nrows * vec;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / vec), c = (int)(i % vec);
        const int64_t o = (int64_t)rows[r] * vec + c;
        if (scatter) dst[o] = src[i]; else dst[i] = src[o];
    }
}
// out = x * (*scalar)   (chain rule with a device-resident upstream scalar)
__global__ __launch_bounds__(256) void scale_by_kernel(const float* __restrict__ x, const float* __restrict__ scalar, float* __restrict__ out, int64_t n) {
    const float s = *scalar;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = x[i] * s;
}

// ---- input pipeline finisher (tools/datasets.py:280-336 ToTensor + Normalize, timm RandomErasing mode 'pixel'):
// uint8 HWC frames (as the CPU decode / augmentation workers leave them) -> fp32 NCHW, (x/255 - mean) / std per channel; inside the
// sample's erase rectangle (y, x, h, w; h == 0: none) every value is replaced by N(0,1) noise (Philox4x32-10 + Box-Muller, keyed
// by seed / step counter / sample / pixel).  One pass, 4 pixels per lane: the fp32 batch is written exactly once.
struct FinishParams { const unsigned char* in; float* out; const int* rects; int B, H, W; float mean[3], inv_std[3]; uint64_t seed; const unsigned long long* state; };
__global__ __launch_bounds__(256) void image_finish_kernel(const FinishParams p) {
    const int64_t hw = (int64_t)p.H * p.W, total4 = (int64_t)p.B * hw / 4;
    const unsigned long long step = p.state ? p.state[0] : 0ull;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int b = (int)((i * 4) / hw);
        const int64_t pix = i * 4 - (int64_t)b * hw;                  // 4 consecutive pixels of one row (W % 4 == 0)
        const int y = (int)(pix / p.W), x = (int)(pix % p.W);
        const unsigned char* src = p.in + ((int64_t)b * hw + pix) * 3;
        const uint32_t w0 = reinterpret_cast<const uint32_t*>(src)[0], w1 = reinterpret_cast<const uint32_t*>(src)[1], w2 = reinterpret_cast<const uint32_t*>(src)[2];
        const unsigned char px[12] = {(unsigned char)w0, (unsigned char)(w0 >> 8), (unsigned char)(w0 >> 16), (unsigned char)(w0 >> 24),
                                      (unsigned char)w1, (unsigned char)(w1 >> 8), (unsigned char)(w1 >> 16), (unsigned char)(w1 >> 24),
                                      (unsigned char)w2, (unsigned char)(w2 >> 8), (unsigned char)(w2 >> 16), (unsigned char)(w2 >> 24)};
        int ry = 0, rx = 0, rh = 0, rw = 0;
        if (p.rects) { ry = p.rects[4 * b]; rx = p.rects[4 * b + 1]; rh = p.rects[4 * b + 2]; rw = p.rects[4 * b + 3]; }
        const bool row_in = rh > 0 && y >= ry && y < ry + rh;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((float)px[3 * j + c] * (1.0f / 255.0f) - p.mean[c]) * p.inv_std[c];
            if (row_in && x + 3 >= rx && x < rx + rw) {
                const uint4 r = philox4x32_10(make_uint4((uint32_t)(i & 0xffffffff), (uint32_t)(i >> 32) * 3u + c, (uint32_t)step, (uint32_t)(step >> 32)),
                                              make_uint2((uint32_t)p.seed, (uint32_t)(p.seed >> 32)));
                const float u0 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = (float)(r.y >> 8) * (1.0f / 16777216.0f);
                const float u2 = ((float)(r.z >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = (float)(r.w >> 8) * (1.0f / 16777216.0f);
                const float m0 = sqrtf(-2.0f * __logf(u0)), m1 = sqrtf(-2.0f * __logf(u2));
                const float n[4] = {m0 * __cosf(6.283185307f * u1), m0 * __sinf(6.283185307f * u1), m1 * __cosf(6.283185307f * u3), m1 * __sinf(6.283185307f * u3)};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (x + j >= rx && x + j < rx + rw) v[j] = n[j];
            }
            *reinterpret_cast<float4*>(p.out + ((int64_t)b * 3 + c) * hw + pix) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

struct HyperVals { float v[20]; int n; };
__global__ void hyper_set_kernel(float* __restrict__ hyper, const HyperVals h) {
    if ((int)threadIdx.x < h.n) hyper[threadIdx.x] = h.v[threadIdx.x];
}

__global__ __launch_bounds__(256) void copy_2d_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int rows, int w16, long long sp16, long long dp16) {
    const long long total = (long long)rows * w16;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / w16, c = i - r * w16;
        dst[r * dp16 + c] = src[r * sp16 + c];
    }
}

inline int grid_for(int64_t work, int per_block = 256) {
    int64_t g = (work + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// dst[c][r] = src[r][c] for n bf16 matrices in one launch: desc[i] = (src element offset, dst element offset, rows, cols); a workgroup
// transposes one 64x64 tile through LDS (coalesced 128-byte rows on both sides).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, const long long* __restrict__ desc, int n) {
    __shared__ bf16_t tile[64][66];
    int b = blockIdx.x, i = 0;
    for (; i < n; ++i) {
        const int tiles = (int)((desc[4 * i + 2] + 63) / 64) * (int)((desc[4 * i + 3] + 63) / 64);
        if (b < tiles) break;
        b -= tiles;
    }
    if (i == n) return;
    const bf16_t* s = src + desc[4 * i];
    bf16_t* d = dst + desc[4 * i + 1];
    const int rows = (int)desc[4 * i + 2], cols = (int)desc[4 * i + 3];
    const int tc = (cols + 63) / 64, r0 = (b / tc) * 64, c0 = (b % tc) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = r0 + ty + 4 * k, c = c0 + tx;
        tile[ty + 4 * k][tx] = (r < rows && c < cols) ? s[(size_t)r * cols + c] : (bf16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + ty + 4 * k, r = r0 + tx;
        if (c < cols && r < rows) d[(size_t)c * rows + r] = tile[tx][ty + 4 * k];
    }
}

}  // namespace

extern "C" {

int ppf_cast_f32_bf16(const float* in, void* out, int64_t n, hipStream_t stream) {
    PPF_CHECK_ARG(n > 0 && (n % 8) == 0, PPF_ERR_SHAPE, "ppf_cast_f32_bf16: n=%lld must be a positive multiple of 8", (long long)n);
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, in, (bf16_t*)out, n / 8);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_cast_bf16_f32(const void* in, float* out, int64_t n, hipStream_t stream) {
    PPF_CHECK_ARG(in && out && n > 0 && (n % 8) == 0, PPF_ERR_SHAPE, "ppf_cast_bf16_f32: n=%lld must be a positive multiple of 8", (long long)n);
    hipLaunchKernelGGL(widen_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16_t*)in, out, n / 8);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_im2col_patch(const float* img, void* cols, int B, int C, int H, int W, int patch, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && C > 0 && patch % 8 == 0 && H % patch == 0 && W % patch == 0, PPF_ERR_SHAPE,
                  "ppf_im2col_patch: bad shape B=%d C=%d H=%d W=%d patch=%d", B, C, H, W, patch);
    const int64_t total8 = (int64_t)B * C * H * W / 8;
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total8)), dim3(256), 0, stream, img, (bf16_t*)cols, B, C, H, W, patch, total8);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_assemble_tokens(const float* tok, const float* cls, const float* pos, float* x, int B, int Np, int D, int lead, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && Np > 0 && D % 4 == 0 && (lead == 0 || lead == 1), PPF_ERR_SHAPE, "ppf_assemble_tokens: bad shape");
    hipLaunchKernelGGL(assemble_kernel, dim3(grid_for((int64_t)B * (Np + lead) * D / 4)), dim3(256), 0, stream, tok, cls, pos, x, B, Np, D, lead);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_assemble_tokens_bwd(const float* dx, void* dtok, float* dpos, float* dcls, int B, int Np, int D, int lead, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && Np > 0 && D % 2 == 0 && (lead == 0 || lead == 1), PPF_ERR_SHAPE, "ppf_assemble_tokens_bwd: bad shape");
    const int T = Np + lead, cblocks = (D / 2 + 63) / 64;
    hipLaunchKernelGGL(assemble_bwd_kernel, dim3(T * cblocks), dim3(256), 0, stream, dx, (bf16_t*)dtok, dpos, dcls, B, Np, D, lead);
    PPF_LAUNCH_CHECK();
    return 0;
}

static inline int sigmoid_rows_per_block(int rows) { int rpb = 16; while ((rows + rpb - 1) / rpb > 2048) rpb *= 2; return rpb; }
int ppf_sigmoid_bwd_blocks(int rows) { const int rpb = sigmoid_rows_per_block(rows); return (rows + rpb - 1) / rpb; }

int ppf_sigmoid_bwd(const float* df, const float* f, void* dz, float* dbias, int rows, int cols, float* partial, size_t partial_bytes,
                    hipStream_t stream) {
    PPF_CHECK_ARG(rows > 0 && cols > 0, PPF_ERR_SHAPE, "ppf_sigmoid_bwd: bad shape");
    const int rpb = sigmoid_rows_per_block(rows), nblk = (rows + rpb - 1) / rpb;
    PPF_CHECK_ARG(partial == nullptr || partial_bytes >= (size_t)nblk * cols * sizeof(float), PPF_ERR_ARG, "ppf_sigmoid_bwd: partial buffer too small");
    if (!dbias) partial = nullptr;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(nblk), dim3(256), 0, stream, df, f, (bf16_t*)dz, dbias, partial, rows, cols, rpb);
    if (partial) hipLaunchKernelGGL(sigmoid_bias_reduce_kernel, dim3((cols + 63) / 64), dim3(1024), 0, stream, partial, dbias, nblk, cols);
    PPF_LAUNCH_CHECK();
    return 0;
}

// seg_bounds: nseg+1 int64 offsets (multiples of 4); seg_lr / seg_wd: nseg floats.  step >= 1.
int ppf_adamw_step(float* p, const float* g, float* m, float* v, float* ema, void* p16, int64_t n, int nseg, const int64_t* seg_bounds,
                   const float* seg_lr, const float* seg_wd, float beta1, float beta2, float eps, int step, float ema_decay,
                   float grad_scale, hipStream_t stream) {
    PPF_CHECK_ARG(n > 0 && (n % 4) == 0 && nseg >= 1 && nseg <= MAX_SEGS && step >= 1, PPF_ERR_ARG, "ppf_adamw_step: bad arguments");
    AdamParams a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.ema = ema; a.p16 = (bf16_t*)p16; a.n = n; a.nseg = nseg;
    for (int s = 0; s < nseg; ++s) {
        PPF_CHECK_ARG(seg_bounds[s] % 4 == 0 && seg_bounds[s + 1] >= seg_bounds[s], PPF_ERR_ALIGN, "ppf_adamw_step: segment bounds must be multiples of 4");
        a.seg[s].begin = seg_bounds[s]; a.seg[s].end = seg_bounds[s + 1]; a.seg[s].lr = seg_lr[s]; a.seg[s].wd = seg_wd[s];
    }
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.bc1 = 1.0f - powf(beta1, (float)step);
    a.bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    a.ema_decay = ema_decay; a.grad_scale = grad_scale; a.hyper = nullptr;
    a.guard = nullptr; a.nonfinite = nullptr;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, a);
    PPF_LAUNCH_CHECK();
    return 0;
}

// The same step with lr / weight decay / bias corrections / gradient scale read from device memory (`hyper`, 20 floats, layout
// above): nothing step-dependent is baked into the launch, so the call can be captured in a HIP graph and replayed.
// With the reference's per-step "Loss is nan/inf, stopping training" check (tools/engine_proto.py:66-70) folded in without a
// host synchronisation: `loss` is the step's loss on the device; when it is not finite NOTHING is updated (the reference exits in front
// of optimizer.step()) and *nonfinite_flag is set to 1 for the host to read at its logging cadence.  loss == NULL: no check.
int ppf_adamw_step_guarded(float* p, const float* g, float* m, float* v, float* ema, void* p16, int64_t n, int nseg,
                           const int64_t* seg_bounds, const float* hyper, float beta1, float beta2, float eps, float ema_decay,
                           const float* loss, int* nonfinite_flag, hipStream_t stream) {
    PPF_CHECK_ARG(n > 0 && (n % 4) == 0 && nseg >= 1 && nseg <= MAX_SEGS && hyper, PPF_ERR_ARG, "ppf_adamw_step_dev: bad arguments");
    PPF_CHECK_ARG((loss == nullptr) == (nonfinite_flag == nullptr), PPF_ERR_ARG, "ppf_adamw_step_guarded: loss and flag go together");
    AdamParams a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.ema = ema; a.p16 = (bf16_t*)p16; a.n = n; a.nseg = nseg;
    for (int s = 0; s < MAX_SEGS; ++s) { a.seg[s].begin = a.seg[s].end = 0; a.seg[s].lr = a.seg[s].wd = 0.f; }
    for (int s = 0; s < nseg; ++s) {
        PPF_CHECK_ARG(seg_bounds[s] % 4 == 0 && seg_bounds[s + 1] >= seg_bounds[s], PPF_ERR_ALIGN, "ppf_adamw_step_dev: segment bounds must be multiples of 4");
        a.seg[s].begin = seg_bounds[s]; a.seg[s].end = seg_bounds[s + 1];
    }
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.bc1 = a.bc2_sqrt = 1.f; a.ema_decay = ema_decay; a.grad_scale = 1.f; a.hyper = hyper;
    a.guard = loss; a.nonfinite = nonfinite_flag;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, a);
    PPF_LAUNCH_CHECK();
    return 0;
}

// The plain form (no loss check).
int ppf_adamw_step_dev(float* p, const float* g, float* m, float* v, float* ema, void* p16, int64_t n, int nseg,
                       const int64_t* seg_bounds, const float* hyper, float beta1, float beta2, float eps, float ema_decay,
                       hipStream_t stream) {
    return ppf_adamw_step_guarded(p, g, m, v, ema, p16, n, nseg, seg_bounds, hyper, beta1, beta2, eps, ema_decay, nullptr, nullptr, stream);
}

// hyper[0..n) <- host_vals[0..n): the values travel as kernel arguments (read from host memory NOW), so the host may run any
// number of steps ahead of the device -- an asynchronous copy from a reused pinned buffer would read whatever the buffer holds later.
int ppf_hyper_set(float* hyper, const float* host_vals, int n, hipStream_t stream) {
    PPF_CHECK_ARG(hyper && host_vals && n >= 1 && n <= 20, PPF_ERR_ARG, "ppf_hyper_set: bad arguments");
    HyperVals h;
    for (int i = 0; i < 20; ++i) h.v[i] = i < n ? host_vals[i] : 0.f;
    h.n = n;
    hipLaunchKernelGGL(hyper_set_kernel, dim3(1), dim3(64), 0, stream, hyper, h);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_clip_grad_blocks(void) { return 512; }

int ppf_clip_grad_scale(const float* g, int64_t n, float max_norm, float pre_scale, float* partial, float* hyper, float* norm_out,
                        hipStream_t stream) {
    PPF_CHECK_ARG(n > 0 && (n % 4) == 0 && max_norm > 0.f && partial && hyper, PPF_ERR_ARG, "ppf_clip_grad_scale: bad arguments");
    const int nblk = ppf_clip_grad_blocks();
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk), dim3(256), 0, stream, g, n / 4, partial);
    hipLaunchKernelGGL(clip_finish_kernel, dim3(1), dim3(256), 0, stream, partial, nblk, max_norm, pre_scale, hyper, norm_out);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_droppath_scales(float* out, const float* keep, int nslot, int B, uint64_t seed, void* state_u64, hipStream_t stream) {
    PPF_CHECK_ARG(nslot > 0 && B > 0 && out && keep && state_u64, PPF_ERR_ARG, "ppf_droppath_scales: bad arguments");
    hipLaunchKernelGGL(droppath_kernel, dim3(1), dim3(256), 0, stream, out, keep, nslot, B, seed, (unsigned long long*)state_u64);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_reserved_rows_map(const int* idx, int* rows, int B, int k, int N, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && k > 0 && N > k, PPF_ERR_SHAPE, "ppf_reserved_rows_map: bad shape B=%d k=%d N=%d", B, k, N);
    hipLaunchKernelGGL(rows_map_kernel, dim3((B * (k + 1) + 255) / 256), dim3(256), 0, stream, idx, rows, B, k, N);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_gather_rows(const void* src, const int* rows, void* dst, int nrows, int row_bytes, hipStream_t stream) {
    PPF_CHECK_ARG(nrows > 0 && row_bytes > 0 && row_bytes % 16 == 0, PPF_ERR_ALIGN, "ppf_gather_rows: row_bytes=%d must be a positive multiple of 16", row_bytes);
    hipLaunchKernelGGL(move_rows_kernel, dim3(grid_for((int64_t)nrows * (row_bytes / 16))), dim3(256), 0, stream, (const uint4*)src, rows, (uint4*)dst,
                       nrows, row_bytes / 16, 0);
    PPF_LAUNCH_CHECK();
    return 0;
}

// dst (nrows_dst rows) = 0 everywhere except dst[rows[r]] = src[r]
int ppf_scatter_rows(const void* src, const int* rows, void* dst, int nrows_src, int nrows_dst, int row_bytes, hipStream_t stream) {
    PPF_CHECK_ARG(nrows_src > 0 && nrows_dst >= nrows_src && row_bytes > 0 && row_bytes % 16 == 0, PPF_ERR_ALIGN, "ppf_scatter_rows: bad arguments");
    hipError_t e = hipMemsetAsync(dst, 0, (size_t)nrows_dst * row_bytes, stream);
    if (e != hipSuccess) { ppf_set_error("ppf_scatter_rows: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(move_rows_kernel, dim3(grid_for((int64_t)nrows_src * (row_bytes / 16))), dim3(256), 0, stream, (const uint4*)src, rows, (uint4*)dst,
                       nrows_src, row_bytes / 16, 1);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_image_finish_u8(const void* in_u8_hwc, float* out_nchw, int B, int H, int W, const float* mean3, const float* std3, const int* rects,
                        uint64_t seed, const void* state_u64, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && H > 0 && W > 0 && W % 4 == 0 && in_u8_hwc && out_nchw && mean3 && std3, PPF_ERR_ARG, "ppf_image_finish_u8: bad arguments (W %% 4 == 0)");
    PPF_CHECK_ARG((((uintptr_t)in_u8_hwc) & 3) == 0 && (((uintptr_t)out_nchw) & 15) == 0, PPF_ERR_ALIGN, "ppf_image_finish_u8: alignment");
    FinishParams p;
    p.in = (const unsigned char*)in_u8_hwc; p.out = out_nchw; p.rects = rects; p.B = B; p.H = H; p.W = W; p.seed = seed;
    p.state = (const unsigned long long*)state_u64;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.inv_std[c] = 1.0f / std3[c]; }
    hipLaunchKernelGGL(image_finish_kernel, dim3(grid_for((int64_t)B * H * W / 4)), dim3(256), 0, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_memset_zero(void* ptr, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return 0;
    hipError_t e = hipMemsetAsync(ptr, 0, bytes, stream);
    if (e != hipSuccess) { ppf_set_error("ppf_memset_zero: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

int ppf_scale_by_scalar(const float* x, const float* scalar_dev, float* out, int64_t n, hipStream_t stream) {
    PPF_CHECK_ARG(n > 0 && x && scalar_dev && out, PPF_ERR_ARG, "ppf_scale_by_scalar: bad arguments");
    hipLaunchKernelGGL(scale_by_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, scalar_dev, out, n);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_transpose_bf16_batched(const void* src, void* dst, const void* desc_i64, int n, int total_tiles, hipStream_t stream) {
    PPF_CHECK_ARG(src && dst && desc_i64 && n > 0 && total_tiles > 0, PPF_ERR_ARG, "ppf_transpose_bf16_batched: bad arguments");
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(total_tiles), dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, (const long long*)desc_i64, n);
    PPF_LAUNCH_CHECK();
    return 0;
}

// dst[r][0 .. width) = src[r][0 .. width) for `rows` rows with independent row pitches (bytes): the strided sub-matrix copies of the CaiT
// class-attention stage (torch.cat of the cls row with the patch rows, cait:314-316; the cls row of a gradient), on the library's
// own launch path so that a recorded step (engine.ReplayedTrainStep) contains them.
int ppf_copy_2d(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width, int64_t rows, hipStream_t stream) {
    PPF_CHECK_ARG(dst && src && width > 0 && rows > 0 && dst_pitch >= width && src_pitch >= width, PPF_ERR_ARG, "ppf_copy_2d: bad arguments");
    if ((((uintptr_t)dst | (uintptr_t)src | (uint64_t)dst_pitch | (uint64_t)src_pitch | (uint64_t)width) & 15) == 0 && rows * (width / 16) < (int64_t)1 << 31) {
        // 16-byte aligned on every side: a plain kernel of this library instead of the runtime's rectangular blit (same step time on
        // cait_xxs24, 10 016 vs 10 028 img/s; one launch path for everything a recorded step contains)
        const int w16 = (int)(width / 16);
        hipLaunchKernelGGL(copy_2d_kernel, dim3(grid_for(rows * w16)), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, (int)rows, w16,
                           (long long)(src_pitch / 16), (long long)(dst_pitch / 16));
        PPF_LAUNCH_CHECK();
        return 0;
    }
    hipError_t e = hipMemcpy2DAsync(dst, (size_t)dst_pitch, src, (size_t)src_pitch, (size_t)width, (size_t)rows, hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) { ppf_set_error("ppf_copy_2d: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

}  // extern "C"
