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

// dx [B][T][D] -> dtok bf16 [B*Np][D] (rows t >= lead), dpos[t] += sum_b dx[b][t], dcls += sum_b dx[b][0] (lead = 1)
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const float* __restrict__ dx, bf16_t* __restrict__ dtok,
                                                           float* __restrict__ dpos, float* __restrict__ dcls, int B, int Np,
                                                           int D, int lead, int bchunk) {
    const int T = Np + lead, d2 = D / 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T * d2) return;
    const int c = (i % d2) * 2, t = i / d2;
    const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
    float2 acc = make_float2(0.f, 0.f);
    for (int b = b0; b < b1; ++b) {
        const float2 v = *reinterpret_cast<const float2*>(dx + ((size_t)b * T + t) * D + c);
        acc.x += v.x; acc.y += v.y;
        if (t >= lead) *reinterpret_cast<uint32_t*>(dtok + ((size_t)b * Np + (t - lead)) * D + c) = pack_bf16x2(v.x, v.y);
    }
    unsafeAtomicAdd(dpos + (size_t)t * D + c, acc.x);
    unsafeAtomicAdd(dpos + (size_t)t * D + c + 1, acc.y);
    if (t < lead && dcls) { unsafeAtomicAdd(dcls + c, acc.x); unsafeAtomicAdd(dcls + c + 1, acc.y); }
}

// Fused AdamW (decoupled weight decay, torch.optim.AdamW semantics) over a flat parameter buffer split in
// segments with their own lr / weight decay; optionally updates an EMA copy and re-emits the bf16 weights.
struct Seg { int64_t begin, end; float lr, wd; };
constexpr int MAX_SEGS = 8;
struct AdamParams {
    float* p; const float* g; float* m; float* v; float* ema; bf16_t* p16;
    int64_t n; int nseg; Seg seg[MAX_SEGS];
    float beta1, beta2, eps, bc1, bc2_sqrt, ema_decay, grad_scale;
};
__global__ __launch_bounds__(256) void adamw_kernel(const AdamParams a) {
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

// dz = bf16(df * f * (1 - f)) for f = sigmoid(z); dbias[c] += column sums of the rounded dz (bias grad of the add-on layer)
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ df, const float* __restrict__ f, bf16_t* __restrict__ dz,
                                                          float* __restrict__ dbias, int rows, int cols, int rows_per_block) {
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
        if (dbias) unsafeAtomicAdd(dbias + c, acc);
    }
}

inline int grid_for(int64_t work, int per_block = 256) {
    int64_t g = (work + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" {

int ppf_cast_f32_bf16(const float* in, void* out, int64_t n, hipStream_t stream) {
    PPF_CHECK_ARG(n > 0 && (n % 8) == 0, PPF_ERR_SHAPE, "ppf_cast_f32_bf16: n=%lld must be a positive multiple of 8", (long long)n);
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, in, (bf16_t*)out, n / 8);
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
    const int T = Np + lead, threads = T * D / 2;
    const int bchunk = 16;
    hipLaunchKernelGGL(assemble_bwd_kernel, dim3((threads + 255) / 256, (B + bchunk - 1) / bchunk), dim3(256), 0, stream, dx, (bf16_t*)dtok, dpos, dcls, B,
                       Np, D, lead, bchunk);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_sigmoid_bwd(const float* df, const float* f, void* dz, float* dbias, int rows, int cols, hipStream_t stream) {
    PPF_CHECK_ARG(rows > 0 && cols > 0, PPF_ERR_SHAPE, "ppf_sigmoid_bwd: bad shape");
    const int rpb = 16;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((rows + rpb - 1) / rpb), dim3(256), 0, stream, df, f, (bf16_t*)dz, dbias, rows, cols, rpb);
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
    a.ema_decay = ema_decay; a.grad_scale = grad_scale;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, a);
    PPF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
