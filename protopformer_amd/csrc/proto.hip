// Prototype layer (protopformer.py:201-247): squared-L2 distance of every reserved token to every prototype,
// log similarity, max-pool over the tokens of a sample -- and its backward.
//
// Forward: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) for the token.prototype contraction -- the expansion
// |x|^2 - 2 x.p + |p|^2 cancels catastrophically for close pairs, exactly where log((d+1)/(d+eps)) is steepest,
// so bf16 operands are not an option here (SURVEY 7 "Distance cancellation").  One workgroup = one sample
// (all its T tokens, padded to TT*32 rows) x 128 prototypes; |x|^2 and |p|^2 are accumulated from the very
// operands that feed the MFMA; relu / log / max+argmax over tokens run in registers (token index lives in the
// accumulator registers, prototype index in the lane).  The full (B,P,T) maps the API returns are transposed
// through LDS so HBM sees token-contiguous runs.
//
// Backward: dL/dd is ~98 % sparse (one arg-max token per (sample, prototype) + the label's prototypes from the
// PPC loss), so it is a gather-scale-accumulate in exact fp32 on (x - p) differences, deterministic, no atomics:
//   proto_bwd_tokens : one workgroup per sample, waves own tokens, scan the prototype axis
//   proto_bwd_protos : one workgroup per prototype, waves own samples, scan the token axis
#include "ppf_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

#ifndef PPF_PROTO_OCC
#define PPF_PROTO_OCC 3
#endif
constexpr int PB = 128;           // prototypes per workgroup (4 waves x 32)
constexpr int BKF = 32;           // contraction chunk (floats)
constexpr int LDP = BKF + 1;      // padded LDS pitch: conflict-free ds_read_b32 for the MFMA operands

struct ProtoFwdParams {
    const float* tok;      // token rows: row(b, i) = tok + b*stride_b + (t0 + i)*Dp      (POOL)
                           //            row(g, i) = tok + (g*rows + i)*stride_b + t0*Dp  (!POOL: one token per sample)
    int64_t stride_b;
    int t0, T;             // POOL: tokens per sample
    const float* protos;   // [P][Dp]
    int B, P, Dp;
    int act_kind;          // 0: log((d+1)/(d+eps))   1: -d
    float eps;
    float* act_max;        // [B][P]
    int* argmax;           // [B][P] (POOL only)
    float* dist_full;      // [B][P][T] or null
    float* act_full;       // [B][P][T] or null
};

__device__ __forceinline__ float activation(float d, int kind, float eps) {
    return kind == 0 ? __logf((d + 1.0f) / (d + eps)) : -d;
}

// Everything after the contraction: distances (protopformer.py:213-216 association order), activations, max-pool + arg-max over the
// tokens, the (B,P) outputs and the full (B,P,T) maps.  acc[t][r] = <token row, prototype of this lane>; lx2 holds |x|^2 per row.
template <int TT, bool POOL>
__device__ __forceinline__ void proto_fwd_epilogue(const ProtoFwdParams& p, f32x16 (&acc)[TT], float* lds, const float* lx2, float p2, int p0,
                                                   int grp, int nrows, int lane, int wave, int hh) {
    constexpr int ROWS = TT * 32;
    const int pidx = p0 + wave * 32 + (lane & 31);
    float best = -INFINITY;
    int besti = 0;
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float d = fmaxf(lx2[row] + (-2.0f * acc[t][r] + p2), 0.0f);      // protopformer.py:213-216 association order
            acc[t][r] = d;
            if (POOL && row < p.T) {
                const float a = activation(d, p.act_kind, p.eps);
                if (a > best) { best = a; besti = row; }
            }
        }
    if (POOL) {
        const float ob = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(besti, 32, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        if (hh == 0 && pidx < p.P) {
            p.act_max[(size_t)grp * p.P + pidx] = best;
            p.argmax[(size_t)grp * p.P + pidx] = besti;
        }
    } else {
        // one token per sample: rows are samples, no pooling
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int smp = grp * ROWS + row;
                if (row < nrows && pidx < p.P) {
                    p.act_max[(size_t)smp * p.P + pidx] = activation(acc[t][r], p.act_kind, p.eps);
                    if (p.dist_full) p.dist_full[(size_t)smp * p.P + pidx] = acc[t][r];
                    if (p.act_full) p.act_full[(size_t)smp * p.P + pidx] = activation(acc[t][r], p.act_kind, p.eps);
                }
            }
        return;
    }
    // full maps (B,P,T): transpose the wave's [token][prototype] tile through LDS, 16 prototypes at a time.  The 16 rows of T floats
    // are ONE contiguous range of the map (prototypes are consecutive), written as 16-byte pieces when the range is so aligned.
    float* xp = lds + wave * 16 * (ROWS + 1);
    const int pw = p0 + wave * 32;                       // first prototype of this wave
    for (int pass = 0; pass < 2; ++pass) {
        float* dst = pass == 0 ? p.dist_full : p.act_full;
        if (!dst) continue;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            if (((lane & 31) >> 4) == half) {
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        xp[(lane & 15) * (ROWS + 1) + row] = pass == 0 ? acc[t][r] : activation(acc[t][r], p.act_kind, p.eps);
                    }
            }
            __syncthreads();
            const int pf = pw + 16 * half;               // first prototype of this half
            const int np = min(16, p.P - pf);            // prototypes of this half that exist
            if (np <= 0) continue;
            float* obase = dst + ((size_t)grp * p.P + pf) * p.T;
            const int total = np * p.T;
            if ((((size_t)grp * p.P + pf) * p.T) % 4 == 0) {
                for (int f = lane * 4; f < total; f += 256) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ff = min(f + e, total - 1), pp = ff / p.T, tt = ff - pp * p.T;
                        v[e] = xp[pp * (ROWS + 1) + tt];
                    }
                    if (f + 4 <= total) *reinterpret_cast<float4*>(obase + f) = make_float4(v[0], v[1], v[2], v[3]);
                    else for (int e = 0; f + e < total; ++e) obase[f + e] = v[e];
                }
            } else {
                for (int f = lane; f < total; f += 64) { const int pp = f / p.T, tt = f - pp * p.T; obase[f] = xp[pp * (ROWS + 1) + tt]; }
            }
        }
    }
}

// ---- split-bf16 contraction ------------------------------------------------------------------------------------------------
// fp32 x = a + b + c with a = bf16(x), b = bf16(x - a), c = bf16(x - a - b): three 8-bit pieces hold the 24-bit significand exactly.
// <x, p> = sum over the six piece products with (piece index of x) + (piece index of p) <= 2 -- each product of two bf16 values is
// exact in fp32, the dropped terms are below 3 * 2^-24 |x||p|, i.e. at the rounding level of an fp32 multiply -- on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6 x 32 cycles per 16 contraction values instead of 8 x 64 for the fp32 MFMA
// (v_mfma_f32_32x32x2_f32), which bounds proto_fwd_kernel below (measured 450 of 561 us).  |x|^2 and |p|^2 stay plain fp32 sums.
struct Split3 { uint32_t a, b, c; };               // two values per register: the pieces of (lo, hi)
__device__ __forceinline__ Split3 split3(float lo, float hi) {
    Split3 r;
    r.a = pack_bf16x2(lo, hi);
    const float l1 = lo - __uint_as_float(r.a << 16), h1 = hi - __uint_as_float(r.a & 0xffff0000u);
    r.b = pack_bf16x2(l1, h1);
    const float l2 = l1 - __uint_as_float(r.b << 16), h2 = h1 - __uint_as_float(r.b & 0xffff0000u);
    r.c = pack_bf16x2(l2, h2);
    return r;
}
__device__ __forceinline__ int p6_off(int r, int c16) { return r * 128 + ((c16 ^ ((r >> 1) & 7)) << 4); }   // 128-byte rows, chunk swizzle

template <int TT, bool POOL>
__global__ __launch_bounds__(256, TT <= 3 ? 3 : 2) void proto_fwd6_kernel(const ProtoFwdParams p) {
    constexpr int ROWS = TT * 32, BK6 = 64;
    constexpr int PLANE = ROWS * 128;                              // bytes: [ROWS][64 bf16], one piece
    constexpr int STAGE = 3 * PLANE / 4;                           // floats
    constexpr int XPOSE = 4 * 16 * (ROWS + 1);
    __shared__ __attribute__((aligned(16))) float lds[(STAGE > XPOSE ? STAGE : XPOSE) + ROWS];
    unsigned char* planes = reinterpret_cast<unsigned char*>(lds);
    float* lx2 = lds + (STAGE > XPOSE ? STAGE : XPOSE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
    const int p0 = blockIdx.x * PB, grp = blockIdx.y;
    const int nrows = POOL ? p.T : min(ROWS, p.B - grp * ROWS);

    f32x16 acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // token rows: global -> registers (next chunk in flight while this one is multiplied) -> three bf16 piece images in LDS
    constexpr int NLD = ROWS * (BK6 / 4) / 256;                    // float4 per thread and chunk (ROWS is a multiple of 32)
    static_assert(ROWS * (BK6 / 4) % 256 == 0, "whole float4 per thread");
    float4 stg[NLD];
    float x2p[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) x2p[j] = 0.f;
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * 256, row = i >> 4, c4 = i & 15;
            const int rr = min(row, nrows - 1);                    // rows past the end repeat the last one (never written out)
            const float* src = POOL ? p.tok + (size_t)grp * p.stride_b + (size_t)(p.t0 + rr) * p.Dp
                                    : p.tok + (size_t)(grp * ROWS + rr) * p.stride_b + (size_t)p.t0 * p.Dp;
            const int kk = min(k0 + c4 * 4, p.Dp - 4);
            float4 v = *reinterpret_cast<const float4*>(src + kk);
            if (k0 + c4 * 4 >= p.Dp) v = make_float4(0.f, 0.f, 0.f, 0.f);
            stg[j] = v;
        }
    };
    auto sstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * 256, row = i >> 4, c4 = i & 15;
            const float4 v = stg[j];
            x2p[j] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            const Split3 s0 = split3(v.x, v.y), s1 = split3(v.z, v.w);
            const int off = p6_off(row, c4 >> 1) + 8 * (c4 & 1);
            *reinterpret_cast<uint2*>(planes + off) = make_uint2(s0.a, s1.a);
            *reinterpret_cast<uint2*>(planes + PLANE + off) = make_uint2(s0.b, s1.b);
            *reinterpret_cast<uint2*>(planes + 2 * PLANE + off) = make_uint2(s0.c, s1.c);
        }
    };
    // prototype row of this lane: straight from global (L2) in fragment order, 8 contraction values per 16-step, split in registers
    const int pidx_c = min(p0 + wave * 32 + l31, p.P - 1);
    const float* prow = p.protos + (size_t)pidx_c * p.Dp + 8 * hh;
    float p2p = 0.f;
    auto pload = [&](int k, float4& u0, float4& u1) __attribute__((always_inline)) {
        const int kk = min(k, p.Dp - 16);                          // Dp % 16 == 0 is checked on the host
        u0 = *reinterpret_cast<const float4*>(prow + kk);
        u1 = *reinterpret_cast<const float4*>(prow + kk + 4);
    };
    float4 q0, q1;
    gload(0);
    pload(0, q0, q1);
    for (int k0 = 0; k0 < p.Dp; k0 += BK6) {
        __syncthreads();                                             // everyone finished reading the previous chunk
        sstore();
        __syncthreads();
        if (k0 + BK6 < p.Dp) gload(k0 + BK6);
#pragma unroll
        for (int ks = 0; ks < BK6 / 16; ++ks) {
            if (k0 + ks * 16 < p.Dp) {
                const float4 u0 = q0, u1 = q1;
                pload(k0 + ks * 16 + 16, q0, q1);                    // next 16-step (clamped re-read at the end)
                p2p += u0.x * u0.x + u0.y * u0.y + u0.z * u0.z + u0.w * u0.w + u1.x * u1.x + u1.y * u1.y + u1.z * u1.z + u1.w * u1.w;
                const Split3 b0 = split3(u0.x, u0.y), b1 = split3(u0.z, u0.w), b2 = split3(u1.x, u1.y), b3 = split3(u1.z, u1.w);
                typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
                const bf16x8 pa = __builtin_bit_cast(bf16x8, (u32x4){b0.a, b1.a, b2.a, b3.a});
                const bf16x8 pb = __builtin_bit_cast(bf16x8, (u32x4){b0.b, b1.b, b2.b, b3.b});
                const bf16x8 pc = __builtin_bit_cast(bf16x8, (u32x4){b0.c, b1.c, b2.c, b3.c});
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    const int off = p6_off(t * 32 + l31, ks * 2 + hh);
                    const bf16x8 xa = *reinterpret_cast<const bf16x8*>(planes + off);
                    const bf16x8 xb = *reinterpret_cast<const bf16x8*>(planes + PLANE + off);
                    const bf16x8 xc = *reinterpret_cast<const bf16x8*>(planes + 2 * PLANE + off);
                    // smallest terms first
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xc, pa, acc[t], 0, 0, 0);     // D[i = token][j = prototype]
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, pc, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, pb, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, pa, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, pb, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, pa, acc[t], 0, 0, 0);
                }
            }
        }
    }
    const float p2 = p2p + __shfl_xor(p2p, 32, 64);
    __syncthreads();
    // |x|^2 per row: the 16 threads that staged a row's float4 pieces are 16 consecutive lanes
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        float v = x2p[j];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        const int i = tid + j * 256;
        if ((i & 15) == 0) lx2[i >> 4] = v;
    }
    __syncthreads();
    proto_fwd_epilogue<TT, POOL>(p, acc, lds, lx2, p2, p0, grp, nrows, lane, wave, hh);
}

template <int TT, bool POOL>
__global__ __launch_bounds__(256, TT <= 3 ? PPF_PROTO_OCC : 2) void proto_fwd_kernel(const ProtoFwdParams p) {
    constexpr int ROWS = TT * 32;
    constexpr int STAGE = (ROWS + PB) * LDP;                       // floats
    constexpr int XPOSE = 4 * 16 * (ROWS + 1);                     // per-wave [16 p][ROWS+1] transpose tiles (two halves per map): with
                                                                   // the operand stage <= 30 KiB this leaves room for five workgroups per CU
    __shared__ float lds[(STAGE > XPOSE ? STAGE : XPOSE) + ROWS];
    float* ltok = lds;
    float* lpro = lds + ROWS * LDP;
    float* lx2 = lds + (STAGE > XPOSE ? STAGE : XPOSE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const int p0 = blockIdx.x * PB, grp = blockIdx.y;
    const int nrows = POOL ? p.T : min(ROWS, p.B - grp * ROWS);

    f32x16 acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float x2p[TT], p2p = 0.f;
#pragma unroll
    for (int t = 0; t < TT; ++t) x2p[t] = 0.f;

    // operand chunks go global -> registers -> LDS; the loads of chunk k+1 are issued before chunk k is multiplied
    constexpr int NLD = ((ROWS + PB) * (BKF / 4) + 255) / 256;
    float4 stg[NLD];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * 256;
            const int row = i / (BKF / 4), c4 = i % (BKF / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < (ROWS + PB) * (BKF / 4) && k0 + c4 * 4 < p.Dp) {
                if (row < ROWS) {
                    if (row < nrows) {
                        const float* src = POOL ? p.tok + (size_t)grp * p.stride_b + (size_t)(p.t0 + row) * p.Dp
                                                : p.tok + (size_t)(grp * ROWS + row) * p.stride_b + (size_t)p.t0 * p.Dp;
                        v = *reinterpret_cast<const float4*>(src + k0 + c4 * 4);
                    }
                } else if (p0 + row - ROWS < p.P) {
                    v = *reinterpret_cast<const float4*>(p.protos + (size_t)(p0 + row - ROWS) * p.Dp + k0 + c4 * 4);
                }
            }
            stg[j] = v;
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * 256;
            if (i < (ROWS + PB) * (BKF / 4)) {
                const int row = i / (BKF / 4), c4 = i % (BKF / 4);
                float* d = lds + row * LDP + c4 * 4;                 // ltok rows are followed by lpro rows at the same pitch
                d[0] = stg[j].x; d[1] = stg[j].y; d[2] = stg[j].z; d[3] = stg[j].w;
            }
        }
    };
    gload(0);
    for (int k0 = 0; k0 < p.Dp; k0 += BKF) {
        __syncthreads();                                             // everyone finished reading the previous chunk
        sstore();
        __syncthreads();
        if (k0 + BKF < p.Dp) gload(k0 + BKF);
#pragma unroll
        for (int kk = 0; kk < BKF / 2; ++kk) {
            const float bv = lpro[(wave * 32 + (lane & 31)) * LDP + kk * 2 + hh];
            p2p += bv * bv;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const float av = ltok[(t * 32 + (lane & 31)) * LDP + kk * 2 + hh];
                x2p[t] += av * av;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);     // D[i = token][j = prototype]
            }
        }
    }
    // |p|^2 for this lane's prototype, |x|^2 per token row (wave 0 publishes them)
    const float p2 = p2p + __shfl_xor(p2p, 32, 64);
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const float x2 = x2p[t] + __shfl_xor(x2p[t], 32, 64);
            if (hh == 0) lx2[t * 32 + lane] = x2;
        }
    }
    __syncthreads();
    proto_fwd_epilogue<TT, POOL>(p, acc, lds, lx2, p2, p0, grp, nrows, lane, wave, hh);
}

// ------------------------------------------------------------------------------------------------ backward
struct ProtoBwdParams {
    const float* tok; int64_t stride_b; int t0, T;
    const float* protos; int B, P, Dp;
    int act_kind; float eps;
    const float* dist_full;    // [B][P][T]
    const float* g_full;       // [B][P][T] upstream grad of act_full, or null
    const float* g_max;        // [B][P]    upstream grad of act_max, or null
    const int* argmax;         // [B][P] (null when T == 1)
    float* dtok;               // written: row(b,i) = dtok + b*dstride_b + (t0+i)*Dp
    int64_t dstride_b;
    float* dprotos;            // [P][Dp], accumulated (+=)
};

__device__ __forceinline__ float dact_dd(float d, int kind, float eps) {
    if (d <= 0.f) return 0.f;                                       // relu clipped (protopformer.py:216)
    return kind == 0 ? (1.0f / (d + 1.0f) - 1.0f / (d + eps)) : -1.0f;
}

// G[b,p,t] = dL/dd
__device__ __forceinline__ float grad_d(const ProtoBwdParams& p, int b, int pi, int t, int amax, float gmax) {
    const size_t o = ((size_t)b * p.P + pi) * p.T + t;
    float g = p.g_full ? p.g_full[o] : 0.f;
    if (t == amax) g += gmax;
    if (g == 0.f) return 0.f;
    return g * dact_dd(p.dist_full[o], p.act_kind, p.eps);
}

// Token gradients in two fully parallel passes (deterministic, no float atomics):
//  mark  : stream the (B,P,T) upstream gradient once (coalesced) and set bit (b,t,p) of a global bitmap [B][T][W]
//          wherever dL/dd != 0 (order-independent atomic OR)
//  gather: one wavefront per token (b,t) walks its bitmap row in ascending prototype order and accumulates
//          2 G (x - p) in exact fp32 -- ~35 non-zeros per token, latency hidden by ~20k independent waves.
// Candidate bitmap for proto_bwd_tokens: bit (b, t, p) is set where the upstream gradient of the activation map is non-zero or t is
// the arg-max token of (b, p).  A superset of the non-zero dL/dd entries is enough (the consumer multiplies by the exact coefficient,
// zero included), so neither the distances nor exact cancellations are looked at: the scan reads g_full once, 16 bytes per lane.
// One workgroup per (sample, 32 prototypes): their rows are one contiguous range of 32 * T floats; word (b, t, p / 32) of the bitmap.
__global__ __launch_bounds__(256) void proto_bwd_mark_kernel(const ProtoBwdParams p, uint32_t* __restrict__ bm, int W) {
    const int w = blockIdx.x, b = blockIdx.y, T = p.T;
    const int np = min(32, p.P - 32 * w);
    const size_t base = ((size_t)b * p.P + 32 * w) * T;
    const int total = np * T;
    const float invT = 1.0f / (float)T;
    if (p.g_full) {
        const float* src = p.g_full + base;
        if ((base & 3) == 0) {
            for (int f = threadIdx.x * 4; f < total; f += 1024) {
                float v[4];
                if (f + 4 <= total) { const float4 q = *reinterpret_cast<const float4*>(src + f); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
                else { for (int e = 0; e < 4; ++e) v[e] = f + e < total ? src[f + e] : 0.f; }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] != 0.f) {
                        int pl = (int)(((float)(f + e) + 0.5f) * invT);                  // exact for f < 2^20
                        int t = f + e - pl * T;
                        if (t < 0) { --pl; t += T; } else if (t >= T) { ++pl; t -= T; }
                        atomicOr(&bm[((size_t)b * T + t) * W + w], 1u << pl);
                    }
            }
        } else {
            for (int f = threadIdx.x; f < total; f += 256)
                if (src[f] != 0.f) { const int pl = f / T, t = f - pl * T; atomicOr(&bm[((size_t)b * T + t) * W + w], 1u << pl); }
        }
    }
    if (p.g_max && threadIdx.x < np) {
        const size_t bp = (size_t)b * p.P + 32 * w + threadIdx.x;
        if (p.g_max[bp] != 0.f) {
            const int am = p.argmax ? p.argmax[bp] : 0;
            atomicOr(&bm[((size_t)b * T + am) * W + w], 1u << threadIdx.x);
        }
    }
}
// NW waves per (sample, token) row, each owning ceil(W / NW) <= 64 words of the row's bitmap.  NW = 1 (P <= 2048): a row is one
// independent wavefront -- no barrier, no cross-wave reduction, eight of them per SIMD to hide the dependent loads (bitmap -> list ->
// coefficients -> prototype rows); the 8-wave form ran 20 rounds of ~9 us workgroups (185 us for 20 736 rows).
template <int NJ, int NW>
__global__ __launch_bounds__(64 * NW) void proto_bwd_tokens_kernel(const ProtoBwdParams p, const uint32_t* __restrict__ bm, int W) {
    __shared__ unsigned short plist[NW][64 * 32];          // up to 64 words per wave
    __shared__ float part[NW > 1 ? NW : 1][NW > 1 ? NJ * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = blockIdx.x;                            // (b, t)
    const int b = row / p.T, t = row % p.T;
    const float* xrow = p.tok + (size_t)b * p.stride_b + (size_t)(p.t0 + t) * p.Dp;
    float x[NJ], acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int d = lane + 64 * j; x[j] = d < p.Dp ? xrow[d] : 0.f; acc[j] = 0.f; }
    const int wpw = (W + NW - 1) / NW;                     // words per wave (<= 64)
    const int w_begin = wave * wpw, w_end = min(W, w_begin + wpw);
    // (1) ascending list of this wave's prototypes with a non-zero gradient for this token
    const uint32_t mine = (w_begin + lane < w_end) ? bm[(size_t)row * W + w_begin + lane] : 0u;
    const int cnt = __popc(mine);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
    const int total = __shfl(incl, 63, 64);
    {
        int pos = incl - cnt;
        uint32_t word = mine;
        while (word) { const int bit = __builtin_ctz(word); word &= word - 1; plist[wave][pos++] = (unsigned short)((w_begin + lane) * 32 + bit - w_begin * 32); }
    }
    // (wave-private LDS region: same-wave program order makes the list visible)
    for (int base = 0; base < total; base += 64) {
        const int n = min(64, total - base);
        int pi = 0; float g2 = 0.f;
        if (lane < n) {
            pi = w_begin * 32 + plist[wave][base + lane];
            const int am = p.argmax ? p.argmax[(size_t)b * p.P + pi] : 0;
            const float gm = p.g_max ? p.g_max[(size_t)b * p.P + pi] : 0.f;
            g2 = 2.0f * grad_d(p, b, pi, t, am, gm);
        }
        for (int e = 0; e < n; e += 4) {
            float gs[4]; const float* pr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int src = min(e + u, n - 1);
                gs[u] = (e + u < n) ? __shfl(g2, src, 64) : 0.f;
                pr[u] = p.protos + (size_t)__shfl(pi, src, 64) * p.Dp;
            }
            float v[4][NJ];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < NJ; ++j) { const int d = lane + 64 * j; v[u][j] = d < p.Dp ? pr[u][d] : 0.f; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] += gs[u] * (x[j] - v[u][j]);
        }
    }
    float* drow = p.dtok + (size_t)b * p.dstride_b + (size_t)(p.t0 + t) * p.Dp;
    if constexpr (NW == 1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int d = lane + 64 * j; if (d < p.Dp) drow[d] = acc[j]; }
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) part[wave][j * 64 + lane] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < NJ * 64; i += 64 * NW) {
        const int d = (i & 63) + 64 * (i >> 6);
        if (d < p.Dp) {
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sacc += part[w][i];
            drow[d] = sacc;
        }
    }
}

// Prototype gradients: one 16-wave workgroup per prototype, two passes over blocks of 16 x 1024 (sample, token) elements.
//  scan   : wave w streams its 1024 consecutive elements of dL/dd (coalesced, 8 independent 64-wide loads in flight) and
//           compacts the non-zeros -- the arg-max token of every sample plus the dense rows of the samples whose label owns
//           this prototype -- into an ascending per-wave LDS list (index, dL/dd).
//  gather : the concatenated list is dealt out evenly to the 16 waves (the dense rows no longer sit on one wave); each wave
//           fetches four token rows per step and accumulates 2 G (p - x) in exact fp32.
// Partial rows are reduced through LDS in fixed order: deterministic, no float atomics.
constexpr int PB_NW = 16, PB_CHUNK = 1024;
template <int NJ>
__global__ __launch_bounds__(1024) void proto_bwd_protos_kernel(const ProtoBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem[];
    float* gl = reinterpret_cast<float*>(pb_smem);                                        // [NW][CHUNK] dL/dd
    unsigned short* il = reinterpret_cast<unsigned short*>(gl + PB_NW * PB_CHUNK);       // [NW][CHUNK] element index inside the wave's range
    float* red = reinterpret_cast<float*>(il + PB_NW * PB_CHUNK);                         // [NW][NJ*64]
    int* cnt = reinterpret_cast<int*>(red + PB_NW * NJ * 64);                             // [NW + 1] exclusive prefix of the list lengths
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pi = blockIdx.x;
    const float* prow = p.protos + (size_t)pi * p.Dp;
    float pv[NJ], acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int d = lane + 64 * j; pv[j] = d < p.Dp ? prow[d] : 0.f; acc[j] = 0.f; }
    const int E = p.B * p.T;
    const float invT = 1.0f / (float)p.T;
    const size_t PT = (size_t)p.P * p.T;
    for (int blk0 = 0; blk0 < E; blk0 += PB_NW * PB_CHUNK) {
        // ---- scan + compact
        const int w0 = blk0 + wave * PB_CHUNK;
        int n = 0;
        for (int s0 = 0; s0 < PB_CHUNK; s0 += 8 * 64) {
            float g[8]; int bb[8], tt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = w0 + s0 + u * 64 + lane;
                int b = __float2int_rz(((float)e + 0.5f) * invT);
                int t = e - b * p.T;
                if (t < 0) { --b; t += p.T; } else if (t >= p.T) { ++b; t -= p.T; }
                bb[u] = b; tt[u] = t;
                g[u] = 0.f;
                if (e < E) {
                    const size_t bp = (size_t)b * p.P + pi;
                    if (p.g_full) g[u] = p.g_full[bp * p.T + t];
                    if (p.g_max) { const int am = p.argmax ? p.argmax[bp] : 0; if (t == am) g[u] += p.g_max[bp]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float G = 0.f;
                if (g[u] != 0.f) G = g[u] * dact_dd(p.dist_full[(size_t)bb[u] * PT + (size_t)pi * p.T + tt[u]], p.act_kind, p.eps);
                const unsigned long long m = __ballot(G != 0.f);
                if (G != 0.f) {
                    const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
                    gl[wave * PB_CHUNK + pos] = G;
                    il[wave * PB_CHUNK + pos] = (unsigned short)(s0 + u * 64 + lane);
                }
                n += __popcll(m);
            }
        }
        if (lane == 0) cnt[wave + 1] = n;
        __syncthreads();
        if (threadIdx.x == 0) { int a = 0; cnt[0] = 0; for (int w = 1; w <= PB_NW; ++w) { a += cnt[w]; cnt[w] = a; } }
        __syncthreads();
        // ---- gather: entries [k0, k1) of the concatenated list
        const int total = cnt[PB_NW];
        const int k0 = (int)(((long long)total * wave) / PB_NW), k1 = (int)(((long long)total * (wave + 1)) / PB_NW);
        for (int kb = k0; kb < k1; kb += 64) {
            const int nn = min(64, k1 - kb);
            float g2 = 0.f; long long roff = 0;
            if (lane < nn) {
                const int k = kb + lane;
                int sw = 0;
#pragma unroll
                for (int w = 1; w < PB_NW; ++w) sw += (k >= cnt[w]) ? 1 : 0;
                const int li = k - cnt[sw];
                g2 = 2.0f * gl[sw * PB_CHUNK + li];
                const int e = blk0 + sw * PB_CHUNK + il[sw * PB_CHUNK + li];
                int b = __float2int_rz(((float)e + 0.5f) * invT);
                int t = e - b * p.T;
                if (t < 0) { --b; t += p.T; } else if (t >= p.T) { ++b; t -= p.T; }
                roff = (long long)b * p.stride_b + (long long)(p.t0 + t) * p.Dp;
            }
            for (int e4 = 0; e4 < nn; e4 += 4) {
                float gs[4]; const float* xr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int src = min(e4 + u, nn - 1);
                    gs[u] = (e4 + u < nn) ? __shfl(g2, src, 64) : 0.f;
                    const int lo = __shfl((int)(roff & 0xffffffffll), src, 64), hi = __shfl((int)(roff >> 32), src, 64);
                    xr[u] = p.tok + (((long long)hi << 32) | (unsigned int)lo);
                }
                float v[4][NJ];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) { const int d = lane + 64 * j; v[u][j] = d < p.Dp ? xr[u][d] : 0.f; }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[j] += gs[u] * (pv[j] - v[u][j]);
            }
        }
        __syncthreads();                                     // lists are rewritten by the next block
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) red[wave * (NJ * 64) + j * 64 + lane] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < NJ * 64; i += 1024) {
        const int d = (i & 63) + 64 * (i >> 6);
        if (d < p.Dp) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < PB_NW; ++w) s += red[w * (NJ * 64) + i];
            p.dprotos[(size_t)pi * p.Dp + d] += s;           // single writer per (prototype, d)
        }
    }
}

// ---- T == 1 (the global / cls branch): every (sample, prototype) pair carries a gradient, so the gather form degenerates into two
// dense fp32 products  dtok[b] = 2 (sum_p G[b,p]) tok[b] - 2 G protos,   dprotos[p] += 2 (sum_b G[b,p]) protos[p] - 2 G^T tok
// with G[b,p] = g[b,p] * dact/dd(dist[b,p]).
__global__ __launch_bounds__(256) void proto_single_gd_rows_kernel(const float* __restrict__ dist, const float* __restrict__ g, int P, int act_kind,
                                                                    float eps, float* __restrict__ G, float* __restrict__ rowsum) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int pi = threadIdx.x; pi < P; pi += 256) {
        const size_t o = (size_t)b * P + pi;
        const float v = g[o] * dact_dd(dist[o], act_kind, eps);
        G[o] = v;
        s += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) rowsum[b] = (red[0] + red[1]) + (red[2] + red[3]);
}
// G and its column sums: a workgroup owns 64 prototypes; its four waves take every fourth sample (coalesced 256-byte rows), their
// partial sums are added in a fixed order
__global__ __launch_bounds__(256) void proto_single_gd_cols_kernel(const float* __restrict__ dist, const float* __restrict__ g, int B, int P, int act_kind,
                                                                    float eps, float* __restrict__ G, float* __restrict__ colsum) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pi = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (pi < P) {
#pragma unroll 4
        for (int b = wave; b < B; b += 4) {
            const size_t o = (size_t)b * P + pi;
            const float v = g[o] * dact_dd(dist[o], act_kind, eps);
            G[o] = v;
            s += v;
        }
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && pi < P) colsum[pi] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
// out[r][:] += 2 * scale[r] * x[r][:]   (rows of Dp floats, independent row strides)
__global__ __launch_bounds__(256) void proto_single_fixup_kernel(float* __restrict__ out, int64_t ostride, const float* __restrict__ x, int64_t xstride,
                                                                  const float* __restrict__ scale, int R, int Dp) {
    const int64_t n = (int64_t)R * Dp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / Dp), d = (int)(i - (int64_t)r * Dp);
        out[r * ostride + d] += 2.0f * scale[r] * x[r * xstride + d];
    }
}

}  // namespace

extern "C" {

// tokens: fp32 rows of Dp values; sample b's token i is at tok + b*stride_b + (t0+i)*Dp, T tokens per sample
// (T == 1: the global/cls branch, no pooling).  protos [P][Dp].  act_kind 0 = 'log', 1 = 'linear'.
// Outputs: act_max [B][P], argmax [B][P] (T > 1), optional dist_full / act_full [B][P][T].
int ppf_proto_fwd(const float* tok, int64_t stride_b, int t0, int T, const float* protos, int B, int P, int Dp, int act_kind, float eps,
                  float* act_max, int* argmax, float* dist_full, float* act_full, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && P > 0 && Dp > 0 && Dp % 4 == 0 && T >= 1 && T <= 128, PPF_ERR_SHAPE, "ppf_proto_fwd: bad shape B=%d P=%d Dp=%d T=%d", B, P, Dp, T);
    PPF_CHECK_ARG(tok && protos && act_max && (T == 1 || argmax), PPF_ERR_ARG, "ppf_proto_fwd: null pointer");
    ProtoFwdParams p;
    p.tok = tok; p.stride_b = stride_b; p.t0 = t0; p.T = T; p.protos = protos; p.B = B; p.P = P; p.Dp = Dp; p.act_kind = act_kind; p.eps = eps;
    p.act_max = act_max; p.argmax = argmax; p.dist_full = dist_full; p.act_full = act_full;
    const int gx = (P + PB - 1) / PB;
    // default: the split-bf16 contraction; PPF_PROTO_FP32=1 (or Dp not a multiple of 16): the fp32-MFMA kernel
    static const int fp32_mfma = getenv("PPF_PROTO_FP32") ? atoi(getenv("PPF_PROTO_FP32")) : 0;
    const bool x6 = !fp32_mfma && Dp % 16 == 0;
    if (T == 1) {
        if (x6) hipLaunchKernelGGL((proto_fwd6_kernel<2, false>), dim3(gx, (B + 63) / 64), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((proto_fwd_kernel<2, false>), dim3(gx, (B + 63) / 64), dim3(256), 0, stream, p);
    } else {
        const int tt = (T + 31) / 32;
        if (x6) {
            switch (tt) {
                case 1: hipLaunchKernelGGL((proto_fwd6_kernel<1, true>), dim3(gx, B), dim3(256), 0, stream, p); break;
                case 2: hipLaunchKernelGGL((proto_fwd6_kernel<2, true>), dim3(gx, B), dim3(256), 0, stream, p); break;
                case 3: hipLaunchKernelGGL((proto_fwd6_kernel<3, true>), dim3(gx, B), dim3(256), 0, stream, p); break;
                default: hipLaunchKernelGGL((proto_fwd6_kernel<4, true>), dim3(gx, B), dim3(256), 0, stream, p); break;
            }
        } else {
            switch (tt) {
                case 1: hipLaunchKernelGGL((proto_fwd_kernel<1, true>), dim3(gx, B), dim3(256), 0, stream, p); break;
                case 2: hipLaunchKernelGGL((proto_fwd_kernel<2, true>), dim3(gx, B), dim3(256), 0, stream, p); break;
                case 3: hipLaunchKernelGGL((proto_fwd_kernel<3, true>), dim3(gx, B), dim3(256), 0, stream, p); break;
                default: hipLaunchKernelGGL((proto_fwd_kernel<4, true>), dim3(gx, B), dim3(256), 0, stream, p); break;
            }
        }
    }
    PPF_LAUNCH_CHECK();
    return 0;
}

// Backward of ppf_proto_fwd given upstream grads of act_max (g_max) and act_full (g_full), either may be null.
// dtok rows are overwritten; dprotos [P][Dp] is accumulated (+=).
int ppf_proto_bwd(const float* tok, int64_t stride_b, int t0, int T, const float* protos, int B, int P, int Dp, int act_kind, float eps,
                  const float* dist_full, const float* g_full, const float* g_max, const int* argmax, float* dtok, int64_t dstride_b,
                  float* dprotos, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && P > 0 && P <= 8192 && Dp > 0 && T >= 1 && Dp <= 512, PPF_ERR_SHAPE, "ppf_proto_bwd: bad shape B=%d P=%d Dp=%d T=%d", B, P, Dp, T);
    PPF_CHECK_ARG(tok && protos && dist_full && (g_full || g_max) && (T == 1 || argmax || !g_max), PPF_ERR_ARG, "ppf_proto_bwd: null pointer");
    const int W = (P + 31) / 32;
    const size_t need = (size_t)B * T * W * sizeof(uint32_t);
    PPF_CHECK_ARG(dtok == nullptr || (workspace != nullptr && workspace_bytes >= need), PPF_ERR_ARG,
                  "ppf_proto_bwd: needs a ZEROED workspace of B*T*ceil(P/32)*4 = %zu bytes", need);
    ProtoBwdParams p;
    p.tok = tok; p.stride_b = stride_b; p.t0 = t0; p.T = T; p.protos = protos; p.B = B; p.P = P; p.Dp = Dp; p.act_kind = act_kind; p.eps = eps;
    p.dist_full = dist_full; p.g_full = g_full; p.g_max = g_max; p.argmax = (T == 1) ? nullptr : argmax; p.dtok = dtok; p.dstride_b = dstride_b; p.dprotos = dprotos;
    const int nj = (Dp + 63) / 64;
    auto run = [&](auto njc) {
        constexpr int NJ = decltype(njc)::value;
        if (dtok) {
            hipLaunchKernelGGL(proto_bwd_mark_kernel, dim3(W, B), dim3(256), 0, stream, p, (uint32_t*)workspace, W);
            static const int tok_nw = getenv("PPF_PROTO_TOK_NW") ? atoi(getenv("PPF_PROTO_TOK_NW")) : 1;
            if (W <= 64 && tok_nw == 1) hipLaunchKernelGGL((proto_bwd_tokens_kernel<NJ, 1>), dim3(B * T), dim3(64), 0, stream, p, (const uint32_t*)workspace, W);
            else hipLaunchKernelGGL((proto_bwd_tokens_kernel<NJ, 8>), dim3(B * T), dim3(512), 0, stream, p, (const uint32_t*)workspace, W);
        }
        if (dprotos) {
            constexpr int lds = PB_NW * PB_CHUNK * 6 + PB_NW * NJ * 64 * 4 + (PB_NW + 1) * 4;
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(proto_bwd_protos_kernel<NJ>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                attr_set = true;
            }
            hipLaunchKernelGGL((proto_bwd_protos_kernel<NJ>), dim3(P), dim3(1024), lds, stream, p);
        }
    };
    if (nj <= 1) run(std::integral_constant<int, 1>());
    else if (nj <= 2) run(std::integral_constant<int, 2>());
    else if (nj <= 3) run(std::integral_constant<int, 3>());
    else if (nj <= 4) run(std::integral_constant<int, 4>());
    else if (nj <= 6) run(std::integral_constant<int, 6>());
    else run(std::integral_constant<int, 8>());
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_sgemm(const float* A, const float* Bm, float* C, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbn, int64_t sbk, int ldc,
              float alpha, float beta, float* workspace, int64_t workspace_floats, hipStream_t stream);       // head.hip

// Backward of ppf_proto_fwd for T == 1 as two dense products (see the kernels above).  g = upstream gradient of the activations [B][P]
// (act_max == act_full when T == 1), dist [B][P].  dtok rows (tok layout) are overwritten when dtok != null; dprotos [P][Dp] is
// accumulated (+=) when dprotos != null.  workspace: ppf_proto_bwd_single_workspace(B, P, Dp) bytes.
size_t ppf_proto_bwd_single_workspace(int B, int P, int Dp) {
    const size_t m = (size_t)(B > P ? B : P);
    return ((size_t)B * P + m + 16 * m * (size_t)Dp) * sizeof(float);
}
int ppf_proto_bwd_single(const float* tok, int64_t stride_b, int t0, const float* protos, int B, int P, int Dp, int act_kind, float eps,
                         const float* dist, const float* g, float* dtok, int64_t dstride_b, float* dprotos, void* workspace, size_t workspace_bytes,
                         hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && P > 0 && Dp > 0, PPF_ERR_SHAPE, "ppf_proto_bwd_single: bad shape B=%d P=%d Dp=%d", B, P, Dp);
    PPF_CHECK_ARG(tok && protos && dist && g && workspace && workspace_bytes >= ppf_proto_bwd_single_workspace(B, P, Dp), PPF_ERR_ARG,
                  "ppf_proto_bwd_single: null pointer or workspace below ppf_proto_bwd_single_workspace() = %zu bytes", ppf_proto_bwd_single_workspace(B, P, Dp));
    float* G = (float*)workspace;
    float* sums = G + (size_t)B * P;
    float* gemm_ws = sums + (B > P ? B : P);
    const float* tok0 = tok + (size_t)t0 * Dp;
    if (dtok) {
        float* out = dtok + (size_t)t0 * Dp;
        hipLaunchKernelGGL(proto_single_gd_rows_kernel, dim3(B), dim3(256), 0, stream, dist, g, P, act_kind, eps, G, sums);
        PPF_LAUNCH_CHECK();
        // out[b][d] = -2 sum_p G[b][p] protos[p][d]
        int rc = ppf_sgemm(G, protos, out, B, Dp, P, P, 1, 1, Dp, (int)dstride_b, -2.0f, 0.0f, gemm_ws, (int64_t)16 * B * Dp, stream);
        if (rc) return rc;
        const int64_t n = (int64_t)B * Dp;
        hipLaunchKernelGGL(proto_single_fixup_kernel, dim3((int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, stream, out, dstride_b,
                           tok0, stride_b, sums, B, Dp);
        PPF_LAUNCH_CHECK();
    }
    if (dprotos) {
        hipLaunchKernelGGL(proto_single_gd_cols_kernel, dim3((P + 63) / 64), dim3(256), 0, stream, dist, g, B, P, act_kind, eps, G, sums);
        PPF_LAUNCH_CHECK();
        // dprotos[p][d] += -2 sum_b G[b][p] tok[b][d]
        int rc = ppf_sgemm(G, tok0, dprotos, P, Dp, B, 1, P, 1, stride_b, Dp, -2.0f, 1.0f, gemm_ws, (int64_t)16 * P * Dp, stream);
        if (rc) return rc;
        const int64_t n = (int64_t)P * Dp;
        hipLaunchKernelGGL(proto_single_fixup_kernel, dim3((int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, stream, dprotos, (int64_t)Dp,
                           protos, (int64_t)Dp, sums, P, Dp);
        PPF_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
