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

constexpr int PROTO_OCC = 3;     // workgroups per CU of the forward kernels at <= 3 token tiles
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
    const uint4* pre;      // proto_fwd6_kernel<.., PRE = true>: the prototypes split once per launch into three bf16 piece planes in FRAGMENT
                           // order [32-prototype block][16-step][piece][lane] x 16 bytes (proto_presplit_kernel), |p|^2 in pre_p2 [P rounded up to 32]
    const float* pre_p2;
    uint32_t t_magic;      // ceil(2^32 / T) (T >= 2): f / T == umulhi(f, t_magic) for the f < 2^11 the map stores divide
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
            // the transpose tile is private to the wave (the callers' barrier separates it from the operand stage it overlays): ordering
            // inside one wave needs no workgroup barrier -- LDS executes a wave's instructions in issue order
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (((lane & 31) >> 4) == half) {
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        xp[(lane & 15) * (ROWS + 1) + row] = pass == 0 ? acc[t][r] : activation(acc[t][r], p.act_kind, p.eps);
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
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
                        const int ff = min(f + e, total - 1), pp = (int)__umulhi((uint32_t)ff, p.t_magic), tt = ff - pp * p.T;   // ff / T
                        v[e] = xp[pp * (ROWS + 1) + tt];
                    }
                    if (f + 4 <= total) *reinterpret_cast<float4*>(obase + f) = make_float4(v[0], v[1], v[2], v[3]);
                    else for (int e = 0; f + e < total; ++e) obase[f + e] = v[e];
                }
            } else {
                for (int f = lane; f < total; f += 64) { const int pp = (int)__umulhi((uint32_t)f, p.t_magic), tt = f - pp * p.T; obase[f] = xp[pp * (ROWS + 1) + tt]; }
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

// Prototype operand of proto_fwd6_kernel<.., PRE>: one wave per 16-step of a 32-prototype block splits that lane's 8 contraction values
// (prototype = block * 32 + (lane & 31), k = step * 16 + 8 (lane >> 5) ..) into the three bf16 pieces and stores them where the main kernel's
// wave reads them with ONE coalesced 1 KiB load per piece (instead of 64 strided 32-byte reads + 44 VALU per 16-step and wave, repeated by
// every one of the B workgroups that share the block); |p|^2 as the plain fp32 sum.  grid = ceil(P / 32) blocks of 256 threads.
__global__ __launch_bounds__(256) void proto_presplit_kernel(const float* __restrict__ protos, int P, int Dp, uint4* __restrict__ pre, float* __restrict__ p2) {
    __shared__ float part[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5, l31 = lane & 31, KS = Dp / 16;
    const int row = min((int)blockIdx.x * 32 + l31, P - 1);
    const float* prow = protos + (size_t)row * Dp + 8 * hh;
    float acc = 0.f;
    for (int s = wave; s < KS; s += 4) {
        const float4 u0 = *reinterpret_cast<const float4*>(prow + s * 16), u1 = *reinterpret_cast<const float4*>(prow + s * 16 + 4);
        acc += u0.x * u0.x + u0.y * u0.y + u0.z * u0.z + u0.w * u0.w + u1.x * u1.x + u1.y * u1.y + u1.z * u1.z + u1.w * u1.w;
        const Split3 b0 = split3(u0.x, u0.y), b1 = split3(u0.z, u0.w), b2 = split3(u1.x, u1.y), b3 = split3(u1.z, u1.w);
        uint4* dst = pre + ((size_t)blockIdx.x * KS + s) * 192 + lane;
        dst[0] = make_uint4(b0.a, b1.a, b2.a, b3.a);
        dst[64] = make_uint4(b0.b, b1.b, b2.b, b3.b);
        dst[128] = make_uint4(b0.c, b1.c, b2.c, b3.c);
    }
    acc += __shfl_xor(acc, 32, 64);
    if (hh == 0) part[wave][l31] = acc;
    __syncthreads();
    if (threadIdx.x < 32) p2[blockIdx.x * 32 + threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

template <int TT, bool POOL, bool PRE>
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
    // PRE: the pieces come pre-split in fragment order (proto_presplit_kernel): three coalesced 16-byte loads per lane and 16-step
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const int KS = p.Dp / 16;
    const uint4* pre = PRE ? p.pre + ((size_t)(blockIdx.x * (PB / 32) + wave) * KS) * 192 + lane : nullptr;
    uint4 ra, rb, rc;
    auto pload3 = [&](int step) __attribute__((always_inline)) {
        const uint4* b = pre + (size_t)min(step, KS - 1) * 192;
        ra = b[0]; rb = b[64]; rc = b[128];
    };
    gload(0);
    if constexpr (PRE) pload3(0); else pload(0, q0, q1);
    for (int k0 = 0; k0 < p.Dp; k0 += BK6) {
        __syncthreads();                                             // everyone finished reading the previous chunk
        sstore();
        __syncthreads();
        if (k0 + BK6 < p.Dp) gload(k0 + BK6);
#pragma unroll
        for (int ks = 0; ks < BK6 / 16; ++ks) {
            if (k0 + ks * 16 < p.Dp) {
                bf16x8 pa, pb, pc;
                if constexpr (PRE) {
                    const uint4 ua = ra, ub = rb, uc = rc;
                    pload3(k0 / 16 + ks + 1);                          // next 16-step (clamped re-read at the end)
                    pa = __builtin_bit_cast(bf16x8, (u32x4){ua.x, ua.y, ua.z, ua.w});
                    pb = __builtin_bit_cast(bf16x8, (u32x4){ub.x, ub.y, ub.z, ub.w});
                    pc = __builtin_bit_cast(bf16x8, (u32x4){uc.x, uc.y, uc.z, uc.w});
                } else {
                    const float4 u0 = q0, u1 = q1;
                    pload(k0 + ks * 16 + 16, q0, q1);                // next 16-step (clamped re-read at the end)
                    p2p += u0.x * u0.x + u0.y * u0.y + u0.z * u0.z + u0.w * u0.w + u1.x * u1.x + u1.y * u1.y + u1.z * u1.z + u1.w * u1.w;
                    const Split3 b0 = split3(u0.x, u0.y), b1 = split3(u0.z, u0.w), b2 = split3(u1.x, u1.y), b3 = split3(u1.z, u1.w);
                    pa = __builtin_bit_cast(bf16x8, (u32x4){b0.a, b1.a, b2.a, b3.a});
                    pb = __builtin_bit_cast(bf16x8, (u32x4){b0.b, b1.b, b2.b, b3.b});
                    pc = __builtin_bit_cast(bf16x8, (u32x4){b0.c, b1.c, b2.c, b3.c});
                }
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
    const float p2 = PRE ? p.pre_p2[p0 + wave * 32 + l31] : p2p + __shfl_xor(p2p, 32, 64);
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
__global__ __launch_bounds__(256, TT <= 3 ? PROTO_OCC : 2) void proto_fwd_kernel(const ProtoFwdParams p) {
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
    const float* dist_full;    // [B][P][T]: the distances (map_is_act == 0) or the ACTIVATIONS the forward wrote (map_is_act == 1)
    int map_is_act;
    const float* g_full;       // [B][P][T] upstream grad of act_full, or null
    const float* g_rows;       // [B][ppc][T] or null: the same gradient in block form -- sample b's rows are the prototypes
    const long long* row_label;//   row_label[b]*ppc ... +ppc-1 (the PPC loss touches nothing else); added to g_full when both are given
    int ppc;
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

// The same derivative from the ACTIVATION a the forward wrote (training keeps only that map): with E = e^a = (d+1)/(d+eps),
// d + eps = (1-eps)/(E-1) and d + 1 = E (1-eps)/(E-1), so da/dd = -(E-1)^2 / ((1-eps) E); the clipped branch d == 0 is the forward's own
// value at d = 0 (same instruction sequence, bit-equal).  Relative error of the result ~ 2 ulp(a) E/(E-1): 1e-6 over the whole range.
__device__ __forceinline__ float dact_from_act(float a, int kind, float eps) {
    if (kind != 0) return a < 0.f ? -1.0f : 0.f;
    if (a >= activation(0.f, 0, eps)) return 0.f;
    const float em1 = expm1f(a);
    return -(em1 * em1) / ((1.0f - eps) * (em1 + 1.0f));
}
__device__ __forceinline__ float dact_map(const ProtoBwdParams& p, size_t o);

// G[b,p,t] = dL/dd
__device__ __forceinline__ float grad_d(const ProtoBwdParams& p, int b, int pi, int t, int amax, float gmax) {
    const size_t o = ((size_t)b * p.P + pi) * p.T + t;
    float g = p.g_full ? p.g_full[o] : 0.f;
    if (p.g_rows) {
        const int k = pi - (int)p.row_label[b] * p.ppc;
        if (k >= 0 && k < p.ppc) g += p.g_rows[((size_t)b * p.ppc + k) * p.T + t];
    }
    if (t == amax) g += gmax;
    if (g == 0.f) return 0.f;
    return g * dact_map(p, o);
}
__device__ __forceinline__ float dact_map(const ProtoBwdParams& p, size_t o) {
    const float v = p.dist_full[o];
    return p.map_is_act ? dact_from_act(v, p.act_kind, p.eps) : dact_dd(v, p.act_kind, p.eps);
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
    if (p.g_rows) {
        const int k0 = 32 * w - (int)p.row_label[b] * p.ppc;                    // block row of this word's first prototype
        if (k0 + np > 0 && k0 < p.ppc)
        for (int i = threadIdx.x; i < np * T; i += 256) {
            const int pl = i / T, t = i - pl * T, k = k0 + pl;
            if (k >= 0 && k < p.ppc && p.g_rows[((size_t)b * p.ppc + k) * T + t] != 0.f) atomicOr(&bm[((size_t)b * T + t) * W + w], 1u << pl);
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
// The same bitmap when there is no dense g_full: one workgroup per sample marks the arg-max token of every prototype and the non-zero
// entries of the sample's block rows (B workgroups instead of B * P / 32 nearly empty ones).
__global__ __launch_bounds__(256) void proto_bwd_mark_sparse_kernel(const ProtoBwdParams p, uint32_t* __restrict__ bm, int W) {
    const int b = blockIdx.x, T = p.T;
    if (p.g_max)
        for (int pi = threadIdx.x; pi < p.P; pi += 256) {
            const size_t bp = (size_t)b * p.P + pi;
            if (p.g_max[bp] != 0.f) atomicOr(&bm[((size_t)b * T + (p.argmax ? p.argmax[bp] : 0)) * W + (pi >> 5)], 1u << (pi & 31));
        }
    if (p.g_rows) {
        const int p0 = (int)p.row_label[b] * p.ppc;
        for (int i = threadIdx.x; i < p.ppc * T; i += 256) {
            const int k = i / T, t = i - k * T, pi = p0 + k;
            if (pi >= 0 && pi < p.P && p.g_rows[((size_t)b * p.ppc + k) * T + t] != 0.f) atomicOr(&bm[((size_t)b * T + t) * W + (pi >> 5)], 1u << (pi & 31));
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
                    if (p.g_rows) {
                        const int k = pi - (int)p.row_label[b] * p.ppc;
                        if (k >= 0 && k < p.ppc) g[u] += p.g_rows[((size_t)b * p.ppc + k) * p.T + t];
                    }
                    if (p.g_max) { const int am = p.argmax ? p.argmax[bp] : 0; if (t == am) g[u] += p.g_max[bp]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float G = 0.f;
                if (g[u] != 0.f) G = g[u] * dact_map(p, (size_t)bb[u] * PT + (size_t)pi * p.T + tt[u]);
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

// ---- prototype gradients, tiled form (round 4).  The per-prototype kernel above gathers one 1.5 KiB token row per non-zero dL/dd
// entry straight from memory: B*P arg-max rows + the label rows, 1.1 GB per step for the 2000 x 256 x 81 case although the token tensor
// itself is 32 MB.  Here the roles are swapped.
//  arg-max terms (proto_bwd_protos_tiled_kernel): a workgroup keeps the accumulators of 256 prototypes in registers (16 waves x 16
//    prototypes x Dp floats), walks a group of samples and has each sample's token block brought into LDS once (LDS-DMA, two chunk
//    images: the next chunk streams in while the current one is used), so a token row is fetched P/256 times instead of once per entry.
//    With  dprotos[p] = sum 2 G (p - x) = 2 (sum G) p - 2 sum G x  a wave accumulates  sum G x  (rows read from LDS) and  sum G  per
//    prototype; the finish kernel adds the sample groups' partials in ascending order and forms the difference.
//  block rows (proto_bwd_protos_rows_kernel, gradient given as g_rows): one workgroup per sample; wave k owns prototype label*ppc + k,
//    the token block is staged the same way and every token contributes 2 G (p - x) to a per-sample partial row; the finish kernel adds
//    the partial rows of a class's samples in ascending order.  These (sample, prototype) pairs are left out of the tiled kernel.
// Deterministic: every prototype's terms are added in a fixed order, no float atomics.  A dense g_full keeps the per-prototype kernel.
typedef __attribute__((address_space(3))) void pt_lds_void;
typedef const __attribute__((address_space(1))) void pt_gbl_void;
constexpr int PT_NW = 16;
// LDS-DMA copy of rows [r0, r0 + rows) of sample b's token block into image `dst` (1 KiB per wave instruction).
__device__ __forceinline__ void pt_issue_chunk(const ProtoBwdParams& p, int b, int r0, int rows, unsigned char* dst, int wave, int lane) {
    const int bytes = rows * p.Dp * 4;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.tok + (size_t)b * p.stride_b + (size_t)(p.t0 + r0) * p.Dp);
    for (int i = wave; i * 1024 < bytes; i += PT_NW) {
        const int off = min(i * 1024 + lane * 16, bytes - 16);                 // the last wave-row re-reads the final 16 bytes
        __builtin_amdgcn_global_load_lds((pt_gbl_void*)(src + off), (pt_lds_void*)(dst + i * 1024), 16, 0, 0);
    }
}
template <int NJ, int SLOTS>
__global__ __launch_bounds__(1024) void proto_bwd_protos_tiled_kernel(const ProtoBwdParams p, float* __restrict__ part, float* __restrict__ psum,
                                                                       int pt, int nsg, int spg, int R, int nchunks, int chunk_pad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pt_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroup -> (prototype tile, sample group): dispatch id L runs on XCD L % 8, and the pt workgroups that read the same samples
    // should share an L2 -- consecutive slots of ONE XCD take the tiles of a group (the token tensor then leaves HBM once, not pt times)
    int tile = blockIdx.x % pt, grp = blockIdx.x / pt;
    if (gridDim.x % 8 == 0 && nsg % 8 == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = j % pt; grp = xcd * (nsg >> 3) + j / pt;
    }
    const int pw0 = tile * (PT_NW * SLOTS) + wave * SLOTS;
    const int nvalid = min(SLOTS, p.P - pw0);                              // <= 0: this wave only helps with the copies
    const int b0 = grp * spg, b1 = min(p.B, b0 + spg);
    const int T = p.T, Dp = p.Dp;
    float acc[SLOTS][NJ];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[s][j] = 0.f;
    float sg = 0.f;                                                        // lane s < SLOTS: sum of G for prototype pw0 + s
    const int nsteps = (b1 - b0) * nchunks;
    if (nsteps > 0) pt_issue_chunk(p, b0, 0, min(R, T), pt_smem, wave, lane);
    int am = 0; float G = 0.f;
    for (int s = 0; s < nsteps; ++s) {
        const int bi = s / nchunks, c = s - bi * nchunks, b = b0 + bi;
        const int r0 = c * R, rows = min(R, T - r0);
        if (c == 0) {                                                      // new sample: coefficients of this wave's prototypes
            am = 0; G = 0.f;
            if (lane < nvalid) {
                const int pi = pw0 + lane;
                const size_t bp = (size_t)b * p.P + pi;
                const float gm = p.g_max[bp];
                bool mine = gm != 0.f;
                if (p.g_rows) { const int k = pi - (int)p.row_label[b] * p.ppc; mine = mine && !(k >= 0 && k < p.ppc); }
                if (mine) {
                    am = p.argmax[bp];
                    G = gm * dact_map(p, bp * T + am);
                    sg += G;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's share of chunk s has landed ...
        __syncthreads();                                                   // ... everyone's has, and image (s+1)&1 is no longer read
        if (s + 1 < nsteps) {
            const int bi2 = (s + 1) / nchunks, c2 = s + 1 - bi2 * nchunks;
            pt_issue_chunk(p, b0 + bi2, c2 * R, min(R, T - c2 * R), pt_smem + ((s + 1) & 1) * chunk_pad, wave, lane);
        }
        const float* X = reinterpret_cast<const float*>(pt_smem + (s & 1) * chunk_pad);
        if (nvalid > 0) {
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int t = __builtin_amdgcn_readlane(am, sl) - r0;
                const float Gs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(G), sl));
                if (Gs != 0.f && t >= 0 && t < rows) {
                    const float* xr = X + t * Dp;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) { const int d = lane + 64 * j; acc[sl][j] += Gs * (d < Dp ? xr[d] : 0.f); }
                }
            }
        }
    }
    if (nvalid > 0) {
        float* out = part + ((size_t)grp * p.P + pw0) * Dp;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl)
            if (sl < nvalid) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) { const int d = lane + 64 * j; if (d < Dp) out[(size_t)sl * Dp + d] = acc[sl][j]; }
            }
        if (lane < nvalid) psum[(size_t)grp * p.P + pw0 + lane] = sg;
    }
}
// Block rows of ONE sample per workgroup: wave k < ppc owns prototype label*ppc + k and writes  sum_t 2 G (p - x[b,t])  to rpart[b][k][:].
template <int NJ>
__global__ __launch_bounds__(1024) void proto_bwd_protos_rows_kernel(const ProtoBwdParams p, float* __restrict__ rpart, int R, int nchunks, int chunk_pad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pt_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x, T = p.T, Dp = p.Dp;
    const int pi = (int)p.row_label[b] * p.ppc + wave;
    const bool owner = wave < p.ppc && pi >= 0 && pi < p.P;
    pt_issue_chunk(p, b, 0, min(R, T), pt_smem, wave, lane);
    float pv[NJ], acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int d = lane + 64 * j; pv[j] = (owner && d < Dp) ? p.protos[(size_t)pi * Dp + d] : 0.f; acc[j] = 0.f; }
    const size_t bp = (size_t)b * p.P + (owner ? pi : 0);
    const int am = (owner && p.argmax && p.g_max) ? p.argmax[bp] : -1;
    const float gm = (owner && p.g_max) ? p.g_max[bp] : 0.f;
    for (int c = 0; c < nchunks; ++c) {
        const int r0 = c * R, rows = min(R, T - r0);
        // coefficients of this chunk's tokens (R <= 128 rows starting anywhere in a 64-block: three per lane), loaded ahead of the
        // wait so they travel with the chunk
        float Gt[3] = {0.f, 0.f, 0.f};
        const int th0 = r0 & ~63;
        if (owner) {
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const int t = th0 + 64 * h + lane;
                const bool in = t >= r0 && t < r0 + rows;
                float g = in ? p.g_rows[((size_t)b * p.ppc + wave) * T + t] : 0.f;
                if (in && t == am) g += gm;
                Gt[h] = (g != 0.f) ? 2.0f * g * dact_map(p, bp * T + t) : 0.f;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 1 < nchunks) pt_issue_chunk(p, b, (c + 1) * R, min(R, T - (c + 1) * R), pt_smem + ((c + 1) & 1) * chunk_pad, wave, lane);
        if (!owner) continue;
        const float* X = reinterpret_cast<const float*>(pt_smem + (c & 1) * chunk_pad);
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            unsigned long long m = __ballot(Gt[h] != 0.f);
            while (m) {
                const int l = __builtin_ctzll(m); m &= m - 1;
                const float Gs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Gt[h]), l));
                const float* xr = X + (th0 + 64 * h + l - r0) * Dp;
#pragma unroll
                for (int j = 0; j < NJ; ++j) { const int d = lane + 64 * j; acc[j] += Gs * (pv[j] - (d < Dp ? xr[d] : 0.f)); }
            }
        }
    }
    if (owner) {
        float* out = rpart + ((size_t)b * p.ppc + wave) * Dp;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int d = lane + 64 * j; if (d < Dp) out[d] = acc[j]; }
    }
}
// dprotos[p][:] += 2 ((sum_g psum[g][p]) protos[p][:] - sum_g part[g][p][:])  (sample groups ascending; skipped when part == null)
//               +  sum over the samples b of class p / ppc, ascending, of rpart[b][p % ppc][:]   (skipped when rpart == null)
__global__ __launch_bounds__(128) void proto_bwd_protos_finish_kernel(const float* __restrict__ part, const float* __restrict__ psum,
                                                                      const float* __restrict__ rpart, const long long* __restrict__ label, int ppc,
                                                                      const float* __restrict__ protos, float* __restrict__ dprotos, int B, int P, int Dp, int SG) {
    const int pi = blockIdx.x, lane = threadIdx.x & 63;
    const size_t n = (size_t)P * Dp;
    float add[3] = {0.f, 0.f, 0.f};                                        // d = threadIdx.x + 128 q  (Dp <= 384)
    if (part) {
        float sgv = 0.f;
        for (int g = 0; g < SG; ++g) sgv += psum[(size_t)g * P + pi];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int d = threadIdx.x + 128 * q;
            if (d < Dp) {
                const size_t i = (size_t)pi * Dp + d;
                float a = 0.f;
                int g = 0;
                for (; g + 8 <= SG; g += 8) {                                  // eight loads in flight, added in ascending order
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(g + u) * n + i];
#pragma unroll
                    for (int u = 0; u < 8; ++u) a += v[u];
                }
                for (; g < SG; ++g) a += part[(size_t)g * n + i];
                add[q] = 2.0f * (sgv * protos[i] - a);
            }
        }
    }
    if (rpart) {
        const int cls = pi / ppc, k = pi - cls * ppc;
        for (int b0 = 0; b0 < B; b0 += 64) {
            unsigned long long m = __ballot(b0 + lane < B && (int)label[b0 + lane] == cls);
            while (m) {
                const int b = b0 + __builtin_ctzll(m); m &= m - 1;
#pragma unroll
                for (int q = 0; q < 3; ++q) { const int d = threadIdx.x + 128 * q; if (d < Dp) add[q] += rpart[((size_t)b * ppc + k) * Dp + d]; }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) { const int d = threadIdx.x + 128 * q; if (d < Dp) dprotos[(size_t)pi * Dp + d] += add[q]; }
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

// Geometry of the tiled prototype-gradient form (proto_bwd_protos_tiled_kernel); ok == false: the per-prototype gather kernel runs.
struct ProtoTiled { bool ok; int slots, pt, sg, spg, R, nchunks, chunk_pad; size_t ws_bytes; };
ProtoTiled proto_tiled_geometry(int B, int T, int P, int Dp, int64_t stride_b, int t0) {
    ProtoTiled g{};
    if (T < 2 || Dp % 4 || Dp > 384 || stride_b % 4 || ((int64_t)t0 * Dp) % 4) return g;
    const int row_bytes = Dp * 4, rmax = min(128, (76 * 1024) / row_bytes);  // two images of <= 76 KiB, <= 128 rows each
    g.slots = 16;                                                            // 16 x 6 accumulator registers per lane at Dp = 384
    g.nchunks = (T + rmax - 1) / rmax;
    g.R = (T + g.nchunks - 1) / g.nchunks;
    g.chunk_pad = ((g.R * row_bytes + 1023) / 1024) * 1024;
    g.pt = (P + PT_NW * g.slots - 1) / (PT_NW * g.slots);
    int sg = (256 + g.pt - 1) / g.pt;                  // about one workgroup per CU
    if (sg > B) sg = B;
    g.spg = (B + sg - 1) / sg;
    g.sg = (B + g.spg - 1) / g.spg;
    g.ws_bytes = ((size_t)g.sg * P * Dp + (size_t)g.sg * P + (size_t)B * PT_NW * Dp) * sizeof(float);     // part, psum, rpart (ppc <= 16)
    g.ok = true;
    return g;
}

}  // namespace

extern "C" {

// tokens: fp32 rows of Dp values; sample b's token i is at tok + b*stride_b + (t0+i)*Dp, T tokens per sample
// (T == 1: the global/cls branch, no pooling).  protos [P][Dp].  act_kind 0 = 'log', 1 = 'linear'.
// Outputs: act_max [B][P], argmax [B][P] (T > 1), optional dist_full / act_full [B][P][T].
// workspace (optional, ppf_proto_fwd_workspace(P, Dp) bytes, any content): the pre-split prototype planes of the pooled branch; NULL or too
// small: every workgroup splits its prototype rows itself (same results: the pieces are the same numbers, |p|^2 is summed in another order).
size_t ppf_proto_fwd_workspace(int P, int Dp) {
    const size_t rows = (size_t)((P + PB - 1) / PB) * PB;
    return rows * (size_t)Dp * 6 + rows * sizeof(float);
}

int ppf_proto_fwd(const float* tok, int64_t stride_b, int t0, int T, const float* protos, int B, int P, int Dp, int act_kind, float eps,
                  float* act_max, int* argmax, float* dist_full, float* act_full, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && P > 0 && Dp > 0 && Dp % 4 == 0 && T >= 1 && T <= 128, PPF_ERR_SHAPE, "ppf_proto_fwd: bad shape B=%d P=%d Dp=%d T=%d", B, P, Dp, T);
    PPF_CHECK_ARG(tok && protos && act_max && (T == 1 || argmax), PPF_ERR_ARG, "ppf_proto_fwd: null pointer");
    ProtoFwdParams p;
    p.tok = tok; p.stride_b = stride_b; p.t0 = t0; p.T = T; p.protos = protos; p.B = B; p.P = P; p.Dp = Dp; p.act_kind = act_kind; p.eps = eps;
    p.act_max = act_max; p.argmax = argmax; p.dist_full = dist_full; p.act_full = act_full;
    p.t_magic = T >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)T - 1) / (uint64_t)T) : 0u;
    const int gx = (P + PB - 1) / PB;
    // algorithmic work: the (B T) x Dp . Dp x P contraction; tokens + prototypes in, the maps that are asked for + max / arg-max out
    PpfProbeScope probe(PPF_PROBE_PROTO_FWD, stream, 2.0 * B * T * (double)Dp * P,
                        4.0 * ((double)B * T * Dp + (double)P * Dp + (double)B * P * T * ((dist_full ? 1 : 0) + (act_full ? 1 : 0)) + 2.0 * B * P));
    // the split-bf16 contraction (six exact products); Dp not a multiple of 16: the fp32-MFMA kernel
    const bool x6 = Dp % 16 == 0;
    const bool pre = x6 && T > 1 && workspace && workspace_bytes >= ppf_proto_fwd_workspace(P, Dp) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
    if (pre) {
        p.pre = reinterpret_cast<const uint4*>(workspace);
        float* p2 = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + (size_t)gx * PB * Dp * 6);
        p.pre_p2 = p2;
        hipLaunchKernelGGL(proto_presplit_kernel, dim3(gx * (PB / 32)), dim3(256), 0, stream, protos, P, Dp, reinterpret_cast<uint4*>(workspace), p2);
    } else { p.pre = nullptr; p.pre_p2 = nullptr; }
    if (T == 1) {
        if (x6) hipLaunchKernelGGL((proto_fwd6_kernel<2, false, false>), dim3(gx, (B + 63) / 64), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((proto_fwd_kernel<2, false>), dim3(gx, (B + 63) / 64), dim3(256), 0, stream, p);
    } else {
        const int tt = (T + 31) / 32;
        if (x6) {
            switch (tt) {
                case 1: if (pre) hipLaunchKernelGGL((proto_fwd6_kernel<1, true, true>), dim3(gx, B), dim3(256), 0, stream, p);
                        else hipLaunchKernelGGL((proto_fwd6_kernel<1, true, false>), dim3(gx, B), dim3(256), 0, stream, p);
                        break;
                case 2: if (pre) hipLaunchKernelGGL((proto_fwd6_kernel<2, true, true>), dim3(gx, B), dim3(256), 0, stream, p);
                        else hipLaunchKernelGGL((proto_fwd6_kernel<2, true, false>), dim3(gx, B), dim3(256), 0, stream, p);
                        break;
                case 3: if (pre) hipLaunchKernelGGL((proto_fwd6_kernel<3, true, true>), dim3(gx, B), dim3(256), 0, stream, p);
                        else hipLaunchKernelGGL((proto_fwd6_kernel<3, true, false>), dim3(gx, B), dim3(256), 0, stream, p);
                        break;
                default: if (pre) hipLaunchKernelGGL((proto_fwd6_kernel<4, true, true>), dim3(gx, B), dim3(256), 0, stream, p);
                        else hipLaunchKernelGGL((proto_fwd6_kernel<4, true, false>), dim3(gx, B), dim3(256), 0, stream, p);
                        break;
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

// Backward of ppf_proto_fwd given upstream grads of act_max (g_max) and act_full (g_full and / or its block form g_rows), any may be null.
// dtok rows are overwritten; dprotos [P][Dp] is accumulated (+=).
static int proto_bwd_launch(ProtoBwdParams p, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const int B = p.B, P = p.P, Dp = p.Dp, T = p.T;
    const int W = (P + 31) / 32;
    const size_t need = p.dtok ? (((size_t)B * T * W * sizeof(uint32_t) + 255) & ~(size_t)255) : 0;
    PPF_CHECK_ARG(p.dtok == nullptr || (workspace != nullptr && workspace_bytes >= (size_t)B * T * W * sizeof(uint32_t)), PPF_ERR_ARG,
                  "ppf_proto_bwd: needs a ZEROED workspace of B*T*ceil(P/32)*4 = %zu bytes", (size_t)B * T * W * sizeof(uint32_t));
    // prototype gradients: the tiled form when there is no dense g_full and the caller's workspace has room for its partial sums
    ProtoTiled tg = (p.dprotos && !p.g_full && (!p.g_max || p.argmax)) ? proto_tiled_geometry(B, T, P, Dp, p.stride_b, p.t0) : ProtoTiled{};
    if (tg.ok && p.g_rows && p.ppc > PT_NW) tg.ok = false;
    if (tg.ok && !(workspace && workspace_bytes >= need + tg.ws_bytes && ((uintptr_t)p.tok & 15) == 0)) tg.ok = false;
    const int nj = (Dp + 63) / 64;
    auto run = [&](auto njc) {
        constexpr int NJ = decltype(njc)::value;
        if (p.dtok) {
            if (p.g_full) hipLaunchKernelGGL(proto_bwd_mark_kernel, dim3(W, B), dim3(256), 0, stream, p, (uint32_t*)workspace, W);
            else hipLaunchKernelGGL(proto_bwd_mark_sparse_kernel, dim3(B), dim3(256), 0, stream, p, (uint32_t*)workspace, W);
            if (W <= 64) hipLaunchKernelGGL((proto_bwd_tokens_kernel<NJ, 1>), dim3(B * T), dim3(64), 0, stream, p, (const uint32_t*)workspace, W);
            else hipLaunchKernelGGL((proto_bwd_tokens_kernel<NJ, 8>), dim3(B * T), dim3(512), 0, stream, p, (const uint32_t*)workspace, W);
        }
        if (p.dprotos && tg.ok) {
            if constexpr (NJ <= 6) {
                const int lds = 2 * tg.chunk_pad;
                static int attr_lds = 0;
                if (lds > attr_lds) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(proto_bwd_protos_tiled_kernel<NJ, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(proto_bwd_protos_rows_kernel<NJ>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    attr_lds = lds;
                }
                float* part = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + need);
                float* psum = part + (size_t)tg.sg * P * Dp;
                float* rpart = psum + (size_t)tg.sg * P;
                if (p.g_max)
                    hipLaunchKernelGGL((proto_bwd_protos_tiled_kernel<NJ, 16>), dim3(tg.pt * tg.sg), dim3(1024), lds, stream, p, part, psum, tg.pt, tg.sg,
                                       tg.spg, tg.R, tg.nchunks, tg.chunk_pad);
                if (p.g_rows)
                    hipLaunchKernelGGL((proto_bwd_protos_rows_kernel<NJ>), dim3(B), dim3(1024), lds, stream, p, rpart, tg.R, tg.nchunks, tg.chunk_pad);
                hipLaunchKernelGGL(proto_bwd_protos_finish_kernel, dim3(P), dim3(128), 0, stream, p.g_max ? part : nullptr, psum, p.g_rows ? rpart : nullptr,
                                   p.row_label, p.ppc, p.protos, p.dprotos, B, P, Dp, tg.sg);
            }
        } else if (p.dprotos) {
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

int ppf_proto_bwd(const float* tok, int64_t stride_b, int t0, int T, const float* protos, int B, int P, int Dp, int act_kind, float eps,
                  const float* dist_full, int map_is_act, const float* g_full, const float* g_max, const int* argmax, float* dtok, int64_t dstride_b,
                  float* dprotos, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && P > 0 && P <= 8192 && Dp > 0 && T >= 1 && Dp <= 512, PPF_ERR_SHAPE, "ppf_proto_bwd: bad shape B=%d P=%d Dp=%d T=%d", B, P, Dp, T);
    PPF_CHECK_ARG(tok && protos && dist_full && (g_full || g_max) && (T == 1 || argmax || !g_max), PPF_ERR_ARG, "ppf_proto_bwd: null pointer");
    ProtoBwdParams p;
    p.tok = tok; p.stride_b = stride_b; p.t0 = t0; p.T = T; p.protos = protos; p.B = B; p.P = P; p.Dp = Dp; p.act_kind = act_kind; p.eps = eps;
    p.dist_full = dist_full; p.map_is_act = map_is_act != 0; p.g_full = g_full; p.g_rows = nullptr; p.row_label = nullptr; p.ppc = 0; p.g_max = g_max;
    p.argmax = (T == 1) ? nullptr : argmax; p.dtok = dtok; p.dstride_b = dstride_b; p.dprotos = dprotos;
    return proto_bwd_launch(p, workspace, workspace_bytes, stream);
}

// The same backward with the activation-map gradient in the block form the PPC loss produces (protopformer.py:259-288: only the ppc
// prototypes of a sample's own class receive a gradient): g_rows [B][ppc][T] holds dL/d act_full[b][label[b]*ppc + k][t], every other
// entry of the (B,P,T) gradient is zero and is never materialised, scanned or stored.  Same workspace contract as ppf_proto_bwd.
int ppf_proto_bwd_rows(const float* tok, int64_t stride_b, int t0, int T, const float* protos, int B, int P, int Dp, int act_kind, float eps,
                       const float* dist_full, int map_is_act, const float* g_rows, const void* label_i64, int ppc, const float* g_max, const int* argmax,
                       float* dtok, int64_t dstride_b, float* dprotos, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && P > 0 && P <= 8192 && Dp > 0 && T >= 2 && Dp <= 512 && ppc >= 1 && ppc <= P, PPF_ERR_SHAPE,
                  "ppf_proto_bwd_rows: bad shape B=%d P=%d Dp=%d T=%d ppc=%d", B, P, Dp, T, ppc);
    PPF_CHECK_ARG(tok && protos && dist_full && g_rows && label_i64 && (argmax || !g_max), PPF_ERR_ARG, "ppf_proto_bwd_rows: null pointer");
    ProtoBwdParams p;
    p.tok = tok; p.stride_b = stride_b; p.t0 = t0; p.T = T; p.protos = protos; p.B = B; p.P = P; p.Dp = Dp; p.act_kind = act_kind; p.eps = eps;
    p.dist_full = dist_full; p.map_is_act = map_is_act != 0; p.g_full = nullptr; p.g_rows = g_rows; p.row_label = (const long long*)label_i64; p.ppc = ppc; p.g_max = g_max;
    p.argmax = argmax; p.dtok = dtok; p.dstride_b = dstride_b; p.dprotos = dprotos;
    return proto_bwd_launch(p, workspace, workspace_bytes, stream);
}

// Bytes of workspace ppf_proto_bwd wants: the zeroed bitmap (when token gradients are asked for) followed by the scratch of the tiled
// prototype-gradient form (need not be initialised; a smaller workspace selects the per-prototype gather kernel instead).
size_t ppf_proto_bwd_workspace(int B, int T, int P, int Dp, int want_dtok, int want_dprotos) {
    const size_t bm = want_dtok ? (((size_t)B * T * ((P + 31) / 32) * sizeof(uint32_t) + 255) & ~(size_t)255) : 0;
    const ProtoTiled tg = want_dprotos ? proto_tiled_geometry(B, T, P, Dp, 0, 0) : ProtoTiled{};
    return bm + (tg.ok ? tg.ws_bytes : 0);
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
