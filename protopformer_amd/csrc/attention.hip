// Multi-head attention with the reference's policy softmax (tools/deit_models_attn.py:29-60), gfx950.
//
// Sequence length is tiny (N = 197 / 196, <= 224), so one wavefront keeps a whole 32-query x N-key score
// strip in registers: no online-softmax rescaling, one pass.  All contractions run on
// v_mfma_f32_32x32x16_bf16 with the "swapped" operand order (first = keys / d, second = queries), so every
// lane owns ONE query row (lane&31) and the softmax reductions are in-lane + one cross-half shuffle.
// K and V of one (batch, head) are staged once per workgroup in LDS as [key][128 B] rows with an XOR chunk
// swizzle that is conflict-free for BOTH access styles used here:
//   ds_read_b128  (K as MFMA A operand for Q.K^T, contraction over d)
//   ds_read_b64_tr_b16 (V / K / Q / dO read "transposed": contraction over keys or queries)
// Kernels:
//   attn_fwd      O = softmax_policy(Q K^T * scale) V, saves row max and 1/(sum+eps) per (b,h,q)
//   attn_headmean mean over heads of the probabilities -> (B, N, NP) fp32 (rollout input), recomputed from
//                 the saved statistics (no B*H*N*N tensor ever exists)
//   attn_bwd_stream / attn_bwd_onepass   dQ, dK, dV in one launch, every (query tile, key tile) pair visited once
#include "ppf_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr float SOFTMAX_EPS = 1e-6f;     // deit:29 eps
constexpr float LOG2E = 1.44269504088896340736f;

struct AttnParams {
    const bf16_t* qkv;     // [B*N][ld] : q | k | v at column offsets 0, D, 2D (+ h*HD)
    int ld;                // 3*D
    bf16_t* out;           // [B*N][D]
    const float* policy;   // [B][N] in {0,1} or null (all ones)
    float* rowmax;         // [B][H][N]
    float* zinv;           // [B][H][N]   1/(sum+eps)
    float* headmean;       // [B][N][NP] fp32
    int NP;
    // backward
    const bf16_t* dout;    // [B*N][D]
    bf16_t* dqkv;          // [B*N][ld]
    float* delta;          // [B][H][N]
    int B, H, N, D;
    int self_keep;         // 1: a masked query still attends to itself (DeiT); 0: CaiT class attention
    float scale;
    float eps_c;           // eps / N_ref: the +eps/N term of the policy softmax; N_ref = tokens BEFORE reservation (compacted blocks pass it)
};

__device__ __forceinline__ int kswz(int row) {
    const int u = row >> 1;
    return ((u & 1) << 2) | (((u >> 2) & 1) << 1) | ((u >> 1) & 1);
}
__device__ __forceinline__ int row_off(int row, int c16) { return row * 128 + ((c16 ^ kswz(row)) << 4); }

// MFMA A/B fragment, contraction-contiguous rows: lane -> row base+(lane&31), 8 values at d = ks*16 + (lane>>5)*8
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* tile, int base, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + row_off(base + (lane & 31), ks * 2 + (lane >> 5)));
}
// MFMA A fragment read transposed: output index = d (dbase + lane&31), contraction slots = rows.
// Slot order matches the natural register order of a swapped-operand score tile:
//   slot (h, jj) <-> row kb + (jj&3) + 8*(jj>>2) + 4*h
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* tile, int kb, int dbase, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int s = lane & 15, g16 = (lane >> 4) & 1, h = lane >> 5;
    const int d = dbase + 16 * g16 + 4 * (s & 3);
    const int r0 = kb + 4 * h + (s >> 2), r1 = r0 + 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + row_off(r0, d >> 3) + ((d & 7) << 1)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + row_off(r1, d >> 3) + ((d & 7) << 1)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int o) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    u32x4 u = {pack_bf16x2(v[o], v[o + 1]), pack_bf16x2(v[o + 2], v[o + 3]), pack_bf16x2(v[o + 4], v[o + 5]), pack_bf16x2(v[o + 6], v[o + 7])};
    return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ int reg_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Stage ROWS rows (HD bf16 each) from src (row stride ld) into a swizzled LDS tile.  Rows beyond nvalid are
// zero-filled, or (CLAMP) replicate the last valid row so that padded keys produce finite scores that never
// exceed the true row maximum (they are then removed by keep = 0).
// Two steps so that a kernel can put ALL its global loads in flight before the first LDS write: the one-loop form compiled to
// load -> s_waitcnt vmcnt(0) -> ds_write per iteration, i.e. 7 dependent HBM round trips per workgroup before any MFMA
// (SQ_WAIT_ANY 43-57 % of the wave cycles, profiles/r2_attn_pmc.txt).
template <int HD, int ROWS, int NTHR>
struct Stage {
    static constexpr int CH = HD / 8, ITER = (ROWS * CH + NTHR - 1) / NTHR;
    uint4 v[ITER];
    template <bool CLAMP>
    __device__ __forceinline__ void load(const bf16_t* src, int ld, int row_begin, int nvalid, int tid) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * NTHR, r = i / CH, c = i - r * CH;
            int gr = row_begin + r;
            if (CLAMP) gr = min(gr, nvalid - 1);
            v[it] = make_uint4(0, 0, 0, 0);
            if (i < ROWS * CH && gr < nvalid) v[it] = *reinterpret_cast<const uint4*>(src + (size_t)gr * ld + c * 8);
        }
    }
    __device__ __forceinline__ void store(unsigned char* tile, int tid) const {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * NTHR, r = i / CH, c = i - r * CH;
            if (i < ROWS * CH) *reinterpret_cast<uint4*>(tile + row_off(r, c)) = v[it];
        }
    }
};
template <int HD, bool CLAMP, int NTHR = 256>
__device__ __forceinline__ void stage_rows(unsigned char* tile, const bf16_t* src, int ld, int row_begin, int rows, int nvalid, int tid) {
    constexpr int CH = HD / 8;
    for (int i = tid; i < rows * CH; i += NTHR) {
        const int r = i / CH, c = i - r * CH;
        uint4 v = make_uint4(0, 0, 0, 0);
        int gr = row_begin + r;
        if (CLAMP) gr = min(gr, nvalid - 1);
        if (gr < nvalid) v = *reinterpret_cast<const uint4*>(src + (size_t)gr * ld + c * 8);
        *reinterpret_cast<uint4*>(tile + row_off(r, c)) = v;
    }
}

// One workgroup per (batch, head): K/V are staged ONCE and shared by all query tiles (8 waves when N > 128: the kernels are
// bound by the K/V/Q traffic and its latency, not by the MFMAs -- profiles/r1_step1_kernel_stats.txt).
template <int NT> struct Geo { static constexpr int NW = NT > 4 ? 8 : 4, NTHR = NW * 64; };

// ------------------------------------------------------------------------------------------------ forward
template <int HD, int NT>
__global__ __launch_bounds__(Geo<NT>::NTHR, Geo<NT>::NW == 8 ? 4 : 2) void attn_fwd_kernel(const AttnParams p) {
    constexpr int NW = Geo<NT>::NW, NTHR = Geo<NT>::NTHR;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * NT * 32 * 128 + NT * 32 * 4];
    unsigned char* tK = lds;
    unsigned char* tV = lds + NT * 32 * 128;
    float* pol = reinterpret_cast<float*>(lds + 2 * NT * 32 * 128);     // keep flag per key (0 for padding)
    constexpr int DT = (HD + 31) / 32, KS = HD / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, N = p.N;
    const bf16_t* base = p.qkv + (size_t)b * N * p.ld + h * HD;
    const float c = p.eps_c;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int q = q0 + (lane & 31), qc = min(q, N - 1);
    bf16x8 qf[KS];
    {
        Stage<HD, NT * 32, NTHR> sk, sv;           // every global load of the prologue in flight before the first LDS write
        sk.template load<true>(base + p.D, p.ld, 0, N, tid);
        sv.template load<false>(base + 2 * p.D, p.ld, 0, N, tid);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)qc * p.ld + ks * 16 + hh * 8);
        for (int i = tid; i < NT * 32; i += NTHR) pol[i] = (i < N) ? (p.policy ? p.policy[(size_t)b * N + i] : 1.0f) : 0.0f;
        sk.store(tK, tid);
        sv.store(tV, tid);
    }
    __syncthreads();
    if (q0 >= N) return;
    const int qself = p.self_keep ? q : -1;
    // Two passes over the keys instead of a 7-tile score strip in registers (112 VGPRs, one workgroup per CU): pass 1 only
    // finds the row maximum, pass 2 recomputes each 32-key score tile, exponentiates it and feeds P.V at once.  The MFMA pipe
    // is idle most of the time here; ~116 VGPRs let two workgroups share a CU so that one stages K/V while the other computes.
    float mraw = -INFINITY;                      // max of the unscaled scores (scale > 0)
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tK, t * 32, ks, lane), qf[ks], s, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) mraw = fmaxf(mraw, s[r]);
    }
    mraw = fmaxf(mraw, __shfl_xor(mraw, 32, 64));
    const float mx = mraw * p.scale;
    const float c1 = p.scale * 1.44269504088896340736f, m1 = mraw * c1;     // exp(x*scale - mx) = exp2(x*c1 - m1)
    float sum = 0.f;
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tK, t * 32, ks, lane), qf[ks], s, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int key0 = t * 32 + 8 * g + 4 * hh;
            const float4 kp = *reinterpret_cast<const float4*>(pol + key0);
            const float keep[4] = {kp.x, kp.y, kp.z, kp.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float k = (key0 + i == qself) ? 1.0f : keep[i];
                const float e = __builtin_amdgcn_exp2f(s[4 * g + i] * c1 - m1) * k;
                sum += e;
                s[4 * g + i] = e + c;                 // unnormalised probability (+eps/N); padded keys hit zero V rows
            }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8 pf = pack8(s, 8 * st);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(tV, t * 32 + 16 * st, dt * 32, lane), pf, o[dt], 0, 0, 0);
        }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float zi = 1.0f / (sum + SOFTMAX_EPS);
    if (hh == 0 && q < N) {
        const size_t si = ((size_t)b * p.H + h) * N + q;
        p.rowmax[si] = mx;
        p.zinv[si] = zi;
    }
    if (q < N) {
        bf16_t* orow = p.out + ((size_t)b * N + q) * p.D + h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * hh;
                if (d < HD)
                    *reinterpret_cast<uint2*>(orow + d) = make_uint2(pack_bf16x2(o[dt][4 * g] * zi, o[dt][4 * g + 1] * zi),
                                                                     pack_bf16x2(o[dt][4 * g + 2] * zi, o[dt][4 * g + 3] * zi));
            }
    }
}

// ------------------------------------------------------------------------------------------- forward + head mean, 16-row geometry
// Round 3: the forward pass and the head-mean map (the rollout's input, deit:104) in ONE launch.  A workgroup of seven waves owns up to
// 112 queries of one sample and loops over the heads; a wave owns 16 queries and keeps the whole score strip of a head in registers
// (v_mfma_f32_16x16x32_bf16, swapped operands: a lane = one query, four consecutive keys per 16-key tile -> 4 x NT16 = 52 registers at
// N = 197), so the softmax is one pass and the probabilities are added to the head-mean accumulators (another 52 registers) on the way to
// the P.V product.  Q.K^T is computed once (attn_headmean_kernel recomputed it from the saved statistics and re-read q / k: 202 MB per
// layer), and N = 197 is tiled as 13 x 16 = 208 keys / queries instead of 7 x 32 = 224.  K and V of the current head sit in LDS
// in one of two buffers (2 x 52 KB) filled by LDS-DMA: head h + 1 lands while head h is multiplied, one barrier per head, one 7-wave
// workgroup per CU.  V is read transposed (ds_read_b64_tr_b16) with the key order of the packed probability tiles.
// (A first version staged through registers with two barriers per head: 110-180 us, all of it staging latency.)
constexpr int F16_WAVES = 7, F16_NTHR = F16_WAVES * 64;       // 7 x 16 = 112 queries per workgroup: N = 197 -> two workgroups per sample (7 + 6 blocks)
template <int HD, int NT16>
struct Fwd16 { static constexpr int ROWS = NT16 * 16, BUF = 2 * ROWS * 128, LDS = 2 * BUF + ROWS * 4; };

// A operand of the P.V product: output index d = dbase + (lane & 15); contraction slots 8 (lane >> 4) + j <-> keys kb + 4 (lane >> 4) + j
// (j < 4) and kb + 16 + 4 (lane >> 4) + (j - 4): the register order of two adjacent 16-key score tiles
__device__ __forceinline__ bf16x8 frag_tr16(const unsigned char* tile, int kb, int dbase, int lane, bool second) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int s = lane & 15, grp = lane >> 4;
    const int d = dbase + 4 * (s & 3);
    const int r0 = kb + 4 * grp + (s >> 2), r1 = r0 + 16;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + row_off(r0, d >> 3) + ((d & 7) << 1)));
    bf16x4 hi = lo;
    if (second) hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + row_off(r1, d >> 3) + ((d & 7) << 1)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// TAIL (round 6, no policy only): N > 16 (NT16 - 1), i.e. only the LAST key tile holds padded keys -- the softmax then runs on the packed fp32
// pipe with no per-element select: exponent and row sum as v_pk_fma_f32 / v_pk_add_f32, the P.V operand is bf16(e) itself, and the policy
// softmax's +eps/N enters the head-mean map as ONE per-row constant (eps/N) sum_h 1/(sum_h + eps) added at the store.  The same term of
// the OUTPUT, (eps/N)/(sum + eps) sum_j v_j, is dropped there: without a policy sum >= 1 (the row maximum contributes e = 1), so it is
// <= N (eps/N) max|v| = 1e-6 max|v| -- 4 000 x below the bf16 rounding of `out`.  With a policy a row's kept mass can be arbitrarily small
// (its maximum may sit on a masked key), so that instantiation keeps the exact per-element form.
template <int HD, int NT16, bool POLICY, bool TAIL>
__global__ __launch_bounds__(F16_NTHR, 1) void attn_fwd16_kernel(const AttnParams p) {
    static_assert(!(POLICY && TAIL), "the packed softmax is the no-policy path");
    static_assert(HD == 64, "128-byte rows");
    constexpr int KS = HD / 32, DB = HD / 16, ROWS = NT16 * 16, BUF = Fwd16<HD, NT16>::BUF;
    constexpr int NPIECE = 2 * ROWS / 8, PPW = (NPIECE + F16_WAVES - 1) / F16_WAVES;      // 1 KiB pieces (8 rows) of K then V; pieces per wave
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];
    float* pol = reinterpret_cast<float*>(lds16 + 2 * BUF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, l15 = lane & 15;
    const int N = p.N;
    // the sample's 16-query blocks are dealt out evenly to the workgroups of this sample (13 blocks on 2 workgroups: 7 + 6).  The two
    // workgroups of a sample read the same K / V: they are placed on the SAME XCD (dispatch ids L and L + 8; observed placement L % 8,
    // speed only) so that the second read is served by that XCD's L2 instead of crossing the fabric again
    int b = blockIdx.y, x = blockIdx.x;
    if (gridDim.x == 2 && (gridDim.y & 7) == 0) {
        const int L = blockIdx.x + 2 * blockIdx.y, r = L & 15;
        b = (L >> 4) * 8 + (r & 7); x = r >> 3;
    }
    const int nb = (N + 15) / 16, nw = gridDim.x, per = nb / nw, rem = nb % nw;
    const int first = x * per + min(x, rem), cnt = per + (x < rem ? 1 : 0);
    const bool active = wave < cnt;
    const int q = (first + wave) * 16 + l15, qc = min(q, N - 1);
    const int qself = p.self_keep ? q : -1;
    const float c = p.eps_c, c1 = p.scale * LOG2E;
    // K / V of a head go global -> LDS by LDS-DMA (no staging registers, asynchronous): piece j = wave + 7 i covers rows 8 j' .. 8 j' + 7 of K
    // (j < ROWS / 8) or V; lane l carries row (l >> 3), LDS chunk slot (l & 7) whose SOURCE chunk is slot ^ kswz(row).  Rows past N repeat
    // the last valid row (finite; their probabilities are forced to zero below).  Two buffers: head h + 1 lands while head h is multiplied.
    int src_off[PPW], dst_off[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int j = min(wave + F16_WAVES * i, NPIECE - 1);
        const int isv = j >= ROWS / 8 ? 1 : 0, r = (j - isv * (ROWS / 8)) * 8 + (lane >> 3);
        src_off[i] = (min(r, N - 1) * p.ld + (1 + isv) * p.D) * 2 + (((lane & 7) ^ kswz(r)) << 4);
        dst_off[i] = j * 1024;
    }
    const unsigned char* gbase = reinterpret_cast<const unsigned char*>(p.qkv + (size_t)b * N * p.ld);
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_offset_of(lds16));
    // (issued through inline assembly, see lds_dma16_hidden: with the builtin the compiler drained head h + 1's pieces in front of head h's
    //  transposed V reads -- the prefetch only covered the Q.K^T + softmax phase)
    auto issue = [&](int h, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (wave + F16_WAVES * i < NPIECE)
                lds_dma16_hidden(gbase + src_off[i] + h * HD * 2, lds_base + buf * BUF + __builtin_amdgcn_readfirstlane(dst_off[i]));
    };
    issue(0, 0);
    for (int i = tid; i < ROWS; i += F16_NTHR) pol[i] = (i < N) ? (p.policy ? p.policy[(size_t)b * N + i] : 1.0f) : 0.0f;
    bf16x8 qn[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qn[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + ((size_t)b * N + qc) * p.ld + 32 * ks + 8 * grp);
    f32x4 mean[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t) mean[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float czsum = 0.f;                                 // TAIL: (eps / N) sum_h 1 / (sum_h + eps), the eps term of this query row's head mean
#pragma unroll 1
    for (int h = 0; h < p.H; ++h) {
        const unsigned char* tK = lds16 + (h & 1) * BUF;
        const unsigned char* tV = tK + ROWS * 128;
        bf16x8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
        // head h's pieces (issued one head ago) and q fragment have landed; every wave is past head h - 1: its buffer is free again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (h + 1 < p.H) {
            issue(h + 1, (h + 1) & 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qn[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + ((size_t)b * N + qc) * p.ld + (h + 1) * HD + 32 * ks + 8 * grp);
        }
        if (!active) continue;
        f32x4 s[NT16];
        float mraw = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(tK + row_off(t * 16 + l15, ks * 4 + grp)), qf[ks], s[t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) mraw = fmaxf(mraw, s[t][i]);
        }
        mraw = fmaxf(mraw, __shfl_xor(mraw, 16, 64));
        mraw = fmaxf(mraw, __shfl_xor(mraw, 32, 64));
        const float m1 = mraw * c1;
        float sum = 0.f;
        if constexpr (TAIL) {
            const ppf_float2 c12 = {c1, c1}, nm = {-m1, -m1};
            ppf_float2 sum2 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                const ppf_float2 x0 = ppf_float2{s[t][0], s[t][1]} * c12 + nm, x1 = ppf_float2{s[t][2], s[t][3]} * c12 + nm;
                ppf_float2 e0, e1;
                e0.x = __builtin_amdgcn_exp2f(x0.x); e0.y = __builtin_amdgcn_exp2f(x0.y);
                e1.x = __builtin_amdgcn_exp2f(x1.x); e1.y = __builtin_amdgcn_exp2f(x1.y);
                if (t == NT16 - 1) {                                   // the only tile with padded keys (their K rows repeat row N - 1)
                    const int key0 = t * 16 + 4 * grp;
                    e0.x = key0 < N ? e0.x : 0.f; e0.y = key0 + 1 < N ? e0.y : 0.f;
                    e1.x = key0 + 2 < N ? e1.x : 0.f; e1.y = key0 + 3 < N ? e1.y : 0.f;
                }
                sum2 += e0; sum2 += e1;
                s[t] = f32x4{e0.x, e0.y, e1.x, e1.y};                  // unnormalised probability WITHOUT the eps / N term (see above)
            }
            sum = sum2.x + sum2.y;
        } else {
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                const int key0 = t * 16 + 4 * grp;
                float keep[4] = {1.f, 1.f, 1.f, 1.f};
                if constexpr (POLICY) {
                    const float4 kp = *reinterpret_cast<const float4*>(pol + key0);
                    keep[0] = kp.x; keep[1] = kp.y; keep[2] = kp.z; keep[3] = kp.w;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool real = key0 + i < N;
                    float e = __builtin_amdgcn_exp2f(s[t][i] * c1 - m1);
                    if constexpr (POLICY) e *= (key0 + i == qself) ? 1.0f : keep[i];
                    else e = real ? e : 0.f;
                    sum += e;
                    s[t][i] = e + (real ? c : 0.f);                    // unnormalised probability (+ eps / N on the real keys)
                }
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float zi = 1.0f / (sum + SOFTMAX_EPS);
        if (grp == 0 && q < N) {
            const size_t si = ((size_t)b * p.H + h) * N + q;
            p.rowmax[si] = mraw * p.scale;
            p.zinv[si] = zi;
        }
        if (p.headmean) {
            if constexpr (TAIL) {
                const ppf_float2 z2 = {zi, zi};
                czsum += c * zi;
#pragma unroll
                for (int t = 0; t < NT16; ++t) {
                    const ppf_float2 lo = ppf_float2{s[t][0], s[t][1]} * z2 + ppf_float2{mean[t][0], mean[t][1]};
                    const ppf_float2 hi = ppf_float2{s[t][2], s[t][3]} * z2 + ppf_float2{mean[t][2], mean[t][3]};
                    mean[t] = f32x4{lo.x, lo.y, hi.x, hi.y};
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT16; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mean[t][i] += s[t][i] * zi;
            }
        }
        f32x4 o[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < (NT16 + 1) / 2; ++tp) {
            constexpr int LAST = NT16 - 1;
            const bool second = 2 * tp + 1 <= LAST;               // compile-time after unrolling
            typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
            const f32x4 a = s[2 * tp], bq = second ? s[(2 * tp + 1 <= LAST) ? 2 * tp + 1 : LAST] : f32x4{0.f, 0.f, 0.f, 0.f};
            const u32x4 u = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(bq[0], bq[1]), pack_bf16x2(bq[2], bq[3])};
            const bf16x8 pf = __builtin_bit_cast(bf16x8, u);
#pragma unroll
            for (int db = 0; db < DB; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr16(tV, 32 * tp, 16 * db, lane, second), pf, o[db], 0, 0, 0);
        }
        if (q < N) {
            bf16_t* orow = p.out + ((size_t)b * N + q) * p.D + h * HD;
#pragma unroll
            for (int db = 0; db < DB; ++db)
                *reinterpret_cast<uint2*>(orow + 16 * db + 4 * grp) = make_uint2(pack_bf16x2(o[db][0] * zi, o[db][1] * zi), pack_bf16x2(o[db][2] * zi, o[db][3] * zi));
        }
    }
    if (p.headmean && active && q < N) {
        const float invH = 1.0f / (float)p.H;
        float* row = p.headmean + ((size_t)b * N + q) * p.NP;
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            const int key0 = t * 16 + 4 * grp;
            float4 v = make_float4(mean[t][0] * invH, mean[t][1] * invH, mean[t][2] * invH, mean[t][3] * invH);
            if constexpr (TAIL) {
                const float cz = czsum * invH;
                if (t < NT16 - 1) { v.x += cz; v.y += cz; v.z += cz; v.w += cz; }
                else { v.x += key0 < N ? cz : 0.f; v.y += key0 + 1 < N ? cz : 0.f; v.z += key0 + 2 < N ? cz : 0.f; v.w += key0 + 3 < N ? cz : 0.f; }
            }
            if (key0 < p.NP) *reinterpret_cast<float4*>(row + key0) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------- head mean
// grid: x = 128-query block, y = KT*32-key block, z = batch.  Loops over heads, K tile restaged per head.
template <int HD, int KT>
__global__ __launch_bounds__(256, 2) void attn_headmean_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char tK[KT * 32 * 128];
    __shared__ __attribute__((aligned(16))) float pol[KT * 32];
    __shared__ __attribute__((aligned(16))) float cv[KT * 32];
    constexpr int KS = HD / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const int b = blockIdx.z, N = p.N, key_begin = blockIdx.y * KT * 32;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int q = q0 + (lane & 31), qc = min(q, N - 1);
    const bool active = q0 < N;
    const float c = p.eps_c;
    for (int i = tid; i < KT * 32; i += 256) {
        const int key = key_begin + i;
        pol[i] = (key < N) ? (p.policy ? p.policy[(size_t)b * N + key] : 1.0f) : 0.0f;
        cv[i] = (key < N) ? c : 0.0f;
    }
    const int qself = p.self_keep ? q : -1;
    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    Stage<HD, KT * 32, 256> sk;                     // next head's K tile travels in registers while this head is computed
    sk.template load<true>(p.qkv + (size_t)b * N * p.ld + p.D, p.ld, key_begin, N, tid);
    for (int h = 0; h < p.H; ++h) {
        const bf16_t* base = p.qkv + (size_t)b * N * p.ld + h * HD;
        __syncthreads();
        sk.store(tK, tid);
        __syncthreads();
        bf16x8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)qc * p.ld + ks * 16 + hh * 8);
        if (h + 1 < p.H) sk.template load<true>(base + HD + p.D, p.ld, key_begin, N, tid);
        if (!active) continue;
        const size_t si = ((size_t)b * p.H + h) * N + qc;
        const float mx = p.rowmax[si], zi = p.zinv[si];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tK, t * 32, ks, lane), qf[ks], s, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kl = t * 32 + 8 * g + 4 * hh;
                const float4 kp = *reinterpret_cast<const float4*>(pol + kl);
                const float4 cq = *reinterpret_cast<const float4*>(cv + kl);
                const float keep[4] = {kp.x, kp.y, kp.z, kp.w}, cc[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float k = (key_begin + kl + i == qself) ? 1.0f : keep[i];
                    acc[t][4 * g + i] += (__expf(s[4 * g + i] * p.scale - mx) * k + cc[i]) * zi;
                }
            }
        }
    }
    if (active && q < N) {
        const float invH = 1.0f / (float)p.H;
        float* row = p.headmean + ((size_t)b * N + q) * p.NP;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key = key_begin + t * 32 + 8 * g + 4 * hh;
                if (key < p.NP)
                    *reinterpret_cast<float4*>(row + key) = make_float4(acc[t][4 * g] * invH, acc[t][4 * g + 1] * invH, acc[t][4 * g + 2] * invH, acc[t][4 * g + 3] * invH);
            }
    }
}

// ------------------------------------------------------------------------------------------------ backward, one pass
// Every (query tile, key tile) pair is visited ONCE: wave w owns key tile w (its K / V fragments stay in registers, dK / dV
// accumulate in registers) and walks the query tiles in the rotated order t = (w + i) mod NT, so that in step i the NT waves work on
// NT different query tiles.  The dQ contribution of a pair is added to that query tile's fp32 accumulator in LDS by plain
// read-modify-write -- the rotation gives every wave exclusive ownership of its tile between two barriers, and the order in which
// the key tiles reach a query tile is fixed (bit-identical from run to run, no atomics).  dS is needed in two layouts (keys in the
// lanes for dK, queries in the lanes for dQ): it is written once, as bf16 [key][query], into a 4 KiB per-wave scratch tile and read
// back transposed (ds_read_b64_tr_b16).  Per pair: 20 MFMA 32x32x16 (S, dP, dV, dK, dQ) and ONE softmax evaluation, against 28 and
// two for the two-phase kernel above.  LDS: Q and dO images (2 x NT x 4 KiB), dQ accumulators (NT x 8 KiB), scratch, row statistics
// = 147.5 KiB at NT = 7; K and V never enter LDS except for the one transposed copy of the wave's own K tile.
template <int HD, int NT>
struct OnePassLds {
    static constexpr int TILE = NT * 32 * 128, DT = (HD + 31) / 32;
    static constexpr int ACC = NT * DT * 4096 + Geo<NT>::NW * 4096;
    static constexpr int MID = ACC > 3 * TILE ? ACC : 3 * TILE;
    static constexpr int BYTES = 2 * TILE + MID + 5 * NT * 32 * 4;
};
template <int HD, int NT>
__global__ __launch_bounds__(Geo<NT>::NTHR, 2) void attn_bwd_onepass_kernel(const AttnParams p) {
    constexpr int NW = Geo<NT>::NW, NTHR = Geo<NT>::NTHR;
    static_assert(NT <= NW, "one key tile per wave");
    constexpr int TILE = NT * 32 * 128;
    constexpr int DT = (HD + 31) / 32, KS = HD / 16;
    constexpr int DQT = DT * 4 * 64 * 4;                        // floats of one query tile's dQ accumulator: [dt][g][lane][4]
    constexpr int MID = OnePassLds<HD, NT>::MID;               // dQ accumulators + scratch tiles; K, V and O images during the prologue
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* tQ = lds;
    unsigned char* tO = lds + TILE;                            // dO
    float* dqacc = reinterpret_cast<float*>(lds + 2 * TILE);
    unsigned char* scr_all = lds + 2 * TILE + NT * DQT * 4;
    float* pol = reinterpret_cast<float*>(lds + 2 * TILE + MID);
    float* st_m = pol + NT * 32;
    float* st_z = st_m + NT * 32;
    float* st_d = st_z + NT * 32;
    float* st_cz = st_d + NT * 32;
    unsigned char* tK = lds + 2 * TILE;                        // prologue only (aliases dqacc / scratch)
    unsigned char* tV = tK + TILE;
    unsigned char* tOut = tV + TILE;                           // O (forward output), for delta
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y, N = p.N;
    const bf16_t* base = p.qkv + (size_t)b * N * p.ld + h * HD;
    const int r0 = wave * 32;                      // this wave's key tile (and the query tile whose delta / dQ rows it handles)
    const bool active = wave < NT && r0 < N;
    const int row = r0 + l31, rc = min(row, N - 1);
    unsigned char* scr = scr_all + wave * 4096;
    const float sc2 = p.scale * LOG2E;
    {
        Stage<HD, NT * 32, NTHR> sq, so, sk, sv, sx;   // every global load of the kernel in flight before the first LDS write
        sq.template load<false>(base, p.ld, 0, N, tid);
        so.template load<false>(p.dout + (size_t)b * N * p.D + h * HD, p.D, 0, N, tid);
        sk.template load<true>(base + p.D, p.ld, 0, N, tid);          // clamped: padded keys repeat the last valid row (finite, keep = 0)
        sv.template load<false>(base + 2 * p.D, p.ld, 0, N, tid);
        sx.template load<false>(p.out + (size_t)b * N * p.D + h * HD, p.D, 0, N, tid);
        for (int i = tid; i < NT * 32; i += NTHR) {
            const size_t si = ((size_t)b * p.H + h) * N + i;
            const float zi = i < N ? p.zinv[si] : 0.f;   // zero => padded queries contribute nothing
            pol[i] = (i < N) ? (p.policy ? p.policy[(size_t)b * N + i] : 1.0f) : 0.0f;
            st_m[i] = i < N ? p.rowmax[si] * LOG2E : 0.f;
            st_z[i] = zi;
            st_cz[i] = p.eps_c * zi;                     // the eps / N term of the policy softmax, per query
            st_d[i] = 0.f;
        }
        sq.store(tQ, tid);
        so.store(tO, tid);
        sk.store(tK, tid);
        sv.store(tV, tid);
        sx.store(tOut, tid);
    }
    __syncthreads();
    // own key tile: K and V rows (second operands of S and dP) and K^T (first operand of the dQ product: output index d, contraction
    // over the 32 keys); delta = rowsum(dO * O) of the own query tile
    bf16x8 kf[KS], vf[KS], ktr[2][DT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { kf[ks] = frag_rows(tK, r0, ks, lane); vf[ks] = frag_rows(tV, r0, ks, lane); }
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ktr[st][dt] = frag_tr(tK, r0 + 16 * st, dt * 32, lane);
    if (active) {
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 dof = frag_rows(tO, r0, ks, lane), ovf = frag_rows(tOut, r0, ks, lane);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)dof[e] * (float)ovf[e];
        }
        dl += __shfl_xor(dl, 32, 64);
        if (hh == 0) st_d[row] = row < N ? dl : 0.f;
    }
    __syncthreads();                               // K / V / O images are dead from here
    {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid; i < NT * DQT / 4; i += NTHR) reinterpret_cast<float4*>(dqacc)[i] = z4;
    }
    __syncthreads();

    const float keep_key = row < N ? pol[rc] : 0.f;
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
#pragma unroll 1
    for (int i = 0; i < NT; ++i) {
        if (active) {
            int t = wave + i;
            if (t >= NT) t -= NT;
            f32x16 s, g;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; g[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tQ, t * 32, ks, lane), kf[ks], s, 0, 0, 0);   // [q][key]
                g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tO, t * 32, ks, lane), vf[ks], g, 0, 0, 0);   // dO.V^T
            }
            // softmax + dS on the packed fp32 pipe (two queries per instruction); exp2 with log2(e) folded into the scale and the saved
            // row maximum.  Only the diagonal pair (t == own tile) contains "a masked query still attends to itself" positions.
            const bool diag = t == wave && p.self_keep;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int qq0 = t * 32 + 8 * gg + 4 * hh;
                const float4 m4 = *reinterpret_cast<const float4*>(st_m + qq0), z4 = *reinterpret_cast<const float4*>(st_z + qq0),
                             d4 = *reinterpret_cast<const float4*>(st_d + qq0), c4 = *reinterpret_cast<const float4*>(st_cz + qq0);
                const ppf_float2 mm[2] = {{m4.x, m4.y}, {m4.z, m4.w}}, zz[2] = {{z4.x, z4.y}, {z4.z, z4.w}}, dd[2] = {{d4.x, d4.y}, {d4.z, d4.w}},
                                 cz[2] = {{c4.x, c4.y}, {c4.z, c4.w}};
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const int r = 4 * gg + 2 * jp;
                    ppf_float2 kk = {keep_key, keep_key};
                    if (diag) {
                        if (qq0 + 2 * jp == row) kk.x = 1.0f;
                        if (qq0 + 2 * jp + 1 == row) kk.y = 1.0f;
                    }
                    const ppf_float2 sv = {s[r], s[r + 1]}, gv = {g[r], g[r + 1]};
                    const ppf_float2 x = sv * sc2 - mm[jp];
                    ppf_float2 e; e.x = __builtin_amdgcn_exp2f(x.x); e.y = __builtin_amdgcn_exp2f(x.y);
                    const ppf_float2 pt = e * zz[jp] * kk;
                    const ppf_float2 ds = pt * (gv - dd[jp]), po = pt + cz[jp];
                    s[r] = ds.x; s[r + 1] = ds.y;                                  // dS[q][key]
                    g[r] = po.x; g[r + 1] = po.y;                                  // out[q][key]
                }
                // dS as bf16 [key][query] for the transposed read below: this lane's key row, queries 8 gg + 4 hh .. + 3
                *reinterpret_cast<uint2*>(scr + row_off(l31, gg) + 8 * hh) =
                    make_uint2(pack_bf16x2(s[4 * gg], s[4 * gg + 1]), pack_bf16x2(s[4 * gg + 2], s[4 * gg + 3]));
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8 dsf = pack8(s, 8 * st), pf = pack8(g, 8 * st);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(tQ, t * 32 + 16 * st, dt * 32, lane), dsf, dk[dt], 0, 0, 0);
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(tO, t * 32 + 16 * st, dt * 32, lane), pf, dv[dt], 0, 0, 0);
                }
            }
            // dQ^T[d][q] of this pair: K^T (registers) x dS^T (scratch, keys in the contraction slots, queries in the lanes)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float* qa = dqacc + t * DQT + lane * 4;
            f32x16 dq[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 v = *reinterpret_cast<const float4*>(qa + (dt * 4 + gq) * 256);
                    dq[dt][4 * gq] = v.x; dq[dt][4 * gq + 1] = v.y; dq[dt][4 * gq + 2] = v.z; dq[dt][4 * gq + 3] = v.w;
                }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8 dst = frag_tr(scr, 16 * st, 0, lane);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktr[st][dt], dst, dq[dt], 0, 0, 0);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<float4*>(qa + (dt * 4 + gq) * 256) = make_float4(dq[dt][4 * gq], dq[dt][4 * gq + 1], dq[dt][4 * gq + 2], dq[dt][4 * gq + 3]);
        }
        __syncthreads();                           // the next step's owner of each query tile sees this step's sums
    }
    // results as bf16 row images (dK over the Q image, dV over the dO image, dQ over the scratch tiles), then 16-byte row-contiguous
    // stores by the whole workgroup: the accumulator layout (a lane = a row, 8-byte pieces) would store 32 scattered pieces per instruction
    unsigned char* tDQ = scr_all;
    if (active) {
        const float* qa = dqacc + wave * DQT + lane * 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c16 = dt * 4 + g;
                if (c16 * 8 < HD) {
                    const float4 q4 = *reinterpret_cast<const float4*>(qa + (dt * 4 + g) * 256);
                    *reinterpret_cast<uint2*>(tDQ + row_off(row, c16) + 8 * hh) =
                        make_uint2(pack_bf16x2(q4.x * p.scale, q4.y * p.scale), pack_bf16x2(q4.z * p.scale, q4.w * p.scale));
                    *reinterpret_cast<uint2*>(tQ + row_off(row, c16) + 8 * hh) =
                        make_uint2(pack_bf16x2(dk[dt][4 * g] * p.scale, dk[dt][4 * g + 1] * p.scale), pack_bf16x2(dk[dt][4 * g + 2] * p.scale, dk[dt][4 * g + 3] * p.scale));
                    *reinterpret_cast<uint2*>(tO + row_off(row, c16) + 8 * hh) =
                        make_uint2(pack_bf16x2(dv[dt][4 * g], dv[dt][4 * g + 1]), pack_bf16x2(dv[dt][4 * g + 2], dv[dt][4 * g + 3]));
                }
            }
    }
    __syncthreads();
    {
        constexpr int CH = HD / 8;
        bf16_t* dst = p.dqkv + (size_t)b * N * p.ld + h * HD;
        for (int i = tid; i < N * CH; i += NTHR) {
            const int r = i / CH, cc = i - r * CH;
            bf16_t* o = dst + (size_t)r * p.ld + cc * 8;
            *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(tDQ + row_off(r, cc));
            *reinterpret_cast<uint4*>(o + p.D) = *reinterpret_cast<const uint4*>(tQ + row_off(r, cc));
            *reinterpret_cast<uint4*>(o + 2 * p.D) = *reinterpret_cast<const uint4*>(tO + row_off(r, cc));
        }
    }
}

// ------------------------------------------------------------------------------------------- backward, one pass, streaming (round 4)
// The pair decomposition of attn_bwd_onepass_kernel (wave w owns key tile w, K / V fragments and dK / dV accumulators in registers,
// dQ accumulated per query tile in LDS by exclusive read-modify-write in a rotated order), rebuilt around what the knock-outs of that
// kernel showed (profiles/r4_attn_bwd.txt): its phases ADD -- 33 us fixed + 20 staging + 85 pair loop + 48 stores of 163 us, and inside
// the pair loop softmax arithmetic (32 us), dK / dV products (17), dQ read-modify-write (11) and the rest (25) add as well, because
// one 147 KiB workgroup per CU leaves nobody to fill a phase and a barrier per step keeps all waves in the same phase.
//   * PERSISTENT workgroups walk items (batch, head) = blockIdx.x, blockIdx.x + gridDim.x, ...: the moment the last pair of item i is
//     done every wave issues item i + 1's loads and only then writes item i's results -- loads and stores overlap each other.
//   * Q and dO images (the only operands every wave needs) arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction,
//     chunk swizzle on the SOURCE address); K / V / O rows of the wave's own 32 keys / queries go straight to registers; K^T fragments
//     come from a transposed read of the wave's own 4 KiB scratch tile.  No K / V / O images, no staging registers.  Results leave
//     through the wave's OWN scratch tile as 16-byte row-contiguous stores, without any workgroup barrier.
//   * NT waves (no idle eighth wave), 1 + NT barriers per item.  (Running the two waves of a SIMD half a step apart -- one in the VALU-heavy
//     first half of a pair, its partner in the MFMA / LDS-heavy second half -- was built and measured: no gain, the waves are latency-bound,
//     not issue-bound: WAIT_ANY 50 %, MFMA 16 %, LDS 32 %, VALU 13 % busy; profiles/r4_attn_bwd.txt.)
//   * 1/sum of the softmax is folded into the exponent (row statistic m' = m log2(e) - log2(1/sum)); the key-keep factor is applied only
//     where a key can be masked (policy given, or the tile holding the padded keys).
// Padded queries (rows >= N of the images repeat row N - 1, finite) are neutralised by their statistics: m' = +3e38 -> exp2 -> 0, eps
// term = 0.  Padded keys: keep = 0 as before.
template <int HD, int NT>
struct StreamLds {
    static constexpr int TILE = NT * 32 * 128, DT = (HD + 31) / 32, DQT = DT * 4 * 64 * 4;
    static constexpr int BYTES = 2 * TILE + NT * DQT * 4 + NT * 4096 + 3 * NT * 32 * 4;
};
template <int HD, int NT>
__global__ __launch_bounds__(NT * 64, 2) void attn_bwd_stream_kernel(const AttnParams p, const int nitems) {
    static_assert(HD == 64, "128-byte rows: one full line per (token, head)");
    constexpr int TILE = NT * 32 * 128;
    constexpr int DT = (HD + 31) / 32, KS = HD / 16;
    constexpr int DQT = DT * 4 * 64 * 4;                        // floats of one query tile's dQ accumulator: [dt][g][lane][4]
    constexpr int PPW = 8;                                      // 1 KiB pieces per wave: 4 of the Q image, 4 of the dO image (4 NT each)
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* tQ = lds;
    unsigned char* tO = lds + TILE;                            // dO
    float* dqacc = reinterpret_cast<float*>(lds + 2 * TILE);
    unsigned char* scr_all = lds + 2 * TILE + NT * DQT * 4;
    float* st_m = reinterpret_cast<float*>(scr_all + NT * 4096);
    float* st_d = st_m + NT * 32;
    float* st_cz = st_d + NT * 32;
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, H = p.H;
    const int r0 = wave * 32;                      // this wave's key tile (and the query tile whose statistics / dQ rows it handles)
    const bool active = r0 < N;
    const bool masked = p.policy != nullptr || r0 + 32 > N;      // wave-uniform: some key of this tile may carry keep = 0
    const int row = r0 + l31, rc = min(row, N - 1);
    unsigned char* scr = scr_all + wave * 4096;
    const float sc2 = p.scale * LOG2E;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_offset_of(lds));

    // LDS-DMA pieces of this wave: piece j = wave + NT i, i < 4: rows 8 j .. 8 j + 7 of the Q image, i >= 4: of the dO image; lane l
    // carries row (l >> 3), LDS chunk slot (l & 7) whose SOURCE chunk is slot ^ kswz(row).  (Offsets are recomputed per item: eight
    // more live registers across the pair loop are eight spills.)
    auto issue_images = [&](int item) {
        const int b = item / H, h = item - b * H;
        const unsigned char* qb = reinterpret_cast<const unsigned char*>(p.qkv + (size_t)b * N * p.ld + h * HD);
        const unsigned char* ob = reinterpret_cast<const unsigned char*>(p.dout + (size_t)b * N * p.D + h * HD);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int j = wave + NT * (i & 3), r = j * 8 + (lane >> 3);
            const int off = min(r, N - 1) * (i < 4 ? p.ld : p.D) * 2 + (((lane & 7) ^ kswz(r)) << 4);
            // (hidden from the compiler's wait-count pass, see lds_dma16_hidden: with the builtin it put s_waitcnt vmcnt(0) between the issue of
            //  item i + 1's images and the stores of item i's results -- the overlap this kernel is built around never happened)
            lds_dma16_hidden((i < 4 ? qb : ob) + off, lds_base + (i < 4 ? 0 : TILE) + j * 1024);
        }
    };
    // rows of the wave's own tile, MFMA fragment layout (lane -> row l31, 8 values at d = 16 ks + 8 hh): K, V (key tile), O (query tile,
    // for delta = rowsum(dO * O)) and the softmax statistics of the own query rows
    bf16x8 kf[KS], vf[KS];
    auto load_rows = [&](int item) {
        const int b = item / H, h = item - b * H;
        const bf16_t* base = p.qkv + ((size_t)b * N + rc) * p.ld + h * HD + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(base + p.D + 16 * ks);
            vf[ks] = *reinterpret_cast<const bf16x8*>(base + 2 * p.D + 16 * ks);
        }
    };
    int item = blockIdx.x;
    if (item < nitems) { issue_images(item); load_rows(item); }
#pragma unroll 1
    for (; item < nitems; item += gridDim.x) {
        const int b = item / H, h = item - b * H;
        // ---- own query rows: O (for delta = rowsum(dO * O)) and the softmax statistics; consumed behind the wait for the images
        // (prefetching them one item ahead costs 19 more live registers = spills: measured slower)
        bf16x8 ovf[KS];
        {
            const bf16_t* orow = p.out + ((size_t)b * N + rc) * p.D + h * HD + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) ovf[ks] = *reinterpret_cast<const bf16x8*>(orow + 16 * ks);
        }
        const size_t si = ((size_t)b * H + h) * N + rc;
        const float s_m = p.rowmax[si], s_z = p.zinv[si];
        const float s_pol = p.policy ? p.policy[(size_t)b * N + rc] : 1.0f;
        // ---- own key tile: K^T fragments through the scratch tile; zeroed dQ accumulator and statistics of the own query tile
        bf16x8 ktr[2][DT];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<bf16x8*>(scr + row_off(l31, ks * 2 + hh)) = kf[ks];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) ktr[st][dt] = frag_tr(scr, 16 * st, dt * 32, lane);
        {
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float4* qz = reinterpret_cast<float4*>(dqacc + wave * DQT) + lane;
#pragma unroll
            for (int i = 0; i < DQT / 256; ++i) qz[i * 64] = z4;
        }
        const float keep_key = row < N ? s_pol : 0.f;
        if (hh == 0) {
            const bool real = row < N;
            // p = exp2(s scale log2e - m'), m' = m log2e - log2(1 / (sum + eps)); padded queries: m' = 3e38 -> p = 0 whatever their scores
            st_m[row] = real ? s_m * LOG2E - __builtin_amdgcn_logf(s_z) : 3.0e38f;
            st_cz[row] = real ? p.eps_c * s_z : 0.f;                    // the eps / N term of the policy softmax, per query
        }
        // ---- images of this item have landed (this wave's pieces: vmcnt; the others': barrier); statistics and zeroes are visible
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // delta of the own query tile from the dO image: the first pair of every wave is its OWN query tile, the other tiles' deltas are
        // read behind at least one more barrier
        {
            float dl = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 dof = frag_rows(tO, r0, ks, lane);
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += (float)dof[e] * (float)ovf[ks][e];
            }
            dl += __shfl_xor(dl, 32, 64);
            if (hh == 0) st_d[row] = row < N ? dl : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        f32x16 dk[DT], dv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
        bf16x8 dsf[2], pf[2];                           // dS and P of the pair between its two halves (bf16: MFMA operands of P2)
#pragma unroll
        for (int st = 0; st < 2; ++st) { dsf[st] = kf[0]; pf[st] = kf[0]; }

        // P1 of step i: scores, dP, softmax, dS
        auto p1 = [&](int i) {
            int t = wave + i;
            if (t >= NT) t -= NT;
            f32x16 s, g;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; g[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tQ, t * 32, ks, lane), kf[ks], s, 0, 0, 0);   // [q][key]
                g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tO, t * 32, ks, lane), vf[ks], g, 0, 0, 0);   // dO.V^T
            }
            // softmax + dS on the packed fp32 pipe (two queries per instruction).  Only the diagonal pair (t == own tile) contains
            // "a masked query still attends to itself" positions.
            const bool diag = masked && t == wave && p.self_keep;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int qq0 = t * 32 + 8 * gg + 4 * hh;
                const float4 m4 = *reinterpret_cast<const float4*>(st_m + qq0), d4 = *reinterpret_cast<const float4*>(st_d + qq0),
                             c4 = *reinterpret_cast<const float4*>(st_cz + qq0);
                const ppf_float2 mm[2] = {{m4.x, m4.y}, {m4.z, m4.w}}, dd[2] = {{d4.x, d4.y}, {d4.z, d4.w}}, cz[2] = {{c4.x, c4.y}, {c4.z, c4.w}};
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const int r = 4 * gg + 2 * jp;
                    const ppf_float2 sv = {s[r], s[r + 1]}, gv = {g[r], g[r + 1]};
                    const ppf_float2 x = sv * sc2 - mm[jp];
                    ppf_float2 pt; pt.x = __builtin_amdgcn_exp2f(x.x); pt.y = __builtin_amdgcn_exp2f(x.y);
                    if (masked) {
                        ppf_float2 kk = {keep_key, keep_key};
                        if (diag) {
                            if (qq0 + 2 * jp == row) kk.x = 1.0f;
                            if (qq0 + 2 * jp + 1 == row) kk.y = 1.0f;
                        }
                        pt = pt * kk;
                    }
                    const ppf_float2 ds = pt * (gv - dd[jp]), po = pt + cz[jp];
                    s[r] = ds.x; s[r + 1] = ds.y;                                  // dS[q][key]
                    g[r] = po.x; g[r + 1] = po.y;                                  // out[q][key]
                }
                // dS as bf16 [key][query] for the transposed read of P2: this lane's key row, queries 8 gg + 4 hh .. + 3
                *reinterpret_cast<uint2*>(scr + row_off(l31, gg) + 8 * hh) =
                    make_uint2(pack_bf16x2(s[4 * gg], s[4 * gg + 1]), pack_bf16x2(s[4 * gg + 2], s[4 * gg + 3]));
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) { dsf[st] = pack8(s, 8 * st); pf[st] = pack8(g, 8 * st); }
        };
        // P2 of step i: dK, dV, dQ
        auto p2 = [&](int i) {
            int t = wave + i;
            if (t >= NT) t -= NT;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(tQ, t * 32 + 16 * st, dt * 32, lane), dsf[st], dk[dt], 0, 0, 0);
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(tO, t * 32 + 16 * st, dt * 32, lane), pf[st], dv[dt], 0, 0, 0);
                }
            // dQ^T[d][q] of this pair: K^T (registers) x dS^T (scratch, keys in the contraction slots, queries in the lanes)
            float* qa = dqacc + t * DQT + lane * 4;
            f32x16 dq[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 v = *reinterpret_cast<const float4*>(qa + (dt * 4 + gq) * 256);
                    dq[dt][4 * gq] = v.x; dq[dt][4 * gq + 1] = v.y; dq[dt][4 * gq + 2] = v.z; dq[dt][4 * gq + 3] = v.w;
                }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8 dst = frag_tr(scr, 16 * st, 0, lane);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktr[st][dt], dst, dq[dt], 0, 0, 0);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<float4*>(qa + (dt * 4 + gq) * 256) = make_float4(dq[dt][4 * gq], dq[dt][4 * gq + 1], dq[dt][4 * gq + 2], dq[dt][4 * gq + 3]);
        };
#pragma unroll 1
        for (int i = 0; i < NT; ++i) {
            if (active) { p1(i); p2(i); }
            __syncthreads();                           // the next step's owner of each query tile sees this step's sums
        }
        // ---- every wave is past the last pair: the images are dead.  The next item's loads go out first, this item's results after them.
        const int nxt = item + gridDim.x;
        if (nxt < nitems) { issue_images(nxt); load_rows(nxt); }
        if (active) {
            bf16_t* dst = p.dqkv + ((size_t)b * N + r0) * p.ld + h * HD;
            const float* qa = dqacc + wave * DQT + lane * 4;
#pragma unroll
            for (int which = 0; which < 3; ++which) {              // 0: dQ (own query tile), 1: dK, 2: dV (own key tile)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v0, v1, v2, v3;
                        if (which == 0) { const float4 q4 = *reinterpret_cast<const float4*>(qa + (dt * 4 + g) * 256); v0 = q4.x * p.scale; v1 = q4.y * p.scale; v2 = q4.z * p.scale; v3 = q4.w * p.scale; }
                        else if (which == 1) { v0 = dk[dt][4 * g] * p.scale; v1 = dk[dt][4 * g + 1] * p.scale; v2 = dk[dt][4 * g + 2] * p.scale; v3 = dk[dt][4 * g + 3] * p.scale; }
                        else { v0 = dv[dt][4 * g]; v1 = dv[dt][4 * g + 1]; v2 = dv[dt][4 * g + 2]; v3 = dv[dt][4 * g + 3]; }
                        *reinterpret_cast<uint2*>(scr + row_off(l31, dt * 4 + g) + 8 * hh) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = (lane >> 3) + 8 * it, cc = lane & 7;
                    const uint4 v = *reinterpret_cast<const uint4*>(scr + row_off(r, cc));
                    if (r0 + r < N) *reinterpret_cast<uint4*>(dst + (size_t)r * p.ld + which * p.D + cc * 8) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <typename F>
int dispatch(int hd, int N, const char* who, F&& f) {
    const int nt = (N + 31) / 32;
#define PPF_HD_CASE(HDV)                                                         \
    if (hd == HDV) {                                                             \
        if (nt <= 1) return f(std::integral_constant<int, HDV>(), std::integral_constant<int, 1>()); \
        if (nt <= 2) return f(std::integral_constant<int, HDV>(), std::integral_constant<int, 2>()); \
        if (nt <= 4) return f(std::integral_constant<int, HDV>(), std::integral_constant<int, 4>()); \
        if (nt <= 7) return f(std::integral_constant<int, HDV>(), std::integral_constant<int, 7>()); \
    }
    PPF_HD_CASE(64)
    PPF_HD_CASE(48)
    PPF_HD_CASE(32)
#undef PPF_HD_CASE
    ppf_set_error("%s: unsupported head_dim=%d / tokens=%d (head_dim in {32,48,64}, tokens <= 224)", who, hd, N);
    return PPF_ERR_SHAPE;
}

int fill(AttnParams& p, const void* qkv, int B, int H, int N, int D, const float* policy, float* rowmax, float* zinv, int self_keep, int eps_n, const char* who) {
    PPF_CHECK_ARG(B > 0 && H > 0 && N > 0 && D > 0 && D % H == 0 && (D % 8) == 0, PPF_ERR_SHAPE, "%s: bad shape B=%d H=%d N=%d D=%d", who, B, H, N, D);
    PPF_CHECK_ARG(qkv && rowmax && zinv, PPF_ERR_ARG, "%s: null pointer", who);
    p = AttnParams();
    p.qkv = (const bf16_t*)qkv; p.ld = 3 * D; p.policy = policy; p.rowmax = rowmax; p.zinv = zinv; p.B = B; p.H = H; p.N = N; p.D = D;
    p.self_keep = self_keep; p.scale = 1.0f / sqrtf((float)(D / H));
    p.eps_c = SOFTMAX_EPS / (float)(eps_n > 0 ? eps_n : N);
    return 0;
}

}  // namespace

extern "C" {

// O[B*N][D] = softmax_policy(Q K^T / sqrt(hd)) V from packed qkv [B*N][3D]; saves rowmax and 1/(sum+eps) [B][H][N].
int ppf_attn_fwd(const void* qkv, void* out, const float* policy, float* rowmax, float* zinv, int B, int H, int N, int D, int self_keep,
                 int eps_n, hipStream_t stream) {
    AttnParams p;
    int rc = fill(p, qkv, B, H, N, D, policy, rowmax, zinv, self_keep, eps_n, "ppf_attn_fwd");
    if (rc) return rc;
    p.out = (bf16_t*)out;
    // algorithmic work: QK^T and PV, 2 N^2 hd flops each per (sample, head); qkv in, out + the two statistics out
    PpfProbeScope probe(PPF_PROBE_ATTN_FWD, stream, 4.0 * B * H * (double)N * N * (D / H), 8.0 * B * N * (double)D + 8.0 * B * H * N);
    return dispatch(D / H, N, "ppf_attn_fwd", [&](auto hd, auto nt) {
        using G = Geo<decltype(nt)::value>;
        hipLaunchKernelGGL((attn_fwd_kernel<decltype(hd)::value, decltype(nt)::value>), dim3((N + G::NW * 32 - 1) / (G::NW * 32), H, B), dim3(G::NTHR), 0, stream, p);
        PPF_LAUNCH_CHECK();
        return 0;
    });
}

// The same forward pass with the head-mean map (NP = N rounded up to a multiple of 4, pad columns 0; NULL: no map) written by the SAME
// launch (attn_fwd16_kernel): ppf_attn_fwd_hm_supported = 1 for head_dim 64 and N <= 208, else callers use ppf_attn_fwd + ppf_attn_headmean.
int ppf_attn_fwd_hm_supported(int H, int N, int D) { return H > 0 && D % H == 0 && D / H == 64 && N > 0 && N <= 208 ? 1 : 0; }
int ppf_attn_fwd_hm(const void* qkv, void* out, const float* policy, float* rowmax, float* zinv, float* headmean, int NP, int B, int H, int N,
                    int D, int self_keep, int eps_n, hipStream_t stream) {
    AttnParams p;
    int rc = fill(p, qkv, B, H, N, D, policy, rowmax, zinv, self_keep, eps_n, "ppf_attn_fwd_hm");
    if (rc) return rc;
    PPF_CHECK_ARG(ppf_attn_fwd_hm_supported(H, N, D), PPF_ERR_SHAPE, "ppf_attn_fwd_hm: head_dim must be 64 and N <= 208 (H=%d N=%d D=%d)", H, N, D);
    PPF_CHECK_ARG(headmean == nullptr || (NP >= N && NP % 4 == 0 && NP < N + 4), PPF_ERR_SHAPE, "ppf_attn_fwd_hm: NP=%d must be N rounded up to a multiple of 4", NP);
    p.out = (bf16_t*)out; p.headmean = headmean; p.NP = NP;
    PpfProbeScope probe(PPF_PROBE_ATTN_FWD, stream, 4.0 * B * H * (double)N * N * (D / H),
                        8.0 * B * N * (double)D + 8.0 * B * H * N + (headmean ? 4.0 * B * N * (double)NP : 0.0));
    const int nb = (N + 15) / 16;
    const dim3 grid((nb + F16_WAVES - 1) / F16_WAVES, B);
    constexpr int lds6 = Fwd16<64, 6>::LDS, lds13 = Fwd16<64, 13>::LDS;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd16_kernel<64, 13, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds13);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd16_kernel<64, 13, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds13);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd16_kernel<64, 13, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds13);
        if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(attn_fwd16): %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    // round 6 (profiles/r6_attn_fwd_packed.txt): the packed softmax where only the last key tile is padded -- 65.8 -> 58.1 us stand-alone, +0.6 % of the step
    if (N <= 96) {
        if (policy) hipLaunchKernelGGL((attn_fwd16_kernel<64, 6, true, false>), grid, dim3(F16_NTHR), lds6, stream, p);
        else if (N > 80) hipLaunchKernelGGL((attn_fwd16_kernel<64, 6, false, true>), grid, dim3(F16_NTHR), lds6, stream, p);
        else hipLaunchKernelGGL((attn_fwd16_kernel<64, 6, false, false>), grid, dim3(F16_NTHR), lds6, stream, p);
    } else {
        if (policy) hipLaunchKernelGGL((attn_fwd16_kernel<64, 13, true, false>), grid, dim3(F16_NTHR), lds13, stream, p);
        else if (N > 192) hipLaunchKernelGGL((attn_fwd16_kernel<64, 13, false, true>), grid, dim3(F16_NTHR), lds13, stream, p);
        else hipLaunchKernelGGL((attn_fwd16_kernel<64, 13, false, false>), grid, dim3(F16_NTHR), lds13, stream, p);
    }
    PPF_LAUNCH_CHECK();
    return 0;
}

// headmean[B][N][NP] = mean_h probabilities (NP = N rounded up to a multiple of 4; pad columns are written as 0).
int ppf_attn_headmean(const void* qkv, const float* policy, const float* rowmax, const float* zinv, float* headmean, int NP, int B, int H,
                      int N, int D, int self_keep, int eps_n, hipStream_t stream) {
    AttnParams p;
    int rc = fill(p, qkv, B, H, N, D, policy, (float*)rowmax, (float*)zinv, self_keep, eps_n, "ppf_attn_headmean");
    if (rc) return rc;
    PPF_CHECK_ARG(NP >= N && NP % 4 == 0 && NP < N + 4, PPF_ERR_SHAPE, "ppf_attn_headmean: NP=%d must be N rounded up to a multiple of 4", NP);
    p.headmean = headmean; p.NP = NP;
    const int hd = D / H;
    constexpr int KT = 4;
    dim3 grid((N + 127) / 128, (N + KT * 32 - 1) / (KT * 32), B);
    if (hd == 64) hipLaunchKernelGGL((attn_headmean_kernel<64, KT>), grid, dim3(256), 0, stream, p);
    else if (hd == 48) hipLaunchKernelGGL((attn_headmean_kernel<48, KT>), grid, dim3(256), 0, stream, p);
    else if (hd == 32) hipLaunchKernelGGL((attn_headmean_kernel<32, KT>), grid, dim3(256), 0, stream, p);
    else { ppf_set_error("ppf_attn_headmean: unsupported head_dim=%d", hd); return PPF_ERR_SHAPE; }
    PPF_LAUNCH_CHECK();
    return 0;
}

// dqkv[B*N][3D] from dout[B*N][D]; needs out, rowmax, zinv of the forward; delta[B][H][N] is scratch.
int ppf_attn_bwd(const void* qkv, const void* out, const void* dout, void* dqkv, const float* policy, const float* rowmax,
                 const float* zinv, float* delta, int B, int H, int N, int D, int self_keep, int eps_n, hipStream_t stream) {
    AttnParams p;
    int rc = fill(p, qkv, B, H, N, D, policy, (float*)rowmax, (float*)zinv, self_keep, eps_n, "ppf_attn_bwd");
    if (rc) return rc;
    PPF_CHECK_ARG(out && dout && dqkv && delta, PPF_ERR_ARG, "ppf_attn_bwd: null pointer");
    p.out = (bf16_t*)out; p.dout = (const bf16_t*)dout; p.dqkv = (bf16_t*)dqkv; p.delta = delta;
    // five products of 2 N^2 hd flops per (sample, head): S, dV, dP, dQ, dK; qkv + out + dout in, dqkv out
    PpfProbeScope probe(PPF_PROBE_ATTN_BWD, stream, 10.0 * B * H * (double)N * N * (D / H), 16.0 * B * N * (double)D + 8.0 * B * H * N);
    return dispatch(D / H, N, "ppf_attn_bwd", [&](auto hd, auto nt) {
        using G = Geo<decltype(nt)::value>;
        constexpr int HDv = decltype(hd)::value, NTv = decltype(nt)::value;
        if constexpr (HDv == 64) {
            {                   // head_dim 64: the streaming form of the one-pass kernel (round 4: 163 -> 134 us, profiles/r4_attn_bwd.txt)
                constexpr int lds_bytes = StreamLds<HDv, NTv>::BYTES;
                auto kern = attn_bwd_stream_kernel<HDv, NTv>;
                static bool attr_set = false;
                static int slots = 0;                          // workgroups the chip holds at once: CUs x (LDS- and wave-limited) per CU
                if (!attr_set) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                    if (e != hipSuccess) { ppf_set_error("ppf_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
                    int dev = 0, cus = 256;
                    (void)hipGetDevice(&dev);
                    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                    int per = (160 * 1024) / lds_bytes;
                    if (per > 8 / NTv) per = 8 / NTv;           // <= 256 VGPRs: two waves per SIMD
                    if (per < 1) per = 1;
                    slots = cus * per;
                    attr_set = true;
                }
                // every workgroup walks the same number of items (the launch lasts as long as its longest walk): 384 items on 256 slots
                // take two rounds either way, 192 workgroups leave the other CUs to the side stream
                const int nitems = B * H, rounds = (nitems + slots - 1) / slots, grid = (nitems + rounds - 1) / rounds;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(NTv * 64), lds_bytes, stream, p, nitems);
                PPF_LAUNCH_CHECK();
                return 0;
            }
        } else {                // other head widths: the non-persistent one-pass kernel
            constexpr int lds_bytes = OnePassLds<HDv, NTv>::BYTES;
            auto kern = attn_bwd_onepass_kernel<HDv, NTv>;
            static bool attr_set = false;                  // one flag per instantiation (the lambda is instantiated per (hd, nt))
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                if (e != hipSuccess) { ppf_set_error("ppf_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
                attr_set = true;
            }
            hipLaunchKernelGGL(kern, dim3(1, H, B), dim3(G::NTHR), lds_bytes, stream, p);
            PPF_LAUNCH_CHECK();
            return 0;
        }
    });
}

}  // extern "C"
