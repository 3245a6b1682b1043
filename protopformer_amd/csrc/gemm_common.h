// Shared declarations of the bf16 MFMA GEMM family (gemm_bf16.hip; the retired 256x256 kernel under scripts/gpu/experiments/gemm_nt256 includes it too).
#pragma once
#include "ppf_common.h"

namespace ppfg {

enum Epi { EPI_BF16 = 0, EPI_F32 = 1, EPI_GELU = 2, EPI_SIGMOID_F32 = 3, EPI_RESID = 4, EPI_DGELU = 5, EPI_ATOMIC = 6, EPI_PARTIAL = 7 };

struct GemmParams {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias;       // [N] (added before activation) or null
    const float* res;        // EPI_RESID: fp32 residual [M][ldres]
    int ldres;
    const float* rowscale;   // EPI_RESID: per-sample scale, index m / rows_per_group (DropPath), or null
    int rows_per_group;
    const float* colscale;   // EPI_RESID: per-column scale (LayerScale gamma), or null
    const bf16_t* aux_in;    // EPI_DGELU: gelu'(pre-activation) [M][ldaux] as 8-bit codes (gelu8_*), as written by EPI_GELU
    bf16_t* aux_out;         // EPI_GELU: gelu'(pre-activation) out, one BYTE per element; EPI_RESID: raw branch output bf16 (optional)
    int ldaux;
    float* colsum;           // EPI_ATOMIC + TA: sum over kc of A(m,kc) accumulated atomically into colsum[m]
    float* ws;               // EPI_PARTIAL: split-K workspace [nsplit][M*N (+M)] fp32 partial tiles (+ partial column sums)
    float alpha;
    // batched problems (blockIdx.y = outer*batch_inner + inner): element offsets added to A / B / C
    int batch_inner;
    long long sa_o, sa_i, sb_o, sb_i, sc_o, sc_i;
    int zslice;              // (device side) this workgroup's K slice
    int nsplit;              // split-K slices; grid.x = tiles * nsplit, slice-major so that an XCD owns whole K slices
    int cs_parts;            // EPI_PARTIAL: partial column sums per slice and row (1)
    int kpad;                // 1: contraction-contiguous operands may read up to the next multiple of 8 beyond K (zero/finite padding)
};

// gelu'(x) of the erf GELU lies in [-0.1290, 1.1290]: it is saved for the backward pass as an 8-bit linear code (round 4):
// q = rint(196 d + 26), d' = (q - 26) / 196, |d' - d| <= 1 / 392 = 2.55e-3 -- the size of the bf16 rounding of a value in [0.5, 1.13]
// (2.0e-3 / 3.9e-3), at half the bytes: 155 -> 77.5 MB written by fc1 and read by the fc2 input-gradient GEMM per deit_small layer.
// Round 5: the grid has 0 and 1 ON code points (26 and 222; 196 * fl(1/196) == 1 in fp32), so a saturated unit -- gelu' = 0 or 1, where
// the reference's gradient is exactly 0 or exactly the upstream value -- decodes exactly; the range is [-0.1327, 1.1684].
constexpr float GELU8_ZERO = 26.0f, GELU8_STEP = 1.0f / 196.0f, GELU8_INV = 196.0f;
__device__ __forceinline__ uint32_t gelu8_pack4(float a, float b, float c, float d) {
    // v_cvt_pk_u8_f32 rounds to nearest and saturates to [0, 255]
    uint32_t r = 0;
    r = __builtin_amdgcn_cvt_pk_u8_f32(a * GELU8_INV + GELU8_ZERO, 0, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(b * GELU8_INV + GELU8_ZERO, 1, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(c * GELU8_INV + GELU8_ZERO, 2, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(d * GELU8_INV + GELU8_ZERO, 3, r);
    return r;
}
__device__ __forceinline__ void gelu8_unpack4(uint32_t q, float (&d)[4]) {
    d[0] = ((float)(q & 0xffu) - GELU8_ZERO) * GELU8_STEP;
    d[1] = ((float)((q >> 8) & 0xffu) - GELU8_ZERO) * GELU8_STEP;
    d[2] = ((float)((q >> 16) & 0xffu) - GELU8_ZERO) * GELU8_STEP;
    d[3] = ((float)((q >> 24) & 0xffu) - GELU8_ZERO) * GELU8_STEP;
}

// The fused epilogues work on 4 consecutive columns of one output row.  On gfx9 loads and stores share the vmcnt counter and
// the compiler waits vmcnt(0) before it uses a loaded value whenever stores are pending too -- a load between two stores makes
// every store wait for the previous one to reach L2.  So the per-column operands (bias, LayerScale) are loaded once per lane
// (EpiCols), the per-row operands (residual, gelu', DropPath scale) of a whole batch of rows are loaded before that batch's
// stores (EpiRow / epi_load_row), and epi_store itself issues no loads.
struct EpiCols { float4 bias, colscale; };
struct EpiRow { float4 res; uint2 aux; float rowscale; };

template <int EPI>
__device__ __forceinline__ EpiCols epi_load_cols(const GemmParams& p, int n0, bool ok) {
    EpiCols c;
    c.bias = make_float4(0.f, 0.f, 0.f, 0.f);
    c.colscale = make_float4(1.f, 1.f, 1.f, 1.f);
    if constexpr (EPI != EPI_ATOMIC && EPI != EPI_PARTIAL) {
        if (ok && p.bias) c.bias = *reinterpret_cast<const float4*>(p.bias + n0);
        if constexpr (EPI == EPI_RESID) { if (ok && p.colscale) c.colscale = *reinterpret_cast<const float4*>(p.colscale + n0); }
    }
    return c;
}

template <int EPI>
__device__ __forceinline__ EpiRow epi_load_row(const GemmParams& p, int m, int n0, bool ok) {
    EpiRow r;
    r.res = make_float4(0.f, 0.f, 0.f, 0.f); r.aux = make_uint2(0, 0); r.rowscale = 1.0f;
    if constexpr (EPI == EPI_RESID) {
        if (ok) {
            r.res = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldres + n0);
            if (p.rowscale) r.rowscale = p.rowscale[m / p.rows_per_group];
        }
    } else if constexpr (EPI == EPI_DGELU) {
        if (ok) {                                 // read once, 10 ms after it was written: streaming load
            const uint32_t t = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(p.aux_in) + (size_t)m * p.ldaux + n0));
            r.aux = make_uint2(t, 0);
        }
    }
    return r;
}

template <int EPI>
__device__ __forceinline__ void epi_store(const GemmParams& p, int m, int n0, float v0, float v1, float v2, float v3, const EpiCols& cc, const EpiRow& rr) {
    float v[4] = {v0 * p.alpha, v1 * p.alpha, v2 * p.alpha, v3 * p.alpha};
    if constexpr (EPI == EPI_ATOMIC) {
        float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n0;
#pragma unroll
        for (int i = 0; i < 4; ++i) unsafeAtomicAdd(c + i, v[i]);
        return;
    }
    if constexpr (EPI == EPI_PARTIAL) {
        const size_t slice = (size_t)p.M * p.N + (p.colsum ? (size_t)p.cs_parts * p.M : 0);
        float* dst = p.ws + p.zslice * slice + (size_t)m * p.N + n0;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        return;
    }
    v[0] += cc.bias.x; v[1] += cc.bias.y; v[2] += cc.bias.z; v[3] += cc.bias.w;
    if constexpr (EPI == EPI_BF16) {
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n0) =
            make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    } else if constexpr (EPI == EPI_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n0) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (EPI == EPI_GELU) {
        ppf_float2 g01, d01, g23, d23;
        gelu_erf_both2(ppf_float2{v[0], v[1]}, g01, d01);
        gelu_erf_both2(ppf_float2{v[2], v[3]}, g23, d23);
        const float g[4] = {g01.x, g01.y, g23.x, g23.y}, d[4] = {d01.x, d01.y, d23.x, d23.y};
        // gelu' is not read again before the backward pass: streaming (non-temporal) store, keeps L2 / MALL for the operands
        __builtin_nontemporal_store(gelu8_pack4(d[0], d[1], d[2], d[3]), reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(p.aux_out) + (size_t)m * p.ldaux + n0));
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n0) = make_uint2(pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3]));
    } else if constexpr (EPI == EPI_SIGMOID_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = 1.0f / (1.0f + __expf(-v[i]));
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n0) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (EPI == EPI_RESID) {
        if (p.aux_out)
            *reinterpret_cast<uint2*>(p.aux_out + (size_t)m * p.ldaux + n0) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        const float s = rr.rowscale;
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n0) =
            make_float4(rr.res.x + v[0] * cc.colscale.x * s, rr.res.y + v[1] * cc.colscale.y * s, rr.res.z + v[2] * cc.colscale.z * s,
                        rr.res.w + v[3] * cc.colscale.w * s);
    } else if constexpr (EPI == EPI_DGELU) {
        float hd[4];
        gelu8_unpack4(rr.aux.x, hd);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n0) =
            make_uint2(pack_bf16x2(v[0] * hd[0], v[1] * hd[1]), pack_bf16x2(v[2] * hd[2], v[3] * hd[3]));
    }
}

// ---- 8-column form of the bf16-output epilogues (EPI_BF16 / EPI_GELU / EPI_DGELU): one 16-byte store per lane instead of two
// 8-byte ones.  The epilogue of these GEMMs is store-ISSUE bound (knock-out timing, profiles/r2_gemm_knockout.txt: the stores of
// fc1+GELU cost 73 of 151 us), so halving the number of store instructions is worth more than any main-loop change.
template <int EPI> struct EpiWide { static constexpr bool value = (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_DGELU); };
struct EpiCols8 { float4 bias[2]; };
struct EpiRow8 { uint4 aux; };

template <int EPI>
__device__ __forceinline__ EpiCols8 epi_load_cols8(const GemmParams& p, int n0, bool ok) {
    EpiCols8 c;
    c.bias[0] = c.bias[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok && p.bias) { c.bias[0] = *reinterpret_cast<const float4*>(p.bias + n0); c.bias[1] = *reinterpret_cast<const float4*>(p.bias + n0 + 4); }
    return c;
}
template <int EPI>
__device__ __forceinline__ EpiRow8 epi_load_row8(const GemmParams& p, int m, int n0, bool ok) {
    EpiRow8 r;
    r.aux = make_uint4(0, 0, 0, 0);
    if constexpr (EPI == EPI_DGELU) {
        if (ok) {                                 // read once, 10 ms after it was written: streaming load
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(p.aux_in) + (size_t)m * p.ldaux + n0));
            r.aux = make_uint4(t.x, t.y, 0, 0);
        }
    }
    return r;
}
template <int EPI>
__device__ __forceinline__ void epi_store8(const GemmParams& p, int m, int n0, const float4& a, const float4& b, const EpiCols8& cc, const EpiRow8& rr) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    float v[8] = {a.x * p.alpha + cc.bias[0].x, a.y * p.alpha + cc.bias[0].y, a.z * p.alpha + cc.bias[0].z, a.w * p.alpha + cc.bias[0].w,
                  b.x * p.alpha + cc.bias[1].x, b.y * p.alpha + cc.bias[1].y, b.z * p.alpha + cc.bias[1].z, b.w * p.alpha + cc.bias[1].w};
    bf16_t* dst = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n0;
    if constexpr (EPI == EPI_BF16) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    } else if constexpr (EPI == EPI_GELU) {
        ppf_float2 g[4], d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gelu_erf_both2(ppf_float2{v[2 * i], v[2 * i + 1]}, g[i], d[i]);
        // gelu' is not read again before the backward pass: streaming (non-temporal) store, keeps L2 / MALL for the operands
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 dv = {gelu8_pack4(d[0].x, d[0].y, d[1].x, d[1].y), gelu8_pack4(d[2].x, d[2].y, d[3].x, d[3].y)};
        __builtin_nontemporal_store(dv, reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(p.aux_out) + (size_t)m * p.ldaux + n0));
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(g[0].x, g[0].y), pack_bf16x2(g[1].x, g[1].y), pack_bf16x2(g[2].x, g[2].y), pack_bf16x2(g[3].x, g[3].y));
    } else if constexpr (EPI == EPI_DGELU) {
        float ha[4], hb[4];
        gelu8_unpack4(rr.aux.x, ha);
        gelu8_unpack4(rr.aux.y, hb);
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(v[0] * ha[0], v[1] * ha[1]), pack_bf16x2(v[2] * ha[2], v[3] * ha[3]),
                                                    pack_bf16x2(v[4] * hb[0], v[5] * hb[1]), pack_bf16x2(v[6] * hb[2], v[7] * hb[3]));
    }
}


}  // namespace ppfg
