// LayerNorm forward / backward (fp32 residual stream in, bf16 activations out) for gfx950.
// One wavefront per row: lane l owns the column pairs c = 2*l + 128*j (coalesced 512-B segments),
// row statistics by wave shuffles, two-pass variance in registers.  HBM-bound streaming kernels.
#include "ppf_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int WAVES = 4;          // rows processed concurrently per 256-thread workgroup
constexpr int MAXJ = 8;           // D <= 1024

template <int NJ>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const int* __restrict__ row_map,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     bf16_t* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2 wv[NJ], bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = 2 * lane + 128 * j;
        wv[j] = c < D ? *reinterpret_cast<const float2*>(w + c) : make_float2(0.f, 0.f);
        bv[j] = c < D ? *reinterpret_cast<const float2*>(b + c) : make_float2(0.f, 0.f);
    }
    const float invD = 1.0f / (float)D;
    for (int r = blockIdx.x * WAVES + wave; r < rows; r += gridDim.x * WAVES) {
        const size_t src = row_map ? (size_t)row_map[r] : (size_t)r;
        const float* xr = x + src * D;
        float2 v[NJ];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = 2 * lane + 128 * j;
            v[j] = c < D ? *reinterpret_cast<const float2*>(xr + c) : make_float2(0.f, 0.f);
            s += v[j].x + v[j].y;
        }
        const float mu = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = 2 * lane + 128 * j;
            if (c < D) { const float a = v[j].x - mu, bb = v[j].y - mu; q += a * a + bb * bb; }
        }
        const float rs = rsqrtf(wave_sum(q) * invD + eps);
        bf16_t* yr = y + (size_t)r * D;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = 2 * lane + 128 * j;
            if (c < D)
                *reinterpret_cast<uint32_t*>(yr + c) =
                    pack_bf16x2((v[j].x - mu) * rs * wv[j].x + bv[j].x, (v[j].y - mu) * rs * wv[j].y + bv[j].y);
        }
        if (lane == 0) { mean_out[r] = mu; rstd_out[r] = rs; }
    }
}

struct LnBwdParams {
    const bf16_t* dy; const float* x; const int* row_map; const float* w; const float* mean; const float* rstd;
    const float* dres_in; float* dx_out; float* dw; float* db;
    bf16_t* cast_out; const float* rowscale; int rows_per_group; const float* colscale; float* dbias_next;
    const bf16_t* branch; float* dcolscale;
    int rows, D;
    float* partial;          // [gridDim.x][4][D] per-workgroup column sums (dw, db, dbias_next, dcolscale) or null: atomics
};

// dx_out[src] = dres_in[src] + LN'(dy);  dw += sum dy*xhat;  db += sum dy;
// optional fused cast for the residual branch below: cast_out = bf16(rowscale*colscale*dx_out), its column
// sums (bias grad of the branch GEMM) and the LayerScale grad sum(rowscale*dx_out*branch).
// With dy == nullptr the LN part is skipped (pure scale/cast/column-sum pass over dres_in).
template <int NJ>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdParams p) {
    __shared__ float red[WAVES][NJ * 2 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = p.D;
    float2 wv[NJ], cs[NJ];
    float2 adw[NJ], adb[NJ], anb[NJ], acs[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = 2 * lane + 128 * j;
        wv[j] = (p.dy && c < D) ? *reinterpret_cast<const float2*>(p.w + c) : make_float2(0.f, 0.f);
        cs[j] = (p.colscale && c < D) ? *reinterpret_cast<const float2*>(p.colscale + c) : make_float2(1.f, 1.f);
        adw[j] = adb[j] = anb[j] = acs[j] = make_float2(0.f, 0.f);
    }
    const float invD = 1.0f / (float)D;
    for (int r = blockIdx.x * WAVES + wave; r < p.rows; r += gridDim.x * WAVES) {
        const size_t src = p.row_map ? (size_t)p.row_map[r] : (size_t)r;
        float2 dx[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = 2 * lane + 128 * j;
            dx[j] = (p.dres_in && c < D) ? *reinterpret_cast<const float2*>(p.dres_in + src * D + c) : make_float2(0.f, 0.f);
        }
        if (p.dy) {
            const float mu = p.mean[r], rs = p.rstd[r];
            float2 xh[NJ], g[NJ];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = 2 * lane + 128 * j;
                if (c < D) {
                    const float2 xv = *reinterpret_cast<const float2*>(p.x + src * D + c);
                    const float2 dyv = unpack_bf16x2(*reinterpret_cast<const uint32_t*>(p.dy + (size_t)r * D + c));
                    xh[j] = make_float2((xv.x - mu) * rs, (xv.y - mu) * rs);
                    g[j] = make_float2(dyv.x * wv[j].x, dyv.y * wv[j].y);
                    s1 += g[j].x + g[j].y;
                    s2 += g[j].x * xh[j].x + g[j].y * xh[j].y;
                    adw[j].x += dyv.x * xh[j].x; adw[j].y += dyv.y * xh[j].y;
                    adb[j].x += dyv.x; adb[j].y += dyv.y;
                } else {
                    xh[j] = g[j] = make_float2(0.f, 0.f);
                }
            }
            const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                dx[j].x += rs * (g[j].x - c1 - xh[j].x * c2);
                dx[j].y += rs * (g[j].y - c1 - xh[j].y * c2);
            }
        }
        const float rsc = p.rowscale ? p.rowscale[src / p.rows_per_group] : 1.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = 2 * lane + 128 * j;
            if (c < D) {
                if (p.dx_out) *reinterpret_cast<float2*>(p.dx_out + src * D + c) = dx[j];
                if (p.cast_out) {
                    const float sx = dx[j].x * rsc, sy = dx[j].y * rsc;
                    if (p.branch) {
                        const float2 br = unpack_bf16x2(*reinterpret_cast<const uint32_t*>(p.branch + src * D + c));
                        acs[j].x += sx * br.x; acs[j].y += sy * br.y;
                    }
                    const uint32_t pk = pack_bf16x2(sx * cs[j].x, sy * cs[j].y);
                    *reinterpret_cast<uint32_t*>(p.cast_out + src * D + c) = pk;
                    const float2 rt = unpack_bf16x2(pk);
                    anb[j].x += rt.x; anb[j].y += rt.y;
                }
            }
        }
    }
    // workgroup reduction of the column partial sums, then one atomic per column per workgroup
    auto flush = [&](float2 (&a)[NJ], float* dst, int which) {
        if (!dst) return;
        float* part = p.partial ? p.partial + ((size_t)blockIdx.x * 4 + which) * D : nullptr;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) { red[wave][(j * 64 + lane) * 2] = a[j].x; red[wave][(j * 64 + lane) * 2 + 1] = a[j].y; }
        __syncthreads();
        for (int i = threadIdx.x; i < NJ * 128; i += 256) {
            const int j = i >> 7, rem = i & 127, l = rem >> 1, e = rem & 1;
            const int c = 2 * l + 128 * j + e;
            if (c < D) {
                float s = 0.f;
#pragma unroll
                for (int wv_ = 0; wv_ < WAVES; ++wv_) s += red[wv_][i];
                if (part) part[c] = s; else unsafeAtomicAdd(dst + c, s);
            }
        }
    };
    if (p.dy) { flush(adw, p.dw, 0); flush(adb, p.db, 1); }
    if (p.cast_out) { flush(anb, p.dbias_next, 2); if (p.branch) flush(acs, p.dcolscale, 3); }
}

// Bandwidth-oriented variant for D = 32*V*NJ (V = 4: D % 128 == 0, V = 2: D % 64 == 0): half a wavefront per row, so a
// wave streams two independent rows at a time with 16-/8-byte accesses (twice the loads in flight of the generic kernel).
// (round 6: non-temporal loads / stores of any of the five streams change nothing, stand-alone or in the step: profiles/r6_ln_bwd_nt.txt)
// LS = false: no LayerScale operands (colscale / branch / dcolscale all null, the DeiT case): 24 fewer live registers -> 4 waves/SIMD.
template <int V, int NJ, bool LS>
__global__ __launch_bounds__(256) void ln_bwd2_kernel(const LnBwdParams p) {
    __shared__ float red[WAVES][NJ * V * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, half = lane >> 5;
    const int D = p.D;
    float wv[NJ][V], cs[NJ][V], adw[NJ][V], adb[NJ][V], anb[NJ][V], acs[NJ][V];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int c = V * l32 + 32 * V * j + e;
            wv[j][e] = p.dy ? p.w[c] : 0.f;
            cs[j][e] = (LS && p.colscale) ? p.colscale[c] : 1.f;
            adw[j][e] = adb[j][e] = anb[j][e] = acs[j][e] = 0.f;
        }
    const float invD = 1.0f / (float)D;
    for (int r0 = (blockIdx.x * WAVES + wave) * 2; r0 < p.rows; r0 += gridDim.x * WAVES * 2) {
        const int r = r0 + half;
        const bool ok = r < p.rows;
        const size_t src = ok ? (p.row_map ? (size_t)p.row_map[r] : (size_t)r) : 0;
        float dx[NJ][V];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = V * l32 + 32 * V * j;
            if (p.dres_in && ok) {
                if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p.dres_in + src * D + c); dx[j][0] = t.x; dx[j][1] = t.y; dx[j][2] = t.z; dx[j][3] = t.w; }
                else { const float2 t = *reinterpret_cast<const float2*>(p.dres_in + src * D + c); dx[j][0] = t.x; dx[j][1] = t.y; }
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) dx[j][e] = 0.f;
            }
        }
        if (p.dy) {
            const float mu = ok ? p.mean[r] : 0.f, rs = ok ? p.rstd[r] : 0.f;
            float xh[NJ][V], g[NJ][V];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = V * l32 + 32 * V * j;
                float xv[V], dyv[V];
                if (ok) {
                    if constexpr (V == 4) {
                        const float4 t = *reinterpret_cast<const float4*>(p.x + src * D + c); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                        const uint2 u = *reinterpret_cast<const uint2*>(p.dy + (size_t)r * D + c);
                        const float2 a = unpack_bf16x2(u.x), b = unpack_bf16x2(u.y); dyv[0] = a.x; dyv[1] = a.y; dyv[2] = b.x; dyv[3] = b.y;
                    } else {
                        const float2 t = *reinterpret_cast<const float2*>(p.x + src * D + c); xv[0] = t.x; xv[1] = t.y;
                        const float2 a = unpack_bf16x2(*reinterpret_cast<const uint32_t*>(p.dy + (size_t)r * D + c)); dyv[0] = a.x; dyv[1] = a.y;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < V; ++e) { xv[e] = 0.f; dyv[e] = 0.f; }
                }
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    xh[j][e] = (xv[e] - mu) * rs;
                    g[j][e] = dyv[e] * wv[j][e];
                    s1 += g[j][e];
                    s2 += g[j][e] * xh[j][e];
                    adw[j][e] += dyv[e] * xh[j][e];
                    adb[j][e] += dyv[e];
                }
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            const float c1 = s1 * invD, c2 = s2 * invD;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < V; ++e) dx[j][e] += rs * (g[j][e] - c1 - xh[j][e] * c2);
        }
        if (!ok) continue;
        const float rsc = p.rowscale ? p.rowscale[src / p.rows_per_group] : 1.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = V * l32 + 32 * V * j;
            if (p.dx_out) {
                if constexpr (V == 4) *reinterpret_cast<float4*>(p.dx_out + src * D + c) = make_float4(dx[j][0], dx[j][1], dx[j][2], dx[j][3]);
                else *reinterpret_cast<float2*>(p.dx_out + src * D + c) = make_float2(dx[j][0], dx[j][1]);
            }
            if (p.cast_out) {
                float sx[V];
#pragma unroll
                for (int e = 0; e < V; ++e) sx[e] = dx[j][e] * rsc;
                if (LS && p.branch) {
                    float br[V];
                    if constexpr (V == 4) {
                        const uint2 u = *reinterpret_cast<const uint2*>(p.branch + src * D + c);
                        const float2 a = unpack_bf16x2(u.x), b = unpack_bf16x2(u.y); br[0] = a.x; br[1] = a.y; br[2] = b.x; br[3] = b.y;
                    } else {
                        const float2 a = unpack_bf16x2(*reinterpret_cast<const uint32_t*>(p.branch + src * D + c)); br[0] = a.x; br[1] = a.y;
                    }
#pragma unroll
                    for (int e = 0; e < V; ++e) acs[j][e] += sx[e] * br[e];
                }
                uint32_t pk[V / 2];
#pragma unroll
                for (int e = 0; e < V; e += 2) {
                    pk[e / 2] = LS ? pack_bf16x2(sx[e] * cs[j][e], sx[e + 1] * cs[j][e + 1]) : pack_bf16x2(sx[e], sx[e + 1]);
                    const float2 rt = unpack_bf16x2(pk[e / 2]);
                    anb[j][e] += rt.x; anb[j][e + 1] += rt.y;
                }
                if constexpr (V == 4) *reinterpret_cast<uint2*>(p.cast_out + src * D + c) = make_uint2(pk[0], pk[1]);
                else *reinterpret_cast<uint32_t*>(p.cast_out + src * D + c) = pk[0];
            }
        }
    }
    auto flush = [&](float (&a)[NJ][V], float* dst, int which) {
        if (!dst) return;
        float* part = p.partial ? p.partial + ((size_t)blockIdx.x * 4 + which) * D : nullptr;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float v = a[j][e] + __shfl_xor(a[j][e], 32, 64);       // the two rows a wave works on
                if (half == 0) red[wave][(j * 32 + l32) * V + e] = v;
            }
        __syncthreads();
        for (int i = threadIdx.x; i < NJ * V * 32; i += 256) {
            float s = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < WAVES; ++w_) s += red[w_][i];
            const int j = i / (32 * V), rem = i % (32 * V);
            if (part) part[32 * V * j + rem] = s;                            // column = V*l32 + e + 32*V*j
            else unsafeAtomicAdd(dst + 32 * V * j + rem, s);
        }
    };
    if (p.dy) { flush(adw, p.dw, 0); flush(adb, p.db, 1); }
    if (p.cast_out) { flush(anb, p.dbias_next, 2); if (LS && p.branch) flush(acs, p.dcolscale, 3); }
}

// dst[c] += sum over workgroups of partial[wg][which][c], in a fixed order (deterministic; replaces ~1.8M fp32 atomics on 1152
// addresses per LayerNorm backward).  ONE launch: a 1024-thread workgroup owns 64 columns of one of the four arrays; its 16 lane
// groups each add every 16th partial (independent loads, eight in flight per lane), the 16 group sums are added in order through LDS.
constexpr int LN_RS = 16;
__global__ __launch_bounds__(1024) void ln_colsum_reduce_kernel(const float* __restrict__ partial, int nblocks, int D, float* d0, float* d1,
                                                                  float* d2, float* d3) {
    __shared__ float red[16][64];
    const int which = blockIdx.y;
    float* dst = which == 0 ? d0 : which == 1 ? d1 : which == 2 ? d2 : d3;
    if (!dst) return;
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < D) {
#pragma unroll 8
        for (int b = g; b < nblocks; b += 16) s += partial[((size_t)b * 4 + which) * D + c];
    }
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        dst[c] += t;
    }
}

int ln_bwd_grid(int rows) {
    // four 4-wave workgroups per CU is what the register budget holds: a grid of exactly that size has no partly filled last round
    // (stand-alone 79 -> 64 us at 50 432 rows) and writes a third fewer column partials
    constexpr int cap = 1024;
    const int g = (rows + WAVES * 8 - 1) / (WAVES * 8);
    return g < cap ? g : cap;
}

template <typename F>
int dispatch_nj(int D, F&& f) {
    const int nj = (D + 127) / 128;
    switch (nj) {
        case 1: return f(std::integral_constant<int, 1>());
        case 2: return f(std::integral_constant<int, 2>());
        case 3: return f(std::integral_constant<int, 3>());
        case 4: return f(std::integral_constant<int, 4>());
        case 6: return f(std::integral_constant<int, 6>());
        case 8: return f(std::integral_constant<int, 8>());
        default: break;
    }
    ppf_set_error("layernorm: unsupported width D=%d (need D even, ceil(D/128) in {1,2,3,4,6,8})", D);
    return PPF_ERR_SHAPE;
}

}  // namespace

extern "C" {

int ppf_layernorm_fwd(const float* x, const int* row_map, const float* w, const float* b, void* y, float* mean, float* rstd,
                      int rows, int D, float eps, hipStream_t stream) {
    PPF_CHECK_ARG(rows > 0 && D > 0 && (D % 2) == 0 && D <= 128 * MAXJ, PPF_ERR_SHAPE, "ppf_layernorm_fwd: bad shape rows=%d D=%d", rows, D);
    const int grid = min((rows + WAVES - 1) / WAVES, 256 * 16);
    return dispatch_nj(D, [&](auto nj) {
        hipLaunchKernelGGL((ln_fwd_kernel<decltype(nj)::value>), dim3(grid), dim3(256), 0, stream, x, row_map, w, b, (bf16_t*)y, mean, rstd, rows, D, eps);
        PPF_LAUNCH_CHECK();
        return 0;
    });
}

int ppf_layernorm_bwd(const void* dy, const float* x, const int* row_map, const float* w, const float* mean, const float* rstd,
                      const float* dres_in, float* dx_out, float* dw, float* db, void* cast_out, const float* rowscale,
                      int rows_per_group, const float* colscale, float* dbias_next, const void* branch, float* dcolscale,
                      int rows, int D, float* partial, size_t partial_bytes, hipStream_t stream) {
    PPF_CHECK_ARG(rows > 0 && D > 0 && (D % 2) == 0 && D <= 128 * MAXJ, PPF_ERR_SHAPE, "ppf_layernorm_bwd: bad shape rows=%d D=%d", rows, D);
    PPF_CHECK_ARG(dy != nullptr || dres_in != nullptr, PPF_ERR_ARG, "ppf_layernorm_bwd: need dy or dres_in");
    LnBwdParams p;
    p.dy = (const bf16_t*)dy; p.x = x; p.row_map = row_map; p.w = w; p.mean = mean; p.rstd = rstd; p.dres_in = dres_in; p.dx_out = dx_out;
    p.dw = dw; p.db = db; p.cast_out = (bf16_t*)cast_out; p.rowscale = rowscale; p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
    p.colscale = colscale; p.dbias_next = dbias_next; p.branch = (const bf16_t*)branch; p.dcolscale = dcolscale; p.rows = rows; p.D = D;
    const int grid = ln_bwd_grid(rows);                                    // >= 8 rows per wave: amortise the column flush
    PPF_CHECK_ARG(partial == nullptr || partial_bytes >= (size_t)(grid + LN_RS) * 4 * D * sizeof(float), PPF_ERR_ARG,
                  "ppf_layernorm_bwd: partial-sum workspace needs ppf_layernorm_bwd_blocks(rows)*4*D*4 = %zu bytes", (size_t)(grid + LN_RS) * 4 * D * sizeof(float));
    p.partial = partial;
    const bool ls = colscale != nullptr || branch != nullptr || dcolscale != nullptr;
#define PPF_LN2(V, NJ) { if (ls) hipLaunchKernelGGL((ln_bwd2_kernel<V, NJ, true>), dim3(grid), dim3(256), 0, stream, p); \
                         else hipLaunchKernelGGL((ln_bwd2_kernel<V, NJ, false>), dim3(grid), dim3(256), 0, stream, p); \
                         PPF_LAUNCH_CHECK(); return 0; }
    if (D % 128 == 0 && D <= 512) { switch (D / 128) { case 1: PPF_LN2(4, 1) case 2: PPF_LN2(4, 2) case 3: PPF_LN2(4, 3) case 4: PPF_LN2(4, 4) } }
    if (D % 64 == 0 && D <= 512) { switch (D / 64) { case 1: PPF_LN2(2, 1) case 3: PPF_LN2(2, 3) case 5: PPF_LN2(2, 5) case 7: PPF_LN2(2, 7) } }
#undef PPF_LN2
    return dispatch_nj(D, [&](auto nj) {
        hipLaunchKernelGGL((ln_bwd_kernel<decltype(nj)::value>), dim3(grid), dim3(256), 0, stream, p);
        PPF_LAUNCH_CHECK();
        return 0;
    });
}

// Rows of the partial-sum workspace for `rows` rows (workgroups + second-level slices): it holds blocks * 4 * D floats.
int ppf_layernorm_bwd_blocks(int rows) { return ln_bwd_grid(rows) + LN_RS; }

// Second half of ppf_layernorm_bwd(partial != NULL): dw / db / dbias_next / dcolscale += column sums of the per-workgroup partials
// (pass the same pointers, NULL where ppf_layernorm_bwd got NULL).  May run on another stream once the first half is complete.
int ppf_layernorm_bwd_reduce(const float* partial, int rows, int D, float* dw, float* db, float* dbias_next, float* dcolscale, hipStream_t stream) {
    PPF_CHECK_ARG(partial && rows > 0 && D > 0, PPF_ERR_ARG, "ppf_layernorm_bwd_reduce: bad arguments");
    const int nb = ln_bwd_grid(rows);
    hipLaunchKernelGGL(ln_colsum_reduce_kernel, dim3((D + 63) / 64, 4), dim3(1024), 0, stream, partial, nb, D, dw, db, dbias_next, dcolscale);
    PPF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
