// Full-row bf16 MFMA GEMM for gfx950 with row-wise fused epilogues (LayerNorm forward / backward):
//
//   acc[m][n] = sum_k A[m*lda + k] * B[n*ldb + k]        m in one TILE of rows, n = 0 .. D-1 (ALL output columns), fp32 accumulate
//
// for the products of a transformer block whose output width is the model width D (deit:58,80 proj / fc2 forward; the input
// gradients of fc1 / qkv / proj): a workgroup owns complete output ROWS, so everything the reference does next to those rows --
// bias, DropPath-scaled residual add and the following LayerNorm (deit:76-81), or the LayerNorm backward in front of the residual
// gradient -- happens in the epilogue while the rows are on chip.  The normalised activations / the branch gradient leave as bf16,
// the residual stream as fp32; the GEMM output itself never goes to HBM, and the LayerNorm kernels (one more read of the residual
// stream each) disappear.
//
// Geometry (MI355X: 256 CUs, 160 KiB LDS, 512 VGPRs per SIMD lane): ONE 512-thread workgroup per CU, one tile of <= 16*MT rows per
// workgroup (the host makes a tile one sample: 197 tokens -> 13 m-tiles of 16, batch 256 = 256 tiles = 256 CUs).  The 8 waves are
// WM x WN: wave (wm, wn) owns m-tiles [wm*MTW, (wm+1)*MTW) x columns [48*wn, 48*wn+48) = MTW x 3 accumulator tiles of
// v_mfma_f32_16x16x32_bf16 (D = 384: 1 x 8 waves, 39 tiles = 156 VGPRs; D = 192: 2 x 4 waves).  Operands are swapped (first = B / n,
// second = A / m) so a lane ends up with 4 consecutive columns of one row.
//
// Main loop: two LDS slots of 64 contraction values ([16*MT + D rows][128 B], 74 KiB at D = 384), double-buffered: the next stage
// is filled by global_load_lds_dwordx4 while the current one is multiplied, ONE s_waitcnt + barrier per stage (78 MFMAs per wave).
// Rows are FULL 128-byte lines: the L2 serves requests, not bytes -- the first version of this kernel (32-value stages, 64-byte
// row segments, three stages in flight) was bound by the L2 request rate at 9 TB/s with the MFMAs completely hidden under the loads
// (knock-out: loads alone 1.03-1.43 us per 32 values, MFMAs alone 0.73; profiles/r3_rowgemm.txt).  The bank swizzle of the
// 128-byte rows (16-byte chunk ^= (row >> 1) & 7: conflict-free ds_read_b128 fragments) is applied to the per-lane SOURCE address,
// the LDS image stays lane-linear.  A wave's 9-10 one-KiB pieces are spread over its MFMA stream (a wave that issues them back to
// back sits in the issue stage until the CU's memory pipeline has taken them all).  A (activations, read once) streams from HBM,
// B (the weights, the same tile for every CU at the same time) from L2, every workgroup walking it from a different start.
//
// Epilogue: 64 rows at a time go through an LDS image [64][D + 4] fp32 (the ring is dead by then) so that 32 (64 for the LayerNorm
// backward) lanes own one row in 16-byte (8-byte) column pieces: row statistics by 5 (6) shuffles, coalesced 512-byte fp32 /
// 256-byte bf16 row segments to and from HBM.
#include "ppf_common.h"
#include "ppf_hip.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int RG_BK = 64, RG_S = 2, RG_NTHR = 512;                  // k per stage, LDS slots (double buffer), threads
constexpr bool RG_NT_A = true;                                       // the A pieces (read once) arrive with the non-temporal hint
constexpr int RG_ISSUE_FRAC = 100;                                   // percent of a stage's MFMA stream over which a wave issues its pieces
enum RgEpi { RG_BF16 = 0, RG_RESID_LN = 1, RG_LNBWD = 2, RG_LNBWD_LS = 3 };

typedef __attribute__((address_space(3))) void rg_lds_t;
typedef const __attribute__((address_space(1))) void rg_gbl_t;

struct RowGemmParams {
    const bf16_t* A; const bf16_t* B;
    int M, K, lda, ldb, rows_per_tile;
    const float* bias;             // [D] added to the accumulator, or null
    // RG_BF16: out16 = bf16(acc + bias)
    bf16_t* out16;
    // RG_RESID_LN: xout = res + rowscale[m / rows_per_group] * colscale[n] * (acc + bias);  ln_out = bf16(LN(xout) * ln_w + ln_b), mean / rstd
    //              per row; aux_out = bf16(acc + bias) (the unscaled branch: LayerScale's gradient needs it), colscale / aux_out optional
    const float* res; float* xout; const float* rowscale; int rows_per_group; const float* colscale; bf16_t* aux_out;
    const float* ln_w; const float* ln_b; bf16_t* ln_out; float* ln_mean; float* ln_rstd; float eps;
    // RG_LNBWD: dn = acc (+ bias); dx_out = dres_in + LN'(dn; x, mean, rstd, w);  cast_out = bf16(rowscale * dx_out);
    //           partial[tile][0][n] = sum_m dn * xhat (d ln weight), partial[tile][1][n] = sum_m dn (d ln bias)
    // RG_LNBWD_LS (CaiT LayerScale below the residual, cait:153-155): cast_out = bf16(rowscale * colscale[n] * dx_out) and
    //           partial[tile][2][n] = sum_m rowscale * dx_out * branch[m][n] (d gamma; branch = the unscaled branch output saved by forward)
    const float* x; const float* mean; const float* rstd; const float* w;
    const float* dres_in; float* dx_out; bf16_t* cast_out; float* partial; const bf16_t* branch;
};

// 16-byte chunk c of a 128-byte row r is stored at chunk c ^ swz(r): the four 16-lane groups of a ds_read_b128 (lanes {0-3,12-15,20-27},
// {4-11,16-19,28-31}, +32) that read 16 rows x 4 chunks then touch 16 different 16-byte slots each (conflict-free)
__device__ __forceinline__ int rg_swz(int row) { return (row >> 1) & 7; }

// VW (2 or 4) consecutive floats <-> registers; bf16 store of VW values
template <int VW> __device__ __forceinline__ void rg_ld(float (&d)[VW], const float* src) {
    if constexpr (VW == 4) { const float4 t = *reinterpret_cast<const float4*>(src); d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
    else { const float2 t = *reinterpret_cast<const float2*>(src); d[0] = t.x; d[1] = t.y; }
}
template <int VW> __device__ __forceinline__ void rg_st(float* dst, const float (&v)[VW]) {
    if constexpr (VW == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    else *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
}
template <int VW> __device__ __forceinline__ void rg_st16(bf16_t* dst, const float (&v)[VW]) {
    if constexpr (VW == 4) *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    else *reinterpret_cast<uint32_t*>(dst) = pack_bf16x2(v[0], v[1]);
}

template <int D, int MT, int EPI>
__global__ __launch_bounds__(RG_NTHR, 2) void rowgemm_kernel(const RowGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WN = D / 48, WM = 8 / WN, MTW = (MT + WM - 1) / WM;      // waves along n / m, m-tiles per wave
    constexpr int MTP = MTW * WM;                                         // m-tiles per workgroup (>= MT)
    constexpr int A_BYTES = MTP * 16 * 128, B_BYTES = D * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int NPA = A_BYTES / 1024, NPB = B_BYTES / 1024, NP = NPA + NPB;       // 1-KiB pieces (8 rows x 128 B) per stage
    constexpr int FR = NP / 8, XP = NP % 8;                                // pieces every wave issues per stage, + one more for waves < XP
    constexpr int LDP = D + 4;                                             // fp32 pitch of the epilogue image (rg_lds_bytes covers it)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wm = wave / WN;
    const int row0 = blockIdx.x * p.rows_per_tile;
    const int rows = min(p.rows_per_tile, p.M - row0);                     // valid rows of this tile (>= 1)

    // ---- LDS-DMA pieces of this wave: piece b = wave + 8 i; lane l carries row (l >> 3) of the piece, LDS chunk (l & 7) ----------------
    const unsigned char* gA = reinterpret_cast<const unsigned char*>(p.A) + (size_t)row0 * p.lda * 2;
    const unsigned char* gB = reinterpret_cast<const unsigned char*>(p.B);
    // every workgroup walks the B pieces from a different start: all CUs multiply the same weight tile at the same time, and without
    // the rotation they would all ask the L2 for the same line in the same cycle
    const int rot = (int)((blockIdx.x * 7u) % (unsigned)NPB);
    const unsigned char* src[FR + 1];
    int dst[FR + 1];
#pragma unroll
    for (int i = 0; i <= FR; ++i) {
        const int b = min(wave + 8 * i, NP - 1);
        int prow;                                                          // first row of the piece inside its operand tile
        if (b < NPA) { prow = b * 8; dst[i] = b * 1024; }
        else { int bb = b - NPA + rot; bb -= bb >= NPB ? NPB : 0; prow = bb * 8; dst[i] = A_BYTES + bb * 1024; }
        const int r = prow + (lane >> 3);
        const int cs = (lane & 7) ^ rg_swz(r);                             // SOURCE chunk of this lane's LDS slot
        if (b < NPA) src[i] = gA + (size_t)min(r, rows - 1) * p.lda * 2 + cs * 16;       // rows past the tile repeat its last row (masked later)
        else src[i] = gB + (size_t)r * p.ldb * 2 + cs * 16;
    }

    // ---- fragment read addresses: lane l of a 16-row tile reads row (l & 15), chunk 4 ks + (l >> 4), stored at chunk ^ swz(row) ----
    int frag[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) frag[ks] = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ rg_swz(lane & 15)) << 4);
    const int a_base = wm * MTW * 2048, b_base = A_BYTES + wn * 48 * 128;

    f32x4 acc[MTW][3];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / RG_BK;

    // L2 touch-prefetch: both operands arrive from beyond the L2 (A is read once, and all CUs reach the same B tile for the first time
    // together), so an LDS-DMA piece takes the memory-side latency (~2 us) and the double buffer runs at slot bytes / latency.  One
    // dword per 128-byte line of the stage AFTER the one being fetched brings those lines into the L2 a stage early: waves 1.. touch
    // the A rows (64 lines per instruction), wave 0 the B rows congruent to this workgroup's index inside its XCD (the 32 CUs of an
    // XCD share the L2: 12 lines each).  The loaded dwords are never used; their registers stay tied until the next counted wait.
    unsigned touch = 0;
    const unsigned char* tsrc = nullptr;
    {
        constexpr int TA_W = (MTP * 16 + 63) / 64;                         // waves 1 .. TA_W touch A
        if (wave == 0) {
            const int n = (int)((blockIdx.x >> 3) & 31) + 32 * lane;
            if (n < D) tsrc = gB + (size_t)n * p.ldb * 2;
        } else if (wave <= TA_W) {
            const int r = (wave - 1) * 64 + lane;
            if (r < rows) tsrc = gA + (size_t)r * p.lda * 2;
        }
    }
    const bool do_touch = tsrc != nullptr;

    // One stage = 64 contraction values out of slot `slot`; with ISSUE the wave's pieces of the next stage go into the other slot,
    // spread over the MFMA stream: the body is cut into one scheduling region per (k-substep, m-tile) -- sched_barrier pins the
    // region order (LDS-DMA writes and fragment reads alias for the compiler, group masks cannot interleave them) -- and region q is
    // [its share of the pieces] [fragment read PF regions ahead] [3 MFMAs].
    auto stage = [&](int slot, int kt_next, auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        constexpr int PF = 3, NR = 2 * MTW;                                // fragment prefetch distance, regions per stage
        const unsigned char* st = smem + slot * STAGE;
        unsigned char* nbase = smem + (slot ^ 1) * STAGE;                   // wave-uniform destination; the hardware adds lane * 16
        const size_t koff = (size_t)kt_next * (RG_BK * 2);
        if constexpr (ISSUE && XP > 0) {
            if (wave < XP) __builtin_amdgcn_global_load_lds((rg_gbl_t*)(src[FR] + koff), (rg_lds_t*)(nbase + dst[FR]), 16, 0, 0);
        }
        if constexpr (ISSUE) {
            if (do_touch && kt_next + 1 < nk) asm volatile("global_load_dword %0, %1, off" : "=v"(touch) : "v"(tsrc + koff + RG_BK * 2) : "memory");
        }
        bf16x8 fb[2][3], fa[NR];
#pragma unroll
        for (int j = 0; j < 3; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(st + b_base + j * 2048 + frag[0]);
#pragma unroll
        for (int q = 0; q < PF && q < NR; ++q) fa[q] = *reinterpret_cast<const bf16x8*>(st + a_base + (q % MTW) * 2048 + frag[q / MTW]);
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ISSUE) {
                // the wave's pieces go out over the first NRI regions only: with two slots the stage ends in a full drain
                // (vmcnt(0)), so a piece issued late in the MFMA stream has its whole latency exposed
                constexpr int NRI = (NR * RG_ISSUE_FRAC + 99) / 100;
                const int qq = q < NRI ? q : NRI;
                const int k0 = (qq * FR) / NRI, k1 = q < NRI ? ((qq + 1) * FR) / NRI : k0;     // this region's pieces [k0, k1)
#pragma unroll
                for (int k = k0; k < k1; ++k) {
                    // A rows are read exactly once, by this CU: streaming (nt) loads keep them from displacing the weight tile in the L2
                    if (RG_NT_A && 8 * k + 7 < NPA) __builtin_amdgcn_global_load_lds((rg_gbl_t*)(src[k] + koff), (rg_lds_t*)(nbase + dst[k]), 16, 0, 2);
                    else __builtin_amdgcn_global_load_lds((rg_gbl_t*)(src[k] + koff), (rg_lds_t*)(nbase + dst[k]), 16, 0, 0);
                }
            }
            if (q == MTW - PF || (MTW < PF && q == 0)) {                    // the second k-substep's B fragments, PF regions ahead too
#pragma unroll
                for (int j = 0; j < 3; ++j) fb[1][j] = *reinterpret_cast<const bf16x8*>(st + b_base + j * 2048 + frag[1]);
            }
            if (q + PF < NR) fa[q + PF] = *reinterpret_cast<const bf16x8*>(st + a_base + ((q + PF) % MTW) * 2048 + frag[(q + PF) / MTW]);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[q % MTW][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[q / MTW][j], fa[q], acc[q % MTW][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    {                                                                      // prologue: stage 0 into slot 0
#pragma unroll
        for (int i = 0; i < FR; ++i) __builtin_amdgcn_global_load_lds((rg_gbl_t*)src[i], (rg_lds_t*)(smem + dst[i]), 16, 0, 0);
        if (XP > 0 && wave < XP) __builtin_amdgcn_global_load_lds((rg_gbl_t*)src[FR], (rg_lds_t*)(smem + dst[FR]), 16, 0, 0);
        if (do_touch && nk > 1) asm volatile("global_load_dword %0, %1, off" : "=v"(touch) : "v"(tsrc + RG_BK * 2) : "memory");
    }
    int slot = 0;
    for (int t = 0; t + 1 < nk; ++t) {
        // own pieces of stage t have landed, then everybody's; everybody has also finished reading stage t-1, whose slot is refilled
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(touch) : : "memory");
        __builtin_amdgcn_s_barrier();
        stage(slot, t + 1, std::true_type());
        slot ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(touch) : : "memory");
    __builtin_amdgcn_s_barrier();
    stage(slot, 0, std::false_type());
    __builtin_amdgcn_s_barrier();                                          // every fragment read is done: the ring becomes the epilogue's image

    // ---- epilogue -------------------------------------------------------------------------------------------------------------------
    float* img = reinterpret_cast<float*>(smem);                           // [64][LDP] fp32
    float* stash = img + 64 * LDP;                                         // [3][D]: bias | ln_w (or w) | ln_b
    constexpr bool LNB = EPI == RG_LNBWD || EPI == RG_LNBWD_LS, LS = EPI == RG_LNBWD_LS;
    constexpr int NPART = LS ? 3 : 2;                                      // column-sum arrays per tile
    float* colp = stash + 3 * D;                                           // [8 waves][NPART][D] column partial sums (LayerNorm backward)
    for (int i = tid; i < D; i += RG_NTHR) {
        stash[i] = p.bias ? p.bias[i] : 0.f;
        if constexpr (EPI == RG_RESID_LN) { stash[D + i] = p.ln_out ? p.ln_w[i] : 0.f; stash[2 * D + i] = p.ln_out ? p.ln_b[i] : 0.f; }
        if constexpr (LNB) stash[D + i] = p.w[i];
        if constexpr (LS) stash[2 * D + i] = p.colscale[i];
    }
    // LPR lanes per row, each owning the VW-column pieces at VW*jl + (D/3)*i (512-byte fp32 / 256-byte bf16 row segments per access at
    // D = 384).  The LayerNorm backward holds five values per column and two column accumulators: 2 columns per piece (64 lanes per
    // row at D = 384) keeps it inside the 256-register budget next to the accumulators of the chunks still waiting; the others use 4.
    constexpr int VW = LNB ? 2 : 4;                                        // columns per piece
    constexpr int NSEG = 3, PIECE = D / NSEG, LPR = PIECE / VW, RPW = 64 / LPR;       // pieces per row, columns a row's lanes cover per piece, lanes per row, rows per wave at a time
    static_assert(D % 3 == 0 && (LPR == 16 || LPR == 32 || LPR == 64), "D = 3 x 64 or 3 x 128");
    const int jl = lane % LPR, rsub = lane / LPR;
    float adw[NSEG][VW], adb[NSEG][VW], adg[NSEG][VW];
    if constexpr (LNB) {
#pragma unroll
        for (int i = 0; i < NSEG; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e) adw[i][e] = adb[i][e] = adg[i][e] = 0.f;
    }
    const float invD = 1.0f / (float)D;
    constexpr int NCH = (MTP + 3) / 4;                                     // 64-row chunks
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c * 64 >= rows) break;                                         // uniform: nothing valid left
        __syncthreads();                                                   // the previous chunk's readers are done (and the stash is written)
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int mtg = wm * MTW + i;                                  // global m-tile
            if ((mtg >> 2) == c) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    *reinterpret_cast<f32x4*>(img + ((mtg & 3) * 16 + (lane & 15)) * LDP + wn * 48 + j * 16 + (lane >> 4) * 4) = acc[i][j];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int g = 0; g < 8 / RPW; ++g) {
            const int r = wave * 8 + g * RPW + rsub;                       // row inside the chunk
            const int rt = c * 64 + r;                                     // row inside the tile
            const bool ok = rt < rows;
            const size_t m = (size_t)row0 + (ok ? rt : 0);
            float v[NSEG][VW];
#pragma unroll
            for (int i = 0; i < NSEG; ++i) {
                const int col = VW * jl + PIECE * i;
                rg_ld<VW>(v[i], img + r * LDP + col);
                float bv[VW];
                rg_ld<VW>(bv, stash + col);
#pragma unroll
                for (int e = 0; e < VW; ++e) v[i][e] += bv[e];
            }
            if constexpr (EPI == RG_BF16) {
                if (ok) {
#pragma unroll
                    for (int i = 0; i < NSEG; ++i) rg_st16<VW>(p.out16 + m * D + VW * jl + PIECE * i, v[i]);
                }
            } else if constexpr (EPI == RG_RESID_LN) {
                const float rsc = (ok && p.rowscale) ? p.rowscale[m / p.rows_per_group] : 1.0f;
                float ra[NSEG][VW];
#pragma unroll
                for (int i = 0; i < NSEG; ++i) {
                    if (ok) rg_ld<VW>(ra[i], p.res + m * D + VW * jl + PIECE * i);
                    else {
#pragma unroll
                        for (int e = 0; e < VW; ++e) ra[i][e] = 0.f;
                    }
                }
                if (p.aux_out && ok) {                                     // p.aux_out: uniform
#pragma unroll
                    for (int i = 0; i < NSEG; ++i) rg_st16<VW>(p.aux_out + m * D + VW * jl + PIECE * i, v[i]);
                }
                if (p.colscale) {                                          // uniform
#pragma unroll
                    for (int i = 0; i < NSEG; ++i) {
                        float cs[VW];
                        rg_ld<VW>(cs, p.colscale + VW * jl + PIECE * i);
#pragma unroll
                        for (int e = 0; e < VW; ++e) v[i][e] *= cs[e];
                    }
                }
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < NSEG; ++i)
#pragma unroll
                    for (int e = 0; e < VW; ++e) { v[i][e] = ra[i][e] + rsc * v[i][e]; s += v[i][e]; }
                if (ok) {
#pragma unroll
                    for (int i = 0; i < NSEG; ++i) rg_st<VW>(p.xout + m * D + VW * jl + PIECE * i, v[i]);
                }
                if (p.ln_out) {                                            // uniform
#pragma unroll
                    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                    const float mu = s * invD;
                    float q = 0.f;
#pragma unroll
                    for (int i = 0; i < NSEG; ++i)
#pragma unroll
                        for (int e = 0; e < VW; ++e) { const float d = v[i][e] - mu; q += d * d; }
#pragma unroll
                    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
                    const float rs = rsqrtf(q * invD + p.eps);
                    if (ok) {
#pragma unroll
                        for (int i = 0; i < NSEG; ++i) {
                            const int col = VW * jl + PIECE * i;
                            float lw[VW], lb[VW], y[VW];
                            rg_ld<VW>(lw, stash + D + col); rg_ld<VW>(lb, stash + 2 * D + col);
#pragma unroll
                            for (int e = 0; e < VW; ++e) y[e] = (v[i][e] - mu) * rs * lw[e] + lb[e];
                            rg_st16<VW>(p.ln_out + m * D + col, y);
                        }
                        if (jl == 0) { p.ln_mean[m] = mu; p.ln_rstd[m] = rs; }
                    }
                }
            } else {                                                       // RG_LNBWD
                const float mu = ok ? p.mean[m] : 0.f, rs = ok ? p.rstd[m] : 0.f;
                const float rsc = (ok && p.rowscale) ? p.rowscale[m / p.rows_per_group] : 1.0f;
                float xh[NSEG][VW], gg[NSEG][VW], dr[NSEG][VW];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < NSEG; ++i) {
                    const size_t o = m * D + VW * jl + PIECE * i;
                    float xv[VW], wv[VW];
                    rg_ld<VW>(wv, stash + D + VW * jl + PIECE * i);
                    if (ok) rg_ld<VW>(xv, p.x + o);
                    if (ok && p.dres_in) rg_ld<VW>(dr[i], p.dres_in + o);
#pragma unroll
                    for (int e = 0; e < VW; ++e) {
                        if (!ok) xv[e] = 0.f;
                        if (!(ok && p.dres_in)) dr[i][e] = 0.f;
                        const float dn = ok ? v[i][e] : 0.f;
                        xh[i][e] = (xv[e] - mu) * rs;
                        gg[i][e] = dn * wv[e];
                        s1 += gg[i][e];
                        s2 += gg[i][e] * xh[i][e];
                        adw[i][e] += dn * xh[i][e];
                        adb[i][e] += dn;
                    }
                }
#pragma unroll
                for (int o = LPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                const float c1 = s1 * invD, c2 = s2 * invD;
                if (ok) {
#pragma unroll
                    for (int i = 0; i < NSEG; ++i) {
                        const size_t o = m * D + VW * jl + PIECE * i;
                        float dx[VW], dc[VW];
#pragma unroll
                        for (int e = 0; e < VW; ++e) { dx[e] = dr[i][e] + rs * (gg[i][e] - c1 - xh[i][e] * c2); dc[e] = dx[e] * rsc; }
                        rg_st<VW>(p.dx_out + o, dx);
                        if constexpr (LS) {
                            static_assert(VW == 2, "branch is read as one bf16 pair");
                            const float2 br = unpack_bf16x2(*reinterpret_cast<const uint32_t*>(p.branch + o));
                            float cs[VW];
                            rg_ld<VW>(cs, stash + 2 * D + VW * jl + PIECE * i);
                            adg[i][0] += dc[0] * br.x; adg[i][1] += dc[1] * br.y;
                            dc[0] *= cs[0]; dc[1] *= cs[1];
                        }
                        if (p.cast_out) rg_st16<VW>(p.cast_out + o, dc);
                    }
                }
            }
        }
    }
    if constexpr (LNB) {
        // column sums of this tile in a fixed order: a lane owns its columns for every row its wave handled; then the 8 waves in order
        __syncthreads();                                                   // an early-exit chunk loop leaves no barrier behind the last image reads
#pragma unroll
        for (int i = 0; i < NSEG; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e)
#pragma unroll
                for (int o = LPR; o < 64; o <<= 1) {
                    adw[i][e] += __shfl_xor(adw[i][e], o, 64); adb[i][e] += __shfl_xor(adb[i][e], o, 64);
                    if constexpr (LS) adg[i][e] += __shfl_xor(adg[i][e], o, 64);
                }
        if (lane < LPR) {
#pragma unroll
            for (int i = 0; i < NSEG; ++i) {
                rg_st<VW>(colp + (wave * NPART + 0) * D + VW * jl + PIECE * i, adw[i]);
                rg_st<VW>(colp + (wave * NPART + 1) * D + VW * jl + PIECE * i, adb[i]);
                if constexpr (LS) rg_st<VW>(colp + (wave * NPART + 2) * D + VW * jl + PIECE * i, adg[i]);
            }
        }
        __syncthreads();
        for (int i = tid; i < NPART * D; i += RG_NTHR) {
            float s = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < 8; ++w_) s += colp[w_ * NPART * D + i];
            p.partial[(size_t)blockIdx.x * NPART * D + i] = s;
        }
    }
}

// dst[c] += sum over tiles of partial[tile][which][c], fixed order (one 1024-thread workgroup per 64 columns and array; 16 lane groups
// each add every 16th tile, the 16 group sums are added in order through LDS)
__global__ __launch_bounds__(1024) void rowgemm_colsum_kernel(const float* __restrict__ partial, int ntiles, int D, int nparts, float* d0, float* d1, float* d2) {
    __shared__ float red[16][64];
    const int which = blockIdx.y;
    float* dst = which == 0 ? d0 : which == 1 ? d1 : d2;
    if (!dst) return;
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < D) {
#pragma unroll 8
        for (int b = g; b < ntiles; b += 16) s += partial[((size_t)b * nparts + which) * D + c];
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

// LDS of one workgroup: the operand ring, or the epilogue's image + stash + column partials if that is larger
template <int D, int MT>
constexpr int rg_lds_bytes() {
    constexpr int WN = D / 48, WM = 8 / WN, MTW = (MT + WM - 1) / WM, MTP = MTW * WM;
    constexpr int ring = RG_S * (MTP * 16 * 128 + D * 128), epi = 64 * (D + 4) * 4 + 3 * D * 4 + 8 * 3 * D * 4;
    return ring > epi ? ring : epi;
}

template <int D, int MT, int EPI>
int rg_launch(const RowGemmParams& p, int tiles, hipStream_t stream) {
    constexpr int lds = rg_lds_bytes<D, MT>();
    auto kern = rowgemm_kernel<D, MT, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(rowgemm): %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(RG_NTHR), lds, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

template <int D, int MT>
int rg_dispatch_epi(const RowGemmParams& p, int epi, int tiles, hipStream_t stream) {
    switch (epi) {
        case RG_BF16: return rg_launch<D, MT, RG_BF16>(p, tiles, stream);
        case RG_RESID_LN: return rg_launch<D, MT, RG_RESID_LN>(p, tiles, stream);
        case RG_LNBWD: return rg_launch<D, MT, RG_LNBWD>(p, tiles, stream);
        case RG_LNBWD_LS: return rg_launch<D, MT, RG_LNBWD_LS>(p, tiles, stream);
        default: break;
    }
    ppf_set_error("ppf_rowgemm: unknown epilogue %d", epi);
    return PPF_ERR_ARG;
}

}  // namespace

extern "C" {

// 1 when ppf_rowgemm_* take this shape: D in {192, 384}, K % 64 == 0, tiles of <= 208 rows.
int ppf_rowgemm_supported(int D, int K, int rows_per_tile) {
    return (D == 384 || D == 192) && K >= 64 && (K % 64) == 0 && rows_per_tile >= 1 && rows_per_tile <= 208;
}

static int rg_run(RowGemmParams& p, int D, int epi, hipStream_t stream) {
    PPF_CHECK_ARG(p.M > 0 && ppf_rowgemm_supported(D, p.K, p.rows_per_tile), PPF_ERR_SHAPE,
                  "ppf_rowgemm: unsupported shape M=%d D=%d K=%d rows_per_tile=%d (D in {192, 384}, K %% 64 == 0, rows_per_tile <= 208)", p.M, D, p.K, p.rows_per_tile);
    PPF_CHECK_ARG((p.lda % 8) == 0 && (p.ldb % 8) == 0 && p.lda >= p.K && p.ldb >= p.K && (((uintptr_t)p.A | (uintptr_t)p.B) & 15) == 0, PPF_ERR_ALIGN,
                  "ppf_rowgemm: operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
    const int tiles = (p.M + p.rows_per_tile - 1) / p.rows_per_tile;
    const bool small = p.rows_per_tile <= 112;
    if (D == 384) return small ? rg_dispatch_epi<384, 7>(p, epi, tiles, stream) : rg_dispatch_epi<384, 13>(p, epi, tiles, stream);
    return small ? rg_dispatch_epi<192, 7>(p, epi, tiles, stream) : rg_dispatch_epi<192, 13>(p, epi, tiles, stream);
}

// out bf16 [M][D] = A [M][K] . B[D][K]^T (+ bias): plain full-row product (input gradient of the attention projection, deit:58)
int ppf_rowgemm_bf16(const void* A, const void* B, int M, int D, int K, int lda, int ldb, int rows_per_tile, const float* bias, void* out,
                     hipStream_t stream) {
    PPF_CHECK_ARG(A && B && out && (((uintptr_t)out) & 15) == 0, PPF_ERR_ARG, "ppf_rowgemm_bf16: null / misaligned pointer");
    RowGemmParams p = {};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.M = M; p.K = K; p.lda = lda; p.ldb = ldb; p.rows_per_tile = rows_per_tile; p.bias = bias;
    p.out16 = (bf16_t*)out; p.rows_per_group = 1;
    return rg_run(p, D, RG_BF16, stream);
}

// Residual branch + the LayerNorm that follows it (deit:76-81, cait:153-155): xout = res + rowscale[m / rows_per_group] * colscale[n] *
// (A B^T + bias)  (fp32, may alias res; colscale = CaiT's LayerScale gamma or NULL); aux_out (optional) = bf16(A B^T + bias), the unscaled
// branch the LayerScale gradient needs; ln_out = bf16(LN(xout) * ln_w + ln_b) with its row statistics (ln_out == NULL: no LayerNorm).
int ppf_rowgemm_resid_ln(const void* A, const void* B, int M, int D, int K, int lda, int ldb, int rows_per_tile, const float* bias,
                         const float* res, float* xout, const float* rowscale, int rows_per_group, const float* colscale, void* aux_out,
                         const float* ln_w, const float* ln_b, void* ln_out, float* ln_mean, float* ln_rstd, float eps, hipStream_t stream) {
    PPF_CHECK_ARG(A && B && res && xout && ((((uintptr_t)res) | ((uintptr_t)xout) | ((uintptr_t)ln_out) | ((uintptr_t)aux_out) | ((uintptr_t)colscale)) & 15) == 0,
                  PPF_ERR_ARG, "ppf_rowgemm_resid_ln: null / misaligned pointer");
    PPF_CHECK_ARG(ln_out == nullptr || (ln_w && ln_b && ln_mean && ln_rstd), PPF_ERR_ARG, "ppf_rowgemm_resid_ln: LayerNorm output needs weight, bias, mean and rstd");
    RowGemmParams p = {};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.M = M; p.K = K; p.lda = lda; p.ldb = ldb; p.rows_per_tile = rows_per_tile; p.bias = bias;
    p.res = res; p.xout = xout; p.rowscale = rowscale; p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
    p.colscale = colscale; p.aux_out = (bf16_t*)aux_out;
    p.ln_w = ln_w; p.ln_b = ln_b; p.ln_out = (bf16_t*)ln_out; p.ln_mean = ln_mean; p.ln_rstd = ln_rstd; p.eps = eps;
    return rg_run(p, D, RG_RESID_LN, stream);
}

// Input gradient of a Linear whose input is a LayerNorm output, fused with that LayerNorm's backward and the residual-gradient add
// (autograd of deit:76-81): dn = A B^T;  dx_out = dres_in + LN'(dn; x, mean, rstd, w) (fp32, may alias dres_in; dres_in == NULL: 0);
// cast_out = bf16(rowscale[m / rows_per_group] * dx_out) (the gradient entering the residual branch below, optional);
// partial[tiles][2][D] receives this call's per-tile column sums (d ln weight, d ln bias): add them up with ppf_rowgemm_colsum.
// colscale != NULL (CaiT LayerScale, cait:153-155; then branch = the bf16 unscaled branch output of the forward pass and cast_out are
// required): cast_out = bf16(rowscale * colscale[n] * dx_out) and partial is [tiles][3][D], the third array = d gamma's per-tile sums.
int ppf_rowgemm_lnbwd(const void* A, const void* B, int M, int D, int K, int lda, int ldb, int rows_per_tile, const float* x, const float* mean,
                      const float* rstd, const float* w, const float* dres_in, float* dx_out, void* cast_out, const float* rowscale,
                      int rows_per_group, const float* colscale, const void* branch, float* partial, size_t partial_bytes, hipStream_t stream) {
    PPF_CHECK_ARG(A && B && x && mean && rstd && w && dx_out && partial, PPF_ERR_ARG, "ppf_rowgemm_lnbwd: null pointer");
    PPF_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)dres_in) | ((uintptr_t)dx_out) | ((uintptr_t)cast_out)) & 15) == 0, PPF_ERR_ALIGN, "ppf_rowgemm_lnbwd: misaligned pointer");
    const int tiles = rows_per_tile > 0 ? (M + rows_per_tile - 1) / rows_per_tile : 0;
    const int nparts = colscale ? 3 : 2;
    PPF_CHECK_ARG(partial_bytes >= (size_t)tiles * nparts * D * sizeof(float), PPF_ERR_ARG, "ppf_rowgemm_lnbwd: partial needs tiles*%d*D*4 = %zu bytes",
                  nparts, (size_t)tiles * nparts * D * sizeof(float));
    PPF_CHECK_ARG(colscale == nullptr || (branch && cast_out && (((uintptr_t)branch) & 15) == 0), PPF_ERR_ARG,
                  "ppf_rowgemm_lnbwd: LayerScale needs the saved branch and cast_out");
    RowGemmParams p = {};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.M = M; p.K = K; p.lda = lda; p.ldb = ldb; p.rows_per_tile = rows_per_tile;
    p.x = x; p.mean = mean; p.rstd = rstd; p.w = w; p.dres_in = dres_in; p.dx_out = dx_out; p.cast_out = (bf16_t*)cast_out; p.rowscale = rowscale;
    p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1; p.partial = partial; p.colscale = colscale; p.branch = (const bf16_t*)branch;
    return rg_run(p, D, colscale ? RG_LNBWD_LS : RG_LNBWD, stream);
}

// dw[c] += sum_tiles partial[t][0][c], db[c] += sum_tiles partial[t][1][c] (and, nparts = 3, dg[c] += sum_tiles partial[t][2][c]) in a
// fixed order (may run on another stream); a NULL destination skips its array
int ppf_rowgemm_colsum(const float* partial, int tiles, int D, int nparts, float* dw, float* db, float* dg, hipStream_t stream) {
    PPF_CHECK_ARG(partial && tiles > 0 && D > 0 && (nparts == 2 || nparts == 3), PPF_ERR_ARG, "ppf_rowgemm_colsum: bad arguments");
    hipLaunchKernelGGL(rowgemm_colsum_kernel, dim3((D + 63) / 64, nparts), dim3(1024), 0, stream, partial, tiles, D, nparts, dw, db, dg);
    PPF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
