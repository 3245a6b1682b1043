// bf16 MFMA GEMM family for gfx950 (wave64, v_mfma_f32_32x32x16_bf16), fp32 accumulate.
//
//   C[m][n] (+)= epilogue( sum_kc  A(m,kc) * B(n,kc) )
//
// Operand storage modes (template TA / TB):
//   T = false : contraction-contiguous,  X(r,kc) = X[r*ld + kc]   (activations x[M,K], weights W[N,K])
//   T = true  : contraction-strided,     X(r,kc) = X[kc*ld + r]   (read through ds_read_b64_tr_b16)
// which covers the three GEMMs of a Linear layer without any transposed copies:
//   forward  y  = x W^T       : TA=0 (x [M][K]),      TB=0 (W [N][K])
//   dgrad    dx = dy W        : TA=0 (dy [M][N]),     TB=1 (W [N=kc][K=n'])
//   wgrad    dW = dy^T x      : TA=1 (dy [M=kc][N]),  TB=1 (x [M=kc][K]), split over kc: fp32 partial tiles in a workspace +
//                               an ordered reduce kernel (deterministic); fp32 atomics only without a workspace
//
// Tiling: 128x128x64 per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 2x2 MFMA 32x32 tiles),
// next K tile prefetched into registers while the current one is multiplied out of a single LDS image (32 KiB: three
// workgroups per CU -- measured faster than an LDS double buffer at two), XOR-swizzled LDS images:
//   mode 0 tile [128 r][64 kc]  (128 B rows): 16-B chunk index ^= (r>>1)&7  -> conflict-free ds_read_b128
//   mode 1 tile [64 kc][128 r]  (256 B rows): 64-B unit index  ^= kc&3      -> conflict-free tr-reads
// MFMA operands are swapped (first = B/n, second = A/m) so each lane ends up with 4 consecutive n of one
// output row m: 8/16-byte row-major stores and vector bias/residual accesses.
// blockIdx is remapped XCD-aware (n-tiles of one m-panel share an XCD's L2).
#include "ppf_common.h"
#include "gemm_common.h"
#include <cstdlib>
#include <utility>
#include <vector>

namespace {
using namespace ppfg;

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile (either mode)
constexpr int STAGE_LD = 68;                     // fp32 pitch of the per-wave epilogue strip (64 + 4: conflict-free float4 rows)
constexpr int STAGE_BYTES = 4 * 32 * STAGE_LD * 4;   // 4 waves x 32 rows


__device__ __forceinline__ int lds_off_mode0(int r, int c16) { return r * 128 + ((c16 ^ ((r >> 1) & 7)) << 4); }
// transposed tile [64 kc][ROWS r]: row pitch ROWS*2 bytes, 64-byte units XOR-swizzled by kc&3 inside each 256-byte group
template <int ROWS>
__device__ __forceinline__ int lds_off_mode1(int kc, int col) {
    return kc * (ROWS * 2) + ((((col >> 5) ^ (kc & 3))) << 6) + ((col & 31) << 1);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));       // a native 128-bit register tuple

// Operand tile of ROWS rows (m or n) x 64 contraction values, staged global -> registers -> LDS by 256 threads.
template <bool T, int ROWS>
struct TileIO {
    static constexpr int NLD = ROWS / 32;            // 16-byte loads per thread
    // Branch-free: every load is issued unconditionally from a clamped (always valid) address and the "outside the matrix"
    // predicate travels as a bit mask that sstore applies.  Loads under per-lane branches made the compiler wait vmcnt(0) at
    // every K step (it cannot count loads across exec-masked branches), which serialised the register pipeline.
    static __device__ __forceinline__ unsigned gload(u32x4 (&reg)[NLD], const bf16_t* __restrict__ X, int ld, int R, int row0,
                                                     int k0, int kend, int tid, int kpad) {
        unsigned mask = 0;
        if constexpr (!T) {
            const int c16 = tid & 7, rb = tid >> 3;
            const bool kok = kpad ? (k0 + c16 * 8) < kend : (k0 + c16 * 8 + 8) <= kend;
            const int kk = kok ? k0 + c16 * 8 : 0;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int r = row0 + rb + 32 * i;
                const bf16_t* src = X + (size_t)min(r, R - 1) * ld + kk;
                reg[i] = *reinterpret_cast<const u32x4*>(src);
                mask |= (kok && r < R) ? (1u << i) : 0u;
            }
        } else {
            constexpr int CPR = ROWS / 8;            // 16-byte chunks per kc row
            const int c16 = tid % CPR, kb = tid / CPR;
            const int col = row0 + c16 * 8;
            const bool cok = kpad ? col < R : (col + 8) <= R;
            const int cc = cok ? col : 0;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int kc = k0 + kb + (256 / CPR) * i;
                const bf16_t* src = X + (size_t)min(kc, kend - 1) * ld + cc;
                reg[i] = *reinterpret_cast<const u32x4*>(src);
                mask |= (cok && kc < kend) ? (1u << i) : 0u;
            }
        }
        return mask;
    }
    static __device__ __forceinline__ void sstore(const u32x4 (&reg)[NLD], unsigned mask, unsigned char* tile, int tid) {
        const u32x4 zero = {0u, 0u, 0u, 0u};
        if constexpr (!T) {
            const int c16 = tid & 7, rb = tid >> 3;
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                *reinterpret_cast<u32x4*>(tile + lds_off_mode0(rb + 32 * i, c16)) = ((mask >> i) & 1u) ? reg[i] : zero;
        } else {
            constexpr int CPR = ROWS / 8;
            const int c16 = tid % CPR, kb = tid / CPR;
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                *reinterpret_cast<u32x4*>(tile + lds_off_mode1<ROWS>(kb + (256 / CPR) * i, c16 * 8)) = ((mask >> i) & 1u) ? reg[i] : zero;
        }
    }
    // Fragment of the 32-row sub-tile starting at rbase for k-substep ks (16 contraction values):
    // lane l holds row rbase+(l&31), kc = ks*16 + (l>>5)*8 + 0..7.
    static __device__ __forceinline__ bf16x8 frag(const unsigned char* tile, int rbase, int ks, int lane) {
        if constexpr (!T) {
            return *reinterpret_cast<const bf16x8*>(tile + lds_off_mode0(rbase + (lane & 31), ks * 2 + (lane >> 5)));
        } else {
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
            const int s = lane & 15, g16 = (lane >> 4) & 1, h = lane >> 5;
            const int col = rbase + 16 * g16 + 4 * (s & 3);
            const int kc = ks * 16 + 8 * h + (s >> 2);
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + lds_off_mode1<ROWS>(kc, col)));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + lds_off_mode1<ROWS>(kc + 4, col)));
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
};


// Epilogue of one 32-row x 64-column block that a wave has just transposed into its LDS strip (row pitch STAGE_LD floats):
// bias / residual / gelu' operands are loaded for the whole block first, then the stores are issued back to back (loads and stores
// share vmcnt on gfx9).  bf16-output epilogues use 8 columns per lane (16-byte stores, 8 rows per pass), the others 4.
template <int EPI>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, const float* stage, int mbase, int nbase, int lane) {
    if (EpiWide<EPI>::value && ((p.ldc | p.ldaux) & 7) == 0) {          // 16-byte stores need 8-element row pitches
        const int col = (lane & 7) * 8;
        const int n = nbase + col;
        if (n + 8 <= p.N) {
            const EpiCols8 cc = epi_load_cols8<EPI>(p, n, true);
            EpiRow8 rr[4];
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int m = mbase + pass * 8 + (lane >> 3);
                rr[pass] = epi_load_row8<EPI>(p, m, n, m < p.M);
            }
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int r = pass * 8 + (lane >> 3);
                const int m = mbase + r;
                const float4 a = *reinterpret_cast<const float4*>(stage + r * STAGE_LD + col);
                const float4 b = *reinterpret_cast<const float4*>(stage + r * STAGE_LD + col + 4);
                if (m < p.M) epi_store8<EPI>(p, m, n, a, b, cc, rr[pass]);
            }
        } else if (n < p.N) {                                    // a 4-column remainder at the right edge (N % 8 == 4)
            const EpiCols cc = epi_load_cols<EPI>(p, n, true);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int r = pass * 8 + (lane >> 3);
                const int m = mbase + r;
                if (m < p.M) {
                    const EpiRow rr = epi_load_row<EPI>(p, m, n, true);
                    const float4 v = *reinterpret_cast<const float4*>(stage + r * STAGE_LD + col);
                    epi_store<EPI>(p, m, n, v.x, v.y, v.z, v.w, cc, rr);
                }
            }
        }
    } else {
        const int col = (lane & 15) * 4;
        const int n = nbase + col;
        const EpiCols cc = epi_load_cols<EPI>(p, n, n < p.N);
        EpiRow rr[8];
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int m = mbase + pass * 4 + (lane >> 4);
            rr[pass] = epi_load_row<EPI>(p, m, n, m < p.M && n < p.N);
        }
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int r = pass * 4 + (lane >> 4);
            const int m = mbase + r;
            const float4 v = *reinterpret_cast<const float4*>(stage + r * STAGE_LD + col);
            if (m < p.M && n < p.N) epi_store<EPI>(p, m, n, v.x, v.y, v.z, v.w, cc, rr[pass]);
        }
    }
}

// MT = 32-row MFMA tiles per wave along m: MT = 2 -> 128x128 workgroup tile (3 workgroups/CU), MT = 4 -> 256x128 (wave tile
// 128x64, 2 workgroups/CU): fewer LDS bytes and barriers per flop for the tall activation GEMMs (M = B*N tokens).
// The next K tile is prefetched into registers while the current one is multiplied (three workgroups per CU).  (A three-deep register
// pipeline with hand-counted waits was measured equal stand-alone and 1.8 % slower in the step -- the kernel is not latency-bound,
// profiles/r2_wgrad_pmc.txt -- and removed in round 4.)
template <bool TA, bool TB, int EPI, bool COLSUM, int MT>
__global__ __launch_bounds__(NTHREADS, MT == 2 ? 3 : 2) void gemm_kernel(const GemmParams p_) {
    constexpr int PD = 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TBM = 64 * MT;                    // workgroup tile rows
    constexpr int A_BYTES = TBM * BK * 2;
    using IOA = TileIO<TA, TBM>;
    using IOB = TileIO<TB, BN>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * (32 * MT), wn = (wave >> 1) * 64;

    // 1-D grid over (K slice, m tile, n tile), n fastest, remapped so that each XCD gets a contiguous range: the n-tiles of an
    // m-panel -- and, for split-K, all tiles of a K slice -- share one XCD's L2.
    const int tiles_n = (p_.N + BN - 1) / BN;
    const int tiles = tiles_n * ((p_.M + TBM - 1) / TBM);
    const int vid_all = xcd_remap(blockIdx.x, gridDim.x);
    const int zslice = vid_all / tiles, vid = vid_all - zslice * tiles;
    const int m0 = (vid / tiles_n) * TBM, n0 = (vid % tiles_n) * BN;
    GemmParams p = p_;
    p.zslice = zslice;
    if (gridDim.y > 1) {
        const int bo = blockIdx.y / p.batch_inner, bi = blockIdx.y % p.batch_inner;
        p.A += bo * p.sa_o + bi * p.sa_i;
        p.B += bo * p.sb_o + bi * p.sb_i;
        const long long co = bo * p.sc_o + bi * p.sc_i;
        p.C = (EPI == EPI_BF16) ? (void*)(reinterpret_cast<bf16_t*>(p.C) + co) : (void*)(reinterpret_cast<float*>(p.C) + co);
    }

    // contraction range of this K slice: multiples of BK except the tail
    const int nsplit = p.nsplit;
    const int kchunk = (((p.K + nsplit - 1) / nsplit) + BK - 1) / BK * BK;
    const int kbeg = zslice * kchunk;
    const int kend = min(p.K, kbeg + kchunk);
    if (kbeg >= kend) return;
    const int nk = (kend - kbeg + BK - 1) / BK;

    u32x4 ra[PD][IOA::NLD], rb[PD][IOB::NLD];
    unsigned ma[PD], mb[PD];
    f32x16 acc[2][MT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x16 accs[MT];
    if constexpr (COLSUM) {
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[j][r] = 0.f;
    }
    const bool do_colsum = COLSUM && p.colsum != nullptr && n0 == 0 && wn == 0;
    unsigned char* tA = smem;
    unsigned char* tB = smem + A_BYTES;

#pragma unroll
    for (int st = 0; st < PD; ++st) {
        const int k0 = min(kbeg + st * BK, kbeg + (nk - 1) * BK);     // st >= nk: a harmless re-read of the last tile, never stored
        ma[st] = IOA::gload(ra[st], p.A, p.lda, p.M, m0, k0, kend, tid, p.kpad);
        mb[st] = IOB::gload(rb[st], p.B, p.ldb, p.N, n0, k0, kend, tid, p.kpad);
    }
    IOA::sstore(ra[0], ma[0], tA, tid);
    IOB::sstore(rb[0], mb[0], tB, tid);
    __syncthreads();

    for (int kt0 = 0; kt0 < nk; kt0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int kt = kt0 + u;
            if (kt >= nk) break;
            {                                                // stage u is in LDS: refill it PD tiles ahead; the loads fly while
                // this and the next PD-1 tiles are multiplied (past the end: re-read of the last tile, never stored -- keeps the
                // loop body branch-free so that the compiler's vmcnt counting leaves PD-1 tiles in flight)
                const int k0 = kbeg + min(kt + PD, nk - 1) * BK;
                ma[u] = IOA::gload(ra[u], p.A, p.lda, p.M, m0, k0, kend, tid, p.kpad);
                mb[u] = IOB::gload(rb[u], p.B, p.ldb, p.N, n0, k0, kend, tid, p.kpad);
            }
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 fa[MT], fb[2];
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = IOA::frag(tA, wm + 32 * i, ks, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i) fb[i] = IOB::frag(tB, wn + 32 * i, ks, lane);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
                if constexpr (COLSUM) {
                    if (do_colsum) {
                        bf16x8 ones;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi)
                            accs[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, fa[mi], accs[mi], 0, 0, 0);
                    }
                }
            }
            __syncthreads();                                 // single LDS operand buffer: everyone finished reading it
            if (kt + 1 < nk) {
                IOA::sstore(ra[(u + 1) % PD], ma[(u + 1) % PD], tA, tid);
                IOB::sstore(rb[(u + 1) % PD], mb[(u + 1) % PD], tB, tid);
            }
            __syncthreads();
        }
    }

    // epilogue.  After the MFMAs a lane holds, for each (ni, mi): row m = wm+32*mi+(lane&31) and, for g = 0..3, the four
    // consecutive columns n = wn+32*ni+8*g+4*(lane>>5).. (acc regs 4g..4g+3): row-strided 8/16-byte pieces.  Each wave
    // therefore transposes its accumulators through a private LDS strip (32 rows x 64 fp32 at a time) so that 16
    // consecutive lanes cover one 64-column row segment: bias / residual reads and the output stores become full
    // 128/256-byte runs.  (The main loop's last barrier has already retired every read of the operand tiles.)
    const int h = lane >> 5;
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * STAGE_LD);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(stage + (lane & 31) * STAGE_LD + 32 * ni + 8 * g + 4 * h) =
                    make_float4(acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]);
        __builtin_amdgcn_wave_barrier();                           // keep the compiler from moving reads above the writes
        epilogue_rows<EPI>(p, stage, m0 + wm + 32 * mi, n0 + wn, lane);
        __builtin_amdgcn_wave_barrier();                           // the next 32-row block overwrites the strip
        if constexpr (COLSUM) {
            const int m = m0 + wm + 32 * mi + (lane & 31);
            if (do_colsum && h == 0 && m < p.M) {
                if constexpr (EPI == EPI_PARTIAL) {
                    float* dst = p.ws + p.zslice * ((size_t)p.M * p.N + (size_t)p.cs_parts * p.M) + (size_t)p.M * p.N + m;
                    *dst = accs[mi][0];
                } else unsafeAtomicAdd(p.colsum + m, accs[mi][0]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM with 256 x 128 (or 128 x 256) tiles and EIGHT waves (round 4).  The 128 x 128 kernel above pulls 0.9 GB per launch
// out of L2 -- 45 GB per step, the largest L2 consumer -- and that traffic, not its arithmetic, is what it costs the main stream
// (profiles/r4_wgrad_l2_knockout.txt: half the loads = +3.3 % of the step).  Two 128 x 128 sub-tiles that share one operand tile in LDS
// read 25 % less; every wave keeps the 64 x 64 accumulator tile (64 VGPRs) of the four-wave kernel, so the footprint per wave is
// unchanged and the launch has half as many, twice as large workgroups (48 KiB of operand tiles each).  Both operands transposed
// (contraction = rows in memory), register-staged with one tile of prefetch, ordered split-K partial tiles (EPI_PARTIAL) only.
template <int ROWS, int NTHR>
struct TileT {
    static constexpr int CPR = ROWS / 8;             // 16-byte chunks per contraction row
    static constexpr int KPP = NTHR / CPR;           // contraction rows per pass
    static constexpr int NLD = BK / KPP;             // passes = 16-byte loads per thread
    static __device__ __forceinline__ unsigned gload(u32x4 (&reg)[NLD], const bf16_t* __restrict__ X, int ld, int R, int row0, int k0, int kend, int tid) {
        const int c16 = tid % CPR, kb = tid / CPR;
        const int col = row0 + c16 * 8;
        const bool cok = (col + 8) <= R;
        const int cc = cok ? col : 0;
        unsigned mask = 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int kc = k0 + kb + KPP * i;
            reg[i] = *reinterpret_cast<const u32x4*>(X + (size_t)min(kc, kend - 1) * ld + cc);
            mask |= (cok && kc < kend) ? (1u << i) : 0u;
        }
        return mask;
    }
    static __device__ __forceinline__ void sstore(const u32x4 (&reg)[NLD], unsigned mask, unsigned char* tile, int tid) {
        const u32x4 zero = {0u, 0u, 0u, 0u};
        const int c16 = tid % CPR, kb = tid / CPR;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            *reinterpret_cast<u32x4*>(tile + lds_off_mode1<ROWS>(kb + KPP * i, c16 * 8)) = ((mask >> i) & 1u) ? reg[i] : zero;
    }
    static __device__ __forceinline__ bf16x8 frag(const unsigned char* tile, int rbase, int ks, int lane) { return TileIO<true, ROWS>::frag(tile, rbase, ks, lane); }
};

template <int WMW, int WNW, bool COLSUM>
__global__ __launch_bounds__(64 * WMW * WNW, 1) void wgrad8_kernel(const GemmParams p_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NTHR = 64 * WMW * WNW, TBM = 64 * WMW, TBN = 64 * WNW;
    using IOA = TileT<TBM, NTHR>;
    using IOB = TileT<TBN, NTHR>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave % WMW) * 64, wn = (wave / WMW) * 64;
    const int tiles_n = (p_.N + TBN - 1) / TBN;
    const int tiles = tiles_n * ((p_.M + TBM - 1) / TBM);
    const int vid_all = xcd_remap(blockIdx.x, gridDim.x);
    const int zslice = vid_all / tiles, vid = vid_all - zslice * tiles;
    const int m0 = (vid / tiles_n) * TBM, n0 = (vid % tiles_n) * TBN;
    GemmParams p = p_;
    p.zslice = zslice;
    const int nsplit = p.nsplit;
    const int kchunk = (((p.K + nsplit - 1) / nsplit) + BK - 1) / BK * BK;
    const int kbeg = zslice * kchunk;
    const int kend = min(p.K, kbeg + kchunk);
    if (kbeg >= kend) return;
    const int nk = (kend - kbeg + BK - 1) / BK;

    u32x4 ra[IOA::NLD], rb[IOB::NLD];
    unsigned ma, mb;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x16 accs[2];
    if constexpr (COLSUM) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[j][r] = 0.f;
    }
    const bool do_colsum = COLSUM && p.colsum != nullptr && n0 == 0 && wn == 0;
    unsigned char* tA = smem;
    unsigned char* tB = smem + TBM * BK * 2;
    ma = IOA::gload(ra, p.A, p.lda, p.M, m0, kbeg, kend, tid);
    mb = IOB::gload(rb, p.B, p.ldb, p.N, n0, kbeg, kend, tid);
    IOA::sstore(ra, ma, tA, tid);
    IOB::sstore(rb, mb, tB, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        {                                                    // next tile into registers while this one is multiplied (past the end: a
            const int k0 = kbeg + min(kt + 1, nk - 1) * BK;  // harmless re-read of the last tile, never stored)
            ma = IOA::gload(ra, p.A, p.lda, p.M, m0, k0, kend, tid);
            mb = IOB::gload(rb, p.B, p.ldb, p.N, n0, k0, kend, tid);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = IOA::frag(tA, wm + 32 * i, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i) fb[i] = IOB::frag(tB, wn + 32 * i, ks, lane);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
            if constexpr (COLSUM) {
                if (do_colsum) {
                    bf16x8 ones;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) accs[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, fa[mi], accs[mi], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            IOA::sstore(ra, ma, tA, tid);
            IOB::sstore(rb, mb, tB, tid);
        }
        __syncthreads();
    }
    // epilogue: as gemm_kernel (a private 32 x 64 fp32 strip per wave, row-contiguous partial-tile stores)
    const int h = lane >> 5;
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * STAGE_LD);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(stage + (lane & 31) * STAGE_LD + 32 * ni + 8 * g + 4 * h) =
                    make_float4(acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]);
        __builtin_amdgcn_wave_barrier();
        epilogue_rows<EPI_PARTIAL>(p, stage, m0 + wm + 32 * mi, n0 + wn, lane);
        __builtin_amdgcn_wave_barrier();
        if constexpr (COLSUM) {
            const int m = m0 + wm + 32 * mi + (lane & 31);
            if (do_colsum && h == 0 && m < p.M) {
                float* dst = p.ws + p.zslice * ((size_t)p.M * p.N + (size_t)p.cs_parts * p.M) + (size_t)p.M * p.N + m;
                *dst = accs[mi][0];
            }
        }
    }
}

template <int WMW, int WNW>
int launch_wgrad8(const GemmParams& p, int splitk, hipStream_t stream) {
    auto kern = wgrad8_kernel<WMW, WNW, true>;
    constexpr int NTHR = 64 * WMW * WNW, TBM = 64 * WMW, TBN = 64 * WNW;
    constexpr int opnd = (TBM + TBN) * BK * 2, strips = (NTHR / 64) * 32 * STAGE_LD * 4;
    constexpr int lds = opnd > strips ? opnd : strips;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(wgrad8): %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tiles = ((p.M + TBM - 1) / TBM) * ((p.N + TBN - 1) / TBN);
    hipLaunchKernelGGL(kern, dim3(tiles * splitk), dim3(NTHR), lds, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// 128x128x64 tiles, both operands contraction-contiguous, FOUR workgroups per CU.  The K=384 GEMMs of this model are bound by
// the length of each workgroup's dependent chain (load round trip -> LDS -> MFMA -> epilogue), i.e. by how many independent
// workgroups a CU interleaves.  Loading with global_load_lds_dwordx4 removes the 32 staging registers (<= 128 VGPRs: four
// waves per SIMD) and the ds_write pass; with a single 32 KiB LDS image per workgroup a K step is
//   barrier -> 8 direct-to-LDS loads per thread -> vmcnt(0) + barrier -> 16 fragment reads + 16 MFMA per wave,
// nothing of ONE workgroup overlaps, four of them do.  Swizzle on the source address (LDS image lane-linear), rows past the
// edge re-read the last row (masked in the epilogue).  Requires K % 64 == 0.
typedef __attribute__((address_space(3))) void g4_lds_t;
typedef const __attribute__((address_space(1))) void g4_gbl_t;

template <int EPI, int MT, int OCC>
__global__ __launch_bounds__(NTHREADS, OCC) void gemm128g_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using IO = TileIO<false, 128>;
    constexpr int TBM = 64 * MT, NLA = 2 * MT;                 // A rows per workgroup, 32-row load instructions per K tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave & 1) * (32 * MT), wn = (wave >> 1) * 64;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (vid / tiles_n) * TBM, n0 = (vid % tiles_n) * BN;
    unsigned char* tA = smem;
    unsigned char* tB = smem + TBM * BK * 2;
    unsigned offA[NLA], offB[4];
    {
        const int rb = tid >> 3, ch = (tid & 7) ^ ((rb >> 1) & 7);       // rows rb + 32 i share the swizzle
#pragma unroll
        for (int i = 0; i < NLA; ++i) offA[i] = ((unsigned)min(m0 + rb + 32 * i, p.M - 1) * (unsigned)p.lda + ch * 8) * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) offB[i] = ((unsigned)min(n0 + rb + 32 * i, p.N - 1) * (unsigned)p.ldb + ch * 8) * 2u;
    }
    const unsigned char* gA = reinterpret_cast<const unsigned char*>(p.A);
    const unsigned char* gB = reinterpret_cast<const unsigned char*>(p.B);
    f32x16 acc[2][MT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt > 0) __syncthreads();                          // everyone finished reading the previous K tile
        const unsigned char* ka = gA + (size_t)kt * (BK * 2);
        const unsigned char* kb = gB + (size_t)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < NLA; ++i)
            __builtin_amdgcn_global_load_lds((g4_gbl_t*)(ka + offA[i]), (g4_lds_t*)(tA + i * 4096 + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((g4_gbl_t*)(kb + offB[i]), (g4_lds_t*)(tB + i * 4096 + wave * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[MT], fb[2];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = IO::frag(tA, wm + 32 * i, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i) fb[i] = IO::frag(tB, wn + 32 * i, ks, lane);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
        }
    }
    __syncthreads();

    const int h = lane >> 5;
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * STAGE_LD);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(stage + (lane & 31) * STAGE_LD + 32 * ni + 8 * g + 4 * h) =
                    make_float4(acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]);
        __builtin_amdgcn_wave_barrier();
        epilogue_rows<EPI>(p, stage, m0 + wm + 32 * mi, n0 + wn, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

template <int EPI, int MT, int OCC>
int launch_g4_mt(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm128g_kernel<EPI, MT, OCC>;
    constexpr int TBM = 64 * MT;
    constexpr int opnd = TBM * BK * 2 + TILE_BYTES;
    constexpr int lds = opnd > STAGE_BYTES ? opnd : STAGE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(gemm128g): %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tiles = ((p.M + TBM - 1) / TBM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHREADS), lds, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// 224 x 128 tiles for the tall products with a narrow output (round 4): the input-gradient GEMMs dx = dy W of fc1 / qkv / proj have
// N = 384 output columns, so 128-row tiles give 394 x 3 = 1 182 workgroups for the 768 slots of the chip (three per CU): two rounds,
// the second 54 % full -- the 23 % tile-quantisation loss of profiles/r2_gemm_knockout.txt.  224-row tiles (seven 32-row MFMA tiles;
// they need not end on sample boundaries) make it 226 x 3 = 678 workgroups: ONE round, and 81 instead of 64 flops per operand byte.
// Both operands contraction-contiguous (A = dy [M][K], B = the TRANSPOSED weight shadow [N][K] of FlatStore.register_transposed),
// fetched by global_load_lds_dwordx4 into a single 44 KiB LDS image (three workgroups per CU overlap each other, as gemm128g);
// the four waves sit side by side along n: a wave owns all 224 rows x 32 columns (7 accumulator tiles, 112 VGPRs), the A fragments
// are read by every wave, the B fragment once.  bf16 output only (no bias / fused epilogue: that is what these products need).
constexpr int G224_ROWS = 224, G224_MT = 7, G224_LD = 36;       // epilogue strip pitch (floats): 32 + 4
__global__ __launch_bounds__(NTHREADS, 3) void gemm224g_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using IO = TileIO<false, 128>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave * 32;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (vid / tiles_n) * G224_ROWS, n0 = (vid % tiles_n) * BN;
    unsigned char* tA = smem;
    unsigned char* tB = smem + G224_ROWS * BK * 2;
    unsigned offA[G224_MT], offB[4];
    {
        const int rb = tid >> 3, ch = (tid & 7) ^ ((rb >> 1) & 7);       // rows rb + 32 i share the swizzle
#pragma unroll
        for (int i = 0; i < G224_MT; ++i) offA[i] = ((unsigned)min(m0 + rb + 32 * i, p.M - 1) * (unsigned)p.lda + ch * 8) * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) offB[i] = ((unsigned)min(n0 + rb + 32 * i, p.N - 1) * (unsigned)p.ldb + ch * 8) * 2u;
    }
    const unsigned char* gA = reinterpret_cast<const unsigned char*>(p.A);
    const unsigned char* gB = reinterpret_cast<const unsigned char*>(p.B);
    f32x16 acc[G224_MT];
#pragma unroll
    for (int j = 0; j < G224_MT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int nk = p.K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt > 0) __syncthreads();                          // everyone finished reading the previous K tile
        const unsigned char* ka = gA + (size_t)kt * (BK * 2);
        const unsigned char* kb = gB + (size_t)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < G224_MT; ++i)
            __builtin_amdgcn_global_load_lds((g4_gbl_t*)(ka + offA[i]), (g4_lds_t*)(tA + i * 4096 + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((g4_gbl_t*)(kb + offB[i]), (g4_lds_t*)(tB + i * 4096 + wave * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const bf16x8 fb = IO::frag(tB, wn, ks, lane);
#pragma unroll
            for (int mi = 0; mi < G224_MT; ++mi)
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, IO::frag(tA, 32 * mi, ks, lane), acc[mi], 0, 0, 0);
        }
    }
    __syncthreads();
    // a lane holds row 32 mi + (lane & 31) and, for g = 0..3, columns wn + 8 g + 4 (lane >> 5) + 0..3: through a per-wave strip so that
    // four lanes store one row's 32 columns as 16-byte pieces
    const int h = lane >> 5;
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * G224_LD);
    bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
    for (int mi = 0; mi < G224_MT; ++mi) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(stage + (lane & 31) * G224_LD + 8 * g + 4 * h) =
                make_float4(acc[mi][4 * g], acc[mi][4 * g + 1], acc[mi][4 * g + 2], acc[mi][4 * g + 3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = pass * 16 + (lane >> 2), col = (lane & 3) * 8;
            const int m = m0 + 32 * mi + r, n = n0 + wn + col;
            const float4 a = *reinterpret_cast<const float4*>(stage + r * G224_LD + col);
            const float4 b = *reinterpret_cast<const float4*>(stage + r * G224_LD + col + 4);
            if (m < p.M && n < p.N)
                *reinterpret_cast<uint4*>(C + (size_t)m * p.ldc + n) = make_uint4(pack_bf16x2(a.x * p.alpha, a.y * p.alpha), pack_bf16x2(a.z * p.alpha, a.w * p.alpha),
                                                                                 pack_bf16x2(b.x * p.alpha, b.y * p.alpha), pack_bf16x2(b.z * p.alpha, b.w * p.alpha));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Takes a plain bf16-output NT product when 224-row tiles (three workgroups per CU) need less CU time than 128-row tiles (three per CU in
// the generic kernel, four in the direct-to-LDS one); a partly filled last round is priced as a whole round (a round lasts a tile's
// latency however many CUs it fills: 1.54 rounds measured as 2), half a round below 25 % fill.  K >= 768 (384 until round 5): with three K tiles per
// workgroup (the D = 192 models) the larger tile's prologue / epilogue outweigh the saved round -- fc1 + GELU and the x gelu' input
// gradient of deit_tiny / cait_xxs24 (N = 768, K = 192; 1 182 vs 678 tiles) were measured 1.1 % SLOWER per step with this kernel.
bool g4_eligible(const GemmParams& p);
int g_force_g224 = 0;              // test hook (ppf_gemm_test_force_g224): 1 = every legal shape takes the 224-row kernel
bool g224_eligible(const GemmParams& p, int epi) {
    if (epi != EPI_BF16 || p.bias != nullptr || p.kpad || p.K % BK != 0 || p.N % 8 != 0 || (p.ldc & 7) != 0) return false;
    if ((long long)p.M * p.lda >= (1ll << 30) || (long long)p.N * p.ldb >= (1ll << 30)) return false;
    if (g_force_g224) return true;
    if (p.K < 768) return false;                 // round 5: the K = 384 input gradient of proj is +0.8 % of the deit_small step on 128 x 128 tiles (six K tiles do not pay for the larger tile's prologue / epilogue)
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const long long tn = (p.N + BN - 1) / BN;
    const long long t128 = (long long)((p.M + BM - 1) / BM) * tn, t224 = (long long)((p.M + G224_ROWS - 1) / G224_ROWS) * tn;
    const int per128 = g4_eligible(p) ? 4 : 3;
    auto cost = [](long long tiles, long long slots, int rows, int per) {          // CU time in row units
        const long long full = tiles / slots, rest = tiles - full * slots;
        double c = (double)full * rows * per;
        if (rest) c += rows * per * (4 * rest >= slots ? 1.0 : 0.5);
        return c;
    };
    return t128 >= (long long)per128 * cus && cost(t224, 3ll * cus, G224_ROWS, 3) < 0.97 * cost(t128, (long long)per128 * cus, BM, per128);
}
int launch_g224(const GemmParams& p, hipStream_t stream) {
    constexpr int lds = G224_ROWS * BK * 2 + TILE_BYTES;           // 28 + 16 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm224g_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(gemm224g): %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tiles = ((p.M + G224_ROWS - 1) / G224_ROWS) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(gemm224g_kernel, dim3(tiles), dim3(NTHREADS), lds, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

// MT = 4 (256x128 workgroup tiles, 128x64 per wave: 25 % fewer LDS fragment bytes per MFMA) was measured SLOWER at three and at two
// workgroups per CU (qkv 60.6 -> 83.6 / 69.0 us, fc1+GELU 132 -> 155 / 146 us, train step -5.4 % / -1.9 %; profiles/r2_gemm_knockout.txt):
// what these K = 384 GEMMs want is more independent workgroups per CU, not fewer LDS bytes.  Only MT = 2 is instantiated.
template <int EPI>
int launch_g4(const GemmParams& p, hipStream_t stream) { return launch_g4_mt<EPI, 2, 4>(p, stream); }

bool g4_eligible(const GemmParams& p) {
    // measured (profiles/r1_gemm_ab.txt): -12 % at K = 384 with N >= 1152 (qkv 74 -> 65 us, fc1+GELU 151 -> 133), neutral at N = 384,
    // +5 % at K = 1536 where the register-prefetched kernel overlaps better inside a workgroup -> short contractions only
    return p.K % BK == 0 && p.K <= 512 && p.N >= 512 && !p.kpad && (long long)p.M * p.lda < (1ll << 30) &&
           (long long)p.N * p.ldb < (1ll << 30) && (long long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) >= 512;
}

template <bool TA, bool TB, int EPI, bool COLSUM, int MT>
int launch_impl(const GemmParams& p, int splitk, hipStream_t stream, int nbatch) {
    constexpr int TBM = 64 * MT;
    const int tiles = ((p.M + TBM - 1) / TBM) * ((p.N + BN - 1) / BN);
    auto kern = gemm_kernel<TA, TB, EPI, COLSUM, MT>;
    constexpr int opnd = TBM * BK * 2 + TILE_BYTES;
    constexpr int lds = opnd > STAGE_BYTES ? opnd : STAGE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(gemm): %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    GemmParams q = p;
    q.nsplit = splitk;
    hipLaunchKernelGGL(kern, dim3(tiles * splitk, nbatch, 1), dim3(NTHREADS), lds, stream, q);
    PPF_LAUNCH_CHECK();
    return 0;
}

// Tile choice: 256x128 when the m extent is tall enough to fill the chip with 256-row tiles (activation GEMMs), else 128x128.
// (The 256x128 form of this kernel was 5-30 % slower at every shape of these models -- profiles/r1_gemm_tile_ab.txt, r4_wgrad_tiles.txt -- and
// is no longer instantiated.)
template <bool TA, bool TB, int EPI, bool COLSUM>
int launch(const GemmParams& p, int splitk, hipStream_t stream, int nbatch = 1) {
    return launch_impl<TA, TB, EPI, COLSUM, 2>(p, splitk, stream, nbatch);
}

// out[i] += sum_z ws[z][i] for the M*N tile elements (row stride ldc) and, when colsum != null, the cs_parts * M partial column sums.
// A workgroup owns 64 float4 positions; its four waves take the slabs z = w, w+4, w+8, ... (four independent 16-byte loads in flight
// per lane) and wave 0 adds the four partial sums in a fixed order, so the result does not depend on scheduling.  The earlier
// one-thread-per-position loop read its 12-98 slabs one dependent load after the other (22 us for a 192x192 gradient of 98 slabs).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, float* __restrict__ colsum,
                                                            int M, int N, int ldc, int nsplit, int cs_parts) {
    // round 4: ONE lane per float4 position with up to sixteen slab loads in flight (a weight gradient has 12-48 slabs; the round-3 form
    // gave each of four waves a quarter of the slabs -- three loads in flight per lane at 12 slabs -- and combined through LDS behind a
    // barrier).  Slabs are added in ascending order within groups of sixteen: fixed order, bit-identical from run to run.
    const size_t mn = (size_t)M * N, slice = mn + (colsum ? (size_t)cs_parts * M : 0);
    const size_t total4 = mn / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total4) {
        const float* p = ws + i * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z0 = 0; z0 < nsplit; z0 += 16) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int z = min(z0 + u, nsplit - 1);                      // branch-free: past the end re-reads the last slab (not added)
                v[u] = *reinterpret_cast<const float4*>(p + (size_t)z * slice);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (z0 + u < nsplit) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        const size_t e = i * 4, m = e / N, n = e - m * N;
        float4* dst = reinterpret_cast<float4*>(C + m * ldc + n);
        const float4 c = *dst;
        *dst = make_float4(c.x + acc.x, c.y + acc.y, c.z + acc.z, c.w + acc.w);
    } else if (colsum && i < total4 + (size_t)M) {
        const size_t m = i - total4;
        float a = 0.f;
        for (int z = 0; z < nsplit; ++z)
            for (int q = 0; q < cs_parts; ++q) a += ws[(size_t)z * slice + mn + (size_t)q * M + m];
        colsum[m] += a;
    }
}

// Launch probe (bench.py's roofline leg): HIP events on the launch stream around the split-K wgrad kernel itself -- not its
// ordered reduce -- so that the live average agrees with rocprofv3's per-kernel average.
// Events come from a pool that is reused from one probe session to the next (nothing is created or destroyed per launch once
// the pool has grown); when the step is captured with the probe on, the records become event-record nodes that every replay
// re-records, so a read after the timed region returns the last replay's launch durations.
struct Probe {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    size_t used = 0;
    double flops = 0.0, bytes = 0.0;
    std::pair<hipEvent_t, hipEvent_t> acquire() {
        if (used == pool.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            // timing events that the host only reads after a device synchronise: no system-scope fence at each record
            const unsigned fl = hipEventDisableSystemFence;
            (void)hipEventCreateWithFlags(&a, fl); (void)hipEventCreateWithFlags(&b, fl);
            pool.emplace_back(a, b);
        }
        return pool[used++];
    }
};
Probe g_probe;

// The first PPF_GEMM_COUNTER_BYTES of a split-K workspace are reserved (they held the arrival counters of the in-kernel reduce of round
// 3; the layout of the ABI is kept): the CALLER hands over a workspace used by ppf_gemm_bf16 calls of ONE stream only.
constexpr size_t PPF_GEMM_COUNTER_BYTES = 16384;

int pick_splitk(int M, int N, int K) {
    // wgrad-style problems (small output, very long contraction)
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // K slices: alone on the GPU the kernel is fastest with as many slices as fit in ONE round of 3 workgroups per CU (768:
    // -17 % vs 540; one slice more spills into a second round and gives it all back).  In the train step these GEMMs run on
    // the side stream under the dgrad chain, where a smaller footprint wins (step time: 432 <= 540 < 768), so that is the default.
    // Narrow layers (an output side <= 256: the D = 192 models) take half as many: their reduce kernel reads every slab back and is a
    // third of the side stream's time there (216: deit_tiny +1.5 %, cait_xxs24 +2.9 % same-box; 144: -2 %; at D = 384 288: -4 %).
    const int target = min(M, N) <= 256 ? 216 : 432;
    int s = target / tiles;
    const int maxs = (K + 4 * BK - 1) / (4 * BK);      // at least 4 K-tiles per slice
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    // slices are BK-aligned chunks: drop the ones that would be empty (a partial-tile slice must always be written)
    const int kchunk = (((K + s - 1) / s) + BK - 1) / BK * BK;
    return (K + kchunk - 1) / kchunk;
}

}  // namespace

extern "C" {

// Test hook: force = 1 routes every shape the 224 x 128 kernel can legally take to it (tests/test_gpu_gemm.py covers its ragged edges at
// shapes the cost model would give to other kernels); 0 restores the cost model.  Not thread-safe; never set on the product path.
int ppf_gemm_test_force_g224(int force) { g_force_g224 = force != 0; return 0; }

// Bytes of split-K workspace ppf_gemm_bf16 needs for an accumulating (epi = 6) problem of this shape.
size_t ppf_gemm_workspace_bytes(int M, int N, int K) {
    return PPF_GEMM_COUNTER_BYTES + (size_t)pick_splitk(M, N, K) * ((size_t)M * N + M) * sizeof(float);
}

// Generic entry. trans_a / trans_b select the storage modes described at the top of this file.
// epi: 0 bf16 out, 1 f32 out, 2 bias+GELU (C = gelu(pre) bf16, aux_out = gelu'(pre) as 8-bit codes, gemm_common.h gelu8_*), 3 sigmoid f32 out,
//      4 residual (C f32 = res + rowscale*colscale*(acc+bias), optional aux_out raw bf16), 5 dGELU (C bf16 = acc*aux_in, aux_in = the gelu' written by epi 2),
//      6 f32 accumulate into C (C += ...; split over the contraction; optional colsum[m] += sum_kc A(m,kc) when trans_a).
//        With a workspace of ppf_gemm_workspace_bytes(M,N,K) the slices write partial tiles that a second kernel reduces in a
//        fixed order (deterministic, no atomics); without one the slices fall back to fp32 atomics on C.
int ppf_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a,
                  int trans_b, int epi, const float* bias, const float* res, int ldres, const float* rowscale,
                  int rows_per_group, const float* colscale, const void* aux_in, void* aux_out, int ldaux, float* colsum,
                  float alpha, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    PPF_CHECK_ARG(M > 0 && N > 0 && K > 0, PPF_ERR_SHAPE, "ppf_gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    PPF_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (ldc % 4) == 0 && (N % 4) == 0, PPF_ERR_ALIGN,
                  "ppf_gemm_bf16: lda/ldb must be multiples of 8, ldc and N of 4 (lda=%d ldb=%d ldc=%d N=%d)", lda, ldb, ldc, N);
    PPF_CHECK_ARG(trans_a ? (M % 8 == 0) : (K % 8 == 0), PPF_ERR_ALIGN, "ppf_gemm_bf16: A inner extent must be a multiple of 8");
    PPF_CHECK_ARG(trans_b ? (N % 8 == 0) : (K % 8 == 0), PPF_ERR_ALIGN, "ppf_gemm_bf16: B inner extent must be a multiple of 8");
    PPF_CHECK_ARG((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, PPF_ERR_ALIGN, "ppf_gemm_bf16: pointers must be 16-byte aligned");
    PPF_CHECK_ARG(!(trans_a && !trans_b), PPF_ERR_ARG, "ppf_gemm_bf16: (trans_a=1, trans_b=0) is not instantiated");
    PPF_CHECK_ARG((((uintptr_t)aux_in | (uintptr_t)aux_out | (uintptr_t)res) & 15) == 0 && (ldaux % 4) == 0 && (ldres % 4) == 0, PPF_ERR_ALIGN,
                  "ppf_gemm_bf16: aux / residual pointers must be 16-byte aligned, ldaux and ldres multiples of 4 (the epilogues use 8/16-byte accesses)");
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = bias; p.res = res; p.ldres = ldres; p.rowscale = rowscale; p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
    p.colscale = colscale; p.aux_in = (const bf16_t*)aux_in; p.aux_out = (bf16_t*)aux_out; p.ldaux = ldaux; p.colsum = colsum; p.alpha = alpha; p.ws = nullptr;
    p.batch_inner = 1; p.sa_o = p.sa_i = p.sb_o = p.sb_i = p.sc_o = p.sc_i = 0; p.kpad = 0; p.cs_parts = 1; p.nsplit = 1;
    if (epi == EPI_RESID) PPF_CHECK_ARG(res != nullptr, PPF_ERR_ARG, "ppf_gemm_bf16: epi=4 needs a residual");
    if (epi == EPI_GELU) PPF_CHECK_ARG(aux_out != nullptr, PPF_ERR_ARG, "ppf_gemm_bf16: epi=2 needs aux_out");
    if (epi == EPI_DGELU) PPF_CHECK_ARG(aux_in != nullptr, PPF_ERR_ARG, "ppf_gemm_bf16: epi=5 needs aux_in");
    if (!trans_a && !trans_b) {
        if (g224_eligible(p, epi)) return launch_g224(p, stream);
        if (g4_eligible(p)) {
            switch (epi) {
                case EPI_BF16: return launch_g4<EPI_BF16>(p, stream);
                case EPI_GELU: return launch_g4<EPI_GELU>(p, stream);
                case EPI_RESID: return launch_g4<EPI_RESID>(p, stream);
                case EPI_DGELU: return launch_g4<EPI_DGELU>(p, stream);
                default: break;
            }
        }
        switch (epi) {
            case EPI_BF16: return launch<false, false, EPI_BF16, false>(p, 1, stream);
            case EPI_F32: return launch<false, false, EPI_F32, false>(p, 1, stream);
            case EPI_GELU: return launch<false, false, EPI_GELU, false>(p, 1, stream);
            case EPI_SIGMOID_F32: return launch<false, false, EPI_SIGMOID_F32, false>(p, 1, stream);
            case EPI_RESID: return launch<false, false, EPI_RESID, false>(p, 1, stream);
            case EPI_DGELU: return launch<false, false, EPI_DGELU, false>(p, 1, stream);
            default: break;
        }
    } else if (!trans_a && trans_b) {
        switch (epi) {
            case EPI_BF16: return launch<false, true, EPI_BF16, false>(p, 1, stream);
            case EPI_F32: return launch<false, true, EPI_F32, false>(p, 1, stream);
            case EPI_DGELU: return launch<false, true, EPI_DGELU, false>(p, 1, stream);
            default: break;
        }
    } else {
        switch (epi) {
            case EPI_ATOMIC: {
                // split over the contraction: with a workspace the K slices leave fp32 partial tiles that splitk_reduce_kernel adds to C in
                // slice order (the 28 MB of slabs are recycled by every weight gradient and stay in the L2 / memory-side cache:
                // profiles/r4_reduce_batch.txt); without one: fp32 atomics.  (The 256x256 weight-gradient kernels of rounds 1-3 and the
                // in-kernel last-arriver reduce were measured slower IN THE STEP once more in round 4 and deleted: profiles/r4_wgrad_tiles.txt.)
                const int ns = pick_splitk(M, N, K);
                const size_t need = PPF_GEMM_COUNTER_BYTES + (size_t)ns * ((size_t)M * N + M) * sizeof(float);
                if (workspace == nullptr || workspace_bytes < need || ns == 1) return launch<true, true, EPI_ATOMIC, true>(p, ns, stream);
                p.ws = (float*)((unsigned char*)workspace + PPF_GEMM_COUNTER_BYTES);
                p.cs_parts = 1;
                p.nsplit = ns;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (g_probe.on) { auto ev = g_probe.acquire(); e0 = ev.first; e1 = ev.second; (void)hipEventRecord(e0, stream); }
                // eight-wave tiles where one output side is long and the other at least 384: 256 x 128 (rows) or 128 x 256 (columns).
                // Same-box: deit_small +0.1 .. +0.4 % in four A/Bs with 19.5 % fewer L2 requests from the weight gradients
                // (profiles/r4_wgrad_l2_knockout.txt); the D = 192 models lose (deit_tiny -2.8 %, cait_xxs24 -0.8 %: a third of their
                // 128-column tiles is padding).
                const bool wide = (M < N ? M : N) >= 320;
                int rc;
                if (wide && M >= 512 && M >= N && (M % 256 == 0 || M >= 1024)) rc = launch_wgrad8<4, 2>(p, ns, stream);
                else if (wide && N >= 512 && (N % 256 == 0 || N >= 1024)) rc = launch_wgrad8<2, 4>(p, ns, stream);
                else rc = launch<true, true, EPI_PARTIAL, true>(p, ns, stream);
                if (rc) return rc;
                if (g_probe.on) {
                    (void)hipEventRecord(e1, stream);
                    g_probe.flops += 2.0 * M * N * (double)K;
                    g_probe.bytes += 2.0 * ((double)M * K + (double)N * K) + 4.0 * M * N;
                }
                const size_t work = (size_t)M * N / 4 + (colsum ? M : 0);
                const int grid = (int)((work + 255) / 256);
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, stream, p.ws, (float*)C, colsum, M, N, ldc, ns, p.cs_parts);
                PPF_LAUNCH_CHECK();
                return 0;
            }
            case EPI_F32: return launch<true, true, EPI_F32, false>(p, 1, stream);
            default: break;
        }
    }
    ppf_set_error("ppf_gemm_bf16: combination trans_a=%d trans_b=%d epi=%d not instantiated", trans_a, trans_b, epi);
    return PPF_ERR_ARG;
}

// Roofline probe of the weight-gradient kernel (epi 6 with a workspace).  enable = 1 clears the counters and starts recording
// (pooled events are reused), 0 stops, 2 stops and destroys the event pool (only when no captured graph references it any more).
int ppf_gemm_probe(int enable) {
    if (enable == 1) { g_probe.used = 0; g_probe.flops = 0.0; g_probe.bytes = 0.0; }
    if (enable == 2) {
        for (auto& e : g_probe.pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        g_probe.pool.clear(); g_probe.used = 0;
    }
    g_probe.on = enable == 1;
    return 0;
}
// Synchronises the recorded events: total kernel milliseconds, launches, algorithmic flops and bytes since ppf_gemm_probe(1).
int ppf_gemm_probe_read(double* ms_total, int64_t* launches, double* flops, double* bytes) {
    double ms = 0.0;
    for (size_t i = 0; i < g_probe.used; ++i) {
        auto& e = g_probe.pool[i];
        float t = 0.f;
        hipError_t rc = hipEventSynchronize(e.second);
        if (rc == hipSuccess) rc = hipEventElapsedTime(&t, e.first, e.second);
        if (rc != hipSuccess) { ppf_set_error("ppf_gemm_probe_read: %s", hipGetErrorString(rc)); return (int)rc; }
        ms += t;
    }
    if (ms_total) *ms_total = ms;
    if (launches) *launches = (int64_t)g_probe.used;
    if (flops) *flops = g_probe.flops;
    if (bytes) *bytes = g_probe.bytes;
    return 0;
}

// Batched plain GEMMs (no bias / fused epilogue): nbatch = batch_outer * batch_inner problems, problem (o, i) uses
// A + o*sa_o + i*sa_i (elements) etc.  out_f32 selects fp32 or bf16 C.  kpad = 1 lets contraction-contiguous operands read the
// zero/finite padding up to the next multiple of 8 beyond K (K need not be a multiple of 8 then).
// Used by the CaiT talking-heads attention: per-(sample, head) P.V, dO.V^T, dS.K, dS^T.Q, A^T.dO products.
int ppf_gemm_bf16_batched(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a, int trans_b,
                          int out_f32, float alpha, int batch_outer, int batch_inner, int64_t sa_o, int64_t sa_i, int64_t sb_o, int64_t sb_i,
                          int64_t sc_o, int64_t sc_i, int kpad, hipStream_t stream) {
    PPF_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch_outer > 0 && batch_inner > 0, PPF_ERR_SHAPE, "ppf_gemm_bf16_batched: bad shape");
    PPF_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (ldc % 4) == 0 && (N % 4) == 0, PPF_ERR_ALIGN, "ppf_gemm_bf16_batched: ld alignment");
    PPF_CHECK_ARG(kpad || (trans_a ? (M % 8 == 0) : (K % 8 == 0)), PPF_ERR_ALIGN, "ppf_gemm_bf16_batched: A inner extent must be a multiple of 8");
    PPF_CHECK_ARG(kpad || (trans_b ? (N % 8 == 0) : (K % 8 == 0)), PPF_ERR_ALIGN, "ppf_gemm_bf16_batched: B inner extent must be a multiple of 8");
    PPF_CHECK_ARG(((sa_o | sa_i | sb_o | sb_i) % 8) == 0 && ((sc_o | sc_i) % 4) == 0, PPF_ERR_ALIGN, "ppf_gemm_bf16_batched: batch strides alignment");
    // the bf16-output epilogue stores 16 bytes per lane whenever ldc % 8 == 0: every problem's C must then start on a 16-byte boundary
    PPF_CHECK_ARG(out_f32 || (ldc % 8) != 0 || ((sc_o | sc_i) % 8) == 0, PPF_ERR_ALIGN,
                  "ppf_gemm_bf16_batched: bf16 output with ldc %% 8 == 0 needs batch strides of C that are multiples of 8 elements (sc_o=%lld sc_i=%lld)",
                  (long long)sc_o, (long long)sc_i);
    PPF_CHECK_ARG((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, PPF_ERR_ALIGN, "ppf_gemm_bf16_batched: pointers must be 16-byte aligned");
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = nullptr; p.res = nullptr; p.ldres = 0; p.rowscale = nullptr; p.rows_per_group = 1; p.colscale = nullptr; p.aux_in = nullptr;
    p.aux_out = nullptr; p.ldaux = 0; p.colsum = nullptr; p.ws = nullptr; p.alpha = alpha;
    p.batch_inner = batch_inner; p.sa_o = sa_o; p.sa_i = sa_i; p.sb_o = sb_o; p.sb_i = sb_i; p.sc_o = sc_o; p.sc_i = sc_i; p.kpad = kpad;
    const int nb = batch_outer * batch_inner;
    const int key = (trans_a ? 4 : 0) | (trans_b ? 2 : 0) | (out_f32 ? 1 : 0);
    switch (key) {
        case 0: return launch<false, false, EPI_BF16, false>(p, 1, stream, nb);
        case 1: return launch<false, false, EPI_F32, false>(p, 1, stream, nb);
        case 2: return launch<false, true, EPI_BF16, false>(p, 1, stream, nb);
        case 3: return launch<false, true, EPI_F32, false>(p, 1, stream, nb);
        case 6: return launch<true, true, EPI_BF16, false>(p, 1, stream, nb);
        case 7: return launch<true, true, EPI_F32, false>(p, 1, stream, nb);
        default: break;
    }
    ppf_set_error("ppf_gemm_bf16_batched: (trans_a=1, trans_b=0) is not instantiated");
    return PPF_ERR_ARG;
}

}  // extern "C"
