// 256x256x64 bf16 MFMA GEMM for gfx950, both operands contraction-contiguous, persistent workgroups:
//
//   C[m][n] = epilogue( sum_k A[m*lda + k] * B[n*ldb + k] )          (x W^T forward GEMMs)
//
// One 512-thread workgroup (8 waves, 2 along m x 4 along n, 128x64 outputs per wave) per CU walks a strided list of
// output tiles; the K-tile stream never drains between tiles:
//
//   * operands go global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write pass).  The LDS image of a
//     128-row x 64-k half-tile is lane-linear, so the bank swizzle (16-byte chunk ^= (row>>1)&7) is applied to the per-lane
//     SOURCE address and again on the fragment reads.
//   * a K tile is four half-tiles (A rows 0-127 / 128-255, B rows 0-127 / 128-255), double-buffered (128 KiB).  Each K tile
//     is multiplied in four phases (one 64x32 accumulator quadrant of every wave x K=64 = 8 MFMA 32x32x16 per phase); every
//     phase also issues one half-tile of a LATER K tile -- of the next output tile near the end of this one -- into a slot
//     whose last fragment read is >= 2 phases old.  Loads are retired with a counted s_waitcnt vmcnt(4) once per K tile
//     (never 0 inside the loop), one phase before the first read.  Epilogue stores share the counter; loads retire in order
//     among loads, so "at most 4 outstanding" still implies that everything older than the last two half-tiles has landed.
//   * the two waves that share a SIMD (wave w and w+4: the two m-halves) run one barrier apart, so one of them is in its
//     ds_read / load-issue half of a phase while the other is in its MFMA half.
//   * the fused epilogues (gemm_common.h) run at the tile boundary with the next tile's operands already in flight: each
//     wave transposes its accumulators 16 rows at a time through a private 4 KiB LDS strip above the operand ring, so the
//     stores are full 256-byte row segments and drain while the next tile is being multiplied.  The first m-half does this
//     after the boundary barrier, the second before it: both epilogues fall into the same barrier interval.
#include "gemm_common.h"
#include <cstdlib>

namespace {
using namespace ppfg;

constexpr int TM = 256, TN = 256, TK = 64, NTHR = 512;
constexpr int HALF = 128 * TK * 2;           // 16 KiB: one half-tile
constexpr int BUFB = 4 * HALF;               // A0 A1 B0 B1 of one K tile
constexpr int RING = 2 * BUFB;               // 128 KiB
constexpr int STRIP = 16 * 64 * 4;           // 4 KiB per wave: 16 rows x 64 fp32, float4 slot ^= row
constexpr int LDS_BYTES = RING + 8 * STRIP;  // 160 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

#define NT_BAR() __builtin_amdgcn_s_barrier()
#define NT_PIN() __builtin_amdgcn_sched_barrier(0)

struct Tile {                    // uniform description of one output tile (and, for split-K, of its slice of the contraction)
    const unsigned char* a;      // first operand byte of the (shifted) tile at its first K tile
    const unsigned char* b;
    int m_lo, n_lo;              // outputs below these belong to the previous tile (edge tiles are shifted to stay in range)
    int m0s, n0s;                // shifted origin
    int nt;                      // K tiles (even)
    int z;                       // split-K slice
};

struct Ctx {
    Tile cur, nxt;
    unsigned ta, tb;             // this thread's byte offset inside one glds instruction's slab (swizzled source chunk)
    unsigned ia, ib;             // second instruction of a half-tile: + 64 rows
    unsigned ha, hb;             // second half-tile: + 128 rows (128*ld*2 bytes)
    unsigned ka, kb;             // next K tile: + 128 bytes
    unsigned char* smem;
    int wave_off;                // wave * 1024: this wave's 64 x 16 B run inside an 8 KiB instruction slab
    int a_half, b_half;          // byte offsets of this wave's A / B half-tile inside a K-tile buffer
    int b_row;                   // first B row (of 128) this wave multiplies
    int rd[4];                   // per-lane swizzled offset of k-substep ks
    bool has_next;               // this workgroup has another output tile after the current one
};

// Edge tiles are shifted back so that all 256 rows exist (M, N >= 256); the rows they share with the previous tile are
// recomputed identically and masked out of the epilogue (m_lo / n_lo).  vid = slice * tiles + tile (n fastest).
__device__ __forceinline__ void tile_desc(const GemmParams& p, int vid, Tile& t) {
    const int tiles_n = (p.N + TN - 1) / TN;
    const int tiles = tiles_n * ((p.M + TM - 1) / TM);
    const int z = vid / tiles, v = vid - z * tiles;
    t.z = z;
    t.m_lo = (v / tiles_n) * TM; t.n_lo = (v % tiles_n) * TN;
    t.m0s = min(t.m_lo, p.M - TM); t.n0s = min(t.n_lo, p.N - TN);
    const int nkt = p.K / TK;
    const int per = ((nkt + p.nsplit - 1) / p.nsplit + 1) & ~1;       // K tiles per slice, even
    const int k0 = z * per;
    t.nt = min(per, nkt - k0);
    t.a = reinterpret_cast<const unsigned char*>(p.A) + ((size_t)t.m0s * p.lda + (size_t)k0 * TK) * 2;
    t.b = reinterpret_cast<const unsigned char*>(p.B) + ((size_t)t.n0s * p.ldb + (size_t)k0 * TK) * 2;
}

// one half-tile (128 rows x 64 k): 1024 chunks of 16 B, two per thread
__device__ __forceinline__ void stage_half(const Ctx& c, unsigned char* slot, const unsigned char* base, unsigned toff, unsigned istep,
                                           unsigned hstep, unsigned kstep, int half, int kt) {
    const unsigned char* bk = base + (size_t)kt * kstep + (size_t)half * hstep;      // uniform
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(bk + (size_t)i * istep + toff), (lds_void_t*)(slot + i * 8192 + c.wave_off), 16, 0, 0);
}
__device__ __forceinline__ void stage_a(const Ctx& c, unsigned char* slot, const Tile& t, int half, int kt) {
    stage_half(c, slot, t.a, c.ta, c.ia, c.ha, c.ka, half, kt);
}
__device__ __forceinline__ void stage_b(const Ctx& c, unsigned char* slot, const Tile& t, int half, int kt) {
    stage_half(c, slot, t.b, c.tb, c.ib, c.hb, c.kb, half, kt);
}
// K tile s of the stream that starts at the current output tile: s >= nt is K tile s - nt of the next output tile
template <bool IS_A>
__device__ __forceinline__ void stage_stream(const Ctx& c, unsigned char* slot, int half, int s) {
    const Tile& t = s < c.cur.nt ? c.cur : c.nxt;
    const int kt = s < c.cur.nt ? s : s - c.cur.nt;
    if (IS_A) stage_a(c, slot, t, half, kt); else stage_b(c, slot, t, half, kt);
}

// MFMA fragment of the 32-row sub-tile at `row` (multiple of 32) of a half-tile [128 r][64 k], k-substep ks: lane l holds row l&31,
// contraction values ks*16 + (l>>5)*8 .. +7 (one ds_read_b128)
__device__ __forceinline__ bf16x8 lds_frag(const Ctx& c, const unsigned char* half, int row, int ks) {
    return *reinterpret_cast<const bf16x8*>(half + row * 128 + c.rd[ks]);
}

// Multiply K tile u (resident in buffer BI) up to the last MFMA of phase 4; the caller issues that phase's closing barrier.
// TAIL = false: K tiles u+1 and u+2 are tiles of the current output tile (no tests in the loop body).
template <int BI, bool TAIL>
__device__ __forceinline__ void ktile(const Ctx& c, int u, f32x16 (&acc)[2][4], bf16x8 (&fa)[2][4], bf16x8 (&fb)[2][4]) {
    unsigned char* cur = c.smem + BI * BUFB;
    unsigned char* oth = c.smem + (BI ^ 1) * BUFB;
    const unsigned char* ca = cur + c.a_half;
    const unsigned char* cb = cur + c.b_half;
    const bool more1 = !TAIL || (u + 1 < c.cur.nt) || c.has_next;
    const bool more2 = !TAIL || (u + 2 < c.cur.nt) || c.has_next;

    // ---- phase 1: B sub-tiles 0/1 and A sub-tile 0 -> registers; quadrant (m 0-63, n 0-31)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fb[0][ks] = lds_frag(c, cb, c.b_row, ks);
        fb[1][ks] = lds_frag(c, cb, c.b_row + 32, ks);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fa[0][ks] = lds_frag(c, ca, 0, ks);
        fa[1][ks] = lds_frag(c, ca, 32, ks);
    }
    if (!TAIL) stage_a(c, oth + 0 * HALF, c.cur, 0, u + 1);
    else if (more1) stage_stream<true>(c, oth + 0 * HALF, 0, u + 1);
    NT_PIN(); NT_BAR(); NT_PIN();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[0][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][ks], fa[f][ks], acc[0][f], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    NT_PIN(); NT_BAR(); NT_PIN();

    // ---- phase 2: quadrant (m 0-63, n 32-63)
    if (!TAIL) stage_a(c, oth + 1 * HALF, c.cur, 1, u + 1);
    else if (more1) stage_stream<true>(c, oth + 1 * HALF, 1, u + 1);
    NT_PIN(); NT_BAR(); NT_PIN();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[1][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][ks], fa[f][ks], acc[1][f], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    NT_PIN(); NT_BAR(); NT_PIN();

    // ---- phase 3: A sub-tile 1 -> registers; quadrant (m 64-127, n 32-63).  The B slots of this buffer were last read in
    // phase 1: restage them with K tile u+2.
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fa[0][ks] = lds_frag(c, ca, 64, ks);
        fa[1][ks] = lds_frag(c, ca, 96, ks);
    }
    if (!TAIL) stage_b(c, cur + 2 * HALF, c.cur, 0, u + 2);
    else if (more2) stage_stream<false>(c, cur + 2 * HALF, 0, u + 2);
    NT_PIN(); NT_BAR(); NT_PIN();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[1][2 + f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][ks], fa[f][ks], acc[1][2 + f], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    NT_PIN(); NT_BAR(); NT_PIN();

    // ---- phase 4: quadrant (m 64-127, n 0-31).  Retire K tile u+1 (everything but the two half-tiles of u+2 just issued)
    // BEFORE this phase's first barrier; it is read from the next phase on.
    if (!TAIL) {
        stage_b(c, cur + 3 * HALF, c.cur, 1, u + 2);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (more2) {
        stage_stream<false>(c, cur + 3 * HALF, 1, u + 2);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    NT_PIN(); NT_BAR(); NT_PIN();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[0][2 + f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][ks], fa[f][ks], acc[0][2 + f], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    NT_PIN();
}

// Fused epilogue of one output tile.  After the MFMAs a lane holds, for each (ni, mi): row 32*mi+(lane&31) and, for g = 0..3,
// the four consecutive columns 32*ni+8*g+4*(lane>>5).. (acc regs 4g..4g+3).  Each wave moves its 128x64 block 16 rows at a
// time through its private strip so that 16 consecutive lanes cover one 64-column row segment; then re-zeroes the accumulators.
template <int EPI>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, float* strip, int m0w, int n0w, int m_lo, int n_lo, int lane, f32x16 (&acc)[2][4]) {
    const int l31 = lane & 31, hh = lane >> 5, r16 = l31 & 15;
    const int col = (lane & 15) * 4;
    const int n = n0w + col;
    const EpiCols cc = epi_load_cols<EPI>(p, n, true);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            if ((l31 >> 4) == rh) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int slot = (8 * ni + 2 * g + hh) ^ r16;
                        *reinterpret_cast<float4*>(strip + r16 * 64 + slot * 4) =
                            make_float4(acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]);
                    }
            }
            // lanes exchange data through the strip without a workgroup barrier: the LDS unit keeps one wave's accesses in order,
            // but the compiler must not move a lane's reads above other lanes' (divergent) writes -> wave_barrier (no instruction;
            // a release/acquire fence pair also works but makes hipcc wait vmcnt(0), i.e. for the previous unit's global stores)
            __builtin_amdgcn_wave_barrier();
            EpiRow rr[4];
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int m = m0w + 32 * mi + 16 * rh + pass * 4 + (lane >> 4);
                rr[pass] = epi_load_row<EPI>(p, m, n, m >= m_lo && n >= n_lo);
            }
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int r = pass * 4 + (lane >> 4);
                const int m = m0w + 32 * mi + 16 * rh + r;
                const float4 v = *reinterpret_cast<const float4*>(strip + r * 64 + (((lane & 15) ^ r) << 2));
                if (m >= m_lo && n >= n_lo) epi_store<EPI>(p, m, n, v.x, v.y, v.z, v.w, cc, rr[pass]);
            }
            __builtin_amdgcn_wave_barrier();                 // the next unit overwrites the strip
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

template <int EPI>
__global__ __launch_bounds__(NTHR) void gemm_nt256_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int tiles_n = (p.N + TN - 1) / TN;
    const int tiles = tiles_n * ((p.M + TM - 1) / TM);
    const int ntiles = tiles * p.nsplit;
    const int G = gridDim.x;
    // round j of the persistent loop covers tiles [j*G, (j+1)*G); inside a round consecutive tiles (n fastest: one A panel)
    // go to workgroups of the same XCD
    int vid = xcd_remap(blockIdx.x, G);
    if (vid >= ntiles) return;

    Ctx c;
    c.smem = smem;
    c.wave_off = wave * 1024;
    {
        const int r = tid >> 3, ch = (tid & 7) ^ ((r >> 1) & 7);      // rows r and 64 + r of a half share the swizzle
        c.ta = (unsigned)r * (unsigned)p.lda * 2u + ch * 16;
        c.tb = (unsigned)r * (unsigned)p.ldb * 2u + ch * 16;
        c.ia = 64u * (unsigned)p.lda * 2u; c.ib = 64u * (unsigned)p.ldb * 2u;
        c.ha = 2u * c.ia; c.hb = 2u * c.ib;
        c.ka = c.kb = TK * 2;
        const int l31 = lane & 31, hh = lane >> 5, sw = (l31 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c.rd[ks] = l31 * 128 + (((ks * 2 + hh) ^ sw) << 4);
    }
    c.a_half = wr * HALF;
    c.b_half = (2 + (wc >> 1)) * HALF;
    c.b_row = (wc & 1) * 64;
    float* strip = reinterpret_cast<float*>(smem + RING + wave * STRIP);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][4], fb[2][4];

    // prologue: K tile 0 of the first output tile complete, the B half-tiles of K tile 1 in flight
    tile_desc(p, vid, c.cur);
    c.nxt = c.cur;
    stage_b(c, smem + 2 * HALF, c.cur, 0, 0);
    stage_b(c, smem + 3 * HALF, c.cur, 1, 0);
    stage_a(c, smem + 0 * HALF, c.cur, 0, 0);
    stage_a(c, smem + 1 * HALF, c.cur, 1, 0);
    stage_b(c, smem + BUFB + 2 * HALF, c.cur, 0, 1);
    stage_b(c, smem + BUFB + 3 * HALF, c.cur, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    NT_PIN(); NT_BAR(); NT_PIN();
    if (wr == 1) { NT_BAR(); }                   // the second m-half runs one barrier behind the first
    NT_PIN();

    for (;;) {
        const int nvid = vid + G;
        c.has_next = nvid < ntiles;
        if (c.has_next) tile_desc(p, nvid, c.nxt);
        int u = 0;
        for (; u + 3 < c.cur.nt; u += 2) {
            ktile<0, false>(c, u, acc, fa, fb);
            NT_BAR(); NT_PIN();
            ktile<1, false>(c, u + 1, acc, fa, fb);
            NT_BAR(); NT_PIN();
        }
        ktile<0, true>(c, u, acc, fa, fb);
        NT_BAR(); NT_PIN();
        ktile<1, true>(c, u + 1, acc, fa, fb);
        // tile boundary: both m-halves run their epilogue in the barrier interval that follows the first half's last MFMA
        const int m0 = c.cur.m0s + wr * 128, n0 = c.cur.n0s + wc * 64;
        GemmParams q = p;
        q.zslice = c.cur.z;
        auto finish = [&]() {
            tile_epilogue<EPI>(q, strip, m0, n0, c.cur.m_lo, c.cur.n_lo, lane, acc);
        };
        if (wr == 1) finish();
        NT_PIN(); NT_BAR(); NT_PIN();
        if (wr == 0) finish();
        NT_PIN();
        if (!c.has_next) break;
        vid = nvid;
        c.cur = c.nxt;
    }
    if (wr == 0) { NT_BAR(); }                   // every wave executes the same number of barriers
}

int num_cus() {
    static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    return ncu;
}

template <int EPI>
int launch_one(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm_nt256_kernel<EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(gemm_nt256): %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tiles = ((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN) * p.nsplit;
    const int ncu = num_cus();
    hipLaunchKernelGGL(kern, dim3(tiles < ncu ? tiles : ncu), dim3(NTHR), LDS_BYTES, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace ppfg {

// Shapes the pipelined kernel takes.  Measured on MI355X (profiles/r1_gemm_nt256.txt): with one workgroup per CU nothing overlaps
// a tile's output stores, so the kernel only wins where the contraction is long enough to amortise them -- ~1.0 PFLOP/s vs
// ~0.75 for the 128x128 kernel at K >= 1536, parity at K = 384 -- and where 256-wide tiles waste little of N.
// (Retired from libppf_hip.so in round 6: no BASELINE configuration has a K >= 768 forward product; kept here with its own entry point.)
bool nt256_eligible(const GemmParams& p, int epi) {
    if (!(epi == EPI_BF16 || epi == EPI_F32 || epi == EPI_GELU || epi == EPI_RESID || epi == EPI_DGELU)) return false;
    if (p.K % (2 * TK) != 0 || p.kpad) return false;                      // an even number of K tiles: buffer parity restarts per tile
    if (p.M < TM || p.N < TN || (long long)p.lda * 256 >= (1ll << 30) || (long long)p.ldb * 256 >= (1ll << 30)) return false;
    return true;
}

int launch_nt256(const GemmParams& p_, int epi, hipStream_t stream) {
    GemmParams p = p_;
    p.nsplit = 1;
    switch (epi) {
        case EPI_BF16: return launch_one<EPI_BF16>(p, stream);
        case EPI_F32: return launch_one<EPI_F32>(p, stream);
        case EPI_GELU: return launch_one<EPI_GELU>(p, stream);
        case EPI_RESID: return launch_one<EPI_RESID>(p, stream);
        case EPI_DGELU: return launch_one<EPI_DGELU>(p, stream);
        default: break;
    }
    ppf_set_error("gemm_nt256: epilogue %d not instantiated", epi);
    return PPF_ERR_ARG;
}

}  // namespace ppfg

// Stand-alone C entry (the product library reached this kernel through ppf_gemm_bf16's dispatcher until round 6):
// C[M][N] = epilogue(A[M][K] . B[N][K]^T + bias), epi 0 bf16 | 1 fp32 | 2 GELU (+ gelu' codes to aux) | 4 fp32 residual | 5 x gelu' codes from aux
extern "C" int ppf_gemm_nt256(const void* A, const void* B, void* C, int M, int N, int K, int epi, const float* bias, const float* res, void* aux,
                              hipStream_t stream) {
    ppfg::GemmParams p{};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N;
    p.bias = bias; p.res = res; p.ldres = N; p.rows_per_group = 1; p.aux_in = (const bf16_t*)aux; p.aux_out = (bf16_t*)aux; p.ldaux = N;
    p.alpha = 1.0f; p.batch_inner = 1; p.cs_parts = 1; p.nsplit = 1;
    if (!ppfg::nt256_eligible(p, epi)) return PPF_ERR_SHAPE;
    return ppfg::launch_nt256(p, epi, stream);
}
