// 256x256x64 bf16 MFMA GEMM for gfx950, both operands contraction-contiguous:
//
//   C[m][n] = epilogue( sum_k A[m*lda + k] * B[n*ldb + k] )          (x W^T forward GEMMs; dgrad through a W^T shadow)
//
// One 512-thread workgroup (8 waves, 2 along m x 4 along n, 128x64 outputs per wave) per CU, 128 KiB of LDS:
//
//   * operands go global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write pass).  The LDS image of a
//     128-row x 64-k half-tile is lane-linear, so the bank swizzle (16-byte chunk ^= (row>>1)&7) is applied to the per-lane
//     SOURCE address and again on the fragment reads.
//   * a K tile is four half-tiles (A rows 0-127 / 128-255, B rows 0-127 / 128-255), double-buffered.  Each K tile is
//     multiplied in four phases (one 64x32 accumulator quadrant of every wave x K=64 = 8 MFMA 32x32x16 per phase); every
//     phase also issues one half-tile of a LATER K tile, into a slot whose last fragment read is >= 2 phases old.  Loads are
//     retired with a counted s_waitcnt vmcnt(4) once per K tile (never 0 inside the loop), one phase before the first read.
//   * the two waves that share a SIMD (wave w and w+4: the two m-halves) run one barrier apart, so one of them is in its
//     ds_read / load-issue half of a phase while the other is in its MFMA half.
//
// The fused epilogues are the ones of gemm_bf16.hip (gemm_common.h); the accumulators leave through per-wave LDS strips so that
// stores, bias and residual accesses are full 256-byte row segments.
#include "gemm_common.h"
#include <cstdlib>

namespace {
using namespace ppfg;

constexpr int TM = 256, TN = 256, TK = 64, NTHR = 512;
constexpr int HALF = 128 * TK * 2;           // 16 KiB: one half-tile
constexpr int BUFB = 4 * HALF;               // A0 A1 B0 B1 of one K tile
constexpr int LDS_BYTES = 2 * BUFB;          // 128 KiB
constexpr int SLD = 68;                      // fp32 pitch of the epilogue strip

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

#define NT_BAR() __builtin_amdgcn_s_barrier()
#define NT_PIN() __builtin_amdgcn_sched_barrier(0)

struct Ctx {
    const bf16_t* srcA[2][2];    // [half][instr] this thread's source chunk at k = 0 (swizzle applied)
    const bf16_t* srcB[2][2];
    unsigned char* smem;
    int wave_off;                // wave * 1024: this wave's 64 x 16 B run inside an 8 KiB instruction slab
    int a_rd, b_rd;              // byte offsets of this wave's A / B fragment rows inside a K-tile buffer
    int rd[4];                   // per-lane swizzled offset of k-substep ks
    int nt;
};

// one half-tile: 1024 chunks of 16 B, two per thread
__device__ __forceinline__ void stage_half(const Ctx& c, unsigned char* slot, const bf16_t* const (&src)[2], int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[i] + (size_t)kt * TK), (lds_void_t*)(slot + i * 8192 + c.wave_off), 16, 0, 0);
}

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// Multiply K tile u (resident in buffer BI).  TAIL = false: tiles u+1 and u+2 both exist (no bounds tests in the loop body).
template <int BI, bool TAIL>
__device__ __forceinline__ void ktile(const Ctx& c, int u, f32x16 (&acc)[2][4], bf16x8 (&fa)[2][4], bf16x8 (&fb)[2][4]) {
    unsigned char* cur = c.smem + BI * BUFB;
    unsigned char* oth = c.smem + (BI ^ 1) * BUFB;
    const unsigned char* ca = cur + c.a_rd;
    const unsigned char* cb = cur + c.b_rd;
    const bool more1 = !TAIL || (u + 1 < c.nt);
    const bool more2 = !TAIL || (u + 2 < c.nt);

    // ---- phase 1: B sub-tiles 0/1 and A sub-tile 0 -> registers; quadrant (m 0-63, n 0-31)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fb[0][ks] = lds_frag(cb + c.rd[ks]);
        fb[1][ks] = lds_frag(cb + 32 * 128 + c.rd[ks]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fa[0][ks] = lds_frag(ca + c.rd[ks]);
        fa[1][ks] = lds_frag(ca + 32 * 128 + c.rd[ks]);
    }
    if (more1) stage_half(c, oth + 0 * HALF, c.srcA[0], u + 1);
    NT_PIN(); NT_BAR(); NT_PIN();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[0][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][ks], fa[f][ks], acc[0][f], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    NT_PIN(); NT_BAR(); NT_PIN();

    // ---- phase 2: quadrant (m 0-63, n 32-63)
    if (more1) stage_half(c, oth + 1 * HALF, c.srcA[1], u + 1);
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
        fa[0][ks] = lds_frag(ca + 64 * 128 + c.rd[ks]);
        fa[1][ks] = lds_frag(ca + 96 * 128 + c.rd[ks]);
    }
    if (more2) stage_half(c, cur + 2 * HALF, c.srcB[0], u + 2);
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
    if (more2) {
        stage_half(c, cur + 3 * HALF, c.srcB[1], u + 2);
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
    NT_PIN(); NT_BAR(); NT_PIN();
}

template <int EPI>
__global__ __launch_bounds__(NTHR) void gemm_nt256_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int tiles_n = (p.N + TN - 1) / TN;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (vid / tiles_n) * TM, n0 = (vid % tiles_n) * TN;

    Ctx c;
    c.smem = smem;
    c.wave_off = wave * 1024;
    c.nt = p.K / TK;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = i * NTHR + tid;
            const int r = q >> 3, ch = (q & 7) ^ ((r >> 1) & 7);
            const int ra = min(m0 + h * 128 + r, p.M - 1), rb = min(n0 + h * 128 + r, p.N - 1);   // edge tiles re-read the last row
            c.srcA[h][i] = p.A + (size_t)ra * p.lda + ch * 8;
            c.srcB[h][i] = p.B + (size_t)rb * p.ldb + ch * 8;
        }
    {
        const int l31 = lane & 31, hh = lane >> 5, sw = (l31 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c.rd[ks] = l31 * 128 + (((ks * 2 + hh) ^ sw) << 4);
    }
    c.a_rd = wr * HALF;
    c.b_rd = (2 + (wc >> 1)) * HALF + (wc & 1) * 64 * 128;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][4], fb[2][4];

    // prologue: K tile 0 complete, the B half-tiles of K tile 1 in flight
    stage_half(c, smem + 2 * HALF, c.srcB[0], 0);
    stage_half(c, smem + 3 * HALF, c.srcB[1], 0);
    stage_half(c, smem + 0 * HALF, c.srcA[0], 0);
    stage_half(c, smem + 1 * HALF, c.srcA[1], 0);
    if (c.nt > 1) {
        stage_half(c, smem + BUFB + 2 * HALF, c.srcB[0], 1);
        stage_half(c, smem + BUFB + 3 * HALF, c.srcB[1], 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    NT_PIN(); NT_BAR(); NT_PIN();
    if (wr == 1) { NT_BAR(); }                   // the second m-half runs one barrier behind the first
    NT_PIN();

    int u = 0;
    for (; u + 3 < c.nt; u += 2) {
        ktile<0, false>(c, u, acc, fa, fb);
        ktile<1, false>(c, u + 1, acc, fa, fb);
    }
    for (; u < c.nt; u += 2) {
        ktile<0, true>(c, u, acc, fa, fb);
        if (u + 1 < c.nt) ktile<1, true>(c, u + 1, acc, fa, fb);
    }
    if (wr == 0) { NT_BAR(); }                   // re-align the two halves: every fragment read has retired past this point
    NT_PIN();

    // epilogue: same strip transpose as gemm_bf16.hip (lane holds row wm+32*mi+(lane&31), columns wn+32*ni+8*g+4*(lane>>5)..+3)
    const int hh = lane >> 5;
    const int wm = wr * 128, wn = wc * 64;
    float* strip = reinterpret_cast<float*>(smem) + wave * (32 * SLD);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(strip + (lane & 31) * SLD + 32 * ni + 8 * g + 4 * hh) =
                    make_float4(acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]);
        const int col = (lane & 15) * 4;
        const int n = n0 + wn + col;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int r = pass * 4 + (lane >> 4);
            const int m = m0 + wm + 32 * mi + r;
            const float4 v = *reinterpret_cast<const float4*>(strip + r * SLD + col);
            if (m < p.M && n < p.N) epilogue4<EPI>(p, m, n, v.x, v.y, v.z, v.w);
        }
    }
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
    const int tiles = ((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHR), LDS_BYTES, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace ppfg {

// Shapes the pipelined kernel takes.  Measured on MI355X (profiles/r1_gemm_nt256.txt): with one workgroup per CU nothing overlaps
// a tile's output stores, so the kernel only wins where the contraction is long enough to amortise them -- ~1.0 PFLOP/s vs
// ~0.75 for the 128x128 kernel at K >= 1536, parity at K = 384 -- and where 256-wide tiles waste little of N.
// PPF_GEMM_NT256 = 0 never, 1 always when legal, unset: the heuristic below.
bool nt256_eligible(const GemmParams& p, int epi) {
    static const int mode = getenv("PPF_GEMM_NT256") ? atoi(getenv("PPF_GEMM_NT256")) : -1;
    if (mode == 0) return false;
    if (!(epi == EPI_BF16 || epi == EPI_F32 || epi == EPI_GELU || epi == EPI_RESID || epi == EPI_DGELU)) return false;
    if (p.K % TK != 0 || p.K < 2 * TK || p.kpad) return false;
    const int tn = (p.N + TN - 1) / TN;
    const long long tiles = (long long)((p.M + TM - 1) / TM) * tn;
    if (mode == 1) return true;
    return tiles >= 192 && p.K >= 768 && (long long)tn * TN * 10 <= (long long)p.N * 11;
}

int launch_nt256(const GemmParams& p, int epi, hipStream_t stream) {
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
