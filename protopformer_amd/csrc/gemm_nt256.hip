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
    unsigned ia, ib;             // second instruction of a half-tile: + 64 rows (NT) / + 32 contraction rows (TT)
    unsigned ha, hb;             // second half-tile: + 128 rows (NT: 128*ld*2 bytes, TT: 256 bytes)
    unsigned ka, kb;             // next K tile: + 128 bytes (NT) / + 64*ld*2 bytes (TT)
    unsigned char* smem;
    int wave_off;                // wave * 1024: this wave's 64 x 16 B run inside an 8 KiB instruction slab
    int a_half, b_half;          // byte offsets of this wave's A / B half-tile inside a K-tile buffer
    int b_row;                   // first B row (of 128) this wave multiplies
    int rd[4];                   // NT: per-lane swizzled offset of k-substep ks.  TT: per-lane offset of 32-row sub-tile r5
    int brd[2];                  // TT: the same for this wave's two B sub-tiles
    bool has_next;               // this workgroup has another output tile after the current one
    bool colsum;                 // TT: accumulate sum_kc A(m, kc) for the current tile (bias gradient)
};

// Edge tiles are shifted back so that all 256 rows exist (M, N >= 256); the rows they share with the previous tile are
// recomputed identically and masked out of the epilogue (m_lo / n_lo).  vid = slice * tiles + tile (n fastest).
template <bool TT>
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
    if (TT) {
        t.a = reinterpret_cast<const unsigned char*>(p.A) + ((size_t)k0 * TK * p.lda + t.m0s) * 2;
        t.b = reinterpret_cast<const unsigned char*>(p.B) + ((size_t)k0 * TK * p.ldb + t.n0s) * 2;
    } else {
        t.a = reinterpret_cast<const unsigned char*>(p.A) + ((size_t)t.m0s * p.lda + (size_t)k0 * TK) * 2;
        t.b = reinterpret_cast<const unsigned char*>(p.B) + ((size_t)t.n0s * p.ldb + (size_t)k0 * TK) * 2;
    }
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

// MFMA fragment of the 32-row sub-tile at `row` (multiple of 32) of a half-tile, k-substep ks: lane l holds row l&31,
// contraction values ks*16 + (l>>5)*8 .. +7.  NT half-tiles are [128 r][64 k] (b128 reads), TT half-tiles [64 k][128 r]
// (two ds_read_b64_tr_b16, 64-byte units XOR-swizzled by k & 3: the layout of gemm_bf16.hip's transposed operands).
template <bool TT>
__device__ __forceinline__ bf16x8 lds_frag(const Ctx& c, const unsigned char* half, int row, int ks, int tt_off) {
    if constexpr (!TT) {
        return *reinterpret_cast<const bf16x8*>(half + row * 128 + c.rd[ks]);
    } else {
        // hipcc waits vmcnt(0) before a ds_read_tr builtin while LDS-DMA loads are in flight (it cannot tell the two apart), which
        // would drain the prefetch ring every phase: issue the pair from inline asm and count lgkmcnt by hand (TT_LGKM_WAIT
        // after the phase's first barrier).
        // one address VGPR per 32-row sub-tile (uniform half-tile offset + per-lane part); the k-substep is an immediate offset
        const unsigned addr = (unsigned)(size_t)(lds_void_t*)half + (unsigned)tt_off;
        bf16x4 lo, hi;
        switch (ks) {
            case 0: asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory"); break;
            case 1: asm volatile("ds_read_b64_tr_b16 %0, %2 offset:4096\n\tds_read_b64_tr_b16 %1, %2 offset:5120" : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory"); break;
            case 2: asm volatile("ds_read_b64_tr_b16 %0, %2 offset:8192\n\tds_read_b64_tr_b16 %1, %2 offset:9216" : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory"); break;
            default: asm volatile("ds_read_b64_tr_b16 %0, %2 offset:12288\n\tds_read_b64_tr_b16 %1, %2 offset:13312" : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory"); break;
        }
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}
#define TT_LGKM_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// sum of the 8 bf16 values of a fragment (fp32)
__device__ __forceinline__ float frag_sum(const bf16x8& f) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const u32x4 u = __builtin_bit_cast(u32x4, f);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += __uint_as_float(u[i] << 16) + __uint_as_float(u[i] & 0xffff0000u);
    return s;
}

// Multiply K tile u (resident in buffer BI) up to the last MFMA of phase 4; the caller issues that phase's closing barrier.
// TAIL = false: K tiles u+1 and u+2 are tiles of the current output tile (no tests in the loop body).
template <int BI, bool TAIL, bool TT>
__device__ __forceinline__ void ktile(const Ctx& c, int u, f32x16 (&acc)[2][4], bf16x8 (&fa)[2][4], bf16x8 (&fb)[2][4], float (&csum)[4], int wc) {
    unsigned char* cur = c.smem + BI * BUFB;
    unsigned char* oth = c.smem + (BI ^ 1) * BUFB;
    const unsigned char* ca = cur + c.a_half;
    const unsigned char* cb = cur + c.b_half;
    const bool more1 = !TAIL || (u + 1 < c.cur.nt) || c.has_next;
    const bool more2 = !TAIL || (u + 2 < c.cur.nt) || c.has_next;

    // ---- phase 1: B sub-tiles 0/1 and A sub-tile 0 -> registers; quadrant (m 0-63, n 0-31)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fb[0][ks] = lds_frag<TT>(c, cb, c.b_row, ks, c.brd[0]);
        fb[1][ks] = lds_frag<TT>(c, cb, c.b_row + 32, ks, c.brd[1]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fa[0][ks] = lds_frag<TT>(c, ca, 0, ks, c.rd[0]);
        fa[1][ks] = lds_frag<TT>(c, ca, 32, ks, c.rd[1]);
    }
    if (!TAIL) stage_a(c, oth + 0 * HALF, c.cur, 0, u + 1);
    else if (more1) stage_stream<true>(c, oth + 0 * HALF, 0, u + 1);
    NT_PIN(); NT_BAR(); NT_PIN();
    if (TT) { TT_LGKM_WAIT(); NT_PIN(); }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[0][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][ks], fa[f][ks], acc[0][f], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (TT && c.colsum) {                        // bias gradient: wave wc owns k-substep wc of every A fragment row
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            if (ks == wc) { csum[0] += frag_sum(fa[0][ks]); csum[1] += frag_sum(fa[1][ks]); }
    }
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
        fa[0][ks] = lds_frag<TT>(c, ca, 64, ks, c.rd[2]);
        fa[1][ks] = lds_frag<TT>(c, ca, 96, ks, c.rd[3]);
    }
    if (!TAIL) stage_b(c, cur + 2 * HALF, c.cur, 0, u + 2);
    else if (more2) stage_stream<false>(c, cur + 2 * HALF, 0, u + 2);
    NT_PIN(); NT_BAR(); NT_PIN();
    if (TT) { TT_LGKM_WAIT(); NT_PIN(); }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[1][2 + f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][ks], fa[f][ks], acc[1][2 + f], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (TT && c.colsum) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            if (ks == wc) { csum[2] += frag_sum(fa[0][ks]); csum[3] += frag_sum(fa[1][ks]); }
    }
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

template <int EPI, bool TT>
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
    if constexpr (!TT) {
        const int r = tid >> 3, ch = (tid & 7) ^ ((r >> 1) & 7);      // rows r and 64 + r of a half share the swizzle
        c.ta = (unsigned)r * (unsigned)p.lda * 2u + ch * 16;
        c.tb = (unsigned)r * (unsigned)p.ldb * 2u + ch * 16;
        c.ia = 64u * (unsigned)p.lda * 2u; c.ib = 64u * (unsigned)p.ldb * 2u;
        c.ha = 2u * c.ia; c.hb = 2u * c.ib;
        c.ka = c.kb = TK * 2;
        const int l31 = lane & 31, hh = lane >> 5, sw = (l31 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c.rd[ks] = l31 * 128 + (((ks * 2 + hh) ^ sw) << 4);
    } else {
        // half-tile image [64 kc][128 r]: chunk q = kc * 16 + c', position c' holds source chunk (((c'>>2) ^ (kc&3)) << 2) | (c'&3)
        const int kc = tid >> 4, cp = tid & 15, ch = ((((cp >> 2) ^ (kc & 3))) << 2) | (cp & 3);
        c.ta = (unsigned)kc * (unsigned)p.lda * 2u + ch * 16;
        c.tb = (unsigned)kc * (unsigned)p.ldb * 2u + ch * 16;
        c.ia = 32u * (unsigned)p.lda * 2u; c.ib = 32u * (unsigned)p.ldb * 2u;
        c.ha = c.hb = 256;
        c.ka = 64u * (unsigned)p.lda * 2u; c.kb = 64u * (unsigned)p.ldb * 2u;
        const int s16 = lane & 15, g16 = (lane >> 4) & 1, hh = lane >> 5;
#pragma unroll
        for (int r5 = 0; r5 < 4; ++r5)           // sub-tile r5 of a half: lane part + the swizzled 64-byte unit
            c.rd[r5] = (8 * hh + (s16 >> 2)) * 256 + ((16 * g16 + 4 * (s16 & 3)) << 1) + ((r5 ^ ((s16 >> 2) & 3)) << 6);
    }
    c.a_half = wr * HALF;
    c.b_half = (2 + (wc >> 1)) * HALF;
    c.b_row = (wc & 1) * 64;
    c.brd[0] = (wc & 1) ? c.rd[2] : c.rd[0];
    c.brd[1] = (wc & 1) ? c.rd[3] : c.rd[1];
    float* strip = reinterpret_cast<float*>(smem + RING + wave * STRIP);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][4], fb[2][4];
    float csum[4] = {0.f, 0.f, 0.f, 0.f};

    // prologue: K tile 0 of the first output tile complete, the B half-tiles of K tile 1 in flight
    tile_desc<TT>(p, vid, c.cur);
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
        if (c.has_next) tile_desc<TT>(p, nvid, c.nxt);
        c.colsum = TT && EPI == EPI_PARTIAL && p.colsum != nullptr && c.cur.n_lo == 0;
        int u = 0;
        for (; u + 3 < c.cur.nt; u += 2) {
            ktile<0, false, TT>(c, u, acc, fa, fb, csum, wc);
            NT_BAR(); NT_PIN();
            ktile<1, false, TT>(c, u + 1, acc, fa, fb, csum, wc);
            NT_BAR(); NT_PIN();
        }
        ktile<0, true, TT>(c, u, acc, fa, fb, csum, wc);
        NT_BAR(); NT_PIN();
        ktile<1, true, TT>(c, u + 1, acc, fa, fb, csum, wc);
        // tile boundary: both m-halves run their epilogue in the barrier interval that follows the first half's last MFMA
        const int m0 = c.cur.m0s + wr * 128, n0 = c.cur.n0s + wc * 64;
        GemmParams q = p;
        q.zslice = c.cur.z;
        auto finish = [&]() {
            tile_epilogue<EPI>(q, strip, m0, n0, c.cur.m_lo, c.cur.n_lo, lane, acc);
            if constexpr (TT && EPI == EPI_PARTIAL) {
                if (c.colsum) {                  // 4 partial column sums per row (one per k-substep owner), added up by the reduce kernel
                    const size_t slice = (size_t)p.M * p.N + (size_t)p.cs_parts * p.M;
                    float* dst = p.ws + c.cur.z * slice + (size_t)p.M * p.N + (size_t)wc * p.M;
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const float v = csum[mi] + __shfl_xor(csum[mi], 32, 64);
                        const int m = m0 + 32 * mi + (lane & 31);
                        if (lane < 32 && m >= c.cur.m_lo) dst[m] = v;
                        csum[mi] = 0.f;
                    }
                }
            }
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

// ------------------------------------------------------------------------------------------------------------------------
// Weight-gradient kernel with a deep ring: C[m][n] = sum_kc A[kc*lda + m] * B[kc*ldb + n] over one slice of a very long
// contraction, both operands streamed from HBM.  What bounds this is bytes in flight per CU (latency ~2 us): five 32 KiB stages
// of [32 kc][256 m] + [32 kc][256 n] fill the whole 160 KiB of LDS, three of them in flight (96 KiB) while one is multiplied.
// 256x256 outputs per workgroup (half the operand bytes per flop of the 128x128 kernel), 8 waves of 128x64, fragments by
// ds_read_b64_tr_b16 from inline asm, two barriers per stage with the two m-halves one barrier apart (one reads fragments and
// issues loads while the other multiplies).  A stage slot is refilled two phases after its last read, the stage read next is
// retired by a counted vmcnt one phase ahead.
constexpr int DK = 32, DNS = 5;
constexpr int DOPER = DK * 256 * 2;          // 16 KiB: one operand of one stage
constexpr int DSLOT = 2 * DOPER;
constexpr int DLDS = DNS * DSLOT;            // 160 KiB

struct DCtx {
    const unsigned char* a; const unsigned char* b;
    unsigned ta, tb, ia, ib, ka, kb;
    unsigned char* smem;
    int wave_off;
    int ard[4], brd[2];                      // per-lane fragment offsets of this wave's 4 A / 2 B 32-row sub-tiles
};

__device__ __forceinline__ void dstage(const DCtx& c, int slot, int kt) {
    unsigned char* s = c.smem + slot * DSLOT + c.wave_off;
    const unsigned char* ba = c.a + (size_t)kt * c.ka;
    const unsigned char* bb = c.b + (size_t)kt * c.kb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(ba + (size_t)i * c.ia + c.ta), (lds_void_t*)(s + i * 8192), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(bb + (size_t)i * c.ib + c.tb), (lds_void_t*)(s + DOPER + i * 8192), 16, 0, 0);
    }
}
template <int KS>
__device__ __forceinline__ bf16x8 dfrag(unsigned addr) {
    bf16x4 lo, hi;
    if constexpr (KS == 0) asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory");
    else asm volatile("ds_read_b64_tr_b16 %0, %2 offset:8192\n\tds_read_b64_tr_b16 %1, %2 offset:10240" : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory");
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(NTHR) void gemm_tt_deep_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = (p.N + TN - 1) / TN;
    const int tiles = tiles_n * ((p.M + TM - 1) / TM);
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int z = vid / tiles, v = vid - z * tiles;
    const int m_lo = (v / tiles_n) * TM, n_lo = (v % tiles_n) * TN;
    const int m0s = min(m_lo, p.M - TM), n0s = min(n_lo, p.N - TN);
    const int nkt = p.K / DK, per = (nkt + p.nsplit - 1) / p.nsplit, k0 = z * per;
    const int nt = min(per, nkt - k0);
    if (nt <= 0) return;

    DCtx c;
    c.smem = smem;
    c.wave_off = wave * 1024;
    c.a = reinterpret_cast<const unsigned char*>(p.A) + ((size_t)k0 * DK * p.lda + m0s) * 2;
    c.b = reinterpret_cast<const unsigned char*>(p.B) + ((size_t)k0 * DK * p.ldb + n0s) * 2;
    {
        // operand image [32 kc][256 r] (512-byte rows): chunk q = kc * 32 + c', position c' holds source chunk
        // (((c'>>2) ^ (kc&3)) << 2) | (c'&3) -- the 64-byte-unit swizzle the transposed reads expect
        const int kc = tid >> 5, cp = tid & 31, ch = ((((cp >> 2) ^ (kc & 3))) << 2) | (cp & 3);
        c.ta = (unsigned)kc * (unsigned)p.lda * 2u + ch * 16;
        c.tb = (unsigned)kc * (unsigned)p.ldb * 2u + ch * 16;
        c.ia = 16u * (unsigned)p.lda * 2u; c.ib = 16u * (unsigned)p.ldb * 2u;
        c.ka = (unsigned)DK * (unsigned)p.lda * 2u; c.kb = (unsigned)DK * (unsigned)p.ldb * 2u;
        const int s16 = lane & 15, g16 = (lane >> 4) & 1, hh = lane >> 5, kq = (s16 >> 2) & 3;
        const int lp = (8 * hh + (s16 >> 2)) * 512 + ((16 * g16 + 4 * (s16 & 3)) << 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) c.ard[i] = lp + (((wr * 4 + i) ^ kq) << 6);
#pragma unroll
        for (int i = 0; i < 2; ++i) c.brd[i] = DOPER + lp + (((wc * 2 + i) ^ kq) << 6);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float csum = 0.f;
    const bool colsum = p.colsum != nullptr && n_lo == 0;

    // prologue: stages 0..2 in flight, stage 0 landed
    dstage(c, 0, 0);
    if (nt > 1) dstage(c, 1, 1);
    if (nt > 2) dstage(c, 2, 2);
    if (nt > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NT_PIN(); NT_BAR(); NT_PIN();
    if (wr == 1) { NT_BAR(); }
    NT_PIN();

    const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
    int slot = 0;
    for (int u = 0; u < nt; ++u) {
        const unsigned sb = lds0 + slot * DSLOT;
        bf16x8 fa[4][2], fb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { fb[i][0] = dfrag<0>(sb + c.brd[i]); fb[i][1] = dfrag<1>(sb + c.brd[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[i][0] = dfrag<0>(sb + c.ard[i]); fa[i][1] = dfrag<1>(sb + c.ard[i]); }
        const int rem = nt - 1 - u;              // stages after this one
        if (rem >= 3) {
            int s3 = slot + 3; if (s3 >= DNS) s3 -= DNS;
            dstage(c, s3, u + 3);                // slot last read two phases ago
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (rem == 2) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else if (rem == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        NT_PIN(); NT_BAR(); NT_PIN();
        TT_LGKM_WAIT(); NT_PIN();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ni][ks], fa[mi][ks], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (colsum) {                            // bias gradient: wave wc owns rows 32*wc.. of its m-half
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                if (mi == wc) csum += frag_sum(fa[mi][0]) + frag_sum(fa[mi][1]);
        }
        NT_PIN(); NT_BAR(); NT_PIN();
        if (++slot == DNS) slot = 0;
    }
    if (wr == 0) { NT_BAR(); }                   // re-align the halves: every fragment read has retired, nothing is in flight
    NT_PIN();

    GemmParams q = p;
    q.zslice = z;
    float* strip = reinterpret_cast<float*>(smem + wave * STRIP);
    const int m0 = m0s + wr * 128, n0 = n0s + wc * 64;
    tile_epilogue<EPI_PARTIAL>(q, strip, m0, n0, m_lo, n_lo, lane, acc);
    if (colsum) {
        const size_t slice = (size_t)p.M * p.N + (size_t)p.cs_parts * p.M;
        const float vsum = csum + __shfl_xor(csum, 32, 64);
        const int m = m0 + 32 * wc + (lane & 31);
        if (lane < 32 && m >= m_lo) p.ws[z * slice + (size_t)p.M * p.N + m] = vsum;
    }
}

int num_cus() {
    static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    return ncu;
}

template <int EPI, bool TT>
int launch_one(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm_nt256_kernel<EPI, TT>;
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
// PPF_GEMM_NT256 = 0 never, 1 always when legal, unset: the heuristic below.
bool nt256_eligible(const GemmParams& p, int epi) {
    static const int mode = getenv("PPF_GEMM_NT256") ? atoi(getenv("PPF_GEMM_NT256")) : -1;
    if (mode == 0) return false;
    if (!(epi == EPI_BF16 || epi == EPI_F32 || epi == EPI_GELU || epi == EPI_RESID || epi == EPI_DGELU)) return false;
    if (p.K % (2 * TK) != 0 || p.kpad) return false;                      // an even number of K tiles: buffer parity restarts per tile
    if (p.M < TM || p.N < TN || (long long)p.lda * 256 >= (1ll << 30) || (long long)p.ldb * 256 >= (1ll << 30)) return false;
    const int tn = (p.N + TN - 1) / TN;
    const long long tiles = (long long)((p.M + TM - 1) / TM) * tn;
    if (mode == 1) return true;
    return tiles >= 192 && p.K >= 768 && (long long)tn * TN * 10 <= (long long)p.N * 11;
}

int launch_nt256(const GemmParams& p_, int epi, hipStream_t stream) {
    GemmParams p = p_;
    p.nsplit = 1;
    switch (epi) {
        case EPI_BF16: return launch_one<EPI_BF16, false>(p, stream);
        case EPI_F32: return launch_one<EPI_F32, false>(p, stream);
        case EPI_GELU: return launch_one<EPI_GELU, false>(p, stream);
        case EPI_RESID: return launch_one<EPI_RESID, false>(p, stream);
        case EPI_DGELU: return launch_one<EPI_DGELU, false>(p, stream);
        default: break;
    }
    ppf_set_error("gemm_nt256: epilogue %d not instantiated", epi);
    return PPF_ERR_ARG;
}

// Weight-gradient problems (both operands contraction-strided, fp32 split-K partials): one (tile, K slice) per CU.
// Returns the slice count, or 0 when the shape does not qualify (the 128x128 kernel takes it).
int nt256_wgrad_slices(int M, int N, int K, int lda, int ldb) {
    // Measured (scripts/gpu/wgrad_check.py, profiles/r1_gemm_nt256.txt): correct and deterministic, but 5-20 % SLOWER than the
    // 128x128 kernel on the train step's weight gradients -- both operands stream from HBM (~2 us latency) and the two-K-tile
    // ring keeps only ~48 KiB per CU in flight (vs 3 x 32 KiB for three 128x128 workgroups).  Opt-in until the ring is deeper.
    static const int mode = getenv("PPF_GEMM_NT256_WGRAD") ? atoi(getenv("PPF_GEMM_NT256_WGRAD")) : 0;
    if (!mode || M < TM || N < TN || K % (2 * TK) != 0 || K < 32 * TK) return 0;
    if ((long long)lda * 64 >= (1ll << 30) || (long long)ldb * 64 >= (1ll << 30) || (lda % 8) || (ldb % 8) || (M % 8) || (N % 8)) return 0;
    const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    const int nkt = K / TK;
    int s = num_cus() / tiles;
    if (s < 1) s = 1;
    if (s > nkt / 8) s = nkt / 8;                        // >= 8 K tiles per slice
    if (s < 1) return 0;
    const int per = ((nkt + s - 1) / s + 1) & ~1;        // even K tiles per slice; drop the slices that would be empty
    return (nkt + per - 1) / per;
}

int launch_nt256_wgrad(const GemmParams& p, hipStream_t stream) { return launch_one<EPI_PARTIAL, true>(p, stream); }

// Deep-ring weight-gradient kernel (gemm_tt_deep_kernel): slices so that (tiles x slices) fills the CUs once; 0 = not eligible.
int tt_deep_slices(int M, int N, int K, int lda, int ldb) {
    // Measured (profiles/r1_gemm_nt256.txt): alone on the GPU -13 % on the fc1 / fc2 weight gradients (95 vs 109 us), equal on
    // qkv, slower on proj; inside the train step, where these GEMMs share the GPU with the dgrad chain, a workgroup that owns a
    // whole CU's LDS costs the main stream more than it saves (12.93k vs 13.02k img/s) -> opt-in (1; 2 also takes small outputs).
    static const int mode = getenv("PPF_GEMM_TT_DEEP") ? atoi(getenv("PPF_GEMM_TT_DEEP")) : 0;
    if (!mode || M < TM || N < TN || K % DK != 0 || K < 64 * DK) return 0;
    if ((long long)lda * 32 >= (1ll << 30) || (long long)ldb * 32 >= (1ll << 30) || (lda % 8) || (ldb % 8) || (M % 8) || (N % 8)) return 0;
    const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    if (tiles < 8 && mode != 2) return 0;                // few output tiles = many short slices: the 128x128 kernel is faster (measured)
    const int nkt = K / DK;
    int s = num_cus() / tiles;
    if (s > nkt / 16) s = nkt / 16;                      // >= 16 stages per slice
    if (s < 1) return 0;
    const int per = (nkt + s - 1) / s;
    return (nkt + per - 1) / per;                        // no empty slices
}

int launch_tt_deep(const GemmParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tt_deep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS);
        if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(gemm_tt_deep): %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tiles = ((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN) * p.nsplit;
    hipLaunchKernelGGL(gemm_tt_deep_kernel, dim3(tiles), dim3(NTHR), DLDS, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ppfg
