// Fused MLP forward for gfx950 (round 5):  x2 = x1 + DropPath(LayerScale(gelu(n2 W1^T + b1) W2^T + b2)),  n' = LayerNorm(x2)
// -- timm's Mlp + the residual add + the LayerNorm that follows it (deit:76-81, cait:153-157) -- in ONE launch.  The hidden layer
// h = gelu(.) and gelu'(.) are still WRITTEN (backward needs them: the fc2 weight gradient and the x gelu' input gradient) but never
// read back in the forward pass: the 155 MB (deit_small, batch 256) that the fc2 product re-reads per layer are gone, and so is a launch.
//
// STATUS: verified (tests/test_gpu_rowgemm.py::test_mlp_fwd_*, tests/test_gpu_e2e.py::test_fused_mlp_forward_matches_two_launch_path) and
// SLOWER than the two launches it replaces: 290-305 us vs 214 us per deit_small layer, -10.3 % of the train step.  The host mirror calls it
// only under PPF_MLP_FUSED=1.  Knock-out table, cost model and the code-generation traps met on the way: profiles/r5_mlp_fused.txt.
//
// Why this shape (profiles/r5_l2_shared_tile.txt): a CU's memory pipe delivers L2 hits at ~109 GB/s and anything that comes over
// the fabric (HBM / memory-side cache) at ~26 GB/s when every CU streams, and it ADDS the two; the L2 of an XCD (4 MiB for 32 CUs)
// cannot keep a 197-row activation tile per CU (156 KiB x 32) next to the weight stream.  So the activation tile lives in LDS:
//   * one 256-thread workgroup (4 waves, ONE per SIMD, up to 512 registers each) per tile of <= 112 rows (7 m-tiles; the host passes half
//     a sample), A = the tile's LayerNorm output [112][D] bf16 resident in LDS (84 KiB at D = 384), fetched ONCE;
//   * the hidden layer in chunks of 64 units: P = A W1c^T (contraction D, MFMA 16x16x32, 2 x 2 waves), bias + GELU in registers,
//     P as bf16 into a 14 KiB LDS image + h / gelu' straight to HBM (16- / 8-byte pieces, 64 / 32 contiguous bytes per row),
//     then out += P W2c^T (contraction 64; a wave owns 48 output columns in each 192-column half, all 7 m-tiles: 168 accumulator registers);
//   * skewed by one chunk: the GELU of chunk c rides on the MFMA stream of chunk c - 1's second product; h / gelu' stores leave BEHIND the
//     next stage boundary (a store in front of an s_waitcnt vmcnt(0) makes it wait for the write acknowledge);
//   * the weights stream through a double-buffered ring of 24 KiB stages (W1c as 64 rows x three 128-byte K slabs, W2c as 192 rows x
//     128 bytes: 24 one-KiB LDS-DMA pieces each, six per wave, issued in the first half of the previous stage), one s_waitcnt + barrier
//     per stage; buffer-descriptor addressing (SGPR base + 32-bit VGPR offset + SGPR offset) and a compile-time operand kind per call site;
//   * the epilogue is csrc/rowgemm.hip's RG_RESID_LN: 64 rows at a time through an fp32 LDS image, row statistics by shuffles.
// W1 rows are fetched in the order that makes a lane's two accumulator tiles 8 CONSECUTIVE hidden units (MFMA row 4q + r of tile j <->
// unit 8q + 4j + r), so h leaves as one 16-byte store per lane and m-tile and lands in the image as one ds_write_b128.
#include "gemm_common.h"
#include "ppf_hip.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int MF_NTHR = 256, MF_HC = 64, MF_MT = 7, MF_STAGE = 24576;
typedef __attribute__((address_space(3))) void mf_lds_t;
typedef const __attribute__((address_space(1))) void mf_gbl_t;

struct MlpFwdParams {
    const bf16_t* A; const bf16_t* W1; const bf16_t* W2;       // A [M][lda] (LayerNorm output), W1 [hid][ldw1], W2 [D][ldw2]
    const float* b1; const float* b2;                           // [hid], [D] (b2 may be null)
    int M, hid, lda, ldw1, ldw2, rows_per_tile;
    bf16_t* h_out; unsigned char* dg_out; int ldh;              // h = gelu(pre) bf16 [M][ldh], gelu'(pre) 8-bit codes [M][ldh]
    const float* res; float* xout; const float* rowscale; int rows_per_group; const float* colscale; bf16_t* aux_out;
    const float* ln_w; const float* ln_b; bf16_t* ln_out; float* ln_mean; float* ln_rstd; float eps;
};

struct MfNx { int kind, so; };                                  // a ring stage: kind 0 = a W1 stage, 1 = a W2 half; so = its SGPR byte offset
__device__ __forceinline__ int mf_swz(int row) { return (row >> 1) & 7; }
template <int VW> __device__ __forceinline__ void mf_ld(float (&d)[VW], const float* src) {
    const float4 t = *reinterpret_cast<const float4*>(src); d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
}

template <int D>
struct MfGeo {
    static constexpr int KS = D / 64;                  // 128-byte K slabs of the activation tile
    static constexpr int P1S = KS / 3;                 // W1 stages per chunk (three slabs each)
    static constexpr int NH = D / 192;                 // 192-row halves of W2 = W2 stages per chunk
    static constexpr int SPC = P1S + NH;               // stages per chunk
    static constexpr int A_SLAB = MF_MT * 16 * 128;    // one K slab of the activation tile
    static constexpr int A_BYTES = KS * A_SLAB;
    static constexpr int IMG_BYTES = MF_MT * 16 * 128; // the P image: [112][64] bf16
    static constexpr int OFF_IMG = A_BYTES, OFF_RING = A_BYTES + IMG_BYTES, OFF_B1 = OFF_RING + 2 * MF_STAGE;
    static constexpr int LDP = D + 4;
    static constexpr int EPI_BYTES = 64 * LDP * 4 + 3 * D * 4;
    static_assert(D == 384 || D == 192, "D = 192 or 384");
};
template <int D> constexpr int mf_lds_bytes(int hid) {
    const int main = MfGeo<D>::OFF_B1 + hid * 4;
    return main > MfGeo<D>::EPI_BYTES ? main : MfGeo<D>::EPI_BYTES;
}

template <int D>
__global__ __launch_bounds__(MF_NTHR, 1) void mlp_fwd_kernel(const MlpFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = MfGeo<D>;
    constexpr int MT = MF_MT, KS = G::KS, P1S = G::P1S, NH = G::NH, SPC = G::SPC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                               // P1 layout: m-tiles 4 wm .. 4 wm + 3, hidden units 32 wn .. 32 wn + 31 of the chunk
    const int row0 = blockIdx.x * p.rows_per_tile;
    const int rows = min(p.rows_per_tile, p.M - row0);
    const int nch = p.hid / MF_HC, nst = nch * SPC;
    unsigned char* const aT = smem;
    unsigned char* const img = smem + G::OFF_IMG;
    unsigned char* const ring = smem + G::OFF_RING;
    float* const b1s = reinterpret_cast<float*>(smem + G::OFF_B1);

    // ---- LDS-DMA source offsets (bytes, per lane): piece q = wave + 4 i (i = 0..5) of a 24-piece stage; lane l carries row (l >> 3) of
    //      the piece into LDS chunk (l & 7), reading SOURCE chunk (l & 7) ^ swz(LDS row) --------------------------------------------------
    const int lr = lane >> 3, lc = lane & 7;
    // W1 stage: sub-tile t = q >> 3 = i >> 1 (K slab), row block rb = q & 7 = wave + 4 (i & 1): LDS row R = 8 rb + lr holds hidden unit
    // 32 (R >> 5) + 8 ((R & 15) >> 2) + 4 ((R >> 4) & 1) + (R & 3) of the chunk
    unsigned offW1[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int R = 8 * (wave + 4 * e) + lr, n = R & 15;
        const int unit = 32 * (R >> 5) + 8 * (n >> 2) + 4 * ((R >> 4) & 1) + (n & 3);
        offW1[e] = (unsigned)unit * (unsigned)p.ldw1 * 2u + (unsigned)((lc ^ mf_swz(R)) << 4);
    }
    // W2 stage: LDS row R = 8 q + lr = W2 row 192 nh + R; swz(R) does not depend on i (32 i rows further)
    const unsigned offW2 = (unsigned)(8 * wave + lr) * (unsigned)p.ldw2 * 2u + (unsigned)((lc ^ mf_swz(8 * wave + lr)) << 4);
    // buffer resources (raw, byte offsets): LDS-DMA and the h / gelu' stores address memory as SGPR descriptor + one 32-bit VGPR offset +
    // an SGPR offset -- no 64-bit per-lane address arithmetic in the main loop (the flat form cost 60 VGPRs of addresses and spilled)
    const unsigned long long aW1 = (unsigned long long)p.W1, aW2 = (unsigned long long)p.W2;
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc((void*)p.h_out, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)p.dg_out, 0, -1, 0x00020000);

    // Stage order of the ring (skewed by one chunk: the GELU of chunk c hides under the second product of chunk c - 1):
    //   P1(0) | P1(1) P2(0) | P1(2) P2(1) | ... | P1(nch-1) P2(nch-2) | P2(nch-1)          P1 = P1S stages, P2 = NH stages
    // A stage is named by (kind, so): kind 0 = a W1 stage, 1 = a W2 half; so = its SGPR byte offset.  issue_piece enqueues this wave's
    // i-th (of six) one-KiB piece -- branch-free (uniform selects), so that the MFMA stream around it stays one basic block.
    auto nx_p1 = [&](int c, int s) __attribute__((always_inline)) { return MfNx{0, c * (MF_HC * 2) * p.ldw1 + s * 384}; };
    auto nx_p2 = [&](int c, int nh) __attribute__((always_inline)) { return MfNx{1, nh * 192 * 2 * p.ldw2 + c * (MF_HC * 2)}; };
    auto issue_piece = [&](unsigned char* slot, MfNx nx, int i) __attribute__((always_inline)) {
        const int dst = nx.kind ? (wave + 4 * i) * 1024 : (i >> 1) * 8192 + (wave + 4 * (i & 1)) * 1024;
        const int so = nx.so + (nx.kind ? i * 64 * p.ldw2 : (i >> 1) * 128);
        const unsigned vo = nx.kind ? offW2 : offW1[i & 1];
        // the operand is chosen by a SCALAR select of the two base addresses (a ?: between the two pointers / descriptors themselves is
        // lowered through a stack slot + flat load + s_waitcnt vmcnt(0) in front of every piece: 97 full drains of the DMA queue per chunk)
        const unsigned long long b = nx.kind ? aW2 : aW1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, -1, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (mf_lds_t*)(slot + dst), 16, vo, so, 0, 0);
    };

    // ---- prologue: the activation tile (KS slabs x 2 MT pieces; rows past the tile repeat its last row), b1, stage 0 ----------------------
    {
        const unsigned char* gA = reinterpret_cast<const unsigned char*>(p.A) + (size_t)row0 * p.lda * 2;
        for (int b = wave; b < KS * 2 * MT; b += 4) {
            const int ks = b / (2 * MT), pr = b - ks * (2 * MT);
            const int r = 8 * pr + lr;
            const unsigned char* src = gA + (size_t)min(r, rows - 1) * p.lda * 2 + ks * 128 + ((lc ^ mf_swz(r)) << 4);
            __builtin_amdgcn_global_load_lds((mf_gbl_t*)src, (mf_lds_t*)(aT + ks * G::A_SLAB + pr * 1024), 16, 0, 2);
        }
        for (int i = tid; i < p.hid; i += MF_NTHR) b1s[i] = p.b1[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // b1 is in LDS before the first barrier
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(ring, nx_p1(0, 0), i);
    }

    // fragment read offsets inside a [rows][128 B] tile: lane l reads row (l & 15), chunk 4 ks2 + (l >> 4), stored at chunk ^ swz(row)
    int frag[2];
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) frag[ks2] = (lane & 15) * 128 + (((ks2 * 4 + (lane >> 4)) ^ mf_swz(lane & 15)) << 4);

    f32x4 acc[MT][3 * NH];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 3 * NH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accP[4][2];
    uint32_t hpk[4][4], dpk[4][2];                                         // chunk c's hidden tile after GELU: bf16 pairs, gelu' codes
    ppf_float2 dhold = {0.f, 0.f};
    const int q4 = lane >> 4, r16 = lane & 15;
    // the second wave row has three m-tiles: its fourth slot re-computes the last one (no branch in the MFMA stream) and is never stored
    int mtile[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mtile[i] = min(4 * wm + i, MT - 1);
    float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bb = ba;
    // element offset of this lane's 8 hidden units in row 16 (4 wm) + r16 of the tile, chunk 0 (h: x 2 bytes, gelu' codes: x 1)
    const unsigned offST = (unsigned)(row0 + 64 * wm + r16) * (unsigned)p.ldh + (unsigned)(32 * wn + 8 * q4);

    // bias + GELU of pair u (0..15, in this order) of the chunk in accP: tile u >> 2, elements 2 (u & 1), 2 (u & 1) + 1 of sub-tile (u >> 1) & 1
    auto gelu_pair = [&](int u) __attribute__((always_inline)) {
        const int i = u >> 2, e = u & 3;
        const float bx = e == 0 ? ba.x : e == 1 ? ba.z : e == 2 ? bb.x : bb.z, by = e == 0 ? ba.y : e == 1 ? ba.w : e == 2 ? bb.y : bb.w;
        ppf_float2 g, d;
        gelu_erf_both2(ppf_float2{accP[i][e >> 1][2 * (e & 1)] + bx, accP[i][e >> 1][2 * (e & 1) + 1] + by}, g, d);
        hpk[i][e] = pack_bf16x2(g.x, g.y);
        if ((e & 1) == 0) dhold = d;
        else dpk[i][e >> 1] = ppfg::gelu8_pack4(dhold.x, dhold.y, d.x, d.y);
    };

    auto image_write = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = 16 * (4 * wm + i) + r16;                           // row inside the tile
            if (4 * wm + i < MT)
                *reinterpret_cast<uint4*>(img + m * 128 + (((4 * wn + q4) ^ mf_swz(m)) << 4)) = make_uint4(hpk[i][0], hpk[i][1], hpk[i][2], hpk[i][3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // the image writes are done before the next barrier releases the readers
    };
    // h and gelu' of chunk c to HBM, called right BEHIND a stage boundary: loads and stores share vmcnt, and a store issued in front of a
    // boundary makes its s_waitcnt vmcnt(0) wait for the write acknowledge (measured: 0.9 us per chunk)
    auto store_hd = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = 16 * (4 * wm + i) + r16;
            if (4 * wm + i < MT && m < rows) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const int so = c * MF_HC + i * 16 * p.ldh;                     // uniform part: the chunk's columns, m-tile i
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{dpk[i][0], dpk[i][1]}, rG, offST, so, 2);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{hpk[i][0], hpk[i][1], hpk[i][2], hpk[i][3]}, rH, 2 * offST, 2 * so, 0);
            }
        }
    };

    int S = 0;                                                             // the stage being multiplied: slot S & 1
    // ---- P(c) = A W1c^T: a stage = six k-steps of 8 MFMAs, fragments one k-step ahead, one next-stage piece per k-step; `after` names
    //      the stage that follows this chunk's last W1 stage -----------------------------------------------------------------------------
    auto p1_block = [&](int c, MfNx after, int store_c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { accP[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accP[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < P1S; ++s, ++S) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned char* slot = ring + (S & 1) * MF_STAGE;
            unsigned char* nslot = ring + ((S + 1) & 1) * MF_STAGE;
            const MfNx nx = s + 1 < P1S ? nx_p1(c, s + 1) : after;
            if (s == 0 && store_c >= 0) store_hd(store_c);
            bf16x8 fb[2][2], fa[2][4];
            auto load = [&](int buf, int ks) __attribute__((always_inline)) {                              // fragments of k-step ks (0..5) of this stage
                const int t = ks >> 1, ks2 = ks & 1;
                fb[buf][0] = *reinterpret_cast<const bf16x8*>(slot + t * 8192 + (32 * wn) * 128 + frag[ks2]);
                fb[buf][1] = *reinterpret_cast<const bf16x8*>(slot + t * 8192 + (32 * wn + 16) * 128 + frag[ks2]);
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[buf][i] = *reinterpret_cast<const bf16x8*>(aT + (s * 3 + t) * G::A_SLAB + mtile[i] * 2048 + frag[ks2]);
            };
            load(0, 0);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                if (ks < 3) { issue_piece(nslot, nx, 2 * ks); issue_piece(nslot, nx, 2 * ks + 1); }   // early: a piece issued late is waited for at the boundary
                if (ks + 1 < 6) load((ks + 1) & 1, ks + 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    accP[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks & 1][0], fa[ks & 1][i], accP[i][0], 0, 0, 0);
                    accP[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks & 1][1], fa[ks & 1][i], accP[i][1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int u0 = c * MF_HC + 32 * wn + 8 * q4;                      // this lane's 8 consecutive hidden units
        ba = *reinterpret_cast<const float4*>(b1s + u0); bb = *reinterpret_cast<const float4*>(b1s + u0 + 4);
    };
    // ---- out += P(cp) W2^T from the image; with GELU the 16 pairs of the chunk in accP ride on the MFMA stream (NH x 14 regions) ----------
    auto p2_block = [&](int cp, const bool GELU, MfNx after, int store_c) __attribute__((always_inline)) {
#pragma unroll
        for (int nh = 0; nh < NH; ++nh, ++S) {
            constexpr int PER = 16 / NH, NR = 2 * MT;                        // GELU pairs of this stage, regions (k-step, m-tile)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned char* slot = ring + (S & 1) * MF_STAGE;
            unsigned char* nslot = ring + ((S + 1) & 1) * MF_STAGE;
            const MfNx nx = nh + 1 < NH ? nx_p2(cp, nh + 1) : after;
            if (nh == 0 && store_c >= 0) store_hd(store_c);
            bf16x8 fb[2][3], fa[2];
#pragma unroll
            for (int j = 0; j < 3; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(slot + (48 * wave + 16 * j) * 128 + frag[0]);
            fa[0] = *reinterpret_cast<const bf16x8*>(img + frag[0]);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                __builtin_amdgcn_sched_barrier(0);
                if (r < 6) issue_piece(nslot, nx, r);
                if (r == MT - 3) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) fb[1][j] = *reinterpret_cast<const bf16x8*>(slot + (48 * wave + 16 * j) * 128 + frag[1]);
                }
                if (r + 1 < NR) fa[(r + 1) & 1] = *reinterpret_cast<const bf16x8*>(img + ((r + 1) % MT) * 2048 + frag[(r + 1) / MT]);
                if (GELU && r < PER) gelu_pair(nh * PER + r);
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[r % MT][3 * nh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[r / MT][j], fa[r & 1], acc[r % MT][3 * nh + j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (GELU) {
#pragma unroll
                for (int u = NR; u < PER; ++u) gelu_pair(nh * PER + u);      // NH = 1: pairs 14, 15
            }
        }
    };
    // ---- chunk c: P (bf16) into the image, h and gelu' to HBM -----------------------------------------------------------------------
    // every `after` below has a compile-time kind (the blocks are inlined at each call site, the operand selects in issue_piece fold away);
    // nch >= 2 (ppf_mlp_fwd_supported)
    p1_block(0, nx_p1(1, 0), -1);
#pragma unroll
    for (int u = 0; u < 16; ++u) gelu_pair(u);                               // chunk 0 has no second product to hide under
    image_write(0);
    for (int c = 1; c + 1 < nch; ++c) {
        p1_block(c, nx_p2(c - 1, 0), c - 1);
        p2_block(c - 1, true, nx_p1(c + 1, 0), -1);
        __builtin_amdgcn_s_barrier();                                        // every wave is done with the image of chunk c - 1
        image_write(c);
    }
    p1_block(nch - 1, nx_p2(nch - 2, 0), nch - 2);
    p2_block(nch - 2, true, nx_p2(nch - 1, 0), -1);
    __builtin_amdgcn_s_barrier();
    image_write(nch - 1);
    p2_block(nch - 1, false, nx_p2(nch - 1, NH - 1), nch - 1);               // (the piece stream ends in a harmless re-fetch of the last stage)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // the last h / gelu' stores; nothing of the ring is in flight
    __builtin_amdgcn_s_barrier();                                              // every fragment read is done: LDS becomes the epilogue's image

    // ---- epilogue (csrc/rowgemm.hip RG_RESID_LN): 64 rows at a time through an fp32 image; 32 lanes own one row in 16-byte pieces ------
    constexpr int LDP = G::LDP;
    float* eimg = reinterpret_cast<float*>(smem);                              // [64][LDP]
    float* stash = eimg + 64 * LDP;                                            // [3][D]: b2 | ln_w | ln_b
    for (int i = tid; i < D; i += MF_NTHR) {
        stash[i] = p.b2 ? p.b2[i] : 0.f;
        stash[D + i] = p.ln_out ? p.ln_w[i] : 0.f; stash[2 * D + i] = p.ln_out ? p.ln_b[i] : 0.f;
    }
    constexpr int VW = 4, NSEG = 3, PIECE = D / NSEG, LPR = PIECE / VW, RPW = 64 / LPR;      // D = 384: 32 lanes per row, 2 rows per wave at a time
    const int jl = lane % LPR, rsub = lane / LPR;
    const float invD = 1.0f / (float)D;
    constexpr int NCH = (MT + 3) / 4;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch * 64 >= rows) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if ((i >> 2) == ch) {
#pragma unroll
                for (int j = 0; j < 3 * NH; ++j)
                    *reinterpret_cast<f32x4*>(eimg + ((i & 3) * 16 + r16) * LDP + 192 * (j / 3) + 48 * wave + 16 * (j % 3) + 4 * q4) = acc[i][j];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int g = 0; g < 16 / RPW; ++g) {
            const int r = wave * 16 + g * RPW + rsub;
            const int rt = ch * 64 + r;
            const bool ok = rt < rows;
            const size_t m = (size_t)row0 + (ok ? rt : 0);
            float v[NSEG][VW];
#pragma unroll
            for (int i = 0; i < NSEG; ++i) {
                const int col = VW * jl + PIECE * i;
                mf_ld<VW>(v[i], eimg + r * LDP + col);
                float bv[VW];
                mf_ld<VW>(bv, stash + col);
#pragma unroll
                for (int e = 0; e < VW; ++e) v[i][e] += bv[e];
            }
            const float rsc = (ok && p.rowscale) ? p.rowscale[m / p.rows_per_group] : 1.0f;
            float ra[NSEG][VW];
#pragma unroll
            for (int i = 0; i < NSEG; ++i) {
                if (ok) mf_ld<VW>(ra[i], p.res + m * D + VW * jl + PIECE * i);
                else {
#pragma unroll
                    for (int e = 0; e < VW; ++e) ra[i][e] = 0.f;
                }
            }
            if (p.aux_out && ok) {
#pragma unroll
                for (int i = 0; i < NSEG; ++i)
                    *reinterpret_cast<uint2*>(p.aux_out + m * D + VW * jl + PIECE * i) = make_uint2(pack_bf16x2(v[i][0], v[i][1]), pack_bf16x2(v[i][2], v[i][3]));
            }
            if (p.colscale) {
#pragma unroll
                for (int i = 0; i < NSEG; ++i) {
                    float cs[VW];
                    mf_ld<VW>(cs, p.colscale + VW * jl + PIECE * i);
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
                for (int i = 0; i < NSEG; ++i) *reinterpret_cast<float4*>(p.xout + m * D + VW * jl + PIECE * i) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
            }
            if (p.ln_out) {
#pragma unroll
                for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                const float mu = s * invD;
                float qq = 0.f;
#pragma unroll
                for (int i = 0; i < NSEG; ++i)
#pragma unroll
                    for (int e = 0; e < VW; ++e) { const float d = v[i][e] - mu; qq += d * d; }
#pragma unroll
                for (int o = LPR / 2; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 64);
                const float rs = rsqrtf(qq * invD + p.eps);
                if (ok) {
#pragma unroll
                    for (int i = 0; i < NSEG; ++i) {
                        const int col = VW * jl + PIECE * i;
                        float lw[VW], lb[VW];
                        mf_ld<VW>(lw, stash + D + col); mf_ld<VW>(lb, stash + 2 * D + col);
                        *reinterpret_cast<uint2*>(p.ln_out + m * D + col) =
                            make_uint2(pack_bf16x2((v[i][0] - mu) * rs * lw[0] + lb[0], (v[i][1] - mu) * rs * lw[1] + lb[1]),
                                       pack_bf16x2((v[i][2] - mu) * rs * lw[2] + lb[2], (v[i][3] - mu) * rs * lw[3] + lb[3]));
                    }
                    if (jl == 0) { p.ln_mean[m] = mu; p.ln_rstd[m] = rs; }
                }
            }
        }
    }
}

template <int D>
int mf_launch(const MlpFwdParams& p, hipStream_t stream) {
    const int lds = mf_lds_bytes<D>(p.hid);
    auto kern = mlp_fwd_kernel<D>;
    static int attr_lds = 0;
    if (attr_lds < lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { ppf_set_error("hipFuncSetAttribute(mlp_fwd): %s", hipGetErrorString(e)); return (int)e; }
        attr_lds = lds;
    }
    const int tiles = (p.M + p.rows_per_tile - 1) / p.rows_per_tile;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(MF_NTHR), lds, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

// 1 when ppf_mlp_fwd takes this shape: D in {192, 384}, hid a multiple of 64 whose bias fits the LDS next to the tile, tiles of <= 112 rows.
int ppf_mlp_fwd_supported(int D, int hid, int rows_per_tile) {
    if (!(D == 384 || D == 192) || hid < 128 || hid % 64 != 0 || rows_per_tile < 1 || rows_per_tile > 16 * MF_MT) return 0;
    const int lds = D == 384 ? mf_lds_bytes<384>(hid) : mf_lds_bytes<192>(hid);
    return lds <= 160 * 1024 ? 1 : 0;
}

// timm Mlp + residual + the following LayerNorm (deit:76-81; cait:153-157 with colscale = gamma_2) in one launch:
//   h = gelu(A W1^T + b1) (bf16 [M][hid], written for backward), dgelu = gelu'(A W1^T + b1) as 8-bit codes (gemm_common.h gelu8_*),
//   xout = res + rowscale[m / rows_per_group] * colscale[n] * (h W2^T + b2)  (fp32, may alias res),  aux_out (optional) = bf16(h W2^T + b2),
//   ln_out = bf16(LN(xout) * ln_w + ln_b), ln_mean, ln_rstd (ln_out == NULL: no LayerNorm).
int ppf_mlp_fwd(const void* A, const void* W1, const float* b1, const void* W2, const float* b2, int M, int D, int hid, int rows_per_tile,
                void* h_out, void* dgelu_out, const float* res, float* xout, const float* rowscale, int rows_per_group, const float* colscale,
                void* aux_out, const float* ln_w, const float* ln_b, void* ln_out, float* ln_mean, float* ln_rstd, float eps, hipStream_t stream) {
    PPF_CHECK_ARG(A && W1 && b1 && W2 && h_out && dgelu_out && res && xout, PPF_ERR_ARG, "ppf_mlp_fwd: null pointer");
    PPF_CHECK_ARG(M > 0 && ppf_mlp_fwd_supported(D, hid, rows_per_tile), PPF_ERR_SHAPE,
                  "ppf_mlp_fwd: unsupported shape M=%d D=%d hid=%d rows_per_tile=%d (D in {192, 384}, hid %% 64 == 0, rows_per_tile <= 112)", M, D, hid, rows_per_tile);
    PPF_CHECK_ARG(((((uintptr_t)A) | ((uintptr_t)W1) | ((uintptr_t)W2) | ((uintptr_t)h_out) | ((uintptr_t)dgelu_out) | ((uintptr_t)res) | ((uintptr_t)xout) |
                    ((uintptr_t)ln_out) | ((uintptr_t)aux_out) | ((uintptr_t)colscale) | ((uintptr_t)b1)) & 15) == 0, PPF_ERR_ALIGN, "ppf_mlp_fwd: misaligned pointer");
    PPF_CHECK_ARG(ln_out == nullptr || (ln_w && ln_b && ln_mean && ln_rstd), PPF_ERR_ARG, "ppf_mlp_fwd: LayerNorm output needs weight, bias, mean and rstd");
    PPF_CHECK_ARG((long long)hid * D * 2 < (1ll << 31), PPF_ERR_SHAPE, "ppf_mlp_fwd: weight matrices beyond 2 GiB");
    MlpFwdParams p = {};
    p.A = (const bf16_t*)A; p.W1 = (const bf16_t*)W1; p.W2 = (const bf16_t*)W2; p.b1 = b1; p.b2 = b2;
    p.M = M; p.hid = hid; p.lda = D; p.ldw1 = D; p.ldw2 = hid; p.rows_per_tile = rows_per_tile;
    p.h_out = (bf16_t*)h_out; p.dg_out = (unsigned char*)dgelu_out; p.ldh = hid;
    p.res = res; p.xout = xout; p.rowscale = rowscale; p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1; p.colscale = colscale;
    p.aux_out = (bf16_t*)aux_out; p.ln_w = ln_w; p.ln_b = ln_b; p.ln_out = (bf16_t*)ln_out; p.ln_mean = ln_mean; p.ln_rstd = ln_rstd; p.eps = eps;
    return D == 384 ? mf_launch<384>(p, stream) : mf_launch<192>(p, stream);
}

}  // extern "C"
