// Operand-delivery micro-benchmark for the GEMM access pattern (round 5): what does the L2 -> LDS path deliver when
// every CU streams K-stages of 128-byte row segments by global_load_lds_dwordx4 -- private activation rows (pitch = the matrix row
// pitch, read once: HBM) next to a weight tile that ALL workgroups read (L2) -- and which property of that pattern costs the 3x
// between the 10 TB/s the GEMMs draw and the 31 TB/s of profiles/r2_l2_bandwidth.txt?  No MFMA: the loop is issue -> counted wait ->
// barrier, i.e. the GEMM main loop with the arithmetic knocked out.
//
// build: hipcc --offload-arch=gfx950 -O3 l2tile.hip -o l2tile ; run: ./l2tile [group]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void lds_t;
typedef const __attribute__((address_space(1))) void gbl_t;

struct TP {
    const unsigned char* A; long long pitchA; int rowsA;        // private rows per workgroup tile (multiple of 8), tile t of workgroup w = rows [(w*ntiles+t)*rowsA, ..)
    const unsigned char* B; long long pitchB; int rowsB;        // weight rows (multiple of 8), shared by all workgroups unless privB
    int nk, ntiles;                                             // 128-byte K stages per tile, tiles per workgroup
    int rot;                                                    // 0: same order everywhere; 1: B piece rotation per workgroup; 2: K-stage rotation per workgroup; 3: both
    int privB;                                                  // 1: every workgroup has its own B rows (B + w * rowsB * pitchB)
    int a_nt;                                                   // 1: A pieces with the nt policy
    int sleep;                                                  // s_sleep units per stage in place of the MFMA stream (0 = none)
    int seg;                                                    // > 0: waves [0, seg) issue ONLY the private (A) pieces, waves [seg, NW) only the shared (B) pieces (NS = 1)
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NT threads, NS stages in flight (1 = issue, drain, barrier: the double-buffered GEMM loop whose arithmetic is free), PPW pieces per
// wave and stage (piece = 1 KiB = 8 rows x 128 B).  Nothing consumes the data, so stages beyond the LDS capacity alias older slots.
template <int NT, int NS, int PPW>
__global__ __launch_bounds__(NT) void tile_stream_kernel(const TP p, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NT / 64, NP = NW * PPW, STAGE = NP * 1024;
    constexpr int SL = NS * STAGE <= 160 * 1024 / (NT == 256 ? 2 : 1) ? NS : (160 * 1024 / (NT == 256 ? 2 : 1)) / STAGE;      // LDS slots
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int npa = p.rowsA / 8, npb = p.rowsB / 8;                // npa + npb == NP (host guarantees)
    const int w = blockIdx.x;
    const int rotp = (p.rot & 1) && npb > 0 ? (int)((w * 7u) % (unsigned)npb) : 0;
    const int rotk = (p.rot & 2) ? (int)((w * 5u) % (unsigned)p.nk) : 0;
    const unsigned char* gB = p.B + (p.privB ? (long long)w * p.rowsB * p.pitchB : 0);
    long long offA[PPW], offB[PPW];                                  // per-lane source offsets of this wave's pieces (A: relative to the tile)
    int dst[PPW]; bool isA[PPW];
    bool live[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        int b = wave + NW * i;
        live[i] = b < npa + npb;
        if (!live[i]) b = 0;
        if (p.seg > 0) {                                              // segregated roles: a wave's i-th piece comes from its own operand only
            if (wave < p.seg) { b = wave + p.seg * i; live[i] = b < npa; }
            else { b = npa + (wave - p.seg) + (NW - p.seg) * i; live[i] = b < npa + npb; }
            if (!live[i]) b = 0;
        }
        dst[i] = b * 1024;
        isA[i] = b < npa;
        if (b < npa) { offA[i] = (long long)(b * 8 + (lane >> 3)) * p.pitchA + (lane & 7) * 16; offB[i] = 0; }
        else { int bb = b - npa + rotp; bb -= bb >= npb ? npb : 0; offB[i] = (long long)(bb * 8 + (lane >> 3)) * p.pitchB + (lane & 7) * 16; offA[i] = 0; }
    }
    const int total = p.ntiles * p.nk;
    auto issue = [&](int s) {
        const int t = s / p.nk;
        int k = s - t * p.nk + rotk; k -= k >= p.nk ? p.nk : 0;
        const unsigned char* gA = p.A + ((long long)w * p.ntiles + t) * p.rowsA * p.pitchA + (long long)k * 128;
        const unsigned char* gBk = gB + (long long)k * 128;
        unsigned char* base = NS == 1 ? smem : smem + (s % (SL > 0 ? SL : 1)) * STAGE;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (!live[i]) continue;
            if (isA[i]) {
                if (p.a_nt) __builtin_amdgcn_global_load_lds((gbl_t*)(gA + offA[i]), (lds_t*)(base + dst[i]), 16, 0, 2);
                else __builtin_amdgcn_global_load_lds((gbl_t*)(gA + offA[i]), (lds_t*)(base + dst[i]), 16, 0, 0);
            } else __builtin_amdgcn_global_load_lds((gbl_t*)(gBk + offB[i]), (lds_t*)(base + dst[i]), 16, 0, 0);
        }
    };
    for (int s = 0; s < NS - 1 && s < total; ++s) issue(s);
    for (int s = 0; s < total; ++s) {
        if (s + NS - 1 < total) { issue(s + NS - 1); wait_vm<(NS - 1) * PPW>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                                // stage s has landed for everybody
        if (p.sleep) { for (int q = 0; q < p.sleep; ++q) __builtin_amdgcn_s_sleep(16); }
        __builtin_amdgcn_s_barrier();                                // everybody is done with stage s: its slot may be refilled
    }
    if (smem[threadIdx.x * 16] == 0x5a && p.nk < 0) sink[0] = 1;
}

static unsigned char *dA, *dB; static unsigned* dsink;
static size_t bytesA = (size_t)3 << 30, bytesB = (size_t)640 << 20;

template <int NT, int NS, int PPW>
static double run(TP p, int grid, int iters = 6) {
    constexpr int STAGE = (NT / 64) * PPW * 1024, CAP = 160 * 1024 / (NT == 256 ? 2 : 1);
    constexpr int lds = NS == 1 ? (STAGE < CAP ? STAGE : CAP) : (NS * STAGE <= CAP ? NS : CAP / STAGE) * STAGE;
    auto kern = tile_stream_kernel<NT, NS, PPW>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    size_t needA = (size_t)grid * p.ntiles * p.rowsA * p.pitchA, needB = (size_t)(p.privB ? grid : 1) * p.rowsB * p.pitchB;
    if (needA > bytesA || needB > bytesB) { fprintf(stderr, "buffer too small (%zu / %zu)\n", needA, needB); return -1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> t;
    for (int it = 0; it < iters + 2; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, p, dsink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) t.push_back(ms);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fprintf(stderr, "launch: %s\n", hipGetErrorString(e)); return -1; }
    std::sort(t.begin(), t.end());
    hipEventDestroy(e0); hipEventDestroy(e1);
    return t[t.size() / 2] * 1e-3;                                   // median seconds per launch
}

static void report(const char* name, const TP& p, int grid, int wgs_per_cu, double sec) {
    const double stage = (double)(p.rowsA + p.rowsB) * 128.0, bytes = stage * p.nk * p.ntiles * grid;
    const double per_stage_us = sec * 1e6 / ((double)p.nk * p.ntiles * ((grid + 256 * wgs_per_cu - 1) / (256 * wgs_per_cu)));
    printf("%-78s %7.1f us  %6.2f TB/s  %6.1f GB/s/CU  %5.2f us/stage(%3.0f KiB)\n", name, sec * 1e6, bytes / sec / 1e12, bytes / sec / 256 / 1e9, per_stage_us,
           stage / 1024);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const char* grp = argc > 1 ? argv[1] : "all";
    auto on = [&](const char* g) { return !strcmp(grp, "all") || !strcmp(grp, g); };
    hipMalloc(&dA, bytesA); hipMalloc(&dB, bytesB); hipMalloc(&dsink, 64);
    hipMemset(dA, 1, bytesA); hipMemset(dB, 2, bytesB);
    hipDeviceSynchronize();
    char name[256];
    const long long pitches[] = {768, 832, 896, 1024, 1536, 1664, 2304, 2432, 3072, 3200, 4096, 4224};
    auto nk_of = [](long long pitch) { return (int)std::min<long long>(pitch / 128, 24); };

    if (on("A")) {
        printf("== A only: 256 private rows per workgroup and stage (32 KiB), read once (HBM), one 512-thread workgroup per CU; 1 / 2 / 4 stages in flight ==\n");
        for (long long pitch : pitches) {
            TP p = {}; p.A = dA; p.pitchA = pitch; p.rowsA = 256; p.B = dB; p.pitchB = 768; p.rowsB = 0;
            p.nk = nk_of(pitch); p.ntiles = std::max(1, 96 / p.nk);
            snprintf(name, sizeof name, "A pitch %4lld B  nk %2d x %2d tiles  1 in flight", pitch, p.nk, p.ntiles);
            report(name, p, 256, 1, run<512, 1, 4>(p, 256));
            snprintf(name, sizeof name, "A pitch %4lld B  nk %2d x %2d tiles  2 in flight", pitch, p.nk, p.ntiles);
            report(name, p, 256, 1, run<512, 2, 4>(p, 256));
            snprintf(name, sizeof name, "A pitch %4lld B  nk %2d x %2d tiles  4 in flight", pitch, p.nk, p.ntiles);
            report(name, p, 256, 1, run<512, 4, 4>(p, 256));
        }
        printf("== the same with 2 x 256-thread workgroups per CU (128 rows = 16 KiB per stage each) ==\n");
        for (long long pitch : {768LL, 3072LL, 3200LL}) {
            TP p = {}; p.A = dA; p.pitchA = pitch; p.rowsA = 128; p.B = dB; p.pitchB = 768; p.rowsB = 0;
            p.nk = nk_of(pitch); p.ntiles = std::max(1, 96 / p.nk);
            snprintf(name, sizeof name, "A pitch %4lld B  2 WG/CU x 256 thr  1 in flight", pitch);
            report(name, p, 512, 2, run<256, 1, 4>(p, 512));
            snprintf(name, sizeof name, "A pitch %4lld B  2 WG/CU x 256 thr  4 in flight", pitch);
            report(name, p, 512, 2, run<256, 4, 4>(p, 512));
        }
    }
    if (on("B")) {
        printf("== B only: 384 weight rows x 128 B per stage (48 KiB), every workgroup reads the SAME rows (L2), one 512-thread workgroup per CU ==\n");
        for (long long pitch : {768LL, 832LL, 3072LL, 3200LL}) {
            for (int rot = 0; rot < 4; ++rot) {
                TP p = {}; p.A = dA; p.pitchA = 768; p.rowsA = 0; p.B = dB; p.pitchB = pitch; p.rowsB = 384; p.rot = rot;
                p.nk = nk_of(pitch); p.ntiles = 96 / p.nk;
                snprintf(name, sizeof name, "B shared pitch %4lld B  rot %d (1 = piece order, 2 = K stage, 3 = both)  1 in flight", pitch, rot);
                report(name, p, 256, 1, run<512, 1, 6>(p, 256));
            }
            for (int rot : {0, 2}) {
                TP p = {}; p.A = dA; p.pitchA = 768; p.rowsA = 0; p.B = dB; p.pitchB = pitch; p.rowsB = 384; p.rot = rot;
                p.nk = nk_of(pitch); p.ntiles = 96 / p.nk;
                snprintf(name, sizeof name, "B shared pitch %4lld B  rot %d  2 in flight", pitch, rot);
                report(name, p, 256, 1, run<512, 2, 6>(p, 256));
                snprintf(name, sizeof name, "B shared pitch %4lld B  rot %d  3 in flight", pitch, rot);
                report(name, p, 256, 1, run<512, 3, 6>(p, 256));
            }
            TP p = {}; p.A = dA; p.pitchA = 768; p.rowsA = 0; p.B = dB; p.pitchB = pitch; p.rowsB = 384;
            p.nk = nk_of(pitch); p.ntiles = 96 / p.nk; p.privB = 1;
            snprintf(name, sizeof name, "B PRIVATE (own rows per workgroup: MALL / HBM) pitch %4lld B  1 in flight", pitch);
            report(name, p, 256, 1, run<512, 1, 6>(p, 256));
            snprintf(name, sizeof name, "B PRIVATE pitch %4lld B  3 in flight", pitch);
            report(name, p, 256, 1, run<512, 3, 6>(p, 256));
        }
        printf("== B shared, 4 x 256-thread workgroups per CU (the 128x128 GEMM's B tile: 128 rows = 16 KiB per stage) ==\n");
        for (long long pitch : {768LL, 3072LL}) {
            for (int rot : {0, 2}) {
                TP p = {}; p.A = dA; p.pitchA = 768; p.rowsA = 0; p.B = dB; p.pitchB = pitch; p.rowsB = 128; p.rot = rot;
                p.nk = nk_of(pitch); p.ntiles = 96 / p.nk;
                snprintf(name, sizeof name, "B shared 128 rows pitch %4lld B  rot %d  4 WG/CU  1 in flight", pitch, rot);
                report(name, p, 1024, 4, run<256, 1, 4>(p, 1024));
            }
        }
    }
    if (on("AB")) {
        printf("== A + B: the full-row GEMM stage (192 private A rows + 384 shared B rows = 72 KiB), K = 384 (pitch 768) and K = 1536 (pitch 3072) ==\n");
        for (long long pitch : {768LL, 832LL, 3072LL, 3200LL}) {
            for (int rot : {0, 3}) {
                for (int nt : {0, 1}) {
                    TP p = {}; p.A = dA; p.pitchA = pitch; p.rowsA = 192; p.B = dB; p.pitchB = pitch; p.rowsB = 384; p.rot = rot; p.a_nt = nt;
                    p.nk = nk_of(pitch); p.ntiles = 96 / p.nk;
                    snprintf(name, sizeof name, "A+B pitch %4lld B  rot %d  A nt %d  1 in flight", pitch, rot, nt);
                    report(name, p, 256, 1, run<512, 1, 9>(p, 256));
                }
            }
            TP p = {}; p.A = dA; p.pitchA = pitch; p.rowsA = 192; p.B = dB; p.pitchB = pitch; p.rowsB = 384; p.rot = 3; p.a_nt = 1;
            p.nk = nk_of(pitch); p.ntiles = 96 / p.nk;
            snprintf(name, sizeof name, "A+B pitch %4lld B  rot 3  A nt 1  2 in flight", pitch);
            report(name, p, 256, 1, run<512, 2, 9>(p, 256));
            snprintf(name, sizeof name, "A+B pitch %4lld B  rot 3  A nt 1  3 in flight (slots alias)", pitch);
            report(name, p, 256, 1, run<512, 3, 9>(p, 256));
        }
        printf("== A + B, 128x128 tile (128 private A rows + 128 shared B rows = 32 KiB per stage), 4 x 256-thread workgroups per CU ==\n");
        for (long long pitch : {768LL, 832LL, 3072LL, 3200LL}) {
            TP p = {}; p.A = dA; p.pitchA = pitch; p.rowsA = 128; p.B = dB; p.pitchB = pitch; p.rowsB = 128; p.rot = 0;
            p.nk = nk_of(pitch); p.ntiles = 96 / p.nk;
            snprintf(name, sizeof name, "A+B 128x128 pitch %4lld B  4 WG/CU  1 in flight", pitch);
            report(name, p, 1024, 4, run<256, 1, 8>(p, 1024));
        }
    }
    if (on("sleep")) {
        printf("== A + B full-row stage with an s_sleep gap (units of 1024 clocks) per stage in place of the MFMA stream ==\n");
        for (int sl : {0, 1, 2, 3, 4}) {
            TP p = {}; p.A = dA; p.pitchA = 3072; p.rowsA = 192; p.B = dB; p.pitchB = 3072; p.rowsB = 384; p.rot = 3; p.a_nt = 1; p.sleep = sl;
            p.nk = 24; p.ntiles = 4;
            snprintf(name, sizeof name, "A+B pitch 3072  sleep %d  1 in flight", sl);
            report(name, p, 256, 1, run<512, 1, 9>(p, 256));
            snprintf(name, sizeof name, "A+B pitch 3072  sleep %d  2 in flight", sl);
            report(name, p, 256, 1, run<512, 2, 9>(p, 256));
        }
    }
    if (on("scale")) {
        printf("== how the per-CU rate depends on how many CUs stream (one 512-thread workgroup per CU, 2 stages in flight) ==\n");
        for (int grid : {8, 32, 64, 128, 256}) {
            TP p = {}; p.A = dA; p.pitchA = 3072; p.rowsA = 256; p.B = dB; p.pitchB = 3072; p.rowsB = 0; p.nk = 24; p.ntiles = 4;
            snprintf(name, sizeof name, "A only (HBM), %3d workgroups", grid);
            const double sec = run<512, 2, 4>(p, grid);
            printf("%-78s %7.1f us  %6.2f TB/s  %6.1f GB/s per streaming CU\n", name, sec * 1e6, 32768.0 * 96 * grid / sec / 1e12, 32768.0 * 96 / sec / 1e9);
            TP q = {}; q.A = dA; q.pitchA = 768; q.rowsA = 0; q.B = dB; q.pitchB = 3072; q.rowsB = 384; q.nk = 24; q.ntiles = 4;
            snprintf(name, sizeof name, "B only (L2, shared), %3d workgroups", grid);
            const double sec2 = run<512, 2, 6>(q, grid);
            printf("%-78s %7.1f us  %6.2f TB/s  %6.1f GB/s per streaming CU\n", name, sec2 * 1e6, 49152.0 * 96 * grid / sec2 / 1e12, 49152.0 * 96 / sec2 / 1e9);
        }
    }
    if (on("mix")) {
        printf("== 72 KiB stages with a varying share of private (HBM) rows, pitch 3072, rot 3, A nt, 2 stages in flight: do the two sources add or overlap? ==\n");
        for (int ra : {0, 64, 128, 192, 288, 384, 576}) {
            TP p = {}; p.A = dA; p.pitchA = 3072; p.rowsA = ra; p.B = dB; p.pitchB = 3072; p.rowsB = 576 - ra; p.rot = 3; p.a_nt = 1; p.nk = 24; p.ntiles = 4;
            snprintf(name, sizeof name, "A %3d private rows (%2d KiB) + B %3d shared rows (%2d KiB)", ra, ra / 8, 576 - ra, (576 - ra) / 8);
            report(name, p, 256, 1, run<512, 2, 9>(p, 256));
        }
    }
    if (on("seg")) {
        printf("== does the HBM-sourced operand block the L2-sourced one because loads return IN ORDER per wave?  192 private A rows (24 KiB, HBM, nt) + 384 shared B rows\n"
               "   (48 KiB, L2) per stage, one stage in flight (issue, drain, barrier); mixed = every wave issues both kinds, seg w = waves [0, w) issue only A, the rest only B ==\n");
        for (int seg : {0, 1, 2, 3, 4}) {
            TP p = {}; p.A = dA; p.pitchA = 3072; p.rowsA = 192; p.B = dB; p.pitchB = 3072; p.rowsB = 384; p.rot = 3; p.a_nt = 1; p.nk = 24; p.ntiles = 4; p.seg = seg;
            snprintf(name, sizeof name, seg ? "segregated: %d wave(s) stream A, %d stream B" : "mixed (every wave issues A and B pieces)", seg, 8 - seg);
            report(name, p, 256, 1, run<512, 1, 24>(p, 256));
        }
        for (int seg : {0, 2}) {
            TP p = {}; p.A = dA; p.pitchA = 3072; p.rowsA = 64; p.B = dB; p.pitchB = 3072; p.rowsB = 512; p.rot = 3; p.a_nt = 1; p.nk = 24; p.ntiles = 4; p.seg = seg;
            snprintf(name, sizeof name, seg ? "A 64 rows (8 KiB) + B 512 rows: segregated, %d wave(s) stream A" : "A 64 rows (8 KiB) + B 512 rows: mixed", seg);
            report(name, p, 256, 1, run<512, 1, 32>(p, 256));
        }
    }
    return 0;
}
