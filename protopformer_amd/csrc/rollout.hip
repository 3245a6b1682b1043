// Attention rollout + token reservation (tools/deit_models_attn.py:99-124,223-234; cait:223-261,328-339).
//
// One 1024-thread workgroup per sample walks the layers from the LAST to the first and only propagates the
// row that is consumed downstream (row 0 of a_{L-1}...a_0 for DeiT, cls_row . a_{L-1}...a_0 for CaiT):
//     r <- r . a_l,   a_l = rownorm((discard90(f_l) + 0.2 I) / 1.2)
// which is 197x fewer FLOPs than the reference's full matrix chain and identical up to fp32 summation order.
// "discard the int(N*N*0.9) smallest entries of the sample" is an exact k-th order statistic: a 4-pass
// 8-bit radix select over order-preserving uint keys (values are held in registers, histograms in LDS).
// Everything is deterministic (integer atomics only; float reductions in fixed order).
// Finally the k largest entries of the resulting cls attention are emitted as ascending indices (top-k +
// sort of the reference), together with the 0/1 key policy for the next block.
#include "ppf_common.h"

namespace {

constexpr int NTHR = 1024, NWAVE = 16;
constexpr int RPW = 14;      // rows per wave  (N <= 224)
constexpr int CPL = 4;       // columns per lane (N <= 256)
constexpr int HCOPIES = 16;  // histogram replicas
constexpr int HSTRIDE = 257; // replica pitch (odd: replicas of one bin fall on different LDS banks)

struct RolloutParams {
    const float* hm;         // [L][B][N][NP]
    int64_t layer_stride;    // B*N*NP
    int L, B, N, NP;
    const float* init_rows;  // CaiT: [n_init][B][N+1] head-mean class-attention rows (or null -> one-hot at 0)
    int n_init;
    int lead;                // 1: matrix index 0 is cls, outputs skip it (DeiT); 0: CaiT
    int kdrop, kdrop_init;   // int(N*N*ratio), int((N+1)*ratio): computed by the host in double like the reference
    float identity;
    int k;
    const uint32_t* thr;     // optional [L][B]: per (layer, sample) discard thresholds computed ahead by rollout_threshold_kernel
    float* cls_attn;         // [B][Nk]
    int* idx;                // [B][k] ascending
    float* policy;           // [B][1+Nk]
};

__device__ __forceinline__ uint32_t order_key(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Finds the key of the element with 1-based ascending rank `target` among the values for which valid(i) holds.
// keys: per-thread register array; hist/scratch: LDS.  All threads must call; returns the key to every thread.
// Padding entries are +inf (largest key): they are counted but can never be among the `target` smallest.
template <int NV>
__device__ uint32_t radix_select(const float (&v)[NV], int target, uint32_t* hist, uint32_t* misc) {
    uint32_t prefix = 0;
    int remaining = target;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = threadIdx.x; i < HCOPIES * HSTRIDE; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        const int hcopy = (threadIdx.x & (HCOPIES - 1)) * HSTRIDE;     // attention values cluster in 2-3 exponent bins: spread the
                                                                    // LDS atomics over HCOPIES histogram replicas
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const uint32_t key = order_key(v[i]);
            const bool match = (pass == 0) || ((key >> (shift + 8)) == prefix);
            if (match) atomicAdd(&hist[hcopy + ((key >> shift) & 255u)], 1u);
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // bound the live key/bin temporaries
        }
        __syncthreads();
        // inclusive scan of the 256 bins by the first 4 waves
        uint32_t cum = 0;
        if (threadIdx.x < 256) {
            const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
            uint32_t v = 0;
#pragma unroll
            for (int c = 0; c < HCOPIES; ++c) v += hist[c * HSTRIDE + threadIdx.x];
            hist[threadIdx.x] = v;                                  // replica 0 now holds the merged bin (own slot only)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t n = __shfl_up(v, o, 64);
                if (lane >= o) v += n;
            }
            if (lane == 63) misc[w] = v;
            cum = v;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            const int w = threadIdx.x >> 6;
            uint32_t off = 0;
            for (int j = 0; j < w; ++j) off += misc[j];
            cum += off;
            const uint32_t before = cum - hist[threadIdx.x];
            if ((uint32_t)remaining > before && (uint32_t)remaining <= cum) { misc[8] = threadIdx.x; misc[9] = before; }
        }
        __syncthreads();
        prefix = (prefix << 8) | misc[8];
        remaining -= (int)misc[9];
        __syncthreads();
    }
    return prefix;
}

// ---- shared by the phases of the chain kernel (all threads of a >= 256-thread workgroup call them)
__device__ void rollout_init_vector(const RolloutParams& p, int b, float* r, float* vals, uint32_t* hist, uint32_t* misc) {
    const int tid = threadIdx.x, N = p.N;
    const float inv_norm = 1.0f / (1.0f + p.identity);
    // ---- initial row vector
    if (p.init_rows) {
        // CaiT: mean over the processed class-attention rows (discard on the N+1-long row, +identity at col 0)
        const int M = N + 1;
        const int kdrop = p.kdrop_init;
        if (tid < 256) r[tid] = 0.f;
        __syncthreads();
        for (int ir = 0; ir < p.n_init; ++ir) {
            const float* row = p.init_rows + ((size_t)ir * p.B + b) * M;
            float v1[1];
            const bool ok = tid < M;
            v1[0] = ok ? row[tid] : INFINITY;
            uint32_t thr = 0;
            if (kdrop > 0) thr = radix_select<1>(v1, kdrop, hist, misc);
            float a = (!ok || (kdrop > 0 && order_key(v1[0]) <= thr)) ? 0.f : v1[0];
            if (tid == 0) a += p.identity;
            a *= inv_norm;
            if (tid < 256) vals[tid] = ok ? a : 0.f;
            __syncthreads();
            if (tid == 0) { float s = 0.f; for (int j = 0; j < M; ++j) s += vals[j]; misc[12] = __float_as_uint(s); }
            __syncthreads();
            const float s = __uint_as_float(misc[12]);
            if (tid >= 1 && tid < M) r[tid - 1] += (a / s) / (float)p.n_init;
            __syncthreads();
        }
    } else {
        if (tid < 256) r[tid] = (tid == 0) ? 1.0f : 0.0f;
        __syncthreads();
    }

}

__device__ void rollout_emit(const RolloutParams& p, int b, const float* r, float* vals, int* sel) {
    const int tid = threadIdx.x, N = p.N;
    // ---- outputs: cls attention over the Nk patch tokens, top-k ascending indices, policy
    const int Nk = N - p.lead;
    if (tid < 256) vals[tid] = tid < Nk ? r[tid + p.lead] : -INFINITY;
    __syncthreads();
    if (tid < Nk) {
        const float mine = vals[tid];
        p.cls_attn[(size_t)b * Nk + tid] = mine;
        int rank = 0;
        for (int j = 0; j < Nk; ++j) {
            const float o = vals[j];
            rank += (o > mine) || (o == mine && j < tid);
        }
        sel[tid] = rank < p.k;
    }
    __syncthreads();
    if (tid < Nk) {
        int pos = 0;
        for (int j = 0; j < tid; ++j) pos += sel[j];
        if (sel[tid]) p.idx[(size_t)b * p.k + pos] = tid;
        p.policy[(size_t)b * (Nk + 1) + 1 + tid] = sel[tid] ? 1.0f : 0.0f;
    }
    if (tid == 0) p.policy[(size_t)b * (Nk + 1)] = 1.0f;
}

__global__ __launch_bounds__(NTHR) void rollout_kernel(const RolloutParams p) {
    __shared__ uint32_t hist[HCOPIES * HSTRIDE];
    __shared__ uint32_t misc[16];
    __shared__ float r[256];               // current row vector
    __shared__ float rnew[NWAVE][256];     // per-wave partial column sums
    __shared__ float vals[256];
    __shared__ int sel[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: scalar row addressing
    const int b = blockIdx.x, N = p.N, NP = p.NP;
    const float inv_norm = 1.0f / (1.0f + p.identity);
    rollout_init_vector(p, b, r, vals, hist, misc);
    const int kdrop = p.kdrop;
    // ---- chain, last layer first
#pragma unroll 1
    for (int l = p.L - 1; l >= 0; --l) {
        const float* f = p.hm + (size_t)l * p.layer_stride + (size_t)b * N * NP;
        float v[RPW * CPL];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = wave + NWAVE * i;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int col = lane + 64 * c;
                v[i * CPL + c] = (row < N && col < N) ? f[(size_t)row * NP + col] : INFINITY;
            }
        }
        uint32_t thr = 0;
        if (kdrop > 0) thr = p.thr ? p.thr[(size_t)l * p.B + b] : radix_select<RPW * CPL>(v, kdrop, hist, misc);
        float colacc[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) colacc[c] = 0.f;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = wave + NWAVE * i;
            float a[CPL]; float s = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int col = lane + 64 * c;
                const bool ok = row < N && col < N;
                float x = (ok && !(kdrop > 0 && order_key(v[i * CPL + c]) <= thr)) ? v[i * CPL + c] : 0.f;
                if (ok && col == row) x += p.identity;
                a[c] = x * inv_norm;
                s += a[c];
            }
            s = wave_sum(s);
            if (row < N) {
                const float coef = r[row] / s;
#pragma unroll
                for (int c = 0; c < CPL; ++c) colacc[c] += coef * a[c];
            }
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) rnew[wave][lane + 64 * c] = colacc[c];
        __syncthreads();
        if (tid < 256) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) s += rnew[w][tid];
            r[tid] = tid < N ? s : 0.f;
        }
        __syncthreads();
    }

    rollout_emit(p, b, r, vals, sel);
}

// The discard threshold of ONE layer's head-mean map per sample (the radix select is 90 % of the rollout's time and does not depend
// on the chain): launched per layer right behind that layer's attn_headmean on the side stream, so that the rollout at the
// reservation layer -- on the critical path -- is left with the row-vector chain only.  Same select, same key: identical results.
__global__ __launch_bounds__(NTHR) void rollout_threshold_kernel(const float* __restrict__ f_layer, int N, int NP, int kdrop, uint32_t* __restrict__ thr_out) {
    __shared__ uint32_t hist[HCOPIES * HSTRIDE];
    __shared__ uint32_t misc[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const float* f = f_layer + (size_t)b * N * NP;
    // the order statistic does not care which thread holds which element: one 16-byte load per (row, lane) -- four consecutive columns,
    // NP is a multiple of 4 and the map 16-byte aligned -- instead of four 4-byte loads (the chain kernel needs the column = lane layout)
    float v[RPW * CPL];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int row = wave + NWAVE * i;
        const int col = lane * 4;
        float4 q = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
        if (row < N && col < NP) q = *reinterpret_cast<const float4*>(f + (size_t)row * NP + col);
        v[i * CPL + 0] = (row < N && col + 0 < N) ? q.x : INFINITY;
        v[i * CPL + 1] = (row < N && col + 1 < N) ? q.y : INFINITY;
        v[i * CPL + 2] = (row < N && col + 2 < N) ? q.z : INFINITY;
        v[i * CPL + 3] = (row < N && col + 3 < N) ? q.w : INFINITY;
    }
    const uint32_t thr = kdrop > 0 ? radix_select<RPW * CPL>(v, kdrop, hist, misc) : 0u;
    if (tid == 0) thr_out[b] = thr;
}

// Standalone top-k + ascending index sort (protopformer.py:157-158, 273-274) for callers that only hold the scores.
__global__ __launch_bounds__(256) void topk_sorted_kernel(const float* scores, int n, int k, int* idx) {
    __shared__ float vals[256];
    __shared__ int sel[256];
    const int tid = threadIdx.x, b = blockIdx.x;
    vals[tid] = tid < n ? scores[(size_t)b * n + tid] : -INFINITY;
    __syncthreads();
    if (tid < n) {
        const float mine = vals[tid];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (vals[j] > mine) || (vals[j] == mine && j < tid);
        sel[tid] = rank < k;
    }
    __syncthreads();
    if (tid < n && sel[tid]) {
        int pos = 0;
        for (int j = 0; j < tid; ++j) pos += sel[j];
        idx[(size_t)b * k + pos] = tid;
    }
}

}  // namespace

extern "C" {

int ppf_topk_sorted(const float* scores, int B, int n, int k, int* idx, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && n >= 1 && n <= 256 && k >= 1 && k <= n, PPF_ERR_SHAPE, "ppf_topk_sorted: bad shape B=%d n=%d k=%d", B, n, k);
    hipLaunchKernelGGL(topk_sorted_kernel, dim3(B), dim3(256), 0, stream, scores, n, k, idx);
    PPF_LAUNCH_CHECK();
    return 0;
}


// hm: [L][B][N][NP] fp32 head-mean attention of the L rollout layers (layer_stride elements apart).
// init_rows: null (DeiT, lead=1) or [n_init][B][N+1] head-mean class-attention rows (CaiT, lead=0).
// Outputs: cls_attn [B][N-lead], idx [B][k] int32 ascending, policy [B][N-lead+1] float {0,1}.
int ppf_rollout_threshold(const float* hm_layer, int B, int N, int NP, int kdrop, void* thr_out_u32, hipStream_t stream) {
    PPF_CHECK_ARG(NP % 4 == 0 && NP <= 64 * CPL && ((uintptr_t)hm_layer & 15) == 0, PPF_ERR_ALIGN,
                  "ppf_rollout_threshold: the map's row pitch NP=%d must be a multiple of 4 (<= 256) and the map 16-byte aligned", NP);
    PPF_CHECK_ARG(B >= 1 && N >= 2 && N <= 16 * RPW && N <= 64 * CPL - 1 && NP >= N && kdrop >= 0 && kdrop < N * N && hm_layer && thr_out_u32, PPF_ERR_SHAPE,
                  "ppf_rollout_threshold: bad arguments B=%d N=%d NP=%d kdrop=%d", B, N, NP, kdrop);
    hipLaunchKernelGGL(rollout_threshold_kernel, dim3(B), dim3(NTHR), 0, stream, hm_layer, N, NP, kdrop, (uint32_t*)thr_out_u32);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_rollout(const float* hm, int64_t layer_stride, int L, int B, int N, int NP, const float* init_rows, int n_init, int lead,
                int kdrop, int kdrop_init, float identity, int k, const void* thr_u32, float* cls_attn, int* idx, float* policy,
                hipStream_t stream) {
    PPF_CHECK_ARG(L >= 1 && B >= 1 && N >= 2 && N <= 16 * RPW && N <= 64 * CPL - 1 && NP >= N, PPF_ERR_SHAPE, "ppf_rollout: bad shape L=%d B=%d N=%d NP=%d", L, B, N, NP);
    PPF_CHECK_ARG((lead == 1 && init_rows == nullptr) || (lead == 0 && init_rows != nullptr && n_init >= 1), PPF_ERR_ARG,
                  "ppf_rollout: lead=1 needs no init rows, lead=0 needs them");
    PPF_CHECK_ARG(k >= 1 && k <= N - lead, PPF_ERR_ARG, "ppf_rollout: bad k=%d", k);
    PPF_CHECK_ARG(kdrop >= 0 && kdrop < N * N && kdrop_init >= 0 && kdrop_init <= N, PPF_ERR_ARG, "ppf_rollout: bad discard counts");
    RolloutParams p;
    p.hm = hm; p.layer_stride = layer_stride; p.L = L; p.B = B; p.N = N; p.NP = NP; p.init_rows = init_rows; p.n_init = n_init; p.lead = lead;
    p.kdrop = kdrop; p.kdrop_init = kdrop_init; p.identity = identity; p.k = k; p.cls_attn = cls_attn; p.idx = idx; p.policy = policy;
    p.thr = (const uint32_t*)thr_u32;
    hipLaunchKernelGGL(rollout_kernel, dim3(B), dim3(NTHR), 0, stream, p);
    PPF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
