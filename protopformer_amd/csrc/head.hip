// Small fp32 kernels of the classification head and the losses:
//   * PPC loss (protopformer.py:249-288): weighted grid mean / covariance per (sample, label prototype),
//     one workgroup per sample, wave-shuffle reductions, analytic gradient emitted alongside the loss
//   * cross-entropy (main.py:390) with its gradient
//   * the frozen +1/-0.5 class-connection linears (protopformer.py:126-131,314-316): a plain strided fp32 GEMM
// All reductions run in a fixed order (per-sample partials + a single-workgroup tree), i.e. deterministic.
#include "ppf_common.h"

namespace {

constexpr int MAXPPC = 16;

struct PpcParams {
    const float* act;        // [B][P][T]
    const int* idx;          // [B][T] ascending patch indices of the reserved tokens
    const long long* label;  // [B]
    int B, P, T, ppc, side;
    float cov_thresh, mean_thresh;
    float* partial;          // [B][2] per-sample sums of the two hinge terms
    float* gcov;             // [B][ppc][T]  d(cov_loss)/d(act)
    float* gmean;            // [B][ppc][T]  d(mean_loss)/d(act)
};

__global__ __launch_bounds__(256) void ppc_kernel(const PpcParams p) {
    __shared__ float mu[MAXPPC][2], S[MAXPPC], hinge_cov[MAXPPC], gmu[MAXPPC][2];
    __shared__ float mean_sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const int N = p.side * p.side;
    const float nfac = (float)N / (float)(N - 1);
    const int lab = (int)p.label[b];
    const float inv_cnt = 1.0f / ((float)p.B * (float)p.ppc);
    // pass 1: moments per label prototype (one wave per prototype, lanes over tokens)
    for (int j = wave; j < p.ppc; j += 4) {
        const float* w = p.act + ((size_t)b * p.P + (size_t)lab * p.ppc + j) * p.T;
        float s = 0.f, sx = 0.f, sy = 0.f;
        for (int t = lane; t < p.T; t += 64) {
            const int id = p.idx[(size_t)b * p.T + t];
            const float wt = w[t], x = (float)(id / p.side), y = (float)(id % p.side);
            s += wt; sx += wt * x; sy += wt * y;
        }
        s = wave_sum(s); sx = wave_sum(sx); sy = wave_sum(sy);
        const float mx = sx / s, my = sy / s;
        float v = 0.f;
        for (int t = lane; t < p.T; t += 64) {
            const int id = p.idx[(size_t)b * p.T + t];
            const float dx = (float)(id / p.side) - mx, dy = (float)(id % p.side) - my;
            v += w[t] * (dx * dx + dy * dy);
        }
        v = wave_sum(v) / s;                                    // weighted second moment (xx + yy)
        const float c = 0.5f * nfac * v;                        // (cov00 + cov11) / 2 with the reference's N/(N-1)
        const bool on = c > p.cov_thresh;
        for (int t = lane; t < p.T; t += 64) {
            const int id = p.idx[(size_t)b * p.T + t];
            const float dx = (float)(id / p.side) - mx, dy = (float)(id % p.side) - my;
            p.gcov[((size_t)b * p.ppc + j) * p.T + t] = on ? inv_cnt * 0.5f * nfac * ((dx * dx + dy * dy) - v) / s : 0.f;
        }
        if (lane == 0) { mu[j][0] = mx; mu[j][1] = my; S[j] = s; hinge_cov[j] = on ? c - p.cov_thresh : 0.f; }
    }
    __syncthreads();
    // pass 2: pairwise mean distances (tiny: ppc x ppc), gradient w.r.t. each mean
    if (threadIdx.x == 0) {
        float ms = 0.f;
        for (int j = 0; j < p.ppc; ++j) { gmu[j][0] = 0.f; gmu[j][1] = 0.f; }
        const float pair_norm = 1.0f / ((float)p.B * (float)p.ppc * (float)p.ppc);
        for (int j = 0; j < p.ppc; ++j)
            for (int k = 0; k < p.ppc; ++k) {
                if (j == k) continue;
                const float dx = mu[j][0] - mu[k][0], dy = mu[j][1] - mu[k][1];
                const float d = sqrtf(dx * dx + dy * dy);
                if (p.mean_thresh - d > 0.f) {
                    ms += p.mean_thresh - d;
                    if (d > 0.f) {      // pair (j,k) and (k,j) both appear in the sum: each contributes -(mu_j-mu_k)/d to j
                        gmu[j][0] -= 2.0f * pair_norm * dx / d;
                        gmu[j][1] -= 2.0f * pair_norm * dy / d;
                    }
                }
            }
        mean_sum = ms;
        float cs = 0.f;
        for (int j = 0; j < p.ppc; ++j) cs += hinge_cov[j];
        p.partial[2 * b] = cs;
        p.partial[2 * b + 1] = ms;
    }
    __syncthreads();
    for (int j = wave; j < p.ppc; j += 4) {
        const float mx = mu[j][0], my = mu[j][1], s = S[j], gx = gmu[j][0], gy = gmu[j][1];
        for (int t = lane; t < p.T; t += 64) {
            const int id = p.idx[(size_t)b * p.T + t];
            const float dx = (float)(id / p.side) - mx, dy = (float)(id % p.side) - my;
            p.gmean[((size_t)b * p.ppc + j) * p.T + t] = (gx * dx + gy * dy) / s;
        }
    }
}

// out[c] = scale[c] * sum_i in[i*stride + c]   (single workgroup, fixed order)
__global__ __launch_bounds__(256) void reduce_cols_kernel(const float* in, int n, int stride, int cols, float s0, float s1, float* out) {
    __shared__ float red[256];
    for (int c = 0; c < cols; ++c) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) s += in[(size_t)i * stride + c];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[c] = red[0] * (c == 0 ? s0 : s1);
        __syncthreads();
    }
}

// g_full[b][label*ppc + j][t] = up[0]*gcov + up[1]*gmean  (g_full must be zero elsewhere)
__global__ __launch_bounds__(256) void ppc_bwd_kernel(const float* gcov, const float* gmean, const float* up_cov, const float* up_mean,
                                                      const long long* label, float* g_full, int B, int P, int T, int ppc) {
    const int b = blockIdx.x;
    const float uc = up_cov ? up_cov[0] : 0.f, um = up_mean ? up_mean[0] : 0.f;
    const int lab = (int)label[b];
    for (int i = threadIdx.x; i < ppc * T; i += 256) {
        const int j = i / T, t = i % T;
        g_full[((size_t)b * P + (size_t)lab * ppc + j) * T + t] = uc * gcov[((size_t)b * ppc + j) * T + t] + um * gmean[((size_t)b * ppc + j) * T + t];
    }
}

__global__ __launch_bounds__(256) void ce_kernel(const float* logits, const long long* label, float* per_sample, float* dlogits, int B, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const float* row = logits + (size_t)b * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(row[c] - mx);
    s = wave_sum(s);
    const int lab = (int)label[b];
    const float lse = mx + __logf(s);
    if (lane == 0) per_sample[b] = lse - row[lab];
    const float invB = 1.0f / (float)B;
    for (int c = lane; c < C; c += 64) dlogits[(size_t)b * C + c] = (__expf(row[c] - lse) - (c == lab ? 1.0f : 0.0f)) * invB;
}

// C[m][n] = alpha * sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk] + beta * C[m][n]; 32x32 tile, 2x2 outputs per thread.
// The problems here are tiny but deep (256 x 200 x 2000): the contraction is split over gridDim.z slices that write partial
// tiles, reduced in fixed order by sgemm_reduce_kernel (deterministic), so that >400 workgroups share the latency-bound K loop.
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                    float* __restrict__ part, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbn,
                                                    int64_t sbk, int ldc, float alpha, float beta) {
    __shared__ float sa[32][33], sb[32][33];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int kchunk = ((K + gridDim.z - 1) / gridDim.z + 31) / 32 * 32;
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    const bool a_kfast = sak == 1, b_kfast = sbk == 1;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        for (int i = threadIdx.x; i < 1024; i += 256) {
            int r, k;
            if (a_kfast) { k = i & 31; r = i >> 5; } else { r = i & 31; k = i >> 5; }
            sa[r][k] = (m0 + r < M && k0 + k < kend) ? A[(size_t)(m0 + r) * sam + (size_t)(k0 + k) * sak] : 0.f;
            if (b_kfast) { k = i & 31; r = i >> 5; } else { r = i & 31; k = i >> 5; }
            sb[r][k] = (n0 + r < N && k0 + k < kend) ? Bm[(size_t)(n0 + r) * sbn + (size_t)(k0 + k) * sbk] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float a0 = sa[ty][k], a1 = sa[ty + 16][k], b0 = sb[tx][k], b1 = sb[tx + 16][k];
            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
            if (m < M && n < N) {
                if (gridDim.z > 1) part[((size_t)blockIdx.z * M + m) * N + n] = acc[i][j];
                else { float* c = C + (size_t)m * ldc + n; *c = alpha * acc[i][j] + (beta != 0.f ? beta * *c : 0.f); }
            }
        }
}

// The same product for LARGE problems (round 6: the fp32 tail of the reference's default 'bottleneck' add-on head, 20 992 x 384 x 384 per
// convolution and pass, ran on the 32 x 32 VALU kernel above at ~15 TF/s = +1.2 ms per deit_small step): exact fp32 on the matrix pipe,
// v_mfma_f32_32x32x2_f32 (an fmaf chain over k, bitwise what the VALU computes in that order).  128 x 128 tile, four waves (2 x 2) with a 64 x 64
// accumulator tile each, k in chunks of 32 staged through LDS as [k][m] / [k][n] images (pitch 129 floats: conflict-free for both the k-fast and
// the m-fast global layouts), the next chunk's 32 scalars per thread prefetched into registers under the 64 MFMAs of the current one.
// Same strided-operand contract and the same split-contraction partials (gridDim.z slices -> sgemm_reduce_kernel) as sgemm_kernel.
constexpr int SM_T = 128, SM_K = 32, SM_P = SM_T + 1;
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                         float* __restrict__ part, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbn,
                                                         int64_t sbk, int ldc, float alpha, float beta) {
    __shared__ float sa[SM_K * SM_P], sb[SM_K * SM_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.y * SM_T, n0 = blockIdx.x * SM_T, wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    const int kchunk = ((K + gridDim.z - 1) / gridDim.z + SM_K - 1) / SM_K * SM_K;
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    // element e = tid + 256 i of a 128 x 32 operand chunk: (row r, contraction k); the fast index follows the operand's unit stride
    const bool a_kfast = sak == 1, b_kfast = sbk == 1;
    float ra[16], rb[16];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i;
            int r = a_kfast ? e >> 5 : e & 127, k = a_kfast ? e & 31 : e >> 7;
            ra[i] = (m0 + r < M && k0 + k < kend) ? A[(size_t)(m0 + r) * sam + (size_t)(k0 + k) * sak] : 0.f;
            r = b_kfast ? e >> 5 : e & 127; k = b_kfast ? e & 31 : e >> 7;
            rb[i] = (n0 + r < N && k0 + k < kend) ? Bm[(size_t)(n0 + r) * sbn + (size_t)(k0 + k) * sbk] : 0.f;
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i;
            int r = a_kfast ? e >> 5 : e & 127, k = a_kfast ? e & 31 : e >> 7;
            sa[k * SM_P + r] = ra[i];
            r = b_kfast ? e >> 5 : e & 127; k = b_kfast ? e & 31 : e >> 7;
            sb[k * SM_P + r] = rb[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += SM_K) {
        __syncthreads();                                       // every wave is done with the previous chunk's images
        sstore();
        __syncthreads();
        if (k0 + SM_K < kend) gload(k0 + SM_K);                // the next chunk travels while this one is multiplied
#pragma unroll
        for (int kk = 0; kk < SM_K; kk += 2) {
            // A operand of the swapped product D[n][m]: lane -> (k = kk + hh, m = .. + l31); B likewise with n
            const float a0 = sa[(kk + hh) * SM_P + wm + l31], a1 = sa[(kk + hh) * SM_P + wm + 32 + l31];
            const float b0 = sb[(kk + hh) * SM_P + wn + l31], b1 = sb[(kk + hh) * SM_P + wn + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);      // rows = m (first operand), lanes' column = n
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // accumulator layout of the 32 x 32 MFMAs: register r of lane l holds row (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the FIRST operand's index (m)
    // and column l & 31 of the second (n)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh, n = n0 + wn + 32 * j + l31;
                if (m < M && n < N) {
                    if (gridDim.z > 1) part[((size_t)blockIdx.z * M + m) * N + n] = acc[i][j][r];
                    else { float* c = C + (size_t)m * ldc + n; *c = alpha * acc[i][j][r] + (beta != 0.f ? beta * *c : 0.f); }
                }
            }
}

__global__ __launch_bounds__(256) void sgemm_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, int M, int N, int ldc, int S,
                                                           float alpha, float beta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += part[(size_t)z * M * N + i];
    float* c = C + (size_t)(i / N) * ldc + (i % N);
    *c = alpha * s + (beta != 0.f ? beta * *c : 0.f);
}

// Two independent strided fp32 products P_i = A_i B_i^T (same M) in ONE launch + ONE ordered reduction: the class-connection
// layers of the global and the local branch (protopformer.py:314-316) and their input gradients.  Slices of both problems share the
// grid (blockIdx.z < S0: problem 0); the reduction writes out_i = alpha_i P_i and, when asked, sum = c0 P0 + c1 P1.
struct SgemmProb {
    const float* A; const float* B; float* out;
    int N, K, ldo, S, kchunk;
    int64_t sam, sak, sbn, sbk;
    float alpha, c;
    int64_t part_off;          // floats: first partial slice of this problem
};
__global__ __launch_bounds__(256) void sgemm_pair_kernel(const SgemmProb q0, const SgemmProb q1, float* __restrict__ part, int M) {
    __shared__ float sa[32][33], sb[32][33];
    const bool second = (int)blockIdx.z >= q0.S;
    const SgemmProb& q = second ? q1 : q0;
    const int z = second ? blockIdx.z - q0.S : blockIdx.z;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    if (n0 >= q.N) return;
    const int kbeg = z * q.kchunk, kend = min(q.K, kbeg + q.kchunk);
    const bool a_kfast = q.sak == 1, b_kfast = q.sbk == 1;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        for (int i = threadIdx.x; i < 1024; i += 256) {
            int r, k;
            if (a_kfast) { k = i & 31; r = i >> 5; } else { r = i & 31; k = i >> 5; }
            sa[r][k] = (m0 + r < M && k0 + k < kend) ? q.A[(size_t)(m0 + r) * q.sam + (size_t)(k0 + k) * q.sak] : 0.f;
            if (b_kfast) { k = i & 31; r = i >> 5; } else { r = i & 31; k = i >> 5; }
            sb[r][k] = (n0 + r < q.N && k0 + k < kend) ? q.B[(size_t)(n0 + r) * q.sbn + (size_t)(k0 + k) * q.sbk] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float a0 = sa[ty][k], a1 = sa[ty + 16][k], b0 = sb[tx][k], b1 = sb[tx + 16][k];
            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
        }
        __syncthreads();
    }
    float* dst = part + q.part_off + (size_t)z * M * q.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
            if (m < M && n < q.N) dst[(size_t)m * q.N + n] = acc[i][j];
        }
}
__global__ __launch_bounds__(256) void sgemm_pair_reduce_kernel(const SgemmProb q0, const SgemmProb q1, const float* __restrict__ part, float* __restrict__ sum,
                                                                int ldsum, int M) {
    const int nmax = q0.N > q1.N ? q0.N : q1.N;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * nmax) return;
    const int m = i / nmax, n = i - m * nmax;
    float p0 = 0.f, p1 = 0.f;
    if (n < q0.N) { for (int z = 0; z < q0.S; ++z) p0 += part[q0.part_off + ((size_t)z * M + m) * q0.N + n]; if (q0.out) q0.out[(size_t)m * q0.ldo + n] = q0.alpha * p0; }
    if (n < q1.N) { for (int z = 0; z < q1.S; ++z) p1 += part[q1.part_off + ((size_t)z * M + m) * q1.N + n]; if (q1.out) q1.out[(size_t)m * q1.ldo + n] = q1.alpha * p1; }
    if (sum) sum[(size_t)m * ldsum + n] = q0.c * p0 + q1.c * p1;
}

__global__ __launch_bounds__(256) void axpbypcz_kernel(const float* x, const float* y, const float* z, float* out, float a, float b, float c, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a * x[i] + b * y[i] + c * z[i];
}
__global__ __launch_bounds__(256) void axpby_kernel(const float* x, const float* y, float* out, float a, float b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a * x[i] + b * y[i];
}

}  // namespace

extern "C" {

// PPC loss. act [B][P][T] (total_proto_act), idx [B][T] int32, label [B] int64.  Writes loss[0] = cov loss,
// loss[1] = mean loss, and the analytic gradients gcov / gmean [B][ppc][T]; partial is [B][2] scratch.
int ppf_ppc_loss(const float* act, const int* idx, const void* label, int B, int P, int T, int ppc, int side, float cov_thresh,
                 float mean_thresh, float* partial, float* gcov, float* gmean, float* loss, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && P > 0 && T > 0 && ppc >= 1 && ppc <= MAXPPC && side >= 2 && P % ppc == 0, PPF_ERR_SHAPE, "ppf_ppc_loss: bad shape");
    PpcParams p;
    p.act = act; p.idx = idx; p.label = (const long long*)label; p.B = B; p.P = P; p.T = T; p.ppc = ppc; p.side = side;
    p.cov_thresh = cov_thresh; p.mean_thresh = mean_thresh; p.partial = partial; p.gcov = gcov; p.gmean = gmean;
    hipLaunchKernelGGL(ppc_kernel, dim3(B), dim3(256), 0, stream, p);
    PPF_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_cols_kernel, dim3(1), dim3(256), 0, stream, partial, B, 2, 2, 1.0f / ((float)B * ppc), 1.0f / ((float)B * ppc * ppc), loss);
    PPF_LAUNCH_CHECK();
    return 0;
}

// Scatter up_cov*gcov + up_mean*gmean into the label rows of g_full [B][P][T] (caller zero-fills g_full).
int ppf_ppc_loss_bwd(const float* gcov, const float* gmean, const float* up_cov, const float* up_mean, const void* label, float* g_full, int B,
                     int P, int T, int ppc, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && P > 0 && T > 0 && ppc >= 1, PPF_ERR_SHAPE, "ppf_ppc_loss_bwd: bad shape");
    hipLaunchKernelGGL(ppc_bwd_kernel, dim3(B), dim3(256), 0, stream, gcov, gmean, up_cov, up_mean, (const long long*)label, g_full, B, P, T, ppc);
    PPF_LAUNCH_CHECK();
    return 0;
}

// Mean cross-entropy over the batch and d(loss)/d(logits); per_sample is [B] scratch.
int ppf_cross_entropy(const float* logits, const void* label, float* per_sample, float* dlogits, float* loss, int B, int C, hipStream_t stream) {
    PPF_CHECK_ARG(B > 0 && C > 0, PPF_ERR_SHAPE, "ppf_cross_entropy: bad shape");
    hipLaunchKernelGGL(ce_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, logits, (const long long*)label, per_sample, dlogits, B, C);
    PPF_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_cols_kernel, dim3(1), dim3(256), 0, stream, per_sample, B, 1, 1, 1.0f / (float)B, 0.f, loss);
    PPF_LAUNCH_CHECK();
    return 0;
}

// Small strided fp32 GEMM for the frozen class-connection layers: C = alpha * A B^T + beta * C.
// workspace (optional): fp32 scratch of ksplit*M*N floats enabling the split contraction (ksplit = workspace_floats / (M*N), <= 16).
int ppf_sgemm(const float* A, const float* Bm, float* C, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbn, int64_t sbk, int ldc,
              float alpha, float beta, float* workspace, int64_t workspace_floats, hipStream_t stream) {
    PPF_CHECK_ARG(M > 0 && N > 0 && K > 0, PPF_ERR_SHAPE, "ppf_sgemm: bad shape");
    int S = workspace ? (int)(workspace_floats / ((int64_t)M * N)) : 1;
    if (S > 16) S = 16;
    if (S > (K + 127) / 128) S = (K + 127) / 128;
    if (S < 1) S = 1;
    const int kchunk = ((K + S - 1) / S + 31) / 32 * 32;
    S = (K + kchunk - 1) / kchunk;                          // no empty slices
    // large problems (the bottleneck add-on head's tail): the fp32-MFMA kernel; the small deep ones (class-connection logits) stay on the VALU tiles
    if ((int64_t)M * N * K >= (int64_t(1) << 28) && M >= 64 && N >= 64) {
        const int tiles = ((N + SM_T - 1) / SM_T) * ((M + SM_T - 1) / SM_T);
        if (tiles >= 128) S = 1;                              // enough workgroups without splitting the contraction
        else if (S > 1) { const int kc = ((K + S - 1) / S + SM_K - 1) / SM_K * SM_K; S = (K + kc - 1) / kc; }
        hipLaunchKernelGGL(sgemm_mfma_kernel, dim3((N + SM_T - 1) / SM_T, (M + SM_T - 1) / SM_T, S), dim3(256), 0, stream, A, Bm, C, workspace, M, N, K,
                           sam, sak, sbn, sbk, ldc, alpha, beta);
    } else {
        hipLaunchKernelGGL(sgemm_kernel, dim3((N + 31) / 32, (M + 31) / 32, S), dim3(256), 0, stream, A, Bm, C, workspace, M, N, K, sam, sak, sbn, sbk,
                           ldc, alpha, beta);
    }
    PPF_LAUNCH_CHECK();
    if (S > 1) {
        hipLaunchKernelGGL(sgemm_reduce_kernel, dim3((M * N + 255) / 256), dim3(256), 0, stream, workspace, C, M, N, ldc, S, alpha, beta);
        PPF_LAUNCH_CHECK();
    }
    return 0;
}

// Two products with the same M in one launch + one ordered reduction (see sgemm_pair_kernel):  out_i = alpha_i A_i B_i^T (i = 0, 1;
// out_i may be NULL), sum = c0 A0 B0^T + c1 A1 B1^T (may be NULL; needs N0 == N1).  A_i(m, k) = A_i[m*sam_i + k*sak_i],
// B_i(n, k) = B_i[n*sbn_i + k*sbk_i].  workspace: ppf_sgemm_pair_workspace(M, N0, K0, N1, K1) floats.
static int pair_slices(int K) { int S = (K + 127) / 128; if (S > 16) S = 16; if (S < 1) S = 1; const int kc = ((K + S - 1) / S + 31) / 32 * 32; return (K + kc - 1) / kc; }
int64_t ppf_sgemm_pair_workspace(int M, int N0, int K0, int N1, int K1) {
    return (int64_t)M * ((int64_t)N0 * pair_slices(K0) + (int64_t)N1 * pair_slices(K1));
}
int ppf_sgemm_pair(const float* A0, const float* B0, float* out0, int N0, int K0, int64_t sam0, int64_t sak0, int64_t sbn0, int64_t sbk0, int ldo0,
                   float alpha0, const float* A1, const float* B1, float* out1, int N1, int K1, int64_t sam1, int64_t sak1, int64_t sbn1,
                   int64_t sbk1, int ldo1, float alpha1, float* sum, int ldsum, float c0, float c1, int M, float* workspace,
                   int64_t workspace_floats, hipStream_t stream) {
    PPF_CHECK_ARG(M > 0 && N0 > 0 && K0 > 0 && N1 > 0 && K1 > 0 && A0 && B0 && A1 && B1, PPF_ERR_SHAPE, "ppf_sgemm_pair: bad shape / null operand");
    PPF_CHECK_ARG(sum == nullptr || N0 == N1, PPF_ERR_ARG, "ppf_sgemm_pair: the combined output needs N0 == N1");
    PPF_CHECK_ARG(workspace && workspace_floats >= ppf_sgemm_pair_workspace(M, N0, K0, N1, K1), PPF_ERR_ARG,
                  "ppf_sgemm_pair: workspace below ppf_sgemm_pair_workspace() = %lld floats", (long long)ppf_sgemm_pair_workspace(M, N0, K0, N1, K1));
    SgemmProb q0, q1;
    q0.A = A0; q0.B = B0; q0.out = out0; q0.N = N0; q0.K = K0; q0.ldo = ldo0; q0.sam = sam0; q0.sak = sak0; q0.sbn = sbn0; q0.sbk = sbk0; q0.alpha = alpha0; q0.c = c0;
    q1.A = A1; q1.B = B1; q1.out = out1; q1.N = N1; q1.K = K1; q1.ldo = ldo1; q1.sam = sam1; q1.sak = sak1; q1.sbn = sbn1; q1.sbk = sbk1; q1.alpha = alpha1; q1.c = c1;
    q0.S = pair_slices(K0); q0.kchunk = ((K0 + q0.S - 1) / q0.S + 31) / 32 * 32;
    q1.S = pair_slices(K1); q1.kchunk = ((K1 + q1.S - 1) / q1.S + 31) / 32 * 32;
    q0.part_off = 0; q1.part_off = (int64_t)M * N0 * q0.S;
    const int nmax = N0 > N1 ? N0 : N1;
    hipLaunchKernelGGL(sgemm_pair_kernel, dim3((nmax + 31) / 32, (M + 31) / 32, q0.S + q1.S), dim3(256), 0, stream, q0, q1, workspace, M);
    PPF_LAUNCH_CHECK();
    hipLaunchKernelGGL(sgemm_pair_reduce_kernel, dim3((M * nmax + 255) / 256), dim3(256), 0, stream, q0, q1, workspace, sum, ldsum, M);
    PPF_LAUNCH_CHECK();
    return 0;
}

// out = a x + b y + c z  (the train loop's loss = CE + 0.1 cov + 0.5 mean in one launch, engine_proto.py:61-64)
int ppf_axpbypcz(const float* x, const float* y, const float* z, float* out, float a, float b, float c, int64_t n, hipStream_t stream) {
    PPF_CHECK_ARG(n > 0 && x && y && z && out, PPF_ERR_ARG, "ppf_axpbypcz: bad arguments");
    const int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(axpbypcz_kernel, dim3((int)(g > 1024 ? 1024 : g)), dim3(256), 0, stream, x, y, z, out, a, b, c, n);
    PPF_LAUNCH_CHECK();
    return 0;
}

int ppf_axpby(const float* x, const float* y, float* out, float a, float b, int64_t n, hipStream_t stream) {
    PPF_CHECK_ARG(n > 0, PPF_ERR_SHAPE, "ppf_axpby: bad length");
    const int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(axpby_kernel, dim3((int)(g > 1024 ? 1024 : g)), dim3(256), 0, stream, x, y, out, a, b, n);
    PPF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
