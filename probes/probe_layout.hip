// GPU probe (test infrastructure, not product): verifies the MFMA fragment layouts and the
// ds_read_b64_tr_b16 lane mapping that the kernels in protopformer_amd/csrc assume.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>
#include <cstring>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline uint16_t f2bf(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
static inline float bf2f(uint16_t h) { uint32_t u = ((uint32_t)h) << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }

// D[n][m] = sum_k B[n][k] A[m][k]  (swapped operands: first operand indexes output rows)
__global__ void k_mfma32(const uint16_t* A, const uint16_t* B, float* D) {
    int l = threadIdx.x;
    bf16x8 a, b;
    const uint16_t* ap = A + (l & 31) * 16 + (l >> 5) * 8;
    const uint16_t* bp = B + (l & 31) * 16 + (l >> 5) * 8;
    a = *(const bf16x8*)ap; b = *(const bf16x8*)bp;
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    // assumed: c[r] = D[i = (r&3)+8*(r>>2)+4*(l>>5)][j = l&31], i indexes first operand rows
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        D[i * 32 + (l & 31)] = c[r];
    }
}
__global__ void k_mfma16(const uint16_t* A, const uint16_t* B, float* D) {
    int l = threadIdx.x;
    bf16x8 a = *(const bf16x8*)(A + (l & 15) * 32 + (l >> 4) * 8);
    bf16x8 b = *(const bf16x8*)(B + (l & 15) * 32 + (l >> 4) * 8);
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void k_mfma32f(const float* A, const float* B, float* D) {
    int l = threadIdx.x;
    float a = A[(l & 31) * 2 + (l >> 5)], b = B[(l & 31) * 2 + (l >> 5)];
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k_mfma16f(const float* A, const float* B, float* D) {
    int l = threadIdx.x;
    float a = A[(l & 15) * 4 + (l >> 4)], b = B[(l & 15) * 4 + (l >> 4)];
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// tr read: lds[i] = i (u16); mode 0: lane address = lane*8 bytes. out[lane*4+j]
__global__ void k_tr(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    int l = threadIdx.x;
    for (int i = l; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    int off;
    if (mode == 0) off = l * 4;                                  // elements
    else if (mode == 1) off = (l & 15) * 64 + (l >> 4) * 4;      // 16 rows of pitch 64 el, 4 el per 16-lane group
    else off = ((l & 3) * 4) + ((l >> 2) & 3) * 64 + (l >> 4) * 256; // 4x16 row-major block per 16 lanes, pitch 64
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + off));
    uint16_t tmp[4]; __builtin_memcpy(tmp, &v, 8);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = tmp[j];
}
// global_load_lds 16B: each lane supplies its own global pointer; LDS dest = uniform base (+ lane*16 by HW)
__global__ void k_glds(const uint32_t* src, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4 * 2];
    int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) lds[i] = 0xdeadbeef;
    __syncthreads();
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(1))) const uint32_t glb_u32;
    // lane l reads 16 B from src + ((63-l)*4) dwords (reversed) to show per-lane source address
    __builtin_amdgcn_global_load_lds((glb_u32*)(src + (63 - l) * 4), (lds_u32*)lds, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 512; i += 64) out[i] = lds[i];
}
__global__ void k_atomic(float* p) {
    unsafeAtomicAdd(p + (threadIdx.x & 7), 1.0f);
}
__global__ void k_permswap(uint32_t* out) {
    int l = threadIdx.x;
    uint32_t a = 1000 + l, b = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l * 2] = r[0]; out[l * 2 + 1] = r[1];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main() {
    srand(1);
    {   // 32x32x16 bf16
        std::vector<uint16_t> A(32 * 16), B(32 * 16); std::vector<float> D(32 * 32), R(32 * 32, 0.f);
        for (auto& x : A) x = f2bf((rand() % 17 - 8) / 4.f);
        for (auto& x : B) x = f2bf((rand() % 13 - 6) / 2.f);
        for (int n = 0; n < 32; ++n) for (int m = 0; m < 32; ++m) { float s = 0; for (int k = 0; k < 16; ++k) s += bf2f(A[n * 16 + k]) * bf2f(B[m * 16 + k]); R[n * 32 + m] = s; }
        uint16_t *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 4096));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        k_mfma32<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        float e = 0, et = 0; for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { e = fmaxf(e, fabsf(D[i * 32 + j] - R[i * 32 + j])); et = fmaxf(et, fabsf(D[i * 32 + j] - R[j * 32 + i])); }
        printf("mfma_32x32x16_bf16: D[i=first-operand-row][j=second] maxerr=%g (transposed-hyp err=%g)\n", e, et);
    }
    {   // 16x16x32 bf16
        std::vector<uint16_t> A(16 * 32), B(16 * 32); std::vector<float> D(256), R(256, 0.f);
        for (auto& x : A) x = f2bf((rand() % 17 - 8) / 4.f);
        for (auto& x : B) x = f2bf((rand() % 13 - 6) / 2.f);
        for (int n = 0; n < 16; ++n) for (int m = 0; m < 16; ++m) { float s = 0; for (int k = 0; k < 32; ++k) s += bf2f(A[n * 32 + k]) * bf2f(B[m * 32 + k]); R[n * 16 + m] = s; }
        uint16_t *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        k_mfma16<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        float e = 0, et = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { e = fmaxf(e, fabsf(D[i * 16 + j] - R[i * 16 + j])); et = fmaxf(et, fabsf(D[i * 16 + j] - R[j * 16 + i])); }
        printf("mfma_16x16x32_bf16: maxerr=%g (transposed-hyp err=%g)\n", e, et);
    }
    {   // f32 32x32x2 and 16x16x4
        std::vector<float> A(64), B(64), D(1024), R(1024);
        for (auto& x : A) x = (rand() % 17 - 8) / 4.f; for (auto& x : B) x = (rand() % 13 - 6) / 2.f;
        for (int n = 0; n < 32; ++n) for (int m = 0; m < 32; ++m) R[n * 32 + m] = A[n * 2] * B[m * 2] + A[n * 2 + 1] * B[m * 2 + 1];
        float *dA, *dB, *dD; CK(hipMalloc(&dA, 256)); CK(hipMalloc(&dB, 256)); CK(hipMalloc(&dD, 4096));
        CK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
        k_mfma32f<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        float e = 0; for (int i = 0; i < 1024; ++i) e = fmaxf(e, fabsf(D[i] - R[i]));
        printf("mfma_32x32x2_f32: maxerr=%g\n", e);
        for (int n = 0; n < 16; ++n) for (int m = 0; m < 16; ++m) { float s = 0; for (int k = 0; k < 4; ++k) s += A[n * 4 + k] * B[m * 4 + k]; R[n * 16 + m] = s; }
        k_mfma16f<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        e = 0; for (int i = 0; i < 256; ++i) e = fmaxf(e, fabsf(D[i] - R[i]));
        printf("mfma_16x16x4_f32: maxerr=%g\n", e);
    }
    for (int mode = 0; mode < 3; ++mode) {
        uint16_t* d; CK(hipMalloc(&d, 512)); std::vector<uint16_t> h(256);
        k_tr<<<1, 64>>>(d, mode); CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
        printf("tr16_b64 mode %d: (lane: 4 source element indices)\n", mode);
        for (int l = 0; l < 64; ++l) printf("  L%02d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 4 == 3) ? "\n" : "");
    }
    {
        std::vector<uint32_t> s(256), o(512); for (int i = 0; i < 256; ++i) s[i] = i;
        uint32_t *ds, *dO; CK(hipMalloc(&ds, 1024)); CK(hipMalloc(&dO, 2048)); CK(hipMemcpy(ds, s.data(), 1024, hipMemcpyHostToDevice));
        k_glds<<<1, 64>>>(ds, dO); CK(hipMemcpy(o.data(), dO, 2048, hipMemcpyDeviceToHost));
        printf("global_load_lds16: lds dwords 0..15: "); for (int i = 0; i < 16; ++i) printf("%u ", o[i]); printf(" ... [252..259]: "); for (int i = 252; i < 260; ++i) printf("%x ", o[i]); printf("\n");
    }
    {
        float* p; CK(hipMalloc(&p, 32)); CK(hipMemset(p, 0, 32)); k_atomic<<<4, 64>>>(p); float h[8]; CK(hipMemcpy(h, p, 32, hipMemcpyDeviceToHost));
        printf("unsafeAtomicAdd: %g (expect 32)\n", h[0]);
    }
    {
        uint32_t* d; CK(hipMalloc(&d, 512)); uint32_t h[128]; k_permswap<<<1, 64>>>(d); CK(hipMemcpy(h, d, 512, hipMemcpyDeviceToHost));
        printf("permlane32_swap: lane0 -> (%u,%u) lane32 -> (%u,%u)\n", h[0], h[1], h[64], h[65]);
    }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s CUs=%d clock=%d MHz lds/block=%zu\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.sharedMemPerBlock);
    return 0;
}
