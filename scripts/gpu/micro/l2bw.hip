// L2 -> CU read bandwidth micro-benchmark: every workgroup re-reads a small (L2-resident) buffer with 16-byte loads.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC l2bw.hip -o l2bw.so
#include <hip/hip_runtime.h>
#include <cstdint>
extern "C" {
__global__ __launch_bounds__(256) void l2_read_kernel(const uint4* __restrict__ buf, int n16, int reps, int per_wg16, uint4* __restrict__ sink) {
    // workgroup w reads [off, off + per_wg16) chunks, reps times
    const int off = (int)(((long long)blockIdx.x * per_wg16) % n16);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r) {
#pragma unroll 8
        for (int i = threadIdx.x; i < per_wg16; i += 256) {
            int j = off + i; if (j >= n16) j -= n16;
            const uint4 v = buf[j];
            acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 42u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void l2_read_lds_kernel(const unsigned char* __restrict__ buf, int nbytes, int reps, int per_wg_bytes, uint4* __restrict__ sink) {
    // the same through global_load_lds_dwordx4 (direct to LDS, 4 KiB per workgroup instruction)
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    typedef __attribute__((address_space(3))) void lds_t;
    typedef const __attribute__((address_space(1))) void gbl_t;
    const long long off0 = ((long long)blockIdx.x * per_wg_bytes) % nbytes;
    const int wave = threadIdx.x >> 6;
    for (int r = 0; r < reps; ++r) {
        for (int o = 0; o < per_wg_bytes; o += 32768) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                long long a = off0 + o + i * 4096 + threadIdx.x * 16; if (a >= nbytes) a -= nbytes;
                __builtin_amdgcn_global_load_lds((gbl_t*)(buf + a), (lds_t*)(lds + i * 4096 + wave * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (lds[threadIdx.x] == 0x7f && reps < 0) sink[0] = make_uint4(1, 2, 3, 4);
}
// host helpers: time `iters` launches with events, return ms per launch
float l2bw_run(int mode, const void* buf, long long nbytes, int reps, long long per_wg_bytes, int grid, int iters, void* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        if (mode == 0) hipLaunchKernelGGL(l2_read_kernel, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, (int)(nbytes / 16), reps, (int)(per_wg_bytes / 16), (uint4*)sink);
        else hipLaunchKernelGGL(l2_read_lds_kernel, dim3(grid), dim3(256), 0, 0, (const unsigned char*)buf, (int)nbytes, reps, (int)per_wg_bytes, (uint4*)sink);
    }
    hipEventRecord(e0, 0);
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) hipLaunchKernelGGL(l2_read_kernel, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, (int)(nbytes / 16), reps, (int)(per_wg_bytes / 16), (uint4*)sink);
        else hipLaunchKernelGGL(l2_read_lds_kernel, dim3(grid), dim3(256), 0, 0, (const unsigned char*)buf, (int)nbytes, reps, (int)per_wg_bytes, (uint4*)sink);
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / iters;
}
}
