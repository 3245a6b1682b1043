// Kernel-boundary cost vs the store policy of the producer: kernel A streams `MB` of output (plain / non-temporal / write-through sc1
// stores), kernel B is a trivial dependent kernel.  Time of the pair (A + boundary + B) minus A alone.
//   hipcc --offload-arch=gfx950 -O3 scripts/gpu/micro/boundary_store.hip -o /tmp/boundary_store && /tmp/boundary_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void writer(f32x4* out, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 val = {v, v + 1.f, v + 2.f, (float)i};
        if (MODE == 0) out[i] = val;
        else if (MODE == 1) __builtin_nontemporal_store(val, out + i);
        else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(out + i), "v"(val) : "memory");
    }
}
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
template <int MODE>
float run(f32x4* buf, size_t n4, float* flag, bool pair, hipStream_t s) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> t;
    for (int it = 0; it < 30; ++it) {
        hipEventRecord(e0, s);
        for (int r = 0; r < 8; ++r) {
            hipLaunchKernelGGL(writer<MODE>, dim3(2048), dim3(256), 0, s, buf, n4, (float)it);
            if (pair) hipLaunchKernelGGL(tiny, dim3(256), dim3(64), 0, s, flag);
        }
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1000.f / 8);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    float* flag; hipMalloc(&flag, 4); hipMemset(flag, 0, 4);
    for (size_t mb : {8, 38, 155, 310}) {
        const size_t n4 = mb * 1000000 / 16;
        f32x4* buf; hipMalloc(&buf, n4 * 16);
        const float a0 = run<0>(buf, n4, flag, false, s), p0 = run<0>(buf, n4, flag, true, s);
        const float a1 = run<1>(buf, n4, flag, false, s), p1 = run<1>(buf, n4, flag, true, s);
        const float a2 = run<2>(buf, n4, flag, false, s), p2 = run<2>(buf, n4, flag, true, s);
        printf("%4zu MB  plain: A %7.1f us, A+tiny %7.1f (+%5.1f)   nt: A %7.1f, A+tiny %7.1f (+%5.1f)   sc1: A %7.1f, A+tiny %7.1f (+%5.1f)\n", mb, a0, p0, p0 - a0, a1,
               p1, p1 - a1, a2, p2, p2 - a2);
        hipFree(buf);
    }
    return 0;
}
