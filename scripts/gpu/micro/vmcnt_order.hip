// Do vector-memory loads and stores of one wave retire from vmcnt IN ORDER on gfx950?  (hipcc assumes not: with a load and a store
// pending it always waits vmcnt(0).)  One wave per CU: [store to a cold HBM line] [load of an L2-hot line] s_waitcnt vmcnt(N) and the
// reverse order; s_memtime around the wait.  In-order: vmcnt(1) returns when the OLDER of the two is done; out of order: when either is.
// build: hipcc --offload-arch=gfx950 -O3 vmcnt_order.hip -o vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k(float* cold, const float* hot, unsigned long long* out, int mode, int stride) {
    const int lane = threadIdx.x;
    float* pc = cold + ((size_t)blockIdx.x * 64 + lane) * stride;      // one line per lane: slow
    const float* ph = hot + lane;                                        // one line per wave, L2 / L1 hot after the warm-up below
    float w = __builtin_nontemporal_load(ph);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t0, t1, t2;
    float v = 0.f, u = 0.f;
    if (mode == 0) {            // store (slow) then load (fast); wait vmcnt(1), then vmcnt(0)
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
        asm volatile("global_store_dword %0, %1, off" :: "v"(pc), "v"(w) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ph) : "memory");
        asm volatile("s_waitcnt vmcnt(1)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2), "+v"(v) :: "memory");
    } else if (mode == 1) {     // load (slow: cold line) then store (to the hot line's neighbour, fast?)
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(pc) : "memory");
        asm volatile("global_store_dword %0, %1, off" :: "v"(cold + (size_t)gridDim.x * 64 * stride + blockIdx.x * 64 + lane), "v"(w) : "memory");
        asm volatile("s_waitcnt vmcnt(1)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2), "+v"(v) :: "memory");
    } else if (mode == 2) {     // two loads: slow then fast (the known in-order case)
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(pc) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(u) : "v"(ph) : "memory");
        asm volatile("s_waitcnt vmcnt(1)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2), "+v"(v), "+v"(u) :: "memory");
    } else {                    // store alone: time to its acknowledge
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
        asm volatile("global_store_dword %0, %1, off" :: "v"(pc), "v"(w) : "memory");
        asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        t2 = t1;
    }
    if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t0; }
    if (v + u == 12345.f) cold[0] = v;
}

int main() {
    const int grid = 256, stride = 4096;                                 // floats between the lanes' cold lines
    float *cold, *hot; unsigned long long* out;
    hipMalloc(&cold, ((size_t)grid * 64 * stride + grid * 64 + 64) * 4 * 8); hipMalloc(&hot, 4096); hipMalloc(&out, grid * 16);
    hipMemset(hot, 0, 4096);
    const char* names[] = {"store(cold) then load(hot)", "load(cold) then store", "load(cold) then load(hot)", "store(cold) alone"};
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<double> a, b;
        for (int it = 0; it < 8; ++it) {
            hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, cold + (size_t)it * grid * 64 * stride, hot, out, mode, stride);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(grid * 2);
            hipMemcpy(h.data(), out, grid * 16, hipMemcpyDeviceToHost);
            if (it < 2) continue;
            for (int i = 0; i < grid; ++i) { a.push_back((double)h[2 * i]); b.push_back((double)h[2 * i + 1]); }
        }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        printf("%-30s  vmcnt(1) passed after %7.0f ticks (median), vmcnt(0) after %7.0f   [s_memtime ticks = 100 MHz? see ratio]\n", names[mode], a[a.size() / 2], b[b.size() / 2]);
    }
    return 0;
}
