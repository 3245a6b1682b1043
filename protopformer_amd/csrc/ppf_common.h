// Shared device/host helpers for the ProtoPFormer gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                              // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define PPF_ERR_SHAPE (-1)
#define PPF_ERR_ALIGN (-2)
#define PPF_ERR_ARG (-3)

#ifdef __cplusplus
extern "C" void ppf_set_error(const char* fmt, ...);
#endif

#define PPF_CHECK_ARG(cond, code, ...)                     \
    do {                                                   \
        if (!(cond)) {                                     \
            ppf_set_error(__VA_ARGS__);                    \
            return (code);                                 \
        }                                                  \
    } while (0)

#define PPF_LAUNCH_CHECK()                                                         \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            ppf_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return (int)e__;                                                       \
        }                                                                          \
    } while (0)

// Path probe (bench.py's roofline.named_path, ppf_runtime.hip): HIP events from a reused pool around the launches of one of the kernels
// the north star names, on the launch stream; off (two loads and a branch) unless ppf_path_probe(1) switched it on.
enum { PPF_PROBE_ATTN_FWD = 0, PPF_PROBE_ATTN_BWD = 1, PPF_PROBE_PROTO_FWD = 2, PPF_PROBE_NTAGS = 3 };
#ifdef __cplusplus
struct PpfProbeScope {
    hipEvent_t stop = nullptr;
    hipStream_t stream;
    PpfProbeScope(int tag, hipStream_t stream, double flops, double bytes);   // records the start event when the probe is on
    ~PpfProbeScope();                                                          // records the stop event
};
#endif

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16 (round-to-nearest-even): native conversion, lowers to v_cvt_pk_bf16_f32 on gfx950 (branch-free)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float2 unpack_bf16x2(uint32_t v) {
    return make_float2(__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7): one exp + one rcp + 5 FMAs instead of libm's erff polynomial
// ladder (the GELU epilogues were VALU-bound on it).  ex2 = exp(-x*x) is returned for reuse by the GELU derivative.
__device__ __forceinline__ float fast_erf(float x, float& ex2) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);      // v_rcp_f32 (1 ulp); __frcp_rn expands to a 10-instruction IEEE divide
    ex2 = __expf(-ax * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    return copysignf(1.0f - poly * ex2, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
    float e;
    return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f, e));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float e;                                                   // e = exp(-x^2/2)
    const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f, e));
    return cdf + x * 0.39894228040143267794f * e;
}
// gelu(x) and gelu'(x) from one erf / exp evaluation: the fc1 epilogue stores both, so the backward epilogue is a plain multiply
__device__ __forceinline__ void gelu_erf_both(float x, float& g, float& d) {
    float e;
    const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f, e));
    g = x * cdf;
    d = cdf + x * 0.39894228040143267794f * e;
}

// The same for two values at once on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32): 12 VALU instructions per element
// instead of 30 -- the fc1 epilogue is VALU-bound on this.
typedef float ppf_float2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_erf_both2(ppf_float2 x, ppf_float2& g, ppf_float2& d) {
    const ppf_float2 z = x * 0.70710678118654752440f;
    ppf_float2 az; az.x = fabsf(z.x); az.y = fabsf(z.y);
    const ppf_float2 den = az * 0.3275911f + 1.0f;
    ppf_float2 t; t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
    const ppf_float2 a2 = -(az * az) * 1.44269504088896340736f;
    ppf_float2 e; e.x = __builtin_amdgcn_exp2f(a2.x); e.y = __builtin_amdgcn_exp2f(a2.y);       // exp(-z^2) = exp(-x^2/2)
    ppf_float2 poly = t * 1.061405429f + (-1.453152027f);
    poly = poly * t + 1.421413741f;
    poly = poly * t + (-0.284496736f);
    poly = poly * t + 0.254829592f;
    poly = poly * t;
    const ppf_float2 ea = 1.0f - poly * e;                                                       // erf(|z|), A&S 7.1.26
    ppf_float2 er; er.x = copysignf(ea.x, z.x); er.y = copysignf(ea.y, z.y);
    const ppf_float2 cdf = er * 0.5f + 0.5f;
    g = x * cdf;
    d = x * (0.39894228040143267794f * e) + cdf;
}

// LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes -> 1 KiB of LDS at `lds_addr`, wave-uniform) issued through inline assembly.
// Why not __builtin_amdgcn_global_load_lds: the compiler's wait-count pass knows that builtin writes LDS asynchronously and puts
// `s_waitcnt vmcnt(0)` in front of every later LDS read it cannot prove disjoint from it -- every ds_read_tr builtin (no memory operand) and
// reads through another pointer into the same dynamic-LDS array.  In attn_fwd16_kernel that drained the NEXT head's K / V before the P.V phase of
// the current one, in attn_bwd_stream_kernel the next item's images before the current item's results were stored (round 6, found in the
// disassembly: scripts/isa_count.py / profiles/r6_lds_dma_hidden.txt).  The kernels that use this order the DMA against their LDS reads themselves
// (s_waitcnt vmcnt + barrier).  The compiler's own vmcnt bookkeeping for ordinary loads stays safe: loads return in order, so operations it does
// not know about only make its counted waits stricter.
__device__ __forceinline__ void lds_dma16_hidden(const void* gptr, uint32_t lds_addr) {
    uint32_t saved_m0;                                   // M0 is the compiler's: saved and restored inside the statement instead of clobbered
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0) : "v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_offset_of(const void* p) { return (uint32_t)(uintptr_t)p; }      // flat LDS address: aperture | offset

// ---- kernel launches that carry their own completion event (round 6) -----------------------------------------------------------------
// Cross-stream ordering used to cost the MAIN queue one event-record packet per dependency (ppf_stream_wait_stream: hipEventRecord on the
// producer's stream): ~70 packets per deit_small step, each a 4-8 us bubble between two dependent kernels (profiles/r6_queue_gaps.txt: 0.74 us
// per boundary with one stream and no events, 4.7 us with two).  hipExtLaunchKernel can attach a stop event to the dispatch packet itself, so
// the producer's own completion signal is what the other stream waits for and no extra packet enters the producer's queue.  A stream is
// "armed" for the duration of one library call by the replay loop (ppf_stream_arm, when the recorded list shows that the call is followed by
// a wait on its stream); every launch of that call then takes an event from the library's ring, and ppf_stream_wait_stream uses the last
// one instead of recording.  Unarmed launches (everything outside a replayed step) are plain <<< >>> launches.
#ifdef __cplusplus
#include <hip/hip_ext.h>
hipEvent_t ppf_take_stop_event(hipStream_t s);          // csrc/ppf_runtime.hip; nullptr unless `s` is armed
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                              \
    do {                                                                                                                               \
        hipEvent_t ppf_ev_ = ppf_take_stop_event(streamId);                                                                            \
        if (ppf_ev_) hipExtLaunchKernelGGL(kernelName, dim3(numBlocks), dim3(numThreads), memPerBlock, streamId, nullptr, ppf_ev_, 0, __VA_ARGS__); \
        else kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);                                          \
    } while (0)
#endif

// Bijective XCD-aware remap of a 1-D block id: consecutive virtual ids land on the same XCD (private L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
