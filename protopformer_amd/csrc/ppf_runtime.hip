// Error channel + version/introspection entry points of the C ABI.
#include "ppf_common.h"
#include "ppf_hip.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

extern "C" {

void ppf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* ppf_last_error(void) { return g_err; }

int ppf_abi_version(void) { return PPF_ABI_VERSION; }

// Device facts the host side uses for launch sizing and reporting.
int ppf_device_info(int* cu_count, int* clock_mhz, char* name, int name_len) {
    hipDeviceProp_t prop;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) { ppf_set_error("ppf_device_info: %s", hipGetErrorString(e)); return (int)e; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    if (name && name_len > 0) snprintf(name, name_len, "%s", prop.gcnArchName);
    return 0;
}

// ---- stream-ordering helpers for the host mirror's two-lane schedule (backbone.WgradLane): cross-stream dependencies through a
// ring of pooled events owned by the library, one C call each instead of several torch stream / event objects per dependency
// (the Python-side bookkeeping was ~4 ms of the 17 ms host time per step).  Events are timing-disabled; the ring is large enough
// that a mark is always consumed within the same train step.
namespace {
constexpr int RING = 4096;
hipEvent_t g_ring[RING];
bool g_ring_made[RING];
int64_t g_seq = 0;
hipEvent_t ring_event(int64_t seq) {
    const int i = (int)(seq % RING);
    if (!g_ring_made[i]) {
        // ordering between two streams of ONE device: no timestamps and no system-scope fence (the host never inspects these events,
        // agent scope is what a kernel boundary gives anyway).  PPF_EVENT_SYSFENCE=1 restores the default system-scope release.
        static const bool sysfence = getenv("PPF_EVENT_SYSFENCE") && atoi(getenv("PPF_EVENT_SYSFENCE")) != 0;
        (void)hipEventCreateWithFlags(&g_ring[i], hipEventDisableTiming | (sysfence ? 0u : hipEventDisableSystemFence));
        g_ring_made[i] = true;
    }
    return g_ring[i];
}
}  // namespace

// everything enqueued on `src` so far happens-before anything enqueued on `dst` after this call
int ppf_stream_wait_stream(hipStream_t dst, hipStream_t src) {
    hipEvent_t ev = ring_event(g_seq++);
    hipError_t e = hipEventRecord(ev, src);
    if (e == hipSuccess) e = hipStreamWaitEvent(dst, ev, 0);
    if (e != hipSuccess) { ppf_set_error("ppf_stream_wait_stream: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}
// records "everything enqueued on `stream` so far" and returns its ticket (>= 0); negative = hipError_t
int64_t ppf_stream_mark(hipStream_t stream) {
    const int64_t seq = g_seq++;
    hipError_t e = hipEventRecord(ring_event(seq), stream);
    if (e != hipSuccess) { ppf_set_error("ppf_stream_mark: %s", hipGetErrorString(e)); return -(int64_t)e; }
    return seq;
}
// `stream` waits for the work recorded under `ticket` (a ticket older than the ring is complete by construction of the caller)
int ppf_stream_wait_mark(hipStream_t stream, int64_t ticket) {
    if (ticket < 0 || g_seq - ticket >= RING) return 0;
    hipError_t e = hipStreamWaitEvent(stream, ring_event(ticket), 0);
    if (e != hipSuccess) { ppf_set_error("ppf_stream_wait_mark: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

}  // extern "C"
