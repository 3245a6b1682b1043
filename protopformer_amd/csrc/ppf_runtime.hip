// Error channel + version/introspection entry points of the C ABI.
#include "ppf_common.h"
#include "ppf_hip.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

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
        // agent scope is what a kernel boundary gives anyway)
        (void)hipEventCreateWithFlags(&g_ring[i], hipEventDisableTiming | hipEventDisableSystemFence);
        g_ring_made[i] = true;
    }
    return g_ring[i];
}
// streams armed for completion events on their launches (ppf_common.h): at most a handful of streams per process
struct ArmSlot { hipStream_t s; bool used, armed, have; hipEvent_t ev; };
ArmSlot g_arm[8];
ArmSlot* arm_slot(hipStream_t s, bool create) {
    for (auto& a : g_arm) if (a.used && a.s == s) return &a;
    if (create) for (auto& a : g_arm) if (!a.used) { a = ArmSlot{s, true, false, false, nullptr}; return &a; }
    return nullptr;
}
}  // namespace

// 1: every kernel launch on `stream` carries a completion event from the ring until the stream is disarmed or the event is consumed by
// ppf_stream_wait_stream(dst, stream); 0: disarm.  Replay loop only (protopformer_amd/_lib.py); never under graph capture.
int ppf_stream_arm(hipStream_t stream, int on) {
    ArmSlot* a = arm_slot(stream, on != 0);
    if (a) { a->armed = on != 0; a->have = false; }
    return 0;
}

// everything enqueued on `src` so far happens-before anything enqueued on `dst` after this call
int ppf_stream_wait_stream(hipStream_t dst, hipStream_t src) {
    if (ArmSlot* a = arm_slot(src, false)) {
        const bool use = a->armed && a->have;
        hipEvent_t aev = a->ev;
        a->armed = false; a->have = false;
        if (use) {                                     // the last launch on src carries this event: no packet enters src's queue
            hipError_t e = hipStreamWaitEvent(dst, aev, 0);
            if (e != hipSuccess) { ppf_set_error("ppf_stream_wait_stream: %s", hipGetErrorString(e)); return (int)e; }
            return 0;
        }
    }
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

// ---- path probe: in-step HIP-event time of the kernels the north star names (attention forward / backward, prototype forward) ----
namespace {
struct PathProbe {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    size_t used = 0;
    double flops = 0.0, bytes = 0.0;
};
PathProbe g_path[PPF_PROBE_NTAGS];
constexpr size_t PATH_PROBE_CAP = 4096;          // event pairs per tag (one step of the largest configuration launches < 64 per tag)
bool g_path_on = false;
}  // namespace

}  // extern "C"

hipEvent_t ppf_take_stop_event(hipStream_t s) {
    for (auto& a : g_arm)
        if (a.used && a.armed && a.s == s) { a.ev = ring_event(g_seq++); a.have = true; return a.ev; }
    return nullptr;
}

PpfProbeScope::PpfProbeScope(int tag, hipStream_t s, double flops, double bytes) : stream(s) {
    if (!g_path_on || tag < 0 || tag >= PPF_PROBE_NTAGS) return;
    // eager launches only: an event recorded into a capturing stream becomes a graph node that hipEventElapsedTime cannot read
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
    PathProbe& pr = g_path[tag];
    if (pr.used >= PATH_PROBE_CAP) return;             // a probe left on for a long run stops counting instead of growing without bound
    if (pr.used == pr.pool.size()) {
        hipEvent_t a = nullptr, b = nullptr;           // timing events the host reads after a device synchronise: agent scope only
        (void)hipEventCreateWithFlags(&a, hipEventDisableSystemFence); (void)hipEventCreateWithFlags(&b, hipEventDisableSystemFence);
        pr.pool.emplace_back(a, b);
    }
    auto& ev = pr.pool[pr.used++];
    pr.flops += flops; pr.bytes += bytes;
    (void)hipEventRecord(ev.first, stream);
    stop = ev.second;
}
PpfProbeScope::~PpfProbeScope() { if (stop) (void)hipEventRecord(stop, stream); }

extern "C" {

// enable = 1 clears the counters and starts recording, 0 stops, 2 stops and destroys the event pools.
int ppf_path_probe(int enable) {
    for (auto& pr : g_path) {
        if (enable == 1) { pr.used = 0; pr.flops = 0.0; pr.bytes = 0.0; }
        if (enable == 2) {
            for (auto& e : pr.pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
            pr.pool.clear(); pr.used = 0;
        }
    }
    g_path_on = enable == 1;
    return 0;
}

// Synchronises tag's events: summed kernel milliseconds, launches, algorithmic flops and bytes since ppf_path_probe(1).
int ppf_path_probe_read(int tag, double* ms_total, int64_t* launches, double* flops, double* bytes) {
    PPF_CHECK_ARG(tag >= 0 && tag < PPF_PROBE_NTAGS, PPF_ERR_ARG, "ppf_path_probe_read: tag %d", tag);
    PathProbe& pr = g_path[tag];
    double ms = 0.0;
    for (size_t i = 0; i < pr.used; ++i) {
        hipError_t rc = hipEventSynchronize(pr.pool[i].second);
        float t = 0.f;
        if (rc == hipSuccess) rc = hipEventElapsedTime(&t, pr.pool[i].first, pr.pool[i].second);
        if (rc != hipSuccess) { ppf_set_error("ppf_path_probe_read: %s", hipGetErrorString(rc)); return (int)rc; }
        ms += t;
    }
    if (ms_total) *ms_total = ms;
    if (launches) *launches = (int64_t)pr.used;
    if (flops) *flops = pr.flops;
    if (bytes) *bytes = pr.bytes;
    return 0;
}

}  // extern "C"
