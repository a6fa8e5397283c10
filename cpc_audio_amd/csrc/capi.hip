// Miscellaneous C-ABI entry points.
#include "cpc_common.h"
#include "cpc_internal.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace cpc {
int g_mfma_mode = 3;

// Events that order a second stream against the caller's inside the *_streams entry points (timing disabled; created
// once per (device, caller stream) and reused: a wait captures the record that precedes it, so re-recording later is
// harmless).  The pool is keyed by the caller's stream and guarded by a mutex: two host threads driving two streams of
// one device (nn.DataParallel-style replicas) each get their own events -- with a shared pool one thread's wait could
// capture the other's record.  Two threads enqueueing on the SAME stream at once is a caller error, as everywhere in HIP.
// The pool assumes long-lived streams (torch's pooled streams are): a caller that creates a stream per step hands each one
// back with cpc_release_stream() before destroying it, or a recycled handle would inherit the old pool.
static std::mutex mu;
static std::map<std::pair<int, hipStream_t>, hipEvent_t*> pools;
hipEvent_t* stream_events(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = pools.find({dev, st});
    if (it != pools.end()) return it->second;
    hipEvent_t* ev = new hipEvent_t[kStreamEvents];
    // These events only ever order streams of ONE device against each other (nobody synchronises the host with them, no other
    // device reads behind them): without the system-scope fence a record does not write the caches back for the host's sake.
    // CPC_EVENT_SYSTEM_FENCE=1 restores the default (A/B runs).
    const char* env = getenv("CPC_EVENT_SYSTEM_FENCE");
    const unsigned flags = hipEventDisableTiming | ((env && env[0] == '1') ? 0u : (unsigned)hipEventDisableSystemFence);
    for (int i = 0; i < kStreamEvents; ++i)
        if (hipEventCreateWithFlags(&ev[i], flags) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(ev[j]);
            delete[] ev;
            return nullptr;
        }
    pools[{dev, st}] = ev;
    return ev;
}
StepHooks& step_hooks() {
    static thread_local StepHooks h;
    return h;
}
}  // namespace cpc

// Drop (and destroy) the events this library keeps for `stream` on the current device; a no-op for a stream it never saw.
// The stream must have no *_streams call of this library in flight.
extern "C" int cpc_release_stream(void* stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return CPC_ERR_ARG;
    std::lock_guard<std::mutex> lock(cpc::mu);
    auto it = cpc::pools.find({dev, (hipStream_t)stream});
    if (it == cpc::pools.end()) return 0;
    for (int i = 0; i < cpc::kStreamEvents; ++i) (void)hipEventDestroy(it->second[i]);
    delete[] it->second;
    cpc::pools.erase(it);
    return 0;
}

extern "C" int cpc_abi_version(void) { return 15; }

// A kernel that keeps one wavefront busy for `ticks` of the 100 MHz wall clock (bounded: it gives up after ~2^14 sleeps).
__global__ void spin_kernel(unsigned long long ticks) {
#ifndef HIPEMU
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 14) && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(64);
#else
    (void)ticks;
#endif
}

// Do kernels on stream `b` run while stream `a` is busy?  The runtime multiplexes hipStreams onto a handful of hardware queues
// (GPU_MAX_HW_QUEUES: 4 per priority level by default) in creation order, and two streams on one queue execute in submission
// order whatever their events say: with 4 queues, pool stream i of torch shares one with stream i + 4 or so, and where the
// step's streams land depends on how many streams the process created before (RCCL creates six).  a runs ONE wavefront for
// 3 ms; b's empty kernel finishes at once unless the two share a queue.  The host side picks the step's streams with this
// probe (ops.pick_concurrent_stream).  Blocks the calling thread for ~3 ms.  *overlap = 1 if b's kernel finished while a's ran.
extern "C" int cpc_streams_overlap(void* stream_a, void* stream_b, int* overlap) {
    CPC_RETURN_IF(!overlap, CPC_ERR_ARG);
#ifdef HIPEMU
    *overlap = stream_a != stream_b;
    return 0;
#else
    hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    hipEvent_t ea = nullptr, eb = nullptr;
    int rc = 1000 + (int)hipErrorUnknown;
    if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&eb, hipEventDisableTiming) == hipSuccess) {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 300000ull);       // 3 ms
        bool ok = hipEventRecord(ea, a) == hipSuccess;
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, 0ull);
        ok = ok && hipEventRecord(eb, b) == hipSuccess && hipEventSynchronize(eb) == hipSuccess;
        if (ok) {
            *overlap = hipEventQuery(ea) == hipErrorNotReady ? 1 : 0;
            rc = hipEventSynchronize(ea) == hipSuccess && hipGetLastError() == hipSuccess ? 0 : rc;
        }
    }
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    return rc;
#endif
}

// Device-side error flags of the current device, accumulated since they were last cleared:
//   bit 0  CPC_DEVERR_GRU_POLL_TIMEOUT   a workgroup of the persistent recurrence gave up waiting for another one (its
//                                        outputs carry NaN from there on)
//   bit 1  CPC_DEVERR_NEGATIVE_INDEX     cpc_nce_prepare was given a draw outside [0,B) x [0,S) (clamped)
//   bit 2  CPC_DEVERR_CONV_EXCHANGE      a workgroup of the N-split conv forward gave up waiting for its partner's row statistics
//                                        (its rows carry NaN)
// Synchronises with the device (two 4-byte reads): for logging points and tests, not for the step path.
// Returns the mask (>= 0), or a negative number if it cannot be read.
extern "C" int cpc_device_error_flags(int clear) {
    unsigned a = 0, b = 0, c = 0;
    if (cpc::gru_error_flag_fetch(clear, &a) != 0 || cpc::nce_error_flag_fetch(clear, &b) != 0 ||
        cpc::enc_error_flag_fetch(clear, &c) != 0) return -1;
    return (int)((a ? 1u : 0u) | (b ? 2u : 0u) | (c ? 4u : 0u));
}

extern "C" int cpc_get_mfma_mode(void) { return cpc::g_mfma_mode; }

extern "C" int cpc_set_mfma_mode(int mode) {
    CPC_RETURN_IF(mode < 0 || mode > 4, CPC_ERR_ARG);
    cpc::g_mfma_mode = mode;
    return 0;
}
