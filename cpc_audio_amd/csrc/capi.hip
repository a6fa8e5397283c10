// Miscellaneous C-ABI entry points.
#include "cpc_common.h"
#include "cpc_internal.h"

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
    for (int i = 0; i < kStreamEvents; ++i)
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(ev[j]);
            delete[] ev;
            return nullptr;
        }
    pools[{dev, st}] = ev;
    return ev;
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

extern "C" int cpc_abi_version(void) { return 11; }

// Device-side error flags of the current device, accumulated since they were last cleared:
//   bit 0  CPC_DEVERR_GRU_POLL_TIMEOUT   a workgroup of the persistent recurrence gave up waiting for another one (its
//                                        outputs carry NaN from there on)
//   bit 1  CPC_DEVERR_NEGATIVE_INDEX     cpc_nce_prepare was given a draw outside [0,B) x [0,S) (clamped)
// Synchronises with the device (two 4-byte reads): for logging points and tests, not for the step path.
// Returns the mask (>= 0), or a negative number if it cannot be read.
extern "C" int cpc_device_error_flags(int clear) {
    unsigned a = 0, b = 0;
    if (cpc::gru_error_flag_fetch(clear, &a) != 0 || cpc::nce_error_flag_fetch(clear, &b) != 0) return -1;
    return (int)((a ? 1u : 0u) | (b ? 2u : 0u));
}

extern "C" int cpc_get_mfma_mode(void) { return cpc::g_mfma_mode; }

extern "C" int cpc_set_mfma_mode(int mode) {
    CPC_RETURN_IF(mode < 0 || mode > 4, CPC_ERR_ARG);
    cpc::g_mfma_mode = mode;
    return 0;
}
