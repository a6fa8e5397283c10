// Miscellaneous C-ABI entry points.
#include "cpc_common.h"
#include "cpc_internal.h"

namespace cpc {
int g_mfma_mode = 2;

// Events that order a second stream against the caller's inside the *_streams entry points (timing disabled; created
// once per device and reused: a wait captures the record that precedes it, so re-recording later is harmless).
hipEvent_t* stream_events() {
    static hipEvent_t ev[16][kStreamEvents];
    static bool made[16] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!made[dev]) {
        for (int i = 0; i < kStreamEvents; ++i)
            if (hipEventCreateWithFlags(&ev[dev][i], hipEventDisableTiming) != hipSuccess) return nullptr;
        made[dev] = true;
    }
    return ev[dev];
}
}  // namespace cpc

extern "C" int cpc_abi_version(void) { return 5; }

extern "C" int cpc_set_mfma_mode(int mode) {
    CPC_RETURN_IF(mode != 0 && mode != 1 && mode != 2, CPC_ERR_ARG);
    cpc::g_mfma_mode = mode;
    return 0;
}
