// Miscellaneous C-ABI entry points.
#include "cpc_common.h"
#include "cpc_internal.h"

namespace cpc { int g_mfma_mode = 2; }

extern "C" int cpc_abi_version(void) { return 4; }

extern "C" int cpc_set_mfma_mode(int mode) {
    CPC_RETURN_IF(mode != 0 && mode != 1 && mode != 2, CPC_ERR_ARG);
    cpc::g_mfma_mode = mode;
    return 0;
}
