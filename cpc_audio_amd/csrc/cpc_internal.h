// Internal (non-ABI) helpers shared between translation units of libcpc_hip.
#pragma once
#include "cpc_common.h"

namespace cpc {

// out[0:n] = sum over `nrows` rows of `part` (row length n), summed in a fixed order.
// tmp must hold 64*n floats.
int rows_sum(const float* part, int nrows, int n, float* tmp, float* out, hipStream_t stream);

}  // namespace cpc
