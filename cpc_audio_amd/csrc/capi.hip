// Miscellaneous C-ABI entry points.
#include "cpc_common.h"

extern "C" int cpc_abi_version(void) { return 1; }
