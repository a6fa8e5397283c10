// Shared device helpers for the CPC hot-path kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cpc_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace cpc {

constexpr int kC = 256;        // encoder channels == GRU hidden == head width (north-star config)
constexpr int kCLog2 = 8;
constexpr float kNormEps = 1e-5f;   // ChannelNorm epsilon, cpc/model.py:29

// status codes returned by every extern "C" entry point (include/cpc_hip.h)
#define CPC_RETURN_IF(cond, code) \
    do { if (cond) return (code); } while (0)
#define CPC_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return 1000 + (int)e__;      \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

// sum over the 32 lanes of each half-wave (lanes 0-31 / 32-63 stay separate)
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

// sum over each group of 16 consecutive lanes
__device__ __forceinline__ float quarter_wave_sum(float v) {
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 32));
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 1));
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// output length of a strided conv (cpc/model.py:83-92 geometry)
static inline int conv_out_len(int lin, int k, int s, int p) { return (lin + 2 * p - k) / s + 1; }

}  // namespace cpc
