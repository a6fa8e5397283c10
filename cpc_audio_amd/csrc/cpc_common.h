// Shared device helpers for the CPC hot-path kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cpc_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

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

// Wave-level sums on the DPP data path.  __shfl_xor lowers to ds_bpermute_b32 (an LDS-crossbar round
// trip plus a full lgkmcnt(0) wait per step); DPP modifiers run at VALU rate with no LDS involvement:
//   quad_perm [1,0,3,2] / [2,3,0,1]  -> totals of each 4 lanes
//   row_half_mirror, row_mirror      -> totals of each 8, then each row of 16 lanes (in every lane)
// and the four row totals are combined through v_readlane (scalar registers).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {      // sum over each group of 16 consecutive lanes
    v += dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);      // row_half_mirror
    v += dpp_mov<0x140>(v);      // row_mirror
    return v;
}
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// sum over the 32 lanes of each half-wave (lanes 0-31 / 32-63 stay separate)
__device__ __forceinline__ float half_wave_sum(float v) {
    v = row16_sum(v);
    const float lo = lane_value(v, 0) + lane_value(v, 16);
    const float hi = lane_value(v, 32) + lane_value(v, 48);
    return (threadIdx.x & 32) ? hi : lo;
}

// sum over each group of 16 consecutive lanes
__device__ __forceinline__ float quarter_wave_sum(float v) { return row16_sum(v); }

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
