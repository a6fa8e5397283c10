// Encoder layer 0:  conv0 (1 -> 256, k=10, s=5, p=3) + bias + ChannelNorm + ReLU, fused.
//
// Reference: cpc/model.py:83 (conv0), :50-58 (ChannelNorm), :100 (relu(norm(conv))).
//
// This is the one HBM-bound layer of the stack (21 MFLOP vs 4.2 MB of output per 1.28 s window, ~5 FLOP/B).  A wavefront
// owns 16 whole time steps: the 10-tap contraction (+ bias) runs on exact-f32 MFMAs with the samples read straight from
// global memory, all 256 channels of a step live in the 16 lanes of one row group (16 per lane), the mean / unbiased
// variance are in-lane sums + one DPP row reduction, and the normalised, rectified row is written exactly once into the
// channels-last (B, L0, C) activation.  Only mean and rstd (8 B per step) are kept for the backward pass, which recomputes
// the 10-tap conv instead of re-reading a 4 MB pre-norm tensor.
#include "cpc_common.h"
#include "cpc_internal.h"
#include "gemm_tile.h"

namespace cpc {

constexpr int K0 = 10, S0 = 5, P0 = 3;     // conv0 geometry, cpc/model.py:83
// 16-step groups per wave (cpc_set_conv0_tuning; a block of four waves covers 64 * groups steps): more amortise the 50
// operand loads of a wave's prologue, fewer leave a shorter tail when the grid is not a multiple of the resident waves
constexpr int C0_MAX_GPW = 8;
constexpr int C0_MAX_NS = S0 * 64 * C0_MAX_GPW + (K0 - S0);      // staged samples per block
static int g_conv0_gpw = 4;
static int g_conv0_nt = 0;          // 1: the activation rows leave as non-temporal stores (measured: plain stores are the faster
                                    // ones, alone -- tools/probe_store_pattern.hip: 6.6 against 5.9 TB/s -- and inside the step)

// The 16 bytes a lane stores of an H2 row when it holds channels 4 * slot .. 4 * slot + 3 of the row (slot = 0..63; any mapping of
// lanes to slots in which lane parity == slot parity and lane ^ 1 holds slot ^ 1): neighbouring lanes swap halves so that the
// even slot owns the 8 h pieces of its 8-channel group and the odd slot the 8 l pieces, and each lane stores 16 contiguous
// bytes at row + 16 * slot.  The pieces come from v_cvt_pk_f16_f32 on explicit products: hipcc forms v_fma_mixlo_f16 from
// (_Float16)(x * s), which issues at less than half the rate of a plain VALU instruction (10.8 against 4.7 cycles per SIMD at 4
// waves, tools/probe_valu_rate.hip) and then converts the same product a second time.
__device__ __forceinline__ unsigned pack_f16x2_rne(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIPEMU)
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return __builtin_bit_cast(unsigned, f16x2{(_Float16)a, (_Float16)b});
#endif
}
__device__ __forceinline__ float f16lo(unsigned p) { return (float)__builtin_bit_cast(f16x2, p).x; }
__device__ __forceinline__ float f16hi(unsigned p) { return (float)__builtin_bit_cast(f16x2, p).y; }
template <bool NT>
__device__ __forceinline__ void h2_store_slot(void* row, int slot, float x0, float x1, float x2, float x3, bool live) {
    // x0..x3: the four values ALREADY multiplied by the storage scale (the caller folds the power of two into its last FMA)
    const unsigned hw0 = pack_f16x2_rne(x0, x1), hw1 = pack_f16x2_rne(x2, x3);
    const unsigned lw0 = pack_f16x2_rne(x0 - f16lo(hw0), x1 - f16hi(hw0)), lw1 = pack_f16x2_rne(x2 - f16lo(hw1), x3 - f16hi(hw1));
    // even lane: {own h, own h, neighbour's h, neighbour's h};  odd lane: {neighbour's l, neighbour's l, own l, own l}
    const bool odd = slot & 1;
    const unsigned r0 = __builtin_bit_cast(unsigned, dpp_mov<0xB1>(__builtin_bit_cast(float, odd ? hw0 : lw0)));   // quad_perm [1,0,3,2]
    const unsigned r1 = __builtin_bit_cast(unsigned, dpp_mov<0xB1>(__builtin_bit_cast(float, odd ? hw1 : lw1)));
    f32x4 o;
    o.x = __builtin_bit_cast(float, odd ? r0 : hw0);
    o.y = __builtin_bit_cast(float, odd ? r1 : hw1);
    o.z = __builtin_bit_cast(float, odd ? lw0 : r0);
    o.w = __builtin_bit_cast(float, odd ? lw1 : r1);
    f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(row) + 16 * slot);
    if (live) {
        if constexpr (NT) __builtin_nontemporal_store(o, dst); else *dst = o;
    }
}

// One wavefront computes 16 time steps x 256 channels at a time on the matrix pipe: per 16 x 16 output tile three
// v_mfma_f32_16x16x4_f32 (exact fp32 products and sums) contract the 10 taps, the bias (tap 10, against a sample of 1) and a
// zero pad:  A[16 steps x 4] = waveform samples s[5 t - 3 + j] from the block's window, staged once in LDS, B[4 x 16 channels] = conv0.weight / bias from a transposed table in LDS.  Tile T of a
// lane (column n = lane & 15) is channel 64 (T >> 2) + 4 n + (T & 3), so a lane ends up with four runs of four consecutive
// channels for each of its four time steps 4 (lane >> 4) + r: ChannelNorm's sums over the 256 channels of a step are 16
// in-lane additions and one DPP row reduction over the 16 lanes of a row group (no cross-row traffic, no LDS at all), and a
// store instruction writes 256 contiguous bytes of each of four rows.  The VALU only normalises, rectifies and encodes --
// the scalar-FMA version of this kernel sat at the VALU issue limit (~105 VALU per 1 KB row, 60-65 us at B = 64 against 44 us
// of HBM time) -- and no LDS-read operand feeds packed fp32 arithmetic any more (the co-residency hazard of DESIGN.md 4.6).
// YK: storage of y -- 0 fp32, 1 H2 (two fp16 pieces scaled by scale_for_amax(*y_amax), cpc_common.h), 2 bf16
template <int YK, bool NT>
__global__ __launch_bounds__(256, 4) void conv0_fwd_kernel(
    const float* __restrict__ wave, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ nw, const float* __restrict__ nb, float* __restrict__ y,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, int L, int L0, const float* __restrict__ y_amax, int gpw) {
    __shared__ float aff[2][kC];                 // ChannelNorm weight / bias: 2 KB, re-read per use (ds_read_b128) instead of
    __shared__ float smp[C0_MAX_NS];             // living in 32 registers; the block's waveform window (zero-padded)
    __shared__ float wT[12][kC];                 // conv0.weight transposed [tap][channel], tap 10 = conv0.bias, tap 11 = 0: the
    const int b = blockIdx.y;                    // B operands of a lane are twelve 16-byte reads of this table (read from global
    const float* wb = wave + (long)b * L;        // memory they were 64 strided 4-byte gathers per lane and wave)
    // (H2 output: times the storage scale, a power of two -- max(fma(xhat, g s, b s), 0) == s max(fma(xhat, g, b), 0) exactly)
    const float sy = YK == 1 ? scale_for_amax(*y_amax) : 1.0f;
    aff[0][threadIdx.x] = nw[threadIdx.x] * sy;
    aff[1][threadIdx.x] = nb[threadIdx.x] * sy;
    wT[K0][threadIdx.x] = bias[threadIdx.x];
    wT[K0 + 1][threadIdx.x] = 0.f;
    for (int e = threadIdx.x; e < kC * K0; e += 256) wT[e % K0][e / K0] = w[e];
    // The window is staged ONCE, up front: vmcnt counts loads and stores in one in-order queue on gfx9, so a global load
    // inside the loop could only be waited for together with every store issued before it -- each 16-step group then waited
    // for the previous group's 16 KB to drain to HBM (6 us per group and wave; 80 us per launch however many waves).
    const int t00 = blockIdx.x * 64 * gpw, ns = S0 * 64 * gpw + (K0 - S0);
    for (int i = threadIdx.x; i < ns; i += 256) {
        const int sidx = t00 * S0 - P0 + i;
        const float v = wb[sidx < 0 ? 0 : (sidx < L ? sidx : L - 1)];       // unconditional load from a clamped index, then
        smp[i] = ((unsigned)sidx < (unsigned)L) ? v : 0.f;                   // a select (a predicated load costs a branch + vmcnt(0))
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int tg0 = (blockIdx.x * 4 + wv) * gpw * 16;
    for (int gi = 0; gi < gpw; ++gi) {
        const int tg = tg0 + gi * 16;                                    // first step of the group
        if (tg >= L0) break;                                             // wave-uniform
        // A operands: step tg + n, tap 4 kk + kq (taps 10 / 11: the bias's 1 / 0)
        float a[3];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int j = 4 * kk + kq;
            const float v = smp[(tg - t00 + n) * S0 + (j < K0 ? j : 0)];
            a[kk] = j < K0 ? v : (j == K0 ? 1.0f : 0.f);
        }
        // B operands: wt[T][kk] = tap 4 kk + kq of channel ch(T) = 64 (T >> 2) + 4 n + (T & 3).  Re-read from LDS for every group
        // (twelve 16-byte reads) rather than held across the loop: 48 registers fewer while the epilogue runs, i.e. four waves
        // per SIMD instead of three -- at B = 64 and 4 groups per wave the 1024 blocks are then all resident at once.
        float wt[16][3];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(&wT[4 * kk + kq][64 * q + 4 * n]);
                wt[4 * q + 0][kk] = v.x; wt[4 * q + 1][kk] = v.y; wt[4 * q + 2][kk] = v.z; wt[4 * q + 3][kk] = v.w;
            }
        f32x4 x[16];
#pragma unroll
        for (int T = 0; T < 16; ++T) {
            x[T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) x[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], wt[T][kk], x[T], 0, 0, 0);
        }
        // x[T][r]: step tg + 4 kq + r, channel ch(T).  Mean / unbiased variance over the 256 channels of a step.
        float mu[4], rs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s0 = (x[0][r] + x[1][r]) + (x[2][r] + x[3][r]), s1 = (x[4][r] + x[5][r]) + (x[6][r] + x[7][r]);
            float s2 = (x[8][r] + x[9][r]) + (x[10][r] + x[11][r]), s3 = (x[12][r] + x[13][r]) + (x[14][r] + x[15][r]);
            mu[r] = row16_sum((s0 + s1) + (s2 + s3)) * (1.0f / kC);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int T = 0; T < 16; T += 2) {
                x[T][r] -= mu[r];
                x[T + 1][r] -= mu[r];
                v0 = fmaf(x[T][r], x[T][r], v0);
                v1 = fmaf(x[T + 1][r], x[T + 1][r], v1);
            }
            rs[r] = __builtin_amdgcn_rsqf(row16_sum(v0 + v1) * (1.0f / (kC - 1)) + kNormEps);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = tg + 4 * kq + r;
            const bool live = t < L0;
            const long row = (long)b * L0 + (live ? t : tg);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 g4 = *reinterpret_cast<const float4*>(&aff[0][64 * q + 4 * n]);
                const float4 n4 = *reinterpret_cast<const float4*>(&aff[1][64 * q + 4 * n]);
                const float gam[4] = {g4.x, g4.y, g4.z, g4.w}, bet[4] = {n4.x, n4.y, n4.z, n4.w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(relu_in(fmaf(x[4 * q + e][r] * rs[r], gam[e], bet[e]), sy), 0.f);
                if constexpr (YK == 1) {
                    h2_store_slot<NT>(y + row * kC, 16 * q + n, o[0], o[1], o[2], o[3], live);
                } else if constexpr (YK == 2) {
                    const unsigned long long w2 = (unsigned long long)(bf16_rne(o[0]) | ((unsigned)bf16_rne(o[1]) << 16)) |
                                                  ((unsigned long long)(bf16_rne(o[2]) | ((unsigned)bf16_rne(o[3]) << 16)) << 32);
                    unsigned long long* dst = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned short*>(y) + row * kC +
                                                                                    64 * q + 4 * n);
                    if (live) {
                        if constexpr (NT) __builtin_nontemporal_store(w2, dst); else *dst = w2;
                    }
                } else {
                    f32x4* dst = reinterpret_cast<f32x4*>(y + row * kC + 64 * q + 4 * n);
                    if (live) {      // streamed once, 268 MB per launch at B = 64 (>> L2)
                        if constexpr (NT) __builtin_nontemporal_store(f32x4{o[0], o[1], o[2], o[3]}, dst); else *dst = f32x4{o[0], o[1], o[2], o[3]};
                    }
                }
            }
            if (n == 0 && live) { mean_out[row] = mu[r]; rstd_out[row] = rs[r]; }
        }
    }
}

// Backward of layer 0, on the layout of the forward kernel above: a wavefront takes 16 time steps x 256 channels at a time,
// recomputes conv0 (+ bias) with the SAME three exact-f32 MFMAs per 16 x 16 tile -- so x, and with the saved (mean, rstd) xhat
// and the ReLU mask, are bit-identical to the forward's -- applies relu' / ChannelNorm backward to the incoming dY (gradient
// w.r.t. the post-ReLU activation, produced by conv1's data gradient), and forms the gradients of conv0.weight (256 x 10) and
// conv0.bias as a second product on the matrix pipe:  D[16 channels x 16 taps] += dX^T[16 x 4 steps] . S[4 steps x 16 taps] per
// tile and step quad, S = the steps' sample windows with a column of ones at tap 10 (the bias gradient) -- the accumulator
// registers of the recomputation, turned into dx in place, ARE the A operand.  The VALU is left with the norm backward itself
// (14 operations per element; the scalar kernel of rounds 1-2 spent 32, 20 of them FMAs of the two 10-tap contractions) and the
// per-channel sums for batchNorm0.weight / .bias.  The waveform needs no gradient (cpc/train.py:81-87), so there is no dgrad.
// Each block reduces its waves through LDS and writes one partial row [13][256] to `part`; a second tiny kernel sums the partial
// rows in a fixed order (deterministic, no float atomics).
constexpr int C0B_GPW = 4;                 // 16-step groups per wave
constexpr int C0B_TT = 64 * C0B_GPW;       // time steps per block of four waves
constexpr int C0B_NS = S0 * C0B_TT + (K0 - S0);
constexpr int C0_NACC = K0 + 3;            // 10 weight taps, conv bias, norm weight, norm bias

// DYB: dy arrives as bf16 (the bf16-storage variant) instead of fp32
template <bool DYB>
__global__ __launch_bounds__(256, 2) void conv0_bwd_kernel(
    const float* __restrict__ wave, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ nw, const float* __restrict__ nb, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, const float* __restrict__ dy, float* __restrict__ part,
    int L, int L0) {
    // One LDS block, two lives.  In the loop: operand tables -- wT[12][kC] (conv0.weight transposed [tap][channel], tap 10 = bias,
    // tap 11 = 0), aff[2][kC] (ChannelNorm weight / bias), smp (the block's waveform window, zero-padded) -- and gb, the running
    // sums for batchNorm0.weight / .bias, one private slice per (wave, lane group): [4][4][2][kC] (in registers they would be the
    // 32 that push the kernel into spilling).  At the end: red[4][13][kC], the waves' sums.
    constexpr int kTab = 12 * kC + 2 * kC + ((C0B_NS + 15) & ~15);
    constexpr int kGb = 4 * 4 * 2 * kC;
    constexpr int kRed = 4 * C0_NACC * kC;
    __shared__ float lds[kTab + kGb > kRed ? kTab + kGb : kRed];
    float (*wT)[kC] = reinterpret_cast<float (*)[kC]>(lds);
    float (*aff)[kC] = reinterpret_cast<float (*)[kC]>(lds + 12 * kC);
    float* smp = lds + 14 * kC;
    float (*red)[C0_NACC][kC] = reinterpret_cast<float (*)[C0_NACC][kC]>(lds);
    const int b = blockIdx.y;
    const int t00 = blockIdx.x * C0B_TT;
    const float* wb = wave + (long)b * L;
    aff[0][threadIdx.x] = nw[threadIdx.x];
    aff[1][threadIdx.x] = nb[threadIdx.x];
    wT[K0][threadIdx.x] = bias[threadIdx.x];
    wT[K0 + 1][threadIdx.x] = 0.f;
    for (int e = threadIdx.x; e < kC * K0; e += 256) wT[e % K0][e / K0] = w[e];
    for (int i = threadIdx.x; i < C0B_NS; i += 256) {
        const int sidx = t00 * S0 - P0 + i;
        const float v = wb[sidx < 0 ? 0 : (sidx < L ? sidx : L - 1)];       // unconditional load from a clamped index, then a select
        smp[i] = ((unsigned)sidx < (unsigned)L) ? v : 0.f;
    }
    for (int i = threadIdx.x; i < kGb; i += 256) lds[kTab + i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    float* gb = lds + kTab + (wv * 4 + kq) * 2 * kC;     // this lane group's sums: [0][c] d batchNorm0.weight, [1][c] d batchNorm0.bias
    // dw[T][e]: d conv0.weight / bias of channel ch(4 kq + e, T), tap n (the D layout of the second product: column = lane & 15 =
    // tap, row 4 kq + e = channel of the tile).  ch(m, T) = 64 (T >> 2) + 4 m + (T & 3).
    f32x4 dw[16];
#pragma unroll
    for (int T = 0; T < 16; ++T) dw[T] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tg0 = t00 + wv * C0B_GPW * 16;
    for (int gi = 0; gi < C0B_GPW; ++gi) {
        const int tg = tg0 + gi * 16;
        if (tg >= L0) break;                                             // wave-uniform
        // ---- conv0 again: the forward kernel's operands and instruction sequence
        float a[3];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int j = 4 * kk + kq;
            const float v = smp[(tg - t00 + n) * S0 + (j < K0 ? j : 0)];
            a[kk] = j < K0 ? v : (j == K0 ? 1.0f : 0.f);
        }
        f32x4 x[16];
        {
            float wt[16][3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(&wT[4 * kk + kq][64 * q + 4 * n]);
                    wt[4 * q + 0][kk] = v.x; wt[4 * q + 1][kk] = v.y; wt[4 * q + 2][kk] = v.z; wt[4 * q + 3][kk] = v.w;
                }
#pragma unroll
            for (int T = 0; T < 16; ++T) {
                x[T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) x[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], wt[T][kk], x[T], 0, 0, 0);
            }
        }
        // ---- the lane's four steps in two halves (r = 0, 1 and r = 2, 3): a step's reductions run over channels only, so the
        // halves are independent, and only half of dy (32 registers) is in flight at a time beside x and the gradient tiles
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            // what this half reads from memory: dy of 2 steps x 16 channels, their mean / rstd.  Steps beyond L0 read a valid
            // row and are masked out below.
            float4 dyv[2][4];                                            // [r2][q]: channels 64 q + 4 n .. of step tg + 4 kq + 2 hf + r2
            float mu[2], rs[2];
            bool live[2];
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                const int t = tg + 4 * kq + 2 * hf + r2;
                live[r2] = t < L0;
                const long row = (long)b * L0 + (live[r2] ? t : tg);
                mu[r2] = mean_in[row];
                rs[r2] = rstd_in[row];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if constexpr (DYB) {
                        const uint2 gbits = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(dy) + row * kC + 64 * q + 4 * n);
                        dyv[r2][q] = make_float4(bf16_val((unsigned short)(gbits.x & 0xFFFFu)), bf16_val((unsigned short)(gbits.x >> 16)),
                                                 bf16_val((unsigned short)(gbits.y & 0xFFFFu)), bf16_val((unsigned short)(gbits.y >> 16)));
                    } else {
                        dyv[r2][q] = *reinterpret_cast<const float4*>(dy + row * kC + 64 * q + 4 * n);
                    }
                }
            }
            // relu' and ChannelNorm backward; x[T][r] becomes xhat, dyv becomes dxhat, then x becomes dx
            float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 g4 = *reinterpret_cast<const float4*>(&aff[0][64 * q + 4 * n]);
                const float4 n4 = *reinterpret_cast<const float4*>(&aff[1][64 * q + 4 * n]);
                float4 dg = *reinterpret_cast<const float4*>(gb + 64 * q + 4 * n);
                float4 db = *reinterpret_cast<const float4*>(gb + kC + 64 * q + 4 * n);
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int T = 4 * q + e, r = 2 * hf + r2;
                        const float xh = (x[T][r] - mu[r2]) * rs[r2];
                        const float yv = relu_in(fmaf(xh, f4c(g4, e), f4c(n4, e)));
                        const float dyh = (live[r2] && yv > 0.f) ? f4c(dyv[r2][q], e) : 0.f;     // relu'
                        (&dg.x)[e] = fmaf(dyh, xh, (&dg.x)[e]);                                  // d batchNorm0.weight
                        (&db.x)[e] += dyh;                                                       // d batchNorm0.bias
                        const float dxh = dyh * f4c(g4, e);
                        s1[r2] += dxh;
                        s2[r2] = fmaf(dxh, xh, s2[r2]);
                        x[T][r] = xh;
                        (&dyv[r2][q].x)[e] = dxh;
                    }
                *reinterpret_cast<float4*>(gb + 64 * q + 4 * n) = dg;
                *reinterpret_cast<float4*>(gb + kC + 64 * q + 4 * n) = db;
            }
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                const int r = 2 * hf + r2;
                const float m1 = row16_sum(s1[r2]) * (1.0f / kC), m2 = row16_sum(s2[r2]) * (1.0f / (kC - 1));
#pragma unroll
                for (int T = 0; T < 16; ++T)
                    x[T][r] = live[r2] ? rs[r2] * (f4c(dyv[r2][T >> 2], T & 3) - m1 - x[T][r] * m2) : 0.f;
                // d conv0.weight / bias: D[channel][tap] += dx[step][channel] * S[step][tap], steps r, 4 + r, 8 + r, 12 + r per MFMA
                const float sv = smp[(tg - t00 + 4 * kq + r) * S0 + (n < K0 ? n : 0)];
                const float bop = n < K0 ? sv : (n == K0 ? 1.0f : 0.f);   // B operand: tap n of step tg + 4 kq + r
#pragma unroll
                for (int T = 0; T < 16; ++T) dw[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[T][r], bop, dw[T], 0, 0, 0);
            }
        }
    }
    // ---- the block's sums.  d batchNorm0.*: the four lane groups of a wave hold different steps of the same channels; lane
    // (n, kq) folds their slices for channels 64 kq + 4 n ..
    float4 dgs = make_float4(0.f, 0.f, 0.f, 0.f), dbs = dgs;
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        const float* g2 = lds + kTab + (wv * 4 + k2) * 2 * kC + 64 * kq + 4 * n;
        const float4 u = *reinterpret_cast<const float4*>(g2), v = *reinterpret_cast<const float4*>(g2 + kC);
        dgs.x += u.x; dgs.y += u.y; dgs.z += u.z; dgs.w += u.w;
        dbs.x += v.x; dbs.y += v.y; dbs.z += v.z; dbs.w += v.w;
    }
    __syncthreads();                                                     // everybody is done with the tables and the slices
    *reinterpret_cast<float4*>(&red[wv][K0 + 1][64 * kq + 4 * n]) = dgs;
    *reinterpret_cast<float4*>(&red[wv][K0 + 2][64 * kq + 4 * n]) = dbs;
    if (n <= K0) {
#pragma unroll
        for (int T = 0; T < 16; ++T)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wv][n][64 * (T >> 2) + 4 * (4 * kq + e) + (T & 3)] = dw[T][e];
    }
    __syncthreads();
    float* prow = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (C0_NACC * kC);
    for (int i = threadIdx.x; i < C0_NACC * kC; i += 256) {
        const int j = i / kC, cc = i - j * kC;
        prow[i] = (red[0][j][cc] + red[1][j][cc]) + (red[2][j][cc] + red[3][j][cc]);
    }
}

// out[g][i] = sum of rows [g*rows_per_group, (g+1)*rows_per_group) of part (row length n).
// Block = 32 columns x 8 row lanes; fixed summation order (deterministic).
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* __restrict__ part, int nrows,
                                                       int n, int rows_per_group,
                                                       float* __restrict__ out, long part_gs, long out_gs) {
    __shared__ float red[8][33];
    part += (long)blockIdx.z * part_gs;                  // blockIdx.z: problem of a group of equally shaped reductions
    out += (long)blockIdx.z * out_gs;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;
    const int r0 = blockIdx.y * rows_per_group;
    const int r1 = min(nrows, r0 + rows_per_group);
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        int r = r0 + ty;
        for (; r + 8 < r1; r += 16) {
            s0 += part[(long)r * n + i];
            s1 += part[(long)(r + 8) * n + i];
        }
        if (r < r1) s0 += part[(long)r * n + i];
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && i < n) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += red[q][tx];
        out[(long)blockIdx.y * n + i] = s;
    }
}

// The same for up to kRowsSumMaxJobs independent reductions per launch (blockIdx.z = job): stage 0 sums each job's row
// groups into its tmp (or straight into out when there is one group), stage 1 folds the groups.
struct RowsSumBatch {
    RowsSumJob j[kRowsSumMaxJobs];
    int groups[kRowsSumMaxJobs], rpg[kRowsSumMaxJobs];
};
__global__ __launch_bounds__(256) void rows_sum_multi_kernel(RowsSumBatch b, int stage) {
    __shared__ float red[8][33];
    const RowsSumJob& jb = b.j[blockIdx.z];
    const int groups = b.groups[blockIdx.z], n = jb.n;
    if ((int)blockIdx.y >= (stage == 0 ? groups : 1) || (stage == 1 && groups == 1) || (int)blockIdx.x * 32 >= n) return;
    const float* part = stage == 0 ? jb.part : jb.tmp;
    float* out = (stage == 1 || groups == 1) ? jb.out : jb.tmp;
    const int nrows = stage == 0 ? jb.nrows : groups;
    const int rpg = stage == 0 ? b.rpg[blockIdx.z] : groups;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;
    const int r0 = blockIdx.y * rpg;
    const int r1 = min(nrows, r0 + rpg);
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        int r = r0 + ty;
        for (; r + 8 < r1; r += 16) {
            s0 += part[(long)r * n + i];
            s1 += part[(long)(r + 8) * n + i];
        }
        if (r < r1) s0 += part[(long)r * n + i];
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && i < n) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += red[q][tx];
        out[(long)blockIdx.y * n + i] = s;
    }
}

__global__ __launch_bounds__(256) void conv0_scatter_kernel(const float* __restrict__ sum,
                                                            float* __restrict__ dW0,
                                                            float* __restrict__ dB0,
                                                            float* __restrict__ dNW0,
                                                            float* __restrict__ dNB0) {
    const int c = threadIdx.x;
#pragma unroll
    for (int j = 0; j < K0; ++j) dW0[c * K0 + j] = sum[j * kC + c];   // conv0.weight is (256,1,10)
    dB0[c] = sum[K0 * kC + c];
    dNW0[c] = sum[(K0 + 1) * kC + c];
    dNB0[c] = sum[(K0 + 2) * kC + c];
}

// sums `nrows` rows of length n in `part` into out[0:n]; `tmp` holds >= kRowsSumGroups*n floats.
int rows_sum(const float* part, int nrows, int n, float* tmp, float* out, hipStream_t stream, int G, long part_gs, long tmp_gs,
             long out_gs) {
    if (nrows <= 0) {
        for (int g = 0; g < G; ++g) (void)hipMemsetAsync(out + g * out_gs, 0, sizeof(float) * n, stream);
        return 0;
    }
    int groups = nrows > 64 ? kRowsSumGroups : 1;
    const int rpg = cdiv(nrows, groups);
    groups = cdiv(nrows, rpg);
    if (groups == 1) {
        hipLaunchKernelGGL(rows_sum_kernel, dim3(cdiv(n, 32), 1, G), dim3(256), 0, stream, part, nrows, n, nrows, out, part_gs, out_gs);
    } else {
        hipLaunchKernelGGL(rows_sum_kernel, dim3(cdiv(n, 32), groups, G), dim3(256), 0, stream, part, nrows, n, rpg, tmp, part_gs, tmp_gs);
        hipLaunchKernelGGL(rows_sum_kernel, dim3(cdiv(n, 32), 1, G), dim3(256), 0, stream, tmp, groups, n, groups, out, tmp_gs, out_gs);
    }
    CPC_LAUNCH_CHECK();
    return 0;
}

// Same arithmetic and summation order as rows_sum for every job (so results do not depend on which one ran).
int rows_sum_multi(const RowsSumJob* jobs, int njobs, hipStream_t stream) {
    if (njobs <= 0) return 0;
    if (njobs > kRowsSumMaxJobs) return CPC_ERR_ARG;
    RowsSumBatch b;
    int gmax = 1, nmax = 0;
    bool two = false;
    for (int q = 0; q < njobs; ++q) {
        b.j[q] = jobs[q];
        if (jobs[q].nrows <= 0 || jobs[q].n <= 0) return CPC_ERR_SHAPE;
        int groups = jobs[q].nrows > 64 ? kRowsSumGroups : 1;
        const int rpg = cdiv(jobs[q].nrows, groups);
        groups = cdiv(jobs[q].nrows, rpg);
        b.groups[q] = groups; b.rpg[q] = rpg;
        gmax = std::max(gmax, groups); nmax = std::max(nmax, jobs[q].n);
        two = two || groups > 1;
    }
    hipLaunchKernelGGL(rows_sum_multi_kernel, dim3(cdiv(nmax, 32), gmax, njobs), dim3(256), 0, stream, b, 0);
    if (two) hipLaunchKernelGGL(rows_sum_multi_kernel, dim3(cdiv(nmax, 32), 1, njobs), dim3(256), 0, stream, b, 1);
    CPC_LAUNCH_CHECK();
    return 0;
}

}  // namespace cpc

using namespace cpc;

extern "C" int cpc_set_conv0_tuning(int groups, int nontemporal) {
    CPC_RETURN_IF(groups < 1 || groups > cpc::C0_MAX_GPW || nontemporal < 0 || nontemporal > 1, CPC_ERR_ARG);
    cpc::g_conv0_gpw = groups;
    cpc::g_conv0_nt = nontemporal;
    return 0;
}

#define CONV0_LAUNCH(YK, st, y, mean, rstd, amax)                                                                             \
    do {                                                                                                                      \
        const dim3 grid(cdiv(L0, 64 * cpc::g_conv0_gpw), B);                                                                  \
        if (cpc::g_conv0_nt)                                                                                                  \
            hipLaunchKernelGGL((cpc::conv0_fwd_kernel<YK, true>), grid, dim3(256), 0, st, wave, w, bias, nw, nb, y, mean,     \
                               rstd, L, L0, amax, cpc::g_conv0_gpw);                                                          \
        else                                                                                                                  \
            hipLaunchKernelGGL((cpc::conv0_fwd_kernel<YK, false>), grid, dim3(256), 0, st, wave, w, bias, nw, nb, y, mean,    \
                               rstd, L, L0, amax, cpc::g_conv0_gpw);                                                          \
    } while (0)

extern "C" int cpc_conv0_forward(const float* wave, const float* w, const float* bias,
                                 const float* nw, const float* nb, float* y, float* mean,
                                 float* rstd, int B, int L, void* stream) {
    return cpc_conv0_forward_h2(wave, w, bias, nw, nb, y, mean, rstd, nullptr, B, L, stream);
}

// y_amax != NULL: y is written in H2 storage (two fp16 pieces per element, cpc_common.h) scaled by
// scale_for_amax(*y_amax); *y_amax must bound |y| (the ChannelNorm bound, norm_bound_kernel).  NULL: fp32.
extern "C" int cpc_conv0_forward_h2(const float* wave, const float* w, const float* bias, const float* nw,
                                    const float* nb, void* y, float* mean, float* rstd, const float* y_amax, int B,
                                    int L, void* stream) {
    CPC_RETURN_IF(B <= 0 || L < K0 - 2 * P0, CPC_ERR_SHAPE);
    const int L0 = conv_out_len(L, K0, S0, P0);
    if (y_amax)
        CONV0_LAUNCH(1, (hipStream_t)stream, reinterpret_cast<float*>(y), mean, rstd, y_amax);
    else
        CONV0_LAUNCH(0, (hipStream_t)stream, reinterpret_cast<float*>(y), mean, rstd, y_amax);
    CPC_LAUNCH_CHECK();
    return 0;
}

namespace cpc {
// bf16-storage variant (mode 4): y as bf16 (B, L0, 256)
int conv0_forward_bf16(const float* wave, const float* w, const float* bias, const float* nw, const float* nb, void* y,
                       float* mean, float* rstd, int B, int L, hipStream_t st) {
    CPC_RETURN_IF(B <= 0 || L < K0 - 2 * P0, CPC_ERR_SHAPE);
    const int L0 = conv_out_len(L, K0, S0, P0);
    CONV0_LAUNCH(2, st, reinterpret_cast<float*>(y), mean, rstd, (const float*)nullptr);
    CPC_LAUNCH_CHECK();
    return 0;
}
}  // namespace cpc

extern "C" long cpc_conv0_backward_scratch_floats(int B, int L) {
    const int L0 = conv_out_len(L, K0, S0, P0);
    return ((long)cdiv(L0, C0B_TT) * B + kRowsSumGroups + 1) * (C0_NACC * kC);
}

// grads: dW0 (256*10), dB0 (256), dNW0 (256), dNB0 (256) -- overwritten, not accumulated.
extern "C" int cpc_conv0_backward(const float* wave, const float* w, const float* bias,
                                  const float* nw, const float* nb, const float* mean,
                                  const float* rstd, const float* dy, float* scratch, float* dW0,
                                  float* dB0, float* dNW0, float* dNB0, int B, int L, void* stream) {
    return cpc::conv0_backward(wave, w, bias, nw, nb, mean, rstd, dy, 0, scratch, dW0, dB0, dNW0, dNB0, B, L, (hipStream_t)stream);
}

// dy_bf16: dy is a bf16 tensor (the bf16-storage variant, mode 4)
int cpc::conv0_backward(const float* wave, const float* w, const float* bias, const float* nw, const float* nb,
                        const float* mean, const float* rstd, const void* dy_any, int dy_bf16, float* scratch, float* dW0,
                        float* dB0, float* dNW0, float* dNB0, int B, int L, hipStream_t stream) {
    const float* dy = reinterpret_cast<const float*>(dy_any);
    CPC_RETURN_IF(B <= 0 || L < K0 - 2 * P0, CPC_ERR_SHAPE);
    const int L0 = conv_out_len(L, K0, S0, P0);
    const int nblk = cdiv(L0, C0B_TT) * B;
    const int n = C0_NACC * kC;
    float* part = scratch;                        // [nblk][13][256]
    float* tmp = scratch + (long)nblk * n;        // rows for the first reduction level
    float* sum = tmp + (long)kRowsSumGroups * n;  // final [13][256]
    hipStream_t st = (hipStream_t)stream;
    if (dy_bf16)
        hipLaunchKernelGGL(conv0_bwd_kernel<true>, dim3(cdiv(L0, C0B_TT), B), dim3(256), 0, st, wave, w, bias,
                           nw, nb, mean, rstd, dy, part, L, L0);
    else
    hipLaunchKernelGGL(conv0_bwd_kernel<false>, dim3(cdiv(L0, C0B_TT), B), dim3(256), 0, st, wave, w, bias,
                       nw, nb, mean, rstd, dy, part, L, L0);
    CPC_LAUNCH_CHECK();
    int rc = rows_sum(part, nblk, n, tmp, sum, st);
    if (rc) return rc;
    hipLaunchKernelGGL(conv0_scatter_kernel, dim3(1), dim3(256), 0, st, sum, dW0, dB0, dNW0, dNB0);
    CPC_LAUNCH_CHECK();
    return 0;
}
