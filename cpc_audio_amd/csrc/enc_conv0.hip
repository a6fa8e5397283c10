// Encoder layer 0:  conv0 (1 -> 256, k=10, s=5, p=3) + bias + ChannelNorm + ReLU, fused.
//
// Reference: cpc/model.py:83 (conv0), :50-58 (ChannelNorm), :100 (relu(norm(conv))).
//
// This is the one HBM-bound layer of the stack (21 MFLOP vs 4.2 MB of output per
// 1.28 s window, ~5 FLOP/B): the waveform window of a tile is staged once in LDS
// (coalesced contiguous read), every wave owns whole time steps so all 256
// channels of a step live in one wavefront (4 per lane), the mean / unbiased
// variance are wavefront-shuffle reductions, and the normalised, rectified row is
// written exactly once as one coalesced 1 KB row of the channels-last (B, L0, C)
// activation.  Only mean and rstd (8 B per step) are kept for the backward pass,
// which recomputes the 10-tap conv instead of re-reading a 4 MB pre-norm tensor.
//
// Measured (B = 64, 276 MB per launch): 60-65 us in the step = 4.3-4.6 TB/s (a plain fill reaches 6.9 TB/s on the same box).
// Ablation: 43 us without the activation stores -- the loop sits near the VALU issue limit (two DPP+readlane wave
// reductions, the normalise/affine/ReLU ops, 40 FMAs per 1 KB row); shuffles (ds_bpermute) were replaced by DPP,
// non-temporal stores are used for the streamed output.  The FMAs are scalar on purpose: issued as 20 v_pk_fma_f32 (round 1)
// the kernel changed its results beside another train loop's 16-bit-MFMA kernels (see the kernel).
#include "cpc_common.h"
#include "cpc_internal.h"
#include "gemm_tile.h"

namespace cpc {

constexpr int K0 = 10, S0 = 5, P0 = 3;     // conv0 geometry, cpc/model.py:83
constexpr int C0_TT = 128;                 // time steps per block (32 per wave, two at a time)
constexpr int C0_NS = S0 * C0_TT + (K0 - S0);   // staged samples per block

// YK: storage of y -- 0 fp32, 1 H2 (two fp16 pieces scaled by scale_for_amax(*y_amax), cpc_common.h), 2 bf16
template <int YK>
__global__ __launch_bounds__(256) void conv0_fwd_kernel(
    const float* __restrict__ wave, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ nw, const float* __restrict__ nb, float* __restrict__ y,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, int L, int L0, const float* __restrict__ y_amax) {
    __shared__ float smp[C0_NS];
    __shared__ float wT[K0][kC];                 // conv0.weight transposed: coalesced global read, float4 LDS reads
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * C0_TT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* wb = wave + (long)b * L;
    const int s_begin = t0 * S0 - P0;
    for (int i = tid; i < C0_NS; i += 256) {
        int s = s_begin + i;
        smp[i] = ((unsigned)s < (unsigned)L) ? wb[s] : 0.f;
    }
    for (int e = tid; e < kC * K0; e += 256) wT[e % K0][e / K0] = w[e];
    __syncthreads();
    // Scalar fp32 arithmetic on purpose (and -fno-slp-vectorize for this file, build.py): with the 40 FMAs of a time step issued
    // as 20 v_pk_fma_f32 whose broadcast operand is an LDS-read sample, this kernel -- like conv0_bwd_kernel before it --
    // returned different bits whenever a 16-bit-MFMA GEMM kernel shared the chip with it (tests/test_gpu_corun.py; inside one
    // train step nothing runs beside it, two train loops on one device do).  1/sqrt is the single v_rsq_f32.
    float4 wj[K0];
    const int c = lane * 4;
#pragma unroll
    for (int j = 0; j < K0; ++j) wj[j] = *reinterpret_cast<const float4*>(&wT[j][c]);
    const float4 b4 = *reinterpret_cast<const float4*>(bias + c);
    const float4 g4 = *reinterpret_cast<const float4*>(nw + c);
    const float4 n4 = *reinterpret_cast<const float4*>(nb + c);
    const float sy = YK == 1 ? scale_for_amax(*y_amax) : 1.0f;
    auto finish = [&](const float4& x, float mu, float rs, long row) __attribute__((always_inline)) {
        float4 o;
        o.x = fmaxf(fmaf((x.x - mu) * rs, g4.x, n4.x), 0.f);
        o.y = fmaxf(fmaf((x.y - mu) * rs, g4.y, n4.y), 0.f);
        o.z = fmaxf(fmaf((x.z - mu) * rs, g4.z, n4.z), 0.f);
        o.w = fmaxf(fmaf((x.w - mu) * rs, g4.w, n4.w), 0.f);
        if constexpr (YK == 1) {
            h2_store_row_nt(y + row * kC, o.x, o.y, o.z, o.w, sy);
        } else if constexpr (YK == 2) {
            const unsigned long long w2 = (unsigned long long)(bf16_rne(o.x) | ((unsigned)bf16_rne(o.y) << 16)) |
                                          ((unsigned long long)(bf16_rne(o.z) | ((unsigned)bf16_rne(o.w) << 16)) << 32);
            __builtin_nontemporal_store(w2, reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned short*>(y) + row * kC + c));
        } else {
            float* yo = y + row * kC + c;       // streamed once, 268 MB per launch at B = 64 (>> L2): non-temporal
            __builtin_nontemporal_store(o.x, yo);
            __builtin_nontemporal_store(o.y, yo + 1);
            __builtin_nontemporal_store(o.z, yo + 2);
            __builtin_nontemporal_store(o.w, yo + 3);
        }
        if (lane == 0) { mean_out[row] = mu; rstd_out[row] = rs; }
    };
    auto sq4 = [](const float4& x, float mu) __attribute__((always_inline)) {
        const float a = x.x - mu, b_ = x.y - mu, c_ = x.z - mu, d = x.w - mu;
        return fmaf(a, a, c_ * c_) + fmaf(b_, b_, d * d);
    };
    // two time steps per iteration: their reduction chains are independent and interleave
    for (int tt = wv; tt < C0_TT; tt += 8) {
        const int ta = t0 + tt, tb = ta + 4;
        if (ta >= L0) break;                     // wave-uniform
        const bool has_b = tb < L0;              // wave-uniform
        float4 xa = b4, xb = b4;
#pragma unroll
        for (int j = 0; j < K0; ++j) {
            const float sa = smp[tt * S0 + j], sb = smp[(tt + 4) * S0 + j];
            xa.x = fmaf(wj[j].x, sa, xa.x); xa.y = fmaf(wj[j].y, sa, xa.y);
            xa.z = fmaf(wj[j].z, sa, xa.z); xa.w = fmaf(wj[j].w, sa, xa.w);
            xb.x = fmaf(wj[j].x, sb, xb.x); xb.y = fmaf(wj[j].y, sb, xb.y);
            xb.z = fmaf(wj[j].z, sb, xb.z); xb.w = fmaf(wj[j].w, sb, xb.w);
        }
        const float mua = wave_sum((xa.x + xa.z) + (xa.y + xa.w)) * (1.0f / kC);
        const float mub = wave_sum((xb.x + xb.z) + (xb.y + xb.w)) * (1.0f / kC);
        const float va = wave_sum(sq4(xa, mua)), vb = wave_sum(sq4(xb, mub));
        const float rsa = __builtin_amdgcn_rsqf(va * (1.0f / (kC - 1)) + kNormEps);
        const float rsb = __builtin_amdgcn_rsqf(vb * (1.0f / (kC - 1)) + kNormEps);
        finish(xa, mua, rsa, (long)b * L0 + ta);
        if (has_b) finish(xb, mub, rsb, (long)b * L0 + tb);
    }
}

// Backward of layer 0.  Per time step it recomputes conv0 -> xhat from the waveform and
// the saved (mean, rstd), applies relu'/norm backward to the incoming dY (gradient
// w.r.t. the post-ReLU activation, produced by conv1's dgrad) and accumulates, in
// registers, the gradients of conv0.weight (256x10), conv0.bias, batchNorm0.weight and
// batchNorm0.bias.  The waveform needs no gradient (cpc/train.py:81-87), so there is
// no dgrad.  Each block reduces its waves through LDS and writes one partial row
// [13][256] to `part`; a second tiny kernel sums the partial rows in a fixed order
// (deterministic, no float atomics).
constexpr int C0B_TT = 256;                // time steps per block in backward (64 per wave); measured alone at B = 64, with
constexpr int C0B_NBW = 2;                 // C0B_NBW rows in flight per wave: 64/1 174 us, 256/1 138, 256/2 140, 256/4 125 (the
                                           // last one slower inside the step, beside the layer-1 weight gradient)
constexpr int C0B_NS = S0 * C0B_TT + (K0 - S0);
constexpr int C0_NACC = K0 + 3;            // 10 weight taps, conv bias, norm weight, norm bias

// DYB: dy arrives as bf16 (the bf16-storage variant) instead of fp32
template <bool DYB>
__global__ __launch_bounds__(256) void conv0_bwd_kernel(
    const float* __restrict__ wave, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ nw, const float* __restrict__ nb, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, const float* __restrict__ dy, float* __restrict__ part,
    int L, int L0) {
    __shared__ float smp[C0B_NS];
    __shared__ float red[4][C0_NACC][kC];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * C0B_TT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* wb = wave + (long)b * L;
    const int s_begin = t0 * S0 - P0;
    for (int i = tid; i < C0B_NS; i += 256) {
        int s = s_begin + i;
        smp[i] = ((unsigned)s < (unsigned)L) ? wb[s] : 0.f;
    }
    // conv0.weight via LDS (coalesced global read); `red` is free until the final reduction
    float* wT = &red[0][0][0];                   // [K0][kC]
    for (int e = tid; e < kC * K0; e += 256) wT[(e % K0) * kC + e / K0] = w[e];
    __syncthreads();
    float wr[4][K0], br[4], gw[4], gb[4];
    const int c = lane * 4;
#pragma unroll
    for (int j = 0; j < K0; ++j) {
        const float4 wv4 = *reinterpret_cast<const float4*>(wT + j * kC + c);
        wr[0][j] = wv4.x; wr[1][j] = wv4.y; wr[2][j] = wv4.z; wr[3][j] = wv4.w;
    }
    {
        const float4 b4 = *reinterpret_cast<const float4*>(bias + c);
        const float4 w4 = *reinterpret_cast<const float4*>(nw + c);
        const float4 n4 = *reinterpret_cast<const float4*>(nb + c);
        br[0] = b4.x; br[1] = b4.y; br[2] = b4.z; br[3] = b4.w;
        gw[0] = w4.x; gw[1] = w4.y; gw[2] = w4.z; gw[3] = w4.w;
        gb[0] = n4.x; gb[1] = n4.y; gb[2] = n4.z; gb[3] = n4.w;
    }
    __syncthreads();                             // all lanes have their weights before `red` is reused
    float acc[4][C0_NACC];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < C0_NACC; ++j) acc[q][j] = 0.f;
    __syncthreads();
    // C0B_NBW rows of this wave at a time: their loads, the two wave reductions each needs and the ~130 VALU operations
    // per row interleave instead of queueing behind one another (one row at a time: 153 us alone at B = 64)
    for (int tt0 = wv; tt0 < C0B_TT; tt0 += 4 * C0B_NBW) {
        if (t0 + tt0 >= L0) break;               // wave-uniform
        float g[C0B_NBW][4], mu[C0B_NBW], rstd[C0B_NBW];
        bool live[C0B_NBW];
#pragma unroll
        for (int u = 0; u < C0B_NBW; ++u) {
            const int tt = tt0 + 4 * u;
            live[u] = tt < C0B_TT && t0 + tt < L0;
            const long row = (long)b * L0 + (live[u] ? t0 + tt : t0 + tt0);
            if constexpr (DYB) {
                const uint2 gb = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(dy) + row * kC + c);
                g[u][0] = bf16_val((unsigned short)(gb.x & 0xFFFFu)); g[u][1] = bf16_val((unsigned short)(gb.x >> 16));
                g[u][2] = bf16_val((unsigned short)(gb.y & 0xFFFFu)); g[u][3] = bf16_val((unsigned short)(gb.y >> 16));
            } else {
                const float4 g4 = *reinterpret_cast<const float4*>(dy + row * kC + c);
                g[u][0] = g4.x; g[u][1] = g4.y; g[u][2] = g4.z; g[u][3] = g4.w;
            }
            mu[u] = mean_in[row]; rstd[u] = rstd_in[row];
        }
        float sv[C0B_NBW][K0], xh[C0B_NBW][4], dxh[C0B_NBW][4], s1[C0B_NBW], s2[C0B_NBW];
#pragma unroll
        for (int u = 0; u < C0B_NBW; ++u) {
            const int tt = live[u] ? tt0 + 4 * u : tt0;
#pragma unroll
            for (int j = 0; j < K0; ++j) sv[u][j] = smp[tt * S0 + j];
            s1[u] = 0.f; s2[u] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x = br[q];
#pragma unroll
                for (int j = 0; j < K0; ++j) x = fmaf(wr[q][j], sv[u][j], x);
                xh[u][q] = (x - mu[u]) * rstd[u];
                const float yv = fmaf(xh[u][q], gw[q], gb[q]);
                const float dyh = (live[u] && yv > 0.f) ? g[u][q] : 0.f;   // relu'
                acc[q][K0 + 1] = fmaf(dyh, xh[u][q], acc[q][K0 + 1]);       // d batchNorm0.weight
                acc[q][K0 + 2] += dyh;                                      // d batchNorm0.bias
                dxh[u][q] = dyh * gw[q];
                s1[u] += dxh[u][q];
                s2[u] = fmaf(dxh[u][q], xh[u][q], s2[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < C0B_NBW; ++u) {
            s1[u] = wave_sum(s1[u]) * (1.0f / kC);
            s2[u] = wave_sum(s2[u]) * (1.0f / (kC - 1));
        }
#pragma unroll
        for (int u = 0; u < C0B_NBW; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dx = live[u] ? rstd[u] * (dxh[u][q] - s1[u] - xh[u][q] * s2[u]) : 0.f;
                acc[q][K0] += dx;                                    // d conv0.bias
#pragma unroll
                for (int j = 0; j < K0; ++j) acc[q][j] = fmaf(dx, sv[u][j], acc[q][j]);   // d conv0.weight
            }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < C0_NACC; ++j) red[wv][j][c + q] = acc[q][j];
    __syncthreads();
    float* prow = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (C0_NACC * kC);
    for (int i = tid; i < C0_NACC * kC; i += 256) {
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
        hipLaunchKernelGGL(conv0_fwd_kernel<1>, dim3(cdiv(L0, C0_TT), B), dim3(256), 0, (hipStream_t)stream,
                           wave, w, bias, nw, nb, reinterpret_cast<float*>(y), mean, rstd, L, L0, y_amax);
    else
        hipLaunchKernelGGL(conv0_fwd_kernel<0>, dim3(cdiv(L0, C0_TT), B), dim3(256), 0, (hipStream_t)stream,
                           wave, w, bias, nw, nb, reinterpret_cast<float*>(y), mean, rstd, L, L0, y_amax);
    CPC_LAUNCH_CHECK();
    return 0;
}

namespace cpc {
// bf16-storage variant (mode 4): y as bf16 (B, L0, 256)
int conv0_forward_bf16(const float* wave, const float* w, const float* bias, const float* nw, const float* nb, void* y,
                       float* mean, float* rstd, int B, int L, hipStream_t st) {
    CPC_RETURN_IF(B <= 0 || L < K0 - 2 * P0, CPC_ERR_SHAPE);
    const int L0 = conv_out_len(L, K0, S0, P0);
    hipLaunchKernelGGL(conv0_fwd_kernel<2>, dim3(cdiv(L0, C0_TT), B), dim3(256), 0, st, wave, w, bias, nw, nb,
                       reinterpret_cast<float*>(y), mean, rstd, L, L0, (const float*)nullptr);
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
