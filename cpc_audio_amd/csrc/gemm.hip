// Plain f32-MFMA GEMMs used around the recurrent and contrastive kernels:
//   nt_gemm:  C[M,N]   = A[M,K] . B[N,K]^T (+ bias[N])         (projections, heads, dX)
//   tn_gemm:  C[N1,N2] (+)= sum_m A[m,N1]^T (x) B[m,N2]         (every weight gradient)
// plus a tiled transpose.  Row addressing of A (and B for tn) goes through RowMap so
// that batch-strided views (c[:, :W], h_{t-1} = y[:, t-1]) need no copies.
#include "cpc_common.h"
#include "cpc_internal.h"
#include "philox.h"
#include "gemm_tile.h"

namespace cpc {

using NtBig = NtTile<128, 128, 2, 2>;     // 128x128 block, 64x64 per wave
using NtSmall = NtTile<64, 64, 2, 2>;      // 64x64 block: used when the big tiling cannot fill 256 CUs
using NtBigX3 = NtTileX3<128, 128, 2, 2>;  // the same on the bf16 pipe with 3-piece split operands
using NtSmallX3 = NtTileX3<64, 64, 2, 2>;
using TnG = TnTile<128, 128, 2, 2>;
using TnGX3 = TnTileX3<128, 128, 2, 2>;
// ... and on the fp16 pipe with 2-piece split operands (3 MFMAs per product instead of 6): needs a bound on max|A| and
// max|B| for the power-of-two operand scales (GemmBounds; gemm_tile.h scale_for_amax)
using NtBigH2 = NtTileX3<128, 128, 2, 2, 32, 1, false, false, 2>;
using NtSmallH2 = NtTileX3<64, 64, 2, 2, 32, 1, false, false, 2>;
using TnGH2 = TnTileX3<128, 128, 2, 2, 32, 1, 2>;
// 128 x 256 tile, 8 waves, 16-k chunks through two LDS stages with four register sets of global prefetch (the conv layers'
// tile, gemm_tile.h): for the wide products (N a multiple of 256, K of 64, enough row tiles to fill the chip) -- a third less
// operand traffic through L2 than 128 x 128, and the load latency of a short K walk stays hidden
using NtWideH2 = NtTileX3<128, 256, 2, 4, 16, 2, true, false, 2>;
using NtWideX3 = NtTileX3<128, 256, 2, 4, 16, 2, true, false, 3>;     // the same without operand bounds: three bf16 pieces

struct OperandScales { float sa, sb, inv; };
__device__ __forceinline__ OperandScales operand_scales(const GemmBounds& gb) {
    OperandScales o;
    o.sa = scale_for_amax(fold_amax(gb.a, gb.a_slots));
    o.sb = scale_for_amax(fold_amax(gb.b, gb.b_slots));
    o.inv = 1.0f / (o.sa * o.sb);                         // powers of two: exact
    return o;
}

// Output row m goes to C + m*ldc, or, when c_R > 0, to C + (m / c_R)*c_bstride + (m % c_R)*ldc
// (a batch-strided view such as dc[:, :W]).
// EPI: GemmEpilogue kind compiled in (0: none -- `ep` is ignored)
template <class NtG, int BMN, bool H2 = false, int BNN = BMN, int EPI = 0>
__global__ __launch_bounds__(NtG::NTHREADS) void nt_gemm_kernel(RowMap am, const float* __restrict__ Bmat,
                                                                int ldb, const float* __restrict__ bias,
                                                                float* __restrict__ C, long ldc, int K,
                                                                int c_R, long c_bstride, GemmBounds gb, GemmGroup grp,
                                                                int ksplit = 1, float* __restrict__ kpart = nullptr,
                                                                GemmEpilogue ep = GemmEpilogue()) {
    __shared__ float smem[NtG::SMEM_FLOATS];
    const int m0 = blockIdx.x * BMN, n0 = blockIdx.y * BNN;
    if (ksplit > 1) {                                    // slice blockIdx.z of the K walk; a dense (M, N) partial per slice
        const long z = blockIdx.z;
        K /= ksplit;
        am.base += z * K; Bmat += z * K;
        C = kpart + z * (long)am.M * (gridDim.y * BNN);
        ldc = gridDim.y * BNN; c_R = 0; bias = nullptr;
    }
    if (grp.G > 1) {                                     // problem blockIdx.z of a group (cpc_internal.h, GemmGroup)
        const long g = blockIdx.z;
        am.base += g * grp.a; Bmat += g * grp.b; C += g * grp.c;
        if (bias) bias += g * grp.bias;
        if (gb.a) gb.a += g * gb.a_gs;
        if (gb.b) gb.b += g * gb.b_gs;
        if constexpr (EPI != 0) {
            ep.seed += (unsigned long long)g;
            if (ep.mask) ep.mask += g * ep.mask_gs;
            if (ep.amax) ep.amax += g * ep.amax_gs;
            if (ep.colsum) ep.colsum += g * ep.colsum_gs;
        }
    }
    f32x16 acc[NtG::TM][NtG::TN];
    zero_acc(acc);
    float inv = 1.0f;
    if constexpr (H2) {
        const OperandScales os = operand_scales(gb);
        inv = os.inv;
        NtG::run(acc, am, m0, Bmat, ldb, n0, K, smem, 0, 16, 0, os.sa, os.sb);
    } else {
        NtG::run(acc, am, m0, Bmat, ldb, n0, K, smem);
    }
    float cmax = 0.f;
    [[maybe_unused]] float csum[NtG::TN];
#pragma unroll
    for (int tn = 0; tn < NtG::TN; ++tn) csum[tn] = 0.f;
    const unsigned th = EPI == 1 ? drop_threshold16(ep.drop_p) : 0u;
    const float keep_scale = EPI == 1 ? 1.0f / (1.0f - ep.drop_p) : 1.0f;
#pragma unroll
    for (int tn = 0; tn < NtG::TN; ++tn) {
        const int col = n0 + NtG::c_col(tn);
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < NtG::TM; ++tm) {
            [[maybe_unused]] float mk[16];            // kind 2: the tile's mask values, requested together and unconditionally
            if constexpr (EPI == 2) {                 // (clamped rows; a load inside the guarded store is a round trip each)
#pragma unroll
                for (int r = 0; r < 16; ++r) mk[r] = ep.mask[(long)min(m0 + NtG::c_row(tm, r), am.M - 1) * ldc + col];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + NtG::c_row(tm, r);
                [[maybe_unused]] Philox4 draw;        // registers 4q .. 4q+3 are four consecutive rows: one Philox block
                if constexpr (EPI == 1)
                    if ((r & 3) == 0 && ep.drop_p > 0.f) draw = philox4x32_10(ep.seed, 1u, ffn_drop_block(m, col));
                if (m < am.M) {
                    long ro = (long)m * ldc;
                    if (c_R > 0) { const int cb = m / c_R; ro = cb * c_bstride + (long)(m - cb * c_R) * ldc; }
                    float v = H2 ? fmaf(acc[tm][tn][r], inv, bv) : acc[tm][tn][r] + bv;
                    if constexpr (EPI == 1) {
                        v = fmaxf(v, 0.f);
                        if (ep.drop_p > 0.f) v = ffn_drop_field(draw, r & 3, col) >= th ? v * keep_scale : 0.f;
                    }
                    if constexpr (EPI == 2) v = mk[r] > 0.f ? v * ep.scale : 0.f;
                    if constexpr (EPI != 0) cmax = fmaxf(cmax, fabsf(v));
                    if constexpr (EPI == 2) csum[tn] += v;
                    C[ro + col] = v;
                }
            }
        }
    }
    if constexpr (EPI != 0) {                         // one atomic per workgroup (cross-XCD atomics on one address: ~1 us each)
        __shared__ float wmax[NtG::NTHREADS / 64];
        cmax = wave_max(cmax);
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = cmax;
        __syncthreads();
        if (threadIdx.x == 0 && ep.amax != nullptr) {
            float m = 0.f;
            for (int w = 0; w < NtG::NTHREADS / 64; ++w) m = fmaxf(m, wmax[w]);
            atomicMax(reinterpret_cast<unsigned*>(ep.amax) + (blockIdx.x + 5u * blockIdx.y) % (unsigned)kAmaxSlots, __float_as_uint(m));
        }
    }
    if constexpr (EPI == 2) {                         // column sums of the tile, in a fixed order: lane halves, then the waves along M
        if (ep.colsum != nullptr) {                   // block-uniform
            constexpr int WAVES_M = BMN / NtG::WM;
            __shared__ float cs[WAVES_M][BNN];
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
            for (int tn = 0; tn < NtG::TN; ++tn) {
                const float s = csum[tn] + __shfl_xor(csum[tn], 32);
                if (lane < 32) cs[wave / (BNN / NtG::WN)][NtG::c_col(tn)] = s;
            }
            __syncthreads();
            for (int c = threadIdx.x; c < BNN; c += NtG::NTHREADS) {
                float s = cs[0][c];
#pragma unroll
                for (int w = 1; w < WAVES_M; ++w) s += cs[w][c];
                ep.colsum[(long)blockIdx.x * (gridDim.y * BNN) + n0 + c] = s;
            }
        }
    }
}

// 1-D grid of 8 * T * ceil(S/8) blocks, T = (N1/128)*(N2/128) output tiles; part[z][N1][N2].
// XCD-aware: all tiles of one row split run on one XCD (see conv_wgrad_kernel).
template <class TnG, bool H2 = false>
__global__ __launch_bounds__(256) void tn_gemm_kernel(RowMap am, RowMap bm, int N1, int N2, int rows_per_split,
                                                      int S, float* __restrict__ part, long zstride, GemmBounds gb,
                                                      GemmGroup grp) {
    __shared__ float smem[TnG::SMEM_FLOATS];
    if (grp.G > 1) {                                     // problem blockIdx.y of a group; its partials behind the previous one's
        const long g = blockIdx.y;
        am.base += g * grp.a; bm.base += g * grp.b; part += grp.part ? g * grp.part : g * S * zstride;
        if (gb.a) gb.a += g * gb.a_gs;
        if (gb.b) gb.b += g * gb.b_gs;
    }
    const int tn2 = N2 / 128, T = (N1 / 128) * tn2;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = slot % T, z = (slot / T) * 8 + xcd;
    if (z >= S) return;                                  // block-uniform
    const int n0 = (tile % tn2) * 128, c0 = (tile / tn2) * 128;
    const int mbeg = z * rows_per_split;
    const int mend = min(am.M, mbeg + rows_per_split);
    f32x16 acc[TnG::TM][TnG::TN];
    zero_acc(acc);
    float inv = 1.0f;
    if constexpr (H2) {
        const OperandScales os = operand_scales(gb);
        inv = os.inv;
        TnG::run(acc, am, c0, bm, n0, mbeg, mend, smem, os.sa, os.sb);
    } else {
        TnG::run(acc, am, c0, bm, n0, mbeg, mend, smem);
    }
    float* out = part + (long)z * zstride;
#pragma unroll
    for (int tm = 0; tm < TnG::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = c0 + TnG::c_row(tm, r);
#pragma unroll
            for (int tn = 0; tn < TnG::TN; ++tn)
                out[(long)row * N2 + n0 + TnG::c_col(tn)] = H2 ? acc[tm][tn][r] * inv : acc[tm][tn][r];
        }
}

__global__ __launch_bounds__(256) void split_reduce_kernel(const float* __restrict__ part, int S,
                                                           long n, float* __restrict__ C, int accumulate, long c_gs,
                                                           long part_gs, const float* __restrict__ out_scale = nullptr,
                                                           long scale_block = 0) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    part += part_gs ? (long)blockIdx.y * part_gs : (long)blockIdx.y * S * n;      // blockIdx.y: problem of a group
    C += (long)blockIdx.y * c_gs;
    if (out_scale != nullptr) {               // (block k of scale_block elements) * out_scale[k]; accumulate adds the scaled sum
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += part[(long)z * n + idx];
        s *= out_scale[idx / scale_block];
        C[idx] = accumulate ? C[idx] + s : s;
        return;
    }
    float s = accumulate ? C[idx] : 0.f;
    for (int z = 0; z < S; ++z) s += part[(long)z * n + idx];
    C[idx] = s;
}

// Up to 4 TN problems of one shape (same M, N1, N2) in one launch: the four weight gradients of a two-layer GRU are
// 3.2 GFLOP each -- alone, each needs 32 row splits to fill the chip and spends its time in prologues and in its
// own split reduction.  Together 8 splits do, with 4x the rows per workgroup.
struct TnBatch { RowMap a[4], b[4]; float* out[4]; };
template <class TnG>
__global__ __launch_bounds__(256) void tn_gemm_batch_kernel(TnBatch bt, int N1, int N2, int rows_per_split, int S,
                                                            float* __restrict__ part, long zstride) {
    __shared__ float smem[TnG::SMEM_FLOATS];
    const int tn2 = N2 / 128, T = (N1 / 128) * tn2;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = slot % T, z = (slot / T) * 8 + xcd;
    if (z >= S) return;                                  // block-uniform
    const int q = blockIdx.y;
    const RowMap& am = bt.a[q];
    const RowMap& bm = bt.b[q];
    const int n0 = (tile % tn2) * 128, c0 = (tile / tn2) * 128;
    const int mbeg = z * rows_per_split;
    const int mend = min(am.M, mbeg + rows_per_split);
    f32x16 acc[TnG::TM][TnG::TN];
    zero_acc(acc);
    TnG::run(acc, am, c0, bm, n0, mbeg, mend, smem);
    float* out = part + ((long)q * S + z) * zstride;
#pragma unroll
    for (int tm = 0; tm < TnG::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = c0 + TnG::c_row(tm, r);
#pragma unroll
            for (int tn = 0; tn < TnG::TN; ++tn)
                out[(long)row * N2 + n0 + TnG::c_col(tn)] = acc[tm][tn][r];
        }
}
__global__ __launch_bounds__(256) void split_reduce_batch_kernel(const float* __restrict__ part, int S, long n,
                                                                 TnBatch bt, int accumulate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    float* __restrict__ C = bt.out[blockIdx.y];
    const float* __restrict__ p = part + (long)blockIdx.y * S * n;
    float s = accumulate ? C[idx] : 0.f;
    for (int z = 0; z < S; ++z) s += p[(long)z * n + idx];
    C[idx] = s;
}

// out[c][r] = in[r][c]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int R, int Cn, long in_gs, long out_gs) {
    __shared__ float tile[32][33];
    in += (long)blockIdx.z * in_gs;                      // blockIdx.z: matrix of a group
    out += (long)blockIdx.z * out_gs;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cn) ? in[(long)r * Cn + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cn && r < R) out[(long)c * R + r] = tile[tx][i];
    }
}

// the same for up to 4 equally shaped matrices in one launch (blockIdx.z picks the pair)
struct TransposeBatch { const float* in[4]; float* out[4]; };
__global__ __launch_bounds__(256) void transpose_batch_kernel(TransposeBatch b, int R, int Cn) {
    __shared__ float tile[32][33];
    const float* __restrict__ in = b.in[blockIdx.z];
    float* __restrict__ out = b.out[blockIdx.z];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cn) ? in[(long)r * Cn + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cn && r < R) out[(long)c * R + r] = tile[tx][i];
    }
}

// One workgroup of 1024 threads per (array, slot): slot s covers elements [s, s+1) * ceil(n / 64 / 4) * 4 of its array,
// four 16-byte loads in flight per thread.
// (a job without an array -- x == NULL -- is a bound known a priori: its slots all carry cval)
struct AbsmaxJobs { const float* x[4]; long n[4]; float cval[4]; };
__global__ __launch_bounds__(1024) void absmax_slots_kernel(AbsmaxJobs jobs, float* __restrict__ out) {
    __shared__ float red[16];
    const float* __restrict__ x = jobs.x[blockIdx.y];
    if (x == nullptr) {                                  // block-uniform
        if (threadIdx.x == 0) out[blockIdx.y * kAmaxSlots + blockIdx.x] = jobs.cval[blockIdx.y];
        return;
    }
    const long n = jobs.n[blockIdx.y];
    const long per = ((n + kAmaxSlots - 1) / kAmaxSlots + 3) & ~3L;
    const long beg = (long)blockIdx.x * per, end = beg + per < n ? beg + per : n;
    float m = 0.f;
    if ((reinterpret_cast<unsigned long>(x) & 15) == 0) {
        long i = beg + 4L * threadIdx.x;
        for (; i + 3 * 4096 + 3 < end; i += 4 * 4096) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + i + u * 4096);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
        }
        for (; i + 3 < end; i += 4096) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (; i < end; ++i) m = fmaxf(m, fabsf(x[i]));         // the one thread that holds a partial last group
    } else {
        for (long i = beg + threadIdx.x; i < end; i += 1024) m = fmaxf(m, fabsf(x[i]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 64) {
        m = wave_max(threadIdx.x < 16 ? red[threadIdx.x] : 0.f);
        if (threadIdx.x == 0) out[blockIdx.y * kAmaxSlots + blockIdx.x] = m;
    }
}

// grid (kAmaxSlots, G, njobs), 256 threads: slot s of array g of job j (weights: a few thousand elements per slot)
struct AbsmaxGroupJobs { const float* x[4]; long n[4]; long gs[4]; };
__global__ __launch_bounds__(256) void absmax_group_kernel(AbsmaxGroupJobs jobs, float* __restrict__ out, long out_gs) {
    __shared__ float red[4];
    const int j = blockIdx.z;
    const float* __restrict__ x = jobs.x[j] + (long)blockIdx.y * jobs.gs[j];
    const long n = jobs.n[j];
    const long per = (n + kAmaxSlots - 1) / kAmaxSlots;
    const long beg = (long)blockIdx.x * per, end = beg + per < n ? beg + per : n;
    float m = 0.f;
    for (long i = beg + threadIdx.x; i < end; i += 256) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        out[(long)j * kAmaxSlots + (long)blockIdx.y * out_gs + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

int g_gemm_wide = 1;
int g_gemm_split = 1;      // cpc_set_gemm_split: 0 keeps every plain GEMM on three bf16 pieces, bounds or not

int absmax_slots(const float* const* x, const long* n, int njobs, float* out, hipStream_t st, const float* cval) {
    if (njobs <= 0 || njobs > 4) return CPC_ERR_ARG;
    AbsmaxJobs jobs;
    for (int j = 0; j < 4; ++j) {
        jobs.x[j] = x[j < njobs ? j : 0]; jobs.n[j] = n[j < njobs ? j : 0];
        jobs.cval[j] = cval ? cval[j < njobs ? j : 0] : 0.f;
    }
    hipLaunchKernelGGL(absmax_slots_kernel, dim3(kAmaxSlots, njobs), dim3(1024), 0, st, jobs, out);
    CPC_LAUNCH_CHECK();
    return 0;
}

// C (rows mapped as nt_gemm_kernel does) = sum over the `splits` dense (M, N) partials, in order, + bias
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, int M, int N,
                                                            const float* __restrict__ bias, float* __restrict__ C, long ldc,
                                                            int c_R, long c_bstride) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // one float4 of a row
    const int n4 = N / 4;
    if (i >= (long)M * n4) return;
    const int m = (int)(i / n4), c = (int)(i - (long)m * n4) * 4;
    float4 s = *reinterpret_cast<const float4*>(part + (long)m * N + c);
    for (int z = 1; z < splits; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((long)z * M + m) * N + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias != nullptr) { s.x += bias[c]; s.y += bias[c + 1]; s.z += bias[c + 2]; s.w += bias[c + 3]; }
    long ro = (long)m * ldc;
    if (c_R > 0) { const int cb = m / c_R; ro = cb * c_bstride + (long)(m - cb * c_R) * ldc; }
    *reinterpret_cast<float4*>(C + ro + c) = s;
}

// The conditions under which nt_gemm() below picks the wide fp16-piece tile without a K split (the one tile whose epilogue is
// compiled with the GemmEpilogue kinds), for a dense C.
int g_gemm_fuse = 1;       // cpc_set_gemm_fuse: 0 keeps the elementwise kernels behind the feed-forward GEMMs (tests, A/B)
bool nt_gemm_fuses(int M, int N, int K, int ldc, const GemmBounds& gb, const GemmGroup& grp) {
    return g_gemm_fuse && M > 0 && g_mfma_mode >= 2 && g_gemm_split && gb.a && gb.b && g_gemm_wide && N % 256 == 0 && K % 64 == 0 && ldc == N &&
           (g_gemm_wide == 2 || (long)cdiv(M, 128) * (N / 256) * grp.G >= 256);
}
int nt_gemm_fused(const RowMap& am, const float* Bmat, int ldb, const float* bias, float* C, long ldc, int N, int K,
                  hipStream_t st, GemmBounds gb, GemmGroup grp, GemmEpilogue ep) {
    if (!nt_gemm_fuses(am.M, N, K, (int)ldc, gb, grp) || (ep.kind != 1 && ep.kind != 2)) return CPC_ERR_ARG;
    if (ep.kind == 1 && (N != kFfnWidth || !(ep.drop_p >= 0.f && ep.drop_p < 1.f))) return CPC_ERR_ARG;
    if (ep.kind == 2 && !ep.mask) return CPC_ERR_ARG;
    const dim3 grid(cdiv(am.M, 128), N / 256, grp.G);
    if (ep.kind == 1)
        hipLaunchKernelGGL((nt_gemm_kernel<NtWideH2, 128, true, 256, 1>), grid, dim3(NtWideH2::NTHREADS), 0, st, am, Bmat, ldb, bias, C,
                           ldc, K, 0, 0L, gb, grp, 1, (float*)nullptr, ep);
    else
        hipLaunchKernelGGL((nt_gemm_kernel<NtWideH2, 128, true, 256, 2>), grid, dim3(NtWideH2::NTHREADS), 0, st, am, Bmat, ldb, bias, C,
                           ldc, K, 0, 0L, gb, grp, 1, (float*)nullptr, ep);
    CPC_LAUNCH_CHECK();
    return 0;
}

int nt_gemm(const RowMap& am, const float* Bmat, int ldb, const float* bias, float* C, long ldc,
            int N, int K, hipStream_t st, int c_R, long c_bstride, GemmBounds bounds, GemmGroup grp, SplitK sk) {
    if (am.M <= 0) return 0;
    if (N % 128 != 0 || K % 16 != 0 || K < 16) return CPC_ERR_SHAPE;
    const bool big = (long)cdiv(am.M, 128) * (N / 128) * grp.G >= 384;
    const bool x3 = g_mfma_mode != 0 && K % 32 == 0;
    const bool h2 = x3 && g_mfma_mode >= 2 && g_gemm_split && bounds.a && bounds.b;      // operand bounds known: fp16 split
    const dim3 gb(cdiv(am.M, 128), N / 128, grp.G), gs(cdiv(am.M, 64), N / 64, grp.G);
    // narrow products on the wide tile with the K walk split (see SplitK): the smallest split that gives ~200 workgroups
    if (x3 && g_gemm_wide >= 1 && grp.G == 1 && sk.part && N % 256 == 0 && am.rstride == K && am.tmul == 0 && ldb == K &&
        (ldc % 4 == 0) && (long)cdiv(am.M, 128) * (N / 256) < 192) {
        const long tiles = (long)cdiv(am.M, 128) * (N / 256);
        int split = 0;
        for (int q = 2; q <= 8 && !split; ++q)
            if (K % (64 * q) == 0 && (tiles * q >= 200 || g_gemm_wide == 2) && (long)q * am.M * N <= sk.floats) split = q;
        if (split) {
            const dim3 g3(cdiv(am.M, 128), N / 256, split);
            if (h2)
                hipLaunchKernelGGL((nt_gemm_kernel<NtWideH2, 128, true, 256>), g3, dim3(NtWideH2::NTHREADS), 0, st, am, Bmat, ldb,
                                   bias, C, ldc, K, c_R, c_bstride, bounds, grp, split, sk.part);
            else
                hipLaunchKernelGGL((nt_gemm_kernel<NtWideX3, 128, false, 256>), g3, dim3(NtWideX3::NTHREADS), 0, st, am, Bmat, ldb,
                                   bias, C, ldc, K, c_R, c_bstride, bounds, grp, split, sk.part);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv((long)am.M * (N / 4), 256)), dim3(256), 0, st, sk.part, split, am.M, N,
                               bias, C, ldc, c_R, c_bstride);
            CPC_LAUNCH_CHECK();
            return 0;
        }
    }
    const bool wide = x3 && g_gemm_wide && N % 256 == 0 && K % 64 == 0 &&
                      (g_gemm_wide == 2 || (long)cdiv(am.M, 128) * (N / 256) * grp.G >= 256);
    if (wide && h2)
        hipLaunchKernelGGL((nt_gemm_kernel<NtWideH2, 128, true, 256>), dim3(cdiv(am.M, 128), N / 256, grp.G), dim3(NtWideH2::NTHREADS), 0,
                           st, am, Bmat, ldb, bias, C, ldc, K, c_R, c_bstride, bounds, grp);
    else if (wide)
        hipLaunchKernelGGL((nt_gemm_kernel<NtWideX3, 128, false, 256>), dim3(cdiv(am.M, 128), N / 256, grp.G), dim3(NtWideX3::NTHREADS), 0,
                           st, am, Bmat, ldb, bias, C, ldc, K, c_R, c_bstride, bounds, grp);
    else if (big && h2)
        hipLaunchKernelGGL((nt_gemm_kernel<NtBigH2, 128, true>), gb, dim3(NtBigH2::NTHREADS), 0, st, am, Bmat, ldb, bias, C,
                           ldc, K, c_R, c_bstride, bounds, grp);
    else if (h2)
        hipLaunchKernelGGL((nt_gemm_kernel<NtSmallH2, 64, true>), gs, dim3(NtSmallH2::NTHREADS), 0, st, am, Bmat, ldb, bias,
                           C, ldc, K, c_R, c_bstride, bounds, grp);
    else if (big && x3)
        hipLaunchKernelGGL((nt_gemm_kernel<NtBigX3, 128>), gb, dim3(NtBigX3::NTHREADS), 0, st, am, Bmat, ldb, bias, C,
                           ldc, K, c_R, c_bstride, bounds, grp);
    else if (big)
        hipLaunchKernelGGL((nt_gemm_kernel<NtBig, 128>), gb, dim3(NtBig::NTHREADS), 0, st, am, Bmat, ldb, bias, C,
                           ldc, K, c_R, c_bstride, bounds, grp);
    else if (x3)
        hipLaunchKernelGGL((nt_gemm_kernel<NtSmallX3, 64>), gs, dim3(NtSmallX3::NTHREADS), 0, st, am, Bmat, ldb, bias,
                           C, ldc, K, c_R, c_bstride, bounds, grp);
    else
        hipLaunchKernelGGL((nt_gemm_kernel<NtSmall, 64>), gs, dim3(NtSmall::NTHREADS), 0, st, am, Bmat, ldb, bias,
                           C, ldc, K, c_R, c_bstride, bounds, grp);
    CPC_LAUNCH_CHECK();
    return 0;
}

long tn_gemm_part_floats(int M, int N1, int N2) {
    int splits, rows;
    tn_gemm_plan(M, N1, N2, &splits, &rows);
    return (long)splits * N1 * N2;
}

void tn_gemm_plan(int M, int N1, int N2, int* splits, int* rows) {
    const int tiles = (N1 / 128) * (N2 / 128);
    int S = cdiv(768, tiles > 0 ? tiles : 1);
    int r = cdiv(cdiv(M, S), 32) * 32;
    if (r < 256) r = 256;
    S = cdiv(M, r);
    if (S < 1) S = 1;
    *splits = S;
    *rows = r;
}

int tn_gemm(const RowMap& am, int N1, const RowMap& bm, int N2, float* part, float* C, int accumulate,
            hipStream_t st, GemmBounds bounds, GemmGroup grp) {
    if (N1 % 128 != 0 || N2 % 128 != 0 || am.M != bm.M) return CPC_ERR_SHAPE;
    const long n = (long)N1 * N2;
    if (am.M <= 0) {
        if (!accumulate) (void)hipMemsetAsync(C, 0, sizeof(float) * n, st);
        return 0;
    }
    int S, rows;
    tn_gemm_plan(am.M, N1, N2, &S, &rows);
    if (grp.G > 1) {                      // the group fills the chip together: fewer row splits per problem (never more than
        const int tiles = grp.G * (N1 / 128) * (N2 / 128);        // the single-problem plan, which sizes `part`)
        int Sg = cdiv(768, tiles);
        if (Sg < 8) Sg = 8;               // the kernel gives split z to XCD z % 8: fewer than 8 splits leave XCDs idle
        int r = cdiv(cdiv(am.M, Sg), 32) * 32;
        if (r < rows) r = rows;
        rows = r;
        S = cdiv(am.M, rows);
    }
    const dim3 grid(8 * (N1 / 128) * (N2 / 128) * cdiv(S, 8), grp.G);
    if (g_mfma_mode >= 2 && g_gemm_split && bounds.a && bounds.b)
        hipLaunchKernelGGL((tn_gemm_kernel<TnGH2, true>), grid, dim3(256), 0, st, am, bm, N1, N2, rows, S, part, n, bounds, grp);
    else if (g_mfma_mode != 0)
        hipLaunchKernelGGL((tn_gemm_kernel<TnGX3>), grid, dim3(256), 0, st, am, bm, N1, N2, rows, S, part, n, bounds, grp);
    else
        hipLaunchKernelGGL((tn_gemm_kernel<TnG>), grid, dim3(256), 0, st, am, bm, N1, N2, rows, S, part, n, bounds, grp);
    hipLaunchKernelGGL(split_reduce_kernel, dim3(cdiv(n, 256), grp.G), dim3(256), 0, st, part, S, n, C, accumulate, grp.c, grp.part,
                       grp.out_scale, (long)grp.out_scale_rows * N2);
    CPC_LAUNCH_CHECK();
    return 0;
}

int absmax_group(const float* const* x, const long* n, const long* x_gs, int njobs, int G, float* out, long out_gs, hipStream_t st) {
    if (njobs <= 0 || njobs > 4 || G <= 0) return CPC_ERR_ARG;
    AbsmaxGroupJobs jobs;
    for (int j = 0; j < 4; ++j) {
        const int u = j < njobs ? j : 0;
        jobs.x[j] = x[u]; jobs.n[j] = n[u]; jobs.gs[j] = x_gs[u];
    }
    hipLaunchKernelGGL(absmax_group_kernel, dim3(kAmaxSlots, G, njobs), dim3(256), 0, st, jobs, out, out_gs);
    CPC_LAUNCH_CHECK();
    return 0;
}

// Splits of the batched plan: aim at ~768 workgroups over all problems, at least 256 rows each.
void tn_gemm_batch_plan(int nprob, int M, int N1, int N2, int* splits, int* rows) {
    const int tiles = nprob * (N1 / 128) * (N2 / 128);
    int S = cdiv(768, tiles > 0 ? tiles : 1);
    int r = cdiv(cdiv(M, S), 32) * 32;
    if (r < 256) r = 256;
    S = cdiv(M, r);
    if (S < 1) S = 1;
    *splits = S;
    *rows = r;
}

long tn_gemm_batch_part_floats(int nprob, int M, int N1, int N2) {
    int S, rows;
    tn_gemm_batch_plan(nprob, M, N1, N2, &S, &rows);
    return (long)nprob * S * N1 * N2;
}

int tn_gemm_batch(int nprob, const RowMap* am, int N1, const RowMap* bm, int N2, float* part, float* const* C,
                  int accumulate, hipStream_t st) {
    if (nprob <= 0 || nprob > 4 || N1 % 128 != 0 || N2 % 128 != 0) return CPC_ERR_SHAPE;
    const int M = am[0].M;
    for (int q = 0; q < nprob; ++q)
        if (am[q].M != M || bm[q].M != M) return CPC_ERR_SHAPE;
    if (M <= 0) return CPC_ERR_SHAPE;
    const long n = (long)N1 * N2;
    TnBatch bt;
    for (int q = 0; q < 4; ++q) { const int u = q < nprob ? q : 0; bt.a[q] = am[u]; bt.b[q] = bm[u]; bt.out[q] = C[u]; }
    int S, rows;
    tn_gemm_batch_plan(nprob, M, N1, N2, &S, &rows);
    const dim3 grid(8 * (N1 / 128) * (N2 / 128) * cdiv(S, 8), nprob);
    if (g_mfma_mode != 0)
        hipLaunchKernelGGL((tn_gemm_batch_kernel<TnGX3>), grid, dim3(256), 0, st, bt, N1, N2, rows, S, part, n);
    else
        hipLaunchKernelGGL((tn_gemm_batch_kernel<TnG>), grid, dim3(256), 0, st, bt, N1, N2, rows, S, part, n);
    hipLaunchKernelGGL(split_reduce_batch_kernel, dim3(cdiv(n, 256), nprob), dim3(256), 0, st, part, S, n, bt, accumulate);
    CPC_LAUNCH_CHECK();
    return 0;
}

int transpose(const float* in, float* out, int R, int Cn, hipStream_t st, int G, long in_gs, long out_gs) {
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(Cn, 32), cdiv(R, 32), G), dim3(256), 0, st, in, out, R, Cn, in_gs, out_gs);
    CPC_LAUNCH_CHECK();
    return 0;
}

int transpose_batch(const float* const* in, float* const* out, int n, int R, int Cn, hipStream_t st) {
    if (n <= 0 || n > 4) return CPC_ERR_ARG;
    TransposeBatch b;
    for (int i = 0; i < 4; ++i) { b.in[i] = in[i < n ? i : 0]; b.out[i] = out[i < n ? i : 0]; }
    hipLaunchKernelGGL(transpose_batch_kernel, dim3(cdiv(Cn, 32), cdiv(R, 32), n), dim3(256), 0, st, b, R, Cn);
    CPC_LAUNCH_CHECK();
    return 0;
}

}  // namespace cpc

using namespace cpc;

// C[M,N] = A[M,K] . B[N,K]^T + bias   (lda/ldb/ldc in floats; bias may be NULL)
extern "C" int cpc_gemm_nt(const float* A, int lda, const float* B, int ldb, const float* bias, float* C,
                           int ldc, int M, int N, int K, void* stream) {
    CPC_RETURN_IF(!A || !B || !C, CPC_ERR_ARG);
    return nt_gemm(plain_rows(A, M, lda), B, ldb, bias, C, ldc, N, K, (hipStream_t)stream);
}

extern "C" long cpc_gemm_tn_scratch_floats(int M, int N1, int N2) { return tn_gemm_part_floats(M, N1, N2); }

// C[N1,N2] (+)= A[M,N1]^T . B[M,N2]
extern "C" int cpc_gemm_tn(const float* A, int lda, const float* B, int ldb, float* part, float* C, int M,
                           int N1, int N2, int accumulate, void* stream) {
    CPC_RETURN_IF(!A || !B || !C || !part, CPC_ERR_ARG);
    return tn_gemm(plain_rows(A, M, lda), N1, plain_rows(B, M, ldb), N2, part, C, accumulate,
                   (hipStream_t)stream);
}

// 1 (default): plain GEMMs whose operand bounds are known run on two fp16 pieces in cpc_set_mfma_mode >= 2; 0: three bf16
// pieces always (A/B measurements, numerical comparisons)
// 1 (default): the transformer layer's feed-forward ReLU (forward, without dropout) and ReLU derivative (backward) run as
// epilogues of their GEMMs where those take the wide fp16-piece tile (GemmEpilogue); 0: always as elementwise kernels behind the
// GEMMs.  Same bits either way (tests/test_gpu_transformer.py).
extern "C" int cpc_set_gemm_fuse(int on) {
    cpc::g_gemm_fuse = on ? 1 : 0;
    return 0;
}
extern "C" int cpc_set_gemm_split(int on) {
    cpc::g_gemm_split = on ? 1 : 0;
    cpc::g_gemm_wide = on == 2 ? 0 : (on == 3 ? 2 : 1);   // 2: two fp16 pieces, but never the 128 x 256 tile (A/B measurements);
                                                           // 3: that tile whenever the shape allows, however few workgroups (tests)
    return 0;
}
