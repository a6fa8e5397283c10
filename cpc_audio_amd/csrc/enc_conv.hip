// Encoder layers 1..4: Conv1d(256 -> 256, k = 2s) + bias + ChannelNorm + ReLU and their
// backward, as implicit GEMMs on the exact-f32 matrix pipe.
//
// Reference: cpc/model.py:85-92 (conv1: k8 s4 p2; conv2-4: k4 s2 p1), :50-58
// (ChannelNorm), :101-104 (relu(norm(conv))).
//
// Layout: activations are channels-last (B, L, C) in HBM, so
//   * the im2col row of output step t is the CONTIGUOUS window x[b, t*s-p : t*s-p+k, :]
//     (k*256 floats): the conv is a plain NT GEMM  out[M=B*Lout, 256] = A[M, k*256] . Wp^T
//     with overlapping A rows and zero padding expressed by RowMap (gemm_tile.h);
//   * a block owns whole rows (all 256 output channels), so ChannelNorm's mean /
//     unbiased variance are reductions of accumulator registers (half-wave
//     shuffles + a 4-entry LDS exchange across the 4 N-waves) and the layer writes
//     its normalised (xhat) and rectified (y) rows straight from registers;
//   * dgrad (k = 2s) is s independent "phase" GEMMs with K = 512: input step tau with
//     (tau+p) = q*s + r receives  dx[q-1] . W[:,:,r+s] + dx[q] . W[:,:,r], again a
//     contiguous 2-row window of the (B, Lout, C) gradient; its epilogue applies the
//     PREVIOUS layer's ReLU'/ChannelNorm backward in registers and emits that layer's
//     pre-norm gradient plus per-block column partials for d(norm weight/bias) and
//     d(conv bias);
//   * wgrad is a TN GEMM  dWp[256, k*256] = dx^T . A  split over row ranges, reduced
//     in a fixed order and permuted back to PyTorch's (O, I, W) layout.
#include <algorithm>
#include <type_traits>

#include "cpc_common.h"
#include "cpc_internal.h"
#include "gemm_tile.h"

namespace cpc {

// ------------------------------------------------------------------ weight re-layouts
// (O,I,W) -> Wp(co, kg = kk*C + ci) for the forward NT GEMM, stored k-blocked: [kg/16][co][kg%16], so that the
// B tile of one 16-k chunk (256 rows x 64 B) is ONE contiguous 16 KB block instead of 256 half cache lines
// 8 KB apart (all CUs walk k in lockstep and would hammer the same few L2 channels).
// split != 0: Wp is written as three bf16 planes [3][total] (the pre-split B operand of NtTileX3).
// split == 2: fp16 two-piece slots [h0..h3 | l0..l3] of w * scale_for_amax(*amax) in place of each 4-float slot.
__device__ __forceinline__ void store_h2(float* base, long dst, float v, float scale) {
    const float x = v * scale;
    const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
    _Float16* o = reinterpret_cast<_Float16*>(base) + ((dst & ~3L) << 1) + (dst & 3);
    o[0] = h;
    o[4] = l;
}
__device__ __forceinline__ void permute_w_fwd_elem(const float* __restrict__ w, float* __restrict__ wp, int k,
                                                   int split, float amax, long idx) {
    const long total = (long)kC * k * kC;
    const int co = (int)(idx / (k * kC));
    const int rem = (int)(idx - (long)co * k * kC);
    const int kk = rem >> kCLog2, ci = rem & (kC - 1);
    const float v = w[((long)co * kC + ci) * k + kk];
    const long dst = ((long)(rem >> 4) * kC + co) * 16 + (rem & 15);
    if (split == 2) {
        store_h2(wp, dst, v, scale_for_amax(amax));
    } else if (split) {
        unsigned h, m, l;
        split3(v, h, m, l);
        unsigned short* o = reinterpret_cast<unsigned short*>(wp);
        o[dst] = (unsigned short)(h >> 16);
        o[total + dst] = (unsigned short)(m >> 16);
        o[2 * total + dst] = (unsigned short)(l >> 16);
    } else {
        wp[dst] = v;
    }
}
__global__ __launch_bounds__(256) void permute_w_fwd_kernel(const float* __restrict__ w,
                                                            float* __restrict__ wp, int k, int split,
                                                            const float* __restrict__ amax) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < (long)kC * k * kC) permute_w_fwd_elem(w, wp, k, split, split == 2 ? *amax : 0.f, idx);
}

// (O,I,W) -> Wd[r](ci, kg = j*C + co) = W[co][ci][r + (1-j)*s],  r < s, j in {0,1}; each phase r k-blocked
// like Wp: [kg/16][ci][kg%16]
__device__ __forceinline__ void permute_w_dgrad_elem(const float* __restrict__ w, float* __restrict__ wd, int s,
                                                     int split, float amax, long idx) {
    const long total = (long)s * kC * 2 * kC;
    const int k = 2 * s;
    const int r = (int)(idx / (kC * 2 * kC));
    const int rem = (int)(idx - (long)r * kC * 2 * kC);
    const int ci = rem / (2 * kC);
    const int jc = rem - ci * 2 * kC;
    const int j = jc >> kCLog2, co = jc & (kC - 1);
    const float v = w[((long)co * kC + ci) * k + r + (1 - j) * s];
    const long dst = (long)r * kC * 2 * kC + ((long)(jc >> 4) * kC + ci) * 16 + (jc & 15);
    if (split == 2) {
        store_h2(wd, dst, v, scale_for_amax(amax));
    } else if (split) {
        unsigned h, m, l;
        split3(v, h, m, l);
        unsigned short* o = reinterpret_cast<unsigned short*>(wd);
        o[dst] = (unsigned short)(h >> 16);
        o[total + dst] = (unsigned short)(m >> 16);
        o[2 * total + dst] = (unsigned short)(l >> 16);
    } else {
        wd[dst] = v;
    }
}
__global__ __launch_bounds__(256) void permute_w_dgrad_kernel(const float* __restrict__ w,
                                                              float* __restrict__ wd, int s, int split,
                                                              const float* __restrict__ amax) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < (long)s * kC * 2 * kC) permute_w_dgrad_elem(w, wd, s, split, split == 2 ? *amax : 0.f, idx);
}

// ------------------------------------------------------------------ operand bounds for the fp16-split mode
// out = max(out, max_i |x_i|): |x| as uint bits is monotone, so an integer atomicMax gives an order-independent,
// exact result.
// max|x| over the slice this workgroup (`blk` of `nblk`) walks; the result is valid in thread 0
__device__ __forceinline__ float block_absmax(const float* __restrict__ x, long n, int blk, int nblk) {
    __shared__ float red[4];
    float m = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blk * 256 + threadIdx.x; i < n4; i += (long)nblk * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (long i = (n4 << 2) + (long)blk * 256 + threadIdx.x; i < n; i += (long)nblk * 256) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    const float m = block_absmax(x, n, blockIdx.x, gridDim.x);
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
}
// Bound of a ChannelNorm output (cpc/model.py:50-58): |xhat| <= sqrt(C-1) (unbiased variance), so
// |y| <= sqrt(C-1) * max|w| + max|b|; ReLU only shrinks it.  One workgroup of 256 threads.
__device__ __forceinline__ void norm_bound_block(const float* __restrict__ nw, const float* __restrict__ nb,
                                                 float* __restrict__ out) {
    __shared__ float red[2][4];
    const float w = wave_max(fabsf(nw[threadIdx.x])), b = wave_max(fabsf(nb[threadIdx.x]));
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = w; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mw = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        const float mb = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        *out = 15.968719f * mw + mb;                             // sqrt(255)
    }
}
__global__ __launch_bounds__(256) void norm_bound_kernel(const float* __restrict__ nw, const float* __restrict__ nb,
                                                         float* __restrict__ out) {
    norm_bound_block(nw, nb, out);
}

// ---- all per-step weight preparation of layers 1..4 in two launches (the composite encoder; the per-layer entry
// points keep their own).  The weights change once per optimiser step, and as separate launches the 4 x (bound,
// max|w|, forward layout, dgrad layout) are ~30 kernels of ~5 us each: more than a whole conv layer.
//   enc_prep_amax_kernel    grid (kPrepParts, 5): rows 0..3 = partial max|w| of layer y+1 (plain stores, no atomics, no
//                           zero fill), row 4 = the ChannelNorm bounds of the four layers' inputs
//   enc_prep_permute_kernel one thread per (layer, layout, element); every workgroup folds its layer's kPrepParts
//                           partials itself (a max is order-independent)
constexpr int kPrepParts = 32;
struct PrepArgs {
    const float* w[4];      // conv{1..4}.weight
    float* wp[4];           // forward layouts   (+ max|w| behind them in the fp16-split mode)
    float* wd[4];           // dgrad layouts     (same)
    const float* nw[4];     // batchNorm{0..3}.weight / .bias: the producers of the four inputs
    const float* nb[4];
    float* bound;           // [4]
    float* partial;         // [4][kPrepParts]
    int k[4];
    int blk0[9];            // first workgroup of segment 2*layer + layout
    int split;
    int fwd_h2[4];          // 1: forward layout of layer y+1 in K-tile-major H2 rows (the DMA kernel's, conv_dma.hip);
                            // 2: forward AND dgrad layouts in K-tile-major bf16 rows (mode 4)
    int dgrad_h2[4];        // 1: dgrad layout of layer y+1 in K-tile-major H2 rows (conv_dgrad_dma_kernel<.., 2>)
    int mask;               // bit i (1..4): layer i is prepared by this launch; bit 0: the input bounds of layers 2..4 (the
                            // ChannelNorm affines of layers 1..3); bit 5: the input bound of layer 1 (layer 0's affine)
};
__global__ __launch_bounds__(256) void enc_prep_amax_kernel(PrepArgs a) {
    const int y = blockIdx.y;
    if (y == 4) {
        if (blockIdx.x < 4 && (a.mask & (blockIdx.x == 0 ? 32 : 1)))
            norm_bound_block(a.nw[blockIdx.x], a.nb[blockIdx.x], a.bound + blockIdx.x);
        return;
    }
    if (!((a.mask >> (y + 1)) & 1)) return;                   // block-uniform
    const float m = block_absmax(a.w[y], (long)kC * a.k[y] * kC, blockIdx.x, kPrepParts);
    if (threadIdx.x == 0) a.partial[y * kPrepParts + blockIdx.x] = m;
}
// row 4 of enc_prep_amax_kernel on its own (4 workgroups): what layer 0 needs of the preparation -- the bound its H2 output is
// scaled by -- when the rest runs on another stream beside it (enc_set_weight_prep_stream)
__global__ __launch_bounds__(256) void enc_prep_bounds_kernel(PrepArgs a) {
    if (a.mask & (blockIdx.x == 0 ? 32 : 1)) norm_bound_block(a.nw[blockIdx.x], a.nb[blockIdx.x], a.bound + blockIdx.x);
}
__global__ __launch_bounds__(256) void enc_prep_permute_kernel(PrepArgs a) {
    const int b = blockIdx.x;
    int seg = 0;
    while (seg < 7 && b >= a.blk0[seg + 1]) ++seg;            // block-uniform
    const int layer = seg >> 1, k = a.k[layer];
    const long total = (long)kC * k * kC;
    static_assert(kPrepParts == 32, "one partial per half-wave lane");
    const float amax = wave_max(a.partial[layer * kPrepParts + (threadIdx.x & 31)]);
    float* dst = (seg & 1) ? a.wd[layer] : a.wp[layer];
    const long idx = (long)(b - a.blk0[seg]) * 256 + threadIdx.x;
    if (a.split == 2 && idx == 0) dst[total] = amax;          // where the GEMM kernels read max|w|
    if (idx >= total) return;
    if ((seg & 1) && a.fwd_h2[layer] == 2) permute_w_dgrad_bf16_elem(a.w[layer], reinterpret_cast<unsigned short*>(dst), k / 2, idx);
    else if ((seg & 1) && a.dgrad_h2[layer]) permute_w_dgrad_h2_elem(a.w[layer], reinterpret_cast<unsigned char*>(dst), k / 2, amax, idx);
    else if (seg & 1) permute_w_dgrad_elem(a.w[layer], dst, k / 2, a.split, amax, idx);
    else if (a.fwd_h2[layer] == 2) permute_w_fwd_bf16_elem(a.w[layer], reinterpret_cast<unsigned short*>(dst), k, idx);
    else if (a.fwd_h2[layer]) permute_w_h2_elem(a.w[layer], reinterpret_cast<unsigned char*>(dst), k, amax, idx);
    else permute_w_fwd_elem(a.w[layer], dst, k, a.split, amax, idx);
}

// max|dx| of a layer as its consumers read it.  The kernel that writes dx accumulates it with integer atomicMax on the float
// bits; 8192 waves hammering ONE address serialise at L2 (measured: the largest norm backward spent ~0.12 ms, as long as its
// 200 MB of traffic should take twice over), so the composite encoder spreads them over kAmaxSlots addresses by workgroup
// and the consuming GEMM kernels fold the slots (a max is order-independent: still exact and deterministic).
// (kAmaxSlots, fold_amax: gemm_tile.h)

// ------------------------------------------------------------------ forward
// MODE: 0 = exact-f32 MFMA, 1 = three bf16 pieces (6 MFMAs per product), 2 = two fp16 pieces (3 MFMAs per
// product; operands scaled by powers of two derived from bounds on their max|.|, see gemm_tile.h)
// AH2 (MODE 2 only): the A operand (the layer's input activation, or its output gradient in the data-gradient kernel) lies in
// H2 storage and is staged without conversion (NtTileX3's AH2)
// PIPE: the software-pipelined 16-k schedule (two LDS stages, four register sets of global loads in flight) for the 32- / 64-row
// tiles as well; without it they run one 32-k LDS stage with ONE chunk prefetched, i.e. a global-load latency per chunk
// with 12 MFMAs to hide it behind (the short layers at B = 64)
template <int BM, int MODE, bool AH2 = false, bool PIPE = false>
struct ConvCfg {
    static_assert(!AH2 || MODE == 2, "H2 operands belong to the fp16-split mode");
    static constexpr bool X3 = MODE != 0;
    static constexpr int NP = MODE == 2 ? 2 : 3;
    static constexpr bool H2 = NP == 2;
    // (64-row tiles on the pipelined schedule: EIGHT waves of 32 x 64, like two 32-row tiles sharing one weight stage -- half the
    // weight bytes through L2 per output row at the same two waves per SIMD; with four 64 x 64 waves the tile lost to fill)
    static constexpr int WAVES_M = BM >= 128 ? 2 : (BM == 64 && PIPE ? 2 : 1);
    // 128-row tiles: two LDS stages of 16 k with the skewed (store-first / MFMA-first) wave schedule
    // (pre-split weight planes, BSPLIT = true, measured slower: 162 vs 176 TF on layer 1 -- three 8-byte loads per
    //  slot instead of one 16-byte load cost more than the VALU they save)
    // mode 2: the weight re-layout kernels also split (same 4 bytes per weight, no VALU left for B in the main loop --
    // what made the 32-row tiles of the short layers slower with on-the-fly fp16 conversion: 108 vs 76 us)
    static constexpr bool kPreSplitW = MODE == 2;
    using X3Tile = typename std::conditional<BM == 128 || PIPE, NtTileX3<BM, kC, WAVES_M, 4, 16, 2, true, kPreSplitW, NP, AH2>,
                                             NtTileX3<BM, kC, WAVES_M, 4, 32, 1, false, kPreSplitW, NP, AH2>>::type;
    using Tile = typename std::conditional<X3, X3Tile, NtTile<BM, kC, WAVES_M, 4>>::type;
};

// One accumulator element per lane, lanes 2j / 2j + 1 holding channels c / c + 1 (c even) of one row: the value's two fp16 pieces
// into H2 storage with ONE dword store per lane -- neighbours swap a piece (quad_perm [1,0,3,2]) so that the even lane owns
// the h pair of the two channels and the odd lane the l pair.  Convergent: every lane of the wave calls it (`live` guards the store).
__device__ __forceinline__ void h2_store_lane(unsigned char* row, int c, float v, float s, bool live) {
    _Float16 h, l;
    h2_split(v, s, h, l);
    const unsigned hb = __builtin_bit_cast(unsigned short, h), lb = __builtin_bit_cast(unsigned short, l);
    const bool odd = c & 1;
    const unsigned got = __builtin_bit_cast(unsigned, dpp_mov<0xB1>(__builtin_bit_cast(float, odd ? hb : lb)));
    const unsigned word = odd ? (got | (lb << 16)) : (hb | (got << 16));
    if (live) *reinterpret_cast<unsigned*>(row + h2_byte_of(c & ~1) + (odd ? 16 : 0)) = word;
}

// x_amax / w_amax (MODE 2 only): device floats holding upper bounds of max|x| and max|w|
// y_amax (MODE 2, may be NULL): y is written in H2 storage scaled for the bound *y_amax (which must bound |y|: the layer's
// ChannelNorm bound) instead of fp32
template <int BM, int MODE, bool AH2 = false, bool PIPE = false>
__global__ __launch_bounds__((ConvCfg<BM, MODE, AH2, PIPE>::Tile::NTHREADS), (BM == 64 && !PIPE ? 2 : 1)) void conv_fwd_kernel(
    RowMap am, const float* __restrict__ wp, int K, const float* __restrict__ bias,
    const float* __restrict__ nw, const float* __restrict__ nb, float* __restrict__ y,
    float* __restrict__ xhat, float* __restrict__ rstd_out, const float* __restrict__ x_amax,
    const float* __restrict__ w_amax, const float* __restrict__ y_amax = nullptr) {
    using Tile = typename ConvCfg<BM, MODE, AH2, PIPE>::Tile;
    constexpr bool X3 = MODE != 0;
    constexpr int TM = Tile::TM, TN = Tile::TN;
    __shared__ float smem[Tile::SMEM_FLOATS];
    __shared__ float red[BM][4];
    const int m0 = blockIdx.x * BM;
    f32x16 acc[TM][TN];
    zero_acc(acc);
    if constexpr (ConvCfg<BM, MODE, AH2>::H2) {
        const float sa = scale_for_amax(*x_amax), sb = scale_for_amax(*w_amax);
        Tile::run(acc, am, m0, wp, 16, 0, K, smem, 0, kC * 16, (int)((blockIdx.x * 4u) % (unsigned)(K / Tile::BK)), sa, sb,
                  ((K >> kCLog2) & ((K >> kCLog2) - 1)) == 0 ? 31 - __builtin_clz(K >> kCLog2) : 0);   // tap-fastest K walk
                                                                                                      // (power-of-two tap counts)
        const float inv = 1.0f / (sa * sb);                      // powers of two: exact
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] *= inv;
    } else if constexpr (X3) {
        Tile::run(acc, am, m0, wp, 16, 0, K, smem, (long)kC * K, kC * 16,    // plane stride used only if pre-split
                  (int)((blockIdx.x * 4u) % (unsigned)(K / Tile::BK)));
    } else {
        Tile::run(acc, am, m0, wp, 16, 0, K, smem, kC * 16);
    }

    const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 3;
    int col[TN];
    float gw[TN], gb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        col[tn] = Tile::c_col(tn);
        const float bc = bias[col[tn]];
        gw[tn] = nw[col[tn]];
        gb[tn] = nb[col[tn]];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] += bc;
    }
    // ---- ChannelNorm statistics, two passes over the accumulators (cpc/model.py:52-54)
    float mean[TM][16], rstd[TM][16];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = 0.f;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) v += acc[tm][tn][r];
            v = half_wave_sum(v);
            if ((lane & 31) == 0) red[Tile::c_row(tm, r)][wn] = v;
        }
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = Tile::c_row(tm, r);
            mean[tm][r] = ((red[row][0] + red[row][1]) + (red[row][2] + red[row][3])) * (1.0f / kC);
        }
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = 0.f;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const float d = acc[tm][tn][r] - mean[tm][r];
                v = fmaf(d, d, v);
            }
            v = half_wave_sum(v);
            if ((lane & 31) == 0) red[Tile::c_row(tm, r)][wn] = v;
        }
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = Tile::c_row(tm, r);
            const float var = ((red[row][0] + red[row][1]) + (red[row][2] + red[row][3])) * (1.0f / (kC - 1));
            rstd[tm][r] = 1.0f / sqrtf(var + kNormEps);
            const int m = m0 + row;
            if constexpr (MODE == 2) {
                if (y_amax != nullptr) {                     // y in H2 storage (block-uniform branch; every lane takes part in the swap)
                    const float sy = scale_for_amax(*y_amax);
                    const bool live = m < am.M;
                    if (live && wn == 0 && (lane & 31) == 0) rstd_out[m] = rstd[tm][r];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const float xh = (acc[tm][tn][r] - mean[tm][r]) * rstd[tm][r];
                        if (live) __builtin_nontemporal_store(xh, xhat + (long)m * kC + col[tn]);
                        h2_store_lane(reinterpret_cast<unsigned char*>(y) + (long)(live ? m : 0) * (kC * 4), col[tn],
                                      fmaxf(relu_in(fmaf(xh, gw[tn], gb[tn])), 0.f), sy, live);
                    }
                    continue;
                }
            }
            if (m < am.M) {
                if (wn == 0 && (lane & 31) == 0) rstd_out[m] = rstd[tm][r];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const float xh = (acc[tm][tn][r] - mean[tm][r]) * rstd[tm][r];
                    __builtin_nontemporal_store(xh, xhat + (long)m * kC + col[tn]);     // read again only in backward
                    y[(long)m * kC + col[tn]] = fmaxf(relu_in(fmaf(xh, gw[tn], gb[tn])), 0.f);
                }
            }
        }
}

// ------------------------------------------------------------------ norm backward helpers
// ReLU' + ChannelNorm backward for one row (cpc/model.py:50-58 differentiated):
//   dyh = dy * [y > 0];  dxh = dyh * w;
//   dx  = rstd * (dxh - mean_c(dxh) - xhat * sum_c(dxh*xhat) / (C-1))      (unbiased variance)
// plus column sums  d(norm w) += dyh*xhat,  d(norm b) += dyh,  d(conv bias) += dx.

// stand-alone version for the top layer (its dy comes from autograd, not from a dgrad GEMM)
constexpr int NB_ROWS = 32;   // rows per block (8 per wave)
// dx_amax (may be NULL): max|dx| is accumulated into it for the fp16-split GEMMs that consume dx
// MASK: where [y > 0] comes from.  0: y in fp32; 1: y in H2 storage (cpc_common.h; the sign of the h piece is the value's);
// 2: recomputed as fmaf(xhat, w, b) > 0 -- the very expression the forward kernels rectify, on the very xhat they stored,
// so the mask is bit-identical and the 4 bytes per element of y are not read at all (the composite encoder's choice).
// The kernel streams 12-16 bytes per element and is latency-bound per row (two dependent wave reductions), so a wave
// keeps the loads of four rows in flight and interleaves their reduction chains (0.25 -> ... ms per step for the four layers).
// DYB / XB: dy / (xhat and the output dx) are bf16 tensors (the bf16-storage variant, mode 4) instead of fp32.
// DXH2: dx is written in H2 storage (cpc_common.h) for the DMA data-gradient kernel and the weight gradient that read it next.
// Its scale must be known before the first element is written, so it comes from a bound instead of the measured maximum:
//   |dx| = rstd |dxh - mean(dxh) - xhat mean'(dxh xhat)| <= rstd max|dxh| (1 + 1 + sqrt(C-1) sqrt(C/(C-1)))
//        (|xhat| <= sqrt(C-1); with the unbiased variance sum|xhat| / (C-1) <= sqrt(C/(C-1)) = 1.002, not 1)
//        <= eps^-1/2 * 18.1 * max|w| * max|dy|
// with max|dy| measured by the kernel that wrote dy (dy_amax, dy_slots partial maxima).  The bound is loose by the ratio of
// 316 to the typical rstd and of 18 to ~2, i.e. 2^8..2^11 -- inside the 2^17 that the two-piece storage absorbs without any
// loss (gemm_tile.h); workgroup 0 leaves it in *dx_bound for the readers.
constexpr float kNormBwdBound = 316.22777f * 18.1f;      // 2 + sqrt(255) * 1.002 = 18.0, rounded up
#ifndef CPC_NBW
#define CPC_NBW 8
#endif
constexpr int NBW = CPC_NBW;        // rows a wave works on at a time (all of its eight: one batch of loads, one round of reductions)
__device__ __forceinline__ f32x4 load4_as_f32(const float* base, long elem, bool bf16) {
    if (bf16) {
        const uint2 d = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem);
        return f32x4{bf16_val((unsigned short)(d.x & 0xFFFFu)), bf16_val((unsigned short)(d.x >> 16)),
                     bf16_val((unsigned short)(d.y & 0xFFFFu)), bf16_val((unsigned short)(d.y >> 16))};
    }
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + elem));       // read once
}
template <int MASK, bool DYB = false, bool XB = false, bool DXH2 = false>
__global__ __launch_bounds__(256) void norm_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ y,
    const float* __restrict__ rstd, const float* __restrict__ nw, const float* __restrict__ nb, float* __restrict__ dx,
    float* __restrict__ colpart, int M, float* __restrict__ dx_amax, int amax_slots,
    const float* __restrict__ dy_amax = nullptr, int dy_slots = 1, float* __restrict__ dx_bound = nullptr) {
    __shared__ float red[4][3][kC];
    __shared__ float wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = lane * 4;
    const float4 w4 = *reinterpret_cast<const float4*>(nw + c);
    const float gw[4] = {w4.x, w4.y, w4.z, w4.w};
    float sdx = 1.0f;
    if constexpr (DXH2) {
        const float wm = wave_max(fmaxf(fmaxf(fabsf(gw[0]), fabsf(gw[1])), fmaxf(fabsf(gw[2]), fabsf(gw[3]))));
        const float bound = kNormBwdBound * wm * fold_amax(dy_amax, dy_slots);
        sdx = scale_for_amax(bound);
        if (blockIdx.x == 0 && tid == 0) *dx_bound = bound;
    }
    float gb[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MASK == 2) {
        const float4 b4 = *reinterpret_cast<const float4*>(nb + c);
        gb[0] = b4.x; gb[1] = b4.y; gb[2] = b4.z; gb[3] = b4.w;
    }
    float cs[3][4];
    float amax = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) cs[a][q] = 0.f;
    static_assert(NB_ROWS % (4 * NBW) == 0, "whole batches of rows per wave");
    for (int r0 = wv; r0 < NB_ROWS; r0 += 4 * NBW) {
        // rows r0, r0 + 4, ... of this block (the four waves interleave); out-of-range rows read row 0 and are dropped
        int m[NBW];
        bool live[NBW];
        f32x4 g4[NBW], x4[NBW];
        float yv[NBW][4], rs[NBW];
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            m[j] = blockIdx.x * NB_ROWS + r0 + 4 * j;
            live[j] = m[j] < M;
            const long row = live[j] ? m[j] : 0;
            g4[j] = load4_as_f32(dy, row * kC + c, DYB);
            x4[j] = load4_as_f32(xhat, row * kC + c, XB);
            rs[j] = rstd[row];
            if constexpr (MASK == 1) {
                uint2 hp, lp;
                h2_load4_raw(y + row * kC, c, hp, lp);
                yv[j][0] = (float)(short)(hp.x & 0xFFFFu); yv[j][1] = (float)(short)(hp.x >> 16);   // > 0 iff the fp16 piece is
                yv[j][2] = (float)(short)(hp.y & 0xFFFFu); yv[j][3] = (float)(short)(hp.y >> 16);
            } else if constexpr (MASK == 0) {
                const float4 y4 = *reinterpret_cast<const float4*>(y + row * kC + c);
                yv[j][0] = y4.x; yv[j][1] = y4.y; yv[j][2] = y4.z; yv[j][3] = y4.w;
            }
        }
        float dxh[NBW][4], s1[NBW], s2[NBW];
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            s1[j] = 0.f; s2[j] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float yq = MASK == 2 ? relu_in(fmaf(x4[j][q], gw[q], gb[q])) : yv[j][q];
                const float dyh = (live[j] && yq > 0.f) ? g4[j][q] : 0.f;
                cs[0][q] = fmaf(dyh, x4[j][q], cs[0][q]);
                cs[1][q] += dyh;
                dxh[j][q] = dyh * gw[q];
                s1[j] += dxh[j][q];
                s2[j] = fmaf(dxh[j][q], x4[j][q], s2[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < NBW; ++j) {          // independent chains: the DPP steps of the rows interleave
            s1[j] = wave_sum(s1[j]) * (1.0f / kC);
            s2[j] = wave_sum(s2[j]) * (1.0f / (kC - 1));
        }
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = live[j] ? rs[j] * (dxh[j][q] - s1[j] - x4[j][q] * s2[j]) : 0.f;
                o[q] = v;
                cs[2][q] += v;
                amax = fmaxf(amax, fabsf(v));
            }
            if (live[j]) {
                if constexpr (DXH2)
                    h2_store_row_nt(reinterpret_cast<unsigned char*>(dx) + (long)m[j] * (kC * 4), o[0], o[1], o[2], o[3], sdx);
                else if constexpr (XB)
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dx) + (long)m[j] * kC + c) =
                        make_uint2(bf16_rne(o[0]) | ((unsigned)bf16_rne(o[1]) << 16), bf16_rne(o[2]) | ((unsigned)bf16_rne(o[3]) << 16));
                else
                    *reinterpret_cast<f32x4*>(dx + (long)m[j] * kC + c) = o;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wv][a][c + q] = cs[a][q];
    __syncthreads();
    float* prow = colpart + (long)blockIdx.x * (3 * kC);
    for (int i = tid; i < 3 * kC; i += 256) {
        const int a = i >> kCLog2, cc = i & (kC - 1);
        prow[i] = (red[0][a][cc] + red[1][a][cc]) + (red[2][a][cc] + red[3][a][cc]);
    }
    if (dx_amax != nullptr) {                  // one atomic per workgroup, spread over amax_slots addresses (fold_amax)
        amax = wave_max(amax);
        if (lane == 0) wmax[wv] = amax;
        __syncthreads();
        if (tid == 0)
            atomicMax(reinterpret_cast<unsigned*>(dx_amax + (int)(blockIdx.x % (unsigned)amax_slots)),
                      __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
    }
}

// ------------------------------------------------------------------ dgrad (+ fused norm backward)
// grid = (row tiles over B*(Lout+1), s phases).  am = 2-row windows over dx of THIS layer.
// MODE 2: dx_amax / w_amax bound the operands; prev_amax (FUSE, may be NULL) receives max|dprev|.
template <int BM, bool FUSE, int MODE, bool AH2 = false, bool PIPE = false>
__global__ __launch_bounds__((ConvCfg<BM, MODE, AH2, PIPE>::Tile::NTHREADS), (BM == 64 && !PIPE ? 2 : 1)) void conv_dgrad_kernel(
    RowMap am, const float* __restrict__ wd, int s, int p, int Lin,
    const float* __restrict__ xhat_prev, const float* __restrict__ y_prev,
    const float* __restrict__ rstd_prev, const float* __restrict__ nw_prev,
    float* __restrict__ dprev, float* __restrict__ colpart, const float* __restrict__ dx_amax,
    const float* __restrict__ w_amax, float* __restrict__ prev_amax, int amax_slots) {
    static_assert(!(AH2 && FUSE), "the H2-input data gradient is the plain (unfused) one");
    using Tile = typename ConvCfg<BM, MODE, AH2, PIPE>::Tile;
    constexpr bool X3 = MODE != 0;
    constexpr int TM = Tile::TM, TN = Tile::TN, WAVES_M = ConvCfg<BM, MODE, AH2, PIPE>::WAVES_M;
    __shared__ float smem[Tile::SMEM_FLOATS];
    __shared__ float red[2][BM][4];
    __shared__ float colsum[3][kC];
    const int m0 = blockIdx.x * BM;
    const int ph = blockIdx.y;
    const int q0 = (am.R == am.Lin && ph < p) ? 1 : 0;              // exact rows (dgrad_rows): phase ph starts at q = q0
    am.off += q0 * kC;
    am.tadd += q0;
    f32x16 acc[TM][TN];
    zero_acc(acc);
    if constexpr (ConvCfg<BM, MODE>::H2) {
        const float sa = scale_for_amax(fold_amax(dx_amax, amax_slots)), sb = scale_for_amax(*w_amax);
        Tile::run(acc, am, m0, wd + (long)ph * kC * 2 * kC, 16, 0, 2 * kC, smem, 0, kC * 16, 0, sa, sb, 1);     // two 256-wide segments
        const float inv = 1.0f / (sa * sb);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] *= inv;
    } else if constexpr (X3 && ConvCfg<BM, MODE>::kPreSplitW)   // wd = 3 bf16 planes of [s][256][512]
        Tile::run(acc, am, m0, reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(wd) + (long)ph * kC * 2 * kC),
                  16, 0, 2 * kC, smem, (long)s * kC * 2 * kC, kC * 16);
    else if constexpr (X3)
        Tile::run(acc, am, m0, wd + (long)ph * kC * 2 * kC, 16, 0, 2 * kC, smem, 0, kC * 16);
    else
        Tile::run(acc, am, m0, wd + (long)ph * kC * 2 * kC, 16, 0, 2 * kC, smem, kC * 16);

    const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 3;
    int col[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) col[tn] = Tile::c_col(tn);

    // output row of every accumulator row: m -> (b, q) -> tau = q*s + ph - p
    int orow[TM][16];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + Tile::c_row(tm, r);
            int o = -1;
            if (m < am.M) {
                const int b = m / am.R, q = m - b * am.R + q0;
                const int tau = q * s + ph - p;
                if ((unsigned)tau < (unsigned)Lin) o = b * Lin + tau;
            }
            orow[tm][r] = o;
        }

    if (!FUSE) {
        float amax = 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (orow[tm][r] >= 0) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        dprev[(long)orow[tm][r] * kC + col[tn]] = acc[tm][tn][r];
                        amax = fmaxf(amax, fabsf(acc[tm][tn][r]));
                    }
                }
        if (prev_amax != nullptr) {          // max|dprev| for a norm backward that writes H2 (norm_bwd_kernel, DXH2)
            amax = wave_max(amax);
            if ((threadIdx.x & 63) == 0)
                atomicMax(reinterpret_cast<unsigned*>(prev_amax + (int)((blockIdx.x + blockIdx.y) % (unsigned)kAmaxSlots)),
                          __float_as_uint(amax));
        }
        return;
    }

    float gw[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) gw[tn] = nw_prev[col[tn]];
    for (int i = threadIdx.x; i < 3 * kC; i += Tile::NTHREADS) (&colsum[0][0])[i] = 0.f;

    f32x16 xh[TM][TN];
    float amax = 0.f;
    float cs_w[TN], cs_b[TN], cs_c[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) { cs_w[tn] = 0.f; cs_b[tn] = 0.f; cs_c[tn] = 0.f; }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = orow[tm][r];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                float xv = 0.f, dxh = 0.f;
                if (o >= 0) {
                    xv = xhat_prev[(long)o * kC + col[tn]];
                    const float yv = y_prev[(long)o * kC + col[tn]];
                    const float dyh = yv > 0.f ? acc[tm][tn][r] : 0.f;
                    cs_w[tn] = fmaf(dyh, xv, cs_w[tn]);
                    cs_b[tn] += dyh;
                    dxh = dyh * gw[tn];
                }
                xh[tm][tn][r] = xv;
                acc[tm][tn][r] = dxh;
                s1 += dxh;
                s2 = fmaf(dxh, xv, s2);
            }
            s1 = half_wave_sum(s1);
            s2 = half_wave_sum(s2);
            if ((lane & 31) == 0) {
                red[0][Tile::c_row(tm, r)][wn] = s1;
                red[1][Tile::c_row(tm, r)][wn] = s2;
            }
        }
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = orow[tm][r];
            const int row = Tile::c_row(tm, r);
            const float S1 = ((red[0][row][0] + red[0][row][1]) + (red[0][row][2] + red[0][row][3])) * (1.0f / kC);
            const float S2 = ((red[1][row][0] + red[1][row][1]) + (red[1][row][2] + red[1][row][3])) * (1.0f / (kC - 1));
            if (o >= 0) {
                const float rs = rstd_prev[o];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const float dx = rs * (acc[tm][tn][r] - S1 - xh[tm][tn][r] * S2);
                    cs_c[tn] += dx;
                    amax = fmaxf(amax, fabsf(dx));
                    dprev[(long)o * kC + col[tn]] = dx;
                }
            }
        }
    // column partials: merge the two half-waves, then the WAVES_M waves sharing a column
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        float a = cs_w[tn], b2 = cs_b[tn], c2 = cs_c[tn];
        a += __shfl_xor(a, 32);
        b2 += __shfl_xor(b2, 32);
        c2 += __shfl_xor(c2, 32);
        if (lane < 32) {
            if (WAVES_M == 1) {
                colsum[0][col[tn]] = a; colsum[1][col[tn]] = b2; colsum[2][col[tn]] = c2;
            } else {          // two contributions per column: the sum is order-independent
                atomicAdd(&colsum[0][col[tn]], a);
                atomicAdd(&colsum[1][col[tn]], b2);
                atomicAdd(&colsum[2][col[tn]], c2);
            }
        }
    }
    __syncthreads();
    float* prow = colpart + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (3 * kC);
    for (int i = threadIdx.x; i < 3 * kC; i += Tile::NTHREADS) prow[i] = (&colsum[0][0])[i];
    if (prev_amax != nullptr) {
        amax = wave_max(amax);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(prev_amax + (int)(blockIdx.x % (unsigned)amax_slots)), __float_as_uint(amax));
    }
}

// The plain H2-fed data gradient on 128 x 128 tiles: two workgroups per 128-row tile, each 128 of the 256 input channels
// (blockIdx.z).  A workgroup's operand traffic through L2 is (rows + columns) x K, so for a fixed number of workgroups -- the chip
// wants >= 256 -- square tiles move the least: layer 3 at B = 64, 16384 rows x 2 phases, goes from 1024 workgroups of 32 x 256
// (537 MB of weight re-reads) to 512 of 128 x 128 (2.25x less).  The data gradient has no ChannelNorm in its epilogue, so
// splitting the channels costs nothing but the second read of the gradient rows.  Pipelined 16-k schedule, fp16 pieces.
using DgradNsTile = NtTileX3<128, 128, 2, 2, 16, 2, true, true, 2, true>;
__global__ __launch_bounds__(DgradNsTile::NTHREADS) void conv_dgrad_nsplit_kernel(
    RowMap am, const float* __restrict__ wd, int s, int p, int Lin, float* __restrict__ dprev,
    const float* __restrict__ dx_bound, const float* __restrict__ w_amax, float* __restrict__ prev_amax) {
    using Tile = DgradNsTile;
    constexpr int TM = Tile::TM, TN = Tile::TN;
    __shared__ float smem[Tile::SMEM_FLOATS];
    const int m0 = blockIdx.x * 128, ph = blockIdx.y, n0 = blockIdx.z * 128;
    const int q0 = (am.R == am.Lin && ph < p) ? 1 : 0;              // exact rows (dgrad_rows): phase ph starts at q = q0
    am.off += q0 * kC;
    am.tadd += q0;
    f32x16 acc[TM][TN];
    zero_acc(acc);
    const float sa = scale_for_amax(*dx_bound), sb = scale_for_amax(*w_amax);
    Tile::run(acc, am, m0, wd + (long)ph * kC * 2 * kC, 16, n0, 2 * kC, smem, 0, kC * 16, 0, sa, sb, 1);
    const float inv = 1.0f / (sa * sb);
    float amax = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + Tile::c_row(tm, r);
            if (m >= am.M) continue;
            const int b = m / am.R, q = m - b * am.R + q0;
            const int tau = q * s + ph - p;
            if ((unsigned)tau >= (unsigned)Lin) continue;
            float* out = dprev + ((long)b * Lin + tau) * kC + n0;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const float v = acc[tm][tn][r] * inv;
                out[Tile::c_col(tn)] = v;
                amax = fmaxf(amax, fabsf(v));
            }
        }
    if (prev_amax != nullptr) {              // max|dprev| for the norm backward below, which writes H2 (64 slots)
        amax = wave_max(amax);
        if ((threadIdx.x & 63) == 0)
            atomicMax(reinterpret_cast<unsigned*>(prev_amax + (int)((blockIdx.x + blockIdx.y + blockIdx.z) % (unsigned)kAmaxSlots)),
                      __float_as_uint(amax));
    }
}

// The forward of a short layer on the same 128 x 128 tiles: two workgroups per 128-row tile, each 128 of the 256 output channels.
// ChannelNorm needs the statistics of the whole row, so the pair exchanges two floats per row through global memory -- the sum
// and the centred sum of squares of its half, combined exactly as a parallel variance is (Chan et al.):
//     mean = (s_lo + s_hi) / 256,  M2 = M2_lo + M2_hi + (mean_hi - mean_lo)^2 * 64,  var = M2 / 255
// both halves evaluate this with the same operand order, i.e. to the same bits.  The exchange buffer is pre-filled with 0xFF bytes
// (no arithmetic result carries that payload: the hand-over scheme of the persistent recurrence, gru.hip); a workgroup stores its
// 8 bytes per row with one agent-scope atomic and polls its partner's.  Partners are 8 workgroup ids apart -- adjacent in
// dispatch order and on one XCD -- so the partner of a resident workgroup is resident or next in line; polling is bounded and a
// partner that never shows up lets NaN through (and sets bit 2 of cpc_device_error_flags) instead of hanging the device.
static __device__ unsigned g_enc_xch_timeout = 0;
constexpr unsigned long long kXchEmpty = 0xFFFFFFFFFFFFFFFFull;
__global__ __launch_bounds__(DgradNsTile::NTHREADS) void conv_fwd_nsplit_kernel(
    RowMap am, const float* __restrict__ wp, int K, const float* __restrict__ bias, const float* __restrict__ nw,
    const float* __restrict__ nb, float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ rstd_out,
    const float* __restrict__ x_amax, const float* __restrict__ w_amax, const float* __restrict__ y_amax,
    unsigned long long* __restrict__ xch, int spin_limit) {
    using Tile = DgradNsTile;
    constexpr int TM = Tile::TM, TN = Tile::TN;
    __shared__ float smem[Tile::SMEM_FLOATS];
    __shared__ float red[128][2];
    __shared__ float stat[128][2];                   // mean, rstd of the whole row
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = (slot >> 1) * 8 + xcd, half = slot & 1;
    const int m0 = tile * 128, n0 = half * 128;
    if (m0 >= am.M) return;                          // block-uniform (both workgroups of the pair)
    f32x16 acc[TM][TN];
    zero_acc(acc);
    const float sa = scale_for_amax(*x_amax), sb = scale_for_amax(*w_amax);
    Tile::run(acc, am, m0, wp, 16, n0, K, smem, 0, kC * 16, (int)((tile * 4u) % (unsigned)(K / Tile::BK)), sa, sb,
              ((K >> kCLog2) & ((K >> kCLog2) - 1)) == 0 ? 31 - __builtin_clz(K >> kCLog2) : 0);
    const float inv = 1.0f / (sa * sb);
    const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 1;
    int col[TN];
    float gw[TN], gb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        col[tn] = n0 + Tile::c_col(tn);
        const float bc = bias[col[tn]];
        gw[tn] = nw[col[tn]];
        gb[tn] = nb[col[tn]];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = fmaf(acc[tm][tn][r], inv, bc);
    }
    // ---- this half's statistics: sum, then the sum of squares around the half's own mean
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = 0.f;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) v += acc[tm][tn][r];
            v = half_wave_sum(v);
            if ((lane & 31) == 0) red[Tile::c_row(tm, r)][wn] = v;
        }
    __syncthreads();
    float hsum[TM][16];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = Tile::c_row(tm, r);
            hsum[tm][r] = red[row][0] + red[row][1];
        }
    const float my_sum = threadIdx.x < 128 ? red[threadIdx.x][0] + red[threadIdx.x][1] : 0.f;     // thread `row` talks for row `row`
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float hm = hsum[tm][r] * (1.0f / 128);
            float v = 0.f;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const float d = acc[tm][tn][r] - hm;
                v = fmaf(d, d, v);
            }
            v = half_wave_sum(v);
            if ((lane & 31) == 0) red[Tile::c_row(tm, r)][wn] = v;
        }
    __syncthreads();
    // ---- exchange with the partner (thread `row` of either workgroup), combine
    if (threadIdx.x < 128) {
        const int row = threadIdx.x;
        const float my_m2 = red[row][0] + red[row][1];
        unsigned long long* mine = xch + ((long)tile * 2 + half) * 128 + row;
        const unsigned long long* theirs = xch + ((long)tile * 2 + (half ^ 1)) * 128 + row;
        const unsigned long long word = (unsigned long long)__float_as_uint(my_sum) | ((unsigned long long)__float_as_uint(my_m2) << 32);
        __hip_atomic_store(mine, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long got = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int budget = spin_limit;
        while (got == kXchEmpty && budget > 0) {          // (a sum / M2 pair of all-ones bits is two NaNs: never a result)
            __builtin_amdgcn_s_sleep(2);
            got = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            --budget;
        }
        if (got == kXchEmpty) atomicOr(&g_enc_xch_timeout, 1u);      // the fill pattern (NaN) goes through to the output
        const float o_sum = __uint_as_float((unsigned)got), o_m2 = __uint_as_float((unsigned)(got >> 32));
        const float s_lo = half ? o_sum : my_sum, s_hi = half ? my_sum : o_sum;
        const float q_lo = half ? o_m2 : my_m2, q_hi = half ? my_m2 : o_m2;
        const float mean = (s_lo + s_hi) * (1.0f / kC);
        const float delta = (s_hi - s_lo) * (1.0f / 128);
        const float m2 = (q_lo + q_hi) + delta * delta * 64.0f;
        const float rs = 1.0f / sqrtf(m2 * (1.0f / (kC - 1)) + kNormEps);
        stat[row][0] = mean;
        stat[row][1] = rs;
        const int m = m0 + row;
        if (half == 0 && m < am.M) rstd_out[m] = rs;
    }
    __syncthreads();
    const bool h2out = y_amax != nullptr;            // block-uniform
    const float sy = h2out ? scale_for_amax(*y_amax) : 1.0f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = Tile::c_row(tm, r);
            const float mean = stat[row][0], rs = stat[row][1];
            const int m = m0 + row;
            const bool live = m < am.M;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const float xh = (acc[tm][tn][r] - mean) * rs;
                const float yv = fmaxf(relu_in(fmaf(xh, gw[tn], gb[tn])), 0.f);
                if (live) __builtin_nontemporal_store(xh, xhat + (long)m * kC + col[tn]);
                if (h2out)       // (every lane takes part in the swap)
                    h2_store_lane(reinterpret_cast<unsigned char*>(y) + (long)(live ? m : 0) * (kC * 4), col[tn], yv, sy, live);
                else if (live)
                    y[(long)m * kC + col[tn]] = yv;
            }
        }
}

// ------------------------------------------------------------------ wgrad
// MODE 3: as 2 with the activation operand (x) in H2 storage; MODE 4: both operands are bf16 tensors, one product;
// MODE 5: as 3 with dx in H2 storage too (scaled for the bound *dx_amax that the norm backward left, norm_bwd_kernel DXH2)
template <int MODE>
struct WgCfg {
    using Tile = typename std::conditional<MODE == 4, TnTileX3<128, 128, 2, 2, 32, 1, 1, false, true>,      // bf16 tensors
                 typename std::conditional<MODE == 5, TnTileX3<128, 128, 2, 2, 32, 1, 2, true, false, true>,   // dx and x in H2
                 typename std::conditional<MODE != 0, TnTileX3<128, 128, 2, 2, 32, 1, MODE >= 2 ? 2 : 3, MODE == 3>,
                                           TnTile<128, 128, 2, 2>>::type>::type>::type;
};
// 1-D grid of 8 * T * ceil(S/8) blocks, T = 2*K/128 output tiles; part[z][co][K].
// XCD-aware mapping: the dispatcher places block b on XCD b % 8 (observed, speed only), and every
// output tile of one row split z re-reads the same dx / activation rows, so all T tiles of a split
// are given to ONE XCD (z % 8): its L2 fetches those rows once instead of eight L2s fetching them
// eight times (the rows do not fit any single L2: 335 MB for layer 1 at B = 64).
template <int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    RowMap dxm, RowMap im, int K, int rows_per_split, int S, float* __restrict__ part,
    const float* __restrict__ dx_amax, const float* __restrict__ x_amax, int amax_slots) {
    using WgTile = typename WgCfg<MODE>::Tile;
    __shared__ float smem[WgTile::SMEM_FLOATS];
    const int T = 2 * (K / 128);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = slot % T, z = (slot / T) * 8 + xcd;
    if (z >= S) return;                                  // block-uniform
    const int n0 = (tile >> 1) * 128, c0 = (tile & 1) * 128;
    const int mbeg = z * rows_per_split;
    const int mend = min(dxm.M, mbeg + rows_per_split);
    f32x16 acc[WgTile::TM][WgTile::TN];
    zero_acc(acc);
    float inv = 1.0f;
    if constexpr (MODE == 4) {
        WgTile::run(acc, dxm, c0, im, n0, mbeg, mend, smem);
    } else if constexpr (MODE >= 2) {
        const float sa = scale_for_amax(fold_amax(dx_amax, amax_slots)), sb = scale_for_amax(*x_amax);
        WgTile::run(acc, dxm, c0, im, n0, mbeg, mend, smem, sa, sb);
        inv = 1.0f / (sa * sb);
    } else {
        WgTile::run(acc, dxm, c0, im, n0, mbeg, mend, smem);
    }
    float* out = part + (long)z * kC * K;
#pragma unroll
    for (int tm = 0; tm < WgTile::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = c0 + WgTile::c_row(tm, r);
#pragma unroll
            for (int tn = 0; tn < WgTile::TN; ++tn)
                out[(long)row * K + n0 + WgTile::c_col(tn)] = (MODE == 2 || MODE == 3 || MODE == 5) ? acc[tm][tn][r] * inv : acc[tm][tn][r];
        }
}

// dW[co][ci][kk] = sum_z part[z][co][kk*C + ci]   (fixed summation order)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int S,
                                                           int k, float* __restrict__ dw) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)kC * k * kC;
    if (idx >= total) return;
    float sum = 0.f;
    for (int z = 0; z < S; ++z) sum += part[(long)z * total + idx];
    const int co = (int)(idx / (k * kC));
    const int rem = (int)(idx - (long)co * k * kC);
    const int kk = rem >> kCLog2, ci = rem & (kC - 1);
    dw[((long)co * kC + ci) * k + kk] = sum;
}

// the same for up to three layers in one launch (the short layers' DMA weight gradients leave their partials in own buffers)
struct WgReduceJobs { const float* part[3]; float* dw[3]; int S[3], k[3], blk0[4]; };
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(WgReduceJobs j) {
    int q = 0;
    while (q < 2 && (int)blockIdx.x >= j.blk0[q + 1]) ++q;           // block-uniform
    const int k = j.k[q], S = j.S[q];
    const long idx = (long)(blockIdx.x - j.blk0[q]) * 256 + threadIdx.x;
    const long total = (long)kC * k * kC;
    if (idx >= total) return;
    const float* part = j.part[q];
    float sum = 0.f;
    for (int z = 0; z < S; ++z) sum += part[(long)z * total + idx];
    const int co = (int)(idx / (k * kC));
    const int rem = (int)(idx - (long)co * k * kC);
    const int kk = rem >> kCLog2, ci = rem & (kC - 1);
    j.dw[q][((long)co * kC + ci) * k + kk] = sum;
}

// Input steps a data-gradient GEMM never reaches.  Its s phases cover the steps tau = q s + r - p, q <= Lout: up to
// s (Lout + 1) - p - 1.  A layer input longer than that -- (Lin + 2p - k) % s leaves steps at the end that no output window
// touches; with this encoder's geometry only layer 1 (k 8, s 4, p 2) can: Lin = 4 Lout + 3 -- has zero gradient there, and the
// buffer the phases write must say so: rows [first, Lin) of every sequence := 0 (16-byte pieces; row16 = pieces per row).
__global__ __launch_bounds__(256) void zero_tail_rows_kernel(int4* __restrict__ dst, int Lin, int first, int row16) {
    const int n = (Lin - first) * row16;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[((long)blockIdx.y * Lin + first) * row16 + i] = make_int4(0, 0, 0, 0);
}
static void zero_uncovered_rows(void* dst, int B, int Lin, int Lout, int s, int p, int row_bytes, hipStream_t st) {
    const int first = s * (Lout + 1) - p;
    if (first >= Lin) return;
    const int row16 = row_bytes / 16;
    hipLaunchKernelGGL(zero_tail_rows_kernel, dim3(cdiv((Lin - first) * row16, 256), B), dim3(256), 0, st,
                       reinterpret_cast<int4*>(dst), Lin, first, row16);
}

struct GradPtrs { float* p[12]; };
// small[i] = [d norm.weight | d norm.bias | d conv.bias] of layer i+1 -> the 12 parameter-gradient tensors
__global__ __launch_bounds__(256) void small_to_grads_kernel(const float* __restrict__ small, GradPtrs g) {
    const int q = blockIdx.x;                 // 0..11 : layer (q/3)+1, item q%3
    g.p[q][threadIdx.x] = small[(long)(q / 3 + 1) * 3 * kC + (q % 3) * kC + threadIdx.x];
}

// ------------------------------------------------------------------ host side
struct ConvGeom { int k, s, p; };
static const ConvGeom kGeom[5] = {{10, 5, 3}, {8, 4, 2}, {4, 2, 1}, {4, 2, 1}, {4, 2, 1}};

static inline long align64(long v) { return (v + 63) & ~63L; }

struct EncLayout {
    int L[5];
    long y[4], xhat[5], rstd[5], mean0;    // offsets (floats) into the saved workspace
    long swd[5], sbound, szero;            // ... dgrad weight layouts (1..4) and input bounds, prepared by the forward
    long saved_total;
    long wp[5];                            // forward scratch: permuted weights (1..4)
    long famax, fwd_total;                 // famax: kPrepParts partial max|w| per layer
    long xch[5], xch_total;                // per-layer exchange buffers of the N-split forward (conv_fwd_nsplit_kernel): 2 x rows x 8 bytes
    // backward scratch
    long dx[5], dy0, part, colpart, tmp, small, conv0, bamax;
    long colp[5], tmpq[5];                 // per-layer partials of the stand-alone norm backwards (summed in one batch)
    long bwd_total;
    int wg_splits[5], wg_rows[5];
    bool h2[4];                            // output of layer i kept in H2 storage (cpc_common.h) -- see act_h2 below
    bool dma[5];                           // layer i (1..4) reads its H2 input -- and its H2 gradient -- on the DMA-fed kernels (conv_dma.hip);
                                           // false with h2[i-1]: on the register-staged tiles, which stage H2 operands as they lie (AH2)
    bool dxh2[5];                          // gradient dx of layer i kept in H2 storage (norm_bwd_kernel DXH2): DMA / AH2 dgrad, DMA wgrad
    long partl[5];                         // weight-gradient partials of layer i (own buffers: one batched reduction for layers 2..4)
    long dxbound, dyamax;                  // [i] the bound dx_i was scaled for; [i][slot] partial max|dy_i| (dy_i: dgrad output)
    bool bf16;                             // mode 4
};
#define act_h2(layer) (e.h2[layer])

// Mode 3: the output of layer 0 (conv1's input; conv1 is 70 % of the stack's FLOPs) lives in H2 storage and conv1 runs on
// the DMA kernel (conv_dma.hip); so does conv2 (a further 17 %) once its 256-row tiles fill the chip (B >= ~100: at B = 64
// its 128-row DMA tiles measured 0.070 ms against 0.067 for the register-staged kernel).  Layers 3 and 4 keep fp32
// activations and the register-staged kernels, whose 32/64-row tiles fill the chip at their small row counts.
static int g_dma_bm = 0;     // 0 = by problem size; 128 / 256 = tuning override of the DMA kernel's rows per workgroup
static int g_wgrad_dma = 1;  // weight gradient of a layer with dx and x in H2 storage: 1 = DMA + transposing LDS reads (conv_dma.hip),
                             // 0 = the register-staged TN tile (conv_wgrad_kernel<5>)
static int g_h2_dx = 1;      // mode 3: layer 1's gradient dx in H2 storage (DMA data gradient); 0 = fp32 dx, register-staged dgrad
static int g_h2_layers = 0;  // 0 = by problem size; 1 / 2 / 4 = tuning / test override: how many layers read H2 input (see enc_layout)
static int g_h2_all = 1;     // what "by problem size" means: 1 (default since round 4) = every layer reads H2 input; 0 = conv1 (conv2 at
                             // B >= ~100) only -- cpc_set_h2_layers(1 / 2) selects the round-3 behaviour explicitly
static int g_force_bm = 0;   // 0 = choose by problem size; 32/64/128 = tuning / test override
static int g_dma_layer2 = 0; // 1: layer 2 on the DMA-fed kernels whatever the batch (cpc_set_dma_layer2; by default from B ~ 100 on)
static int g_small_bm = 64;  // rows of the small tile pick_bm() chooses below 32000 rows (cpc_set_conv_small_tile): 32, or 64 (default
                             // since round 5: eight 32 x 64 waves per 64-row tile on the pipelined schedule -- half the weight bytes
                             // through L2 per output row at two waves per SIMD; where 64-row tiles would leave CUs empty, 32)
static int g_small_pipe = 1;
static int g_dgrad_nsplit = 256;  // > 0: the H2-fed data gradient of the short layers on 128 x 128 tiles when that gives at least this many
                                // workgroups (cpc_set_dgrad_nsplit; 0 = off) // 1: the 32- / 64-row tiles of the H2-fed register-staged kernels on the pipelined 16-k schedule (ConvCfg PIPE)
static constexpr int g_unfuse_big = 2;   // 2: every dgrad runs unfused + streaming norm backward, 1: only the 128-row tiles,
                                         // 0: fused epilogue (cpc_conv_layer_dgrad fuse=1).  Measured 4.69 / 4.76 / 4.79 ms per step
static int pick_bm(int M) {
    if (g_force_bm) return g_force_bm;
    // measured on MI355X (tools/bench_kernels.py): 128-row tiles win as soon as they give ~256 blocks
    // (the 256-column weight tile is re-read from L2 once per block); below that the 32-row tile wins.
    if (M >= 32000) return 128;
    // 64-row tiles (cpc_set_conv_small_tile(64)) only where they still give the chip ~a workgroup per CU
    return (g_small_bm == 64 && cdiv(M, 64) < 192) ? 32 : g_small_bm;
}

static bool enc_layout(int B, int Lw, EncLayout& e) {
    int lin = Lw;
    for (int i = 0; i < 5; ++i) {
        if (lin + 2 * kGeom[i].p < kGeom[i].k) return false;
        e.L[i] = conv_out_len(lin, kGeom[i].k, kGeom[i].s, kGeom[i].p);
        if (e.L[i] <= 0) return false;
        lin = e.L[i];
    }
    e.bf16 = g_mfma_mode == 4;             // bf16-storage variant: y0..y3, xhat1..4 and every gradient tensor as bf16
    // layers reading H2 input: 1 = conv1, 2 = conv1 and conv2 (both on the DMA kernel), 4 = all four -- conv1 on the DMA kernel,
    // conv2 too once its 256-row tiles fill the chip, the short layers on the register-staged tiles fed H2 rows as they lie
    const bool big2 = (long)B * e.L[2] >= 256L * 200;
    const int nh2 = g_mfma_mode != 3 ? 0 : (g_h2_layers ? g_h2_layers : (g_h2_all ? 4 : (big2 ? 2 : 1)));
    for (int i = 0; i < 4; ++i) e.h2[i] = i < nh2;
    e.dma[0] = false;
    for (int i = 1; i < 5; ++i) e.dma[i] = e.h2[i - 1] && (i == 1 || (i == 2 && (nh2 == 2 || big2 || g_dma_layer2)));
    // the gradient of a layer whose input is kept in H2 storage is kept so too: its data gradient reads the pieces as stored
    // (DMA kernel or AH2 tile), its weight gradient reads both operands as pieces by DMA
    for (int i = 0; i < 5; ++i) e.dxh2[i] = i >= 1 && g_h2_dx && e.h2[i - 1];
    long o = 0;
    for (int i = 0; i < 4; ++i) { e.y[i] = o; o += align64((long)B * e.L[i] * kC); }
    // Layer 0's output twice: the composite train step alternates between the two (StepHooks::parity) so that the NEXT step's
    // layer 0 may run while this step's last kernel -- layer 1's weight gradient, which reads y0 -- is still at work
    // (cpc_train_step, open tail).  Every other caller uses buffer 0.
    const long y0b = o; o += align64((long)B * e.L[0] * kC);
    const int parity = step_hooks().parity & 1;
    if (parity) e.y[0] = y0b;
    e.xhat[0] = -1;
    for (int i = 1; i < 5; ++i) { e.xhat[i] = o; o += align64((long)B * e.L[i] * kC); }
    for (int i = 0; i < 5; ++i) { e.rstd[i] = o; o += align64((long)B * e.L[i]); }
    e.mean0 = o; o += align64((long)B * e.L[0]);
    e.swd[0] = -1;
    for (int i = 1; i < 5; ++i) { e.swd[i] = o; o += align64((long)kC * kGeom[i].k * kC * 3 / 2); }
    e.sbound = o + 16 * parity; o += 64;   // [i] = bound of layer i's input (fp16-split mode), i = 1..4; one set per parity: the next
                                           // step's bounds (new norm parameters) are written while layer 1's weight gradient reads this step's
    e.szero = o; o += 64;                  // zeros: what the DMA kernel reads for the conv's padding rows
    e.saved_total = o;

    o = 0;
    e.wp[0] = -1;
    // 1.5x: in split-bf16 mode the re-laid-out weight is three bf16 planes (6 bytes per weight)
    for (int i = 1; i < 5; ++i) { e.wp[i] = o; o += align64((long)kC * kGeom[i].k * kC * 3 / 2); }
    e.famax = o; o += align64(4 * kPrepParts);
    e.xch[0] = e.xch[1] = -1;
    const long xch0 = o;
    for (int i = 2; i < 5; ++i) { e.xch[i] = o; o += align64(4L * cdiv(B * e.L[i], 128) * 128); }     // (sized whatever the switches say)
    e.xch_total = o - xch0;
    e.fwd_total = o;

    o = 0;
    e.dx[0] = -1;
    for (int i = 1; i < 5; ++i) { e.dx[i] = o; o += align64((long)B * e.L[i] * kC); }
    e.dy0 = o; o += align64((long)B * e.L[0] * kC);
    long part_max = 0, col_max = 0;
    for (int i = 1; i < 5; ++i) {
        const int M = B * e.L[i], K = kGeom[i].k * kC;
        const int tiles = 2 * (K / 128);
        int S = cdiv(768, tiles);                       // aim for >= 768 blocks
        int rows = cdiv(cdiv(M, S), 32) * 32;
        if (rows < 256) rows = 256;
        S = cdiv(M, rows);
        e.wg_splits[i] = S; e.wg_rows[i] = rows;
        part_max = std::max(part_max, (long)S * kC * K);
        {                      // the DMA weight gradient splits the rows finer (conv_wgrad_dma_plan).  Sized for it whatever the
            int Sd, rd;        // switches say: callers cache the layout per shape (ops._layout), a size must not depend on a knob
            conv_wgrad_dma_plan(M, kGeom[i].k, &Sd, &rd, 512);
            part_max = std::max(part_max, (long)Sd * kC * K);
        }
        // dgrad of layer i (i >= 2) writes colpart of layer i-1; norm_bwd writes layer 4's
        if (i >= 2) {
            const int Md = B * (e.L[i] + 1);
            col_max = std::max(col_max, (long)cdiv(Md, 32) * kGeom[i].s);     // (the smallest tile: a size must not depend on a knob)
        }
    }
    for (int i = 1; i < 5; ++i) col_max = std::max(col_max, (long)cdiv(B * e.L[i], NB_ROWS));   // stand-alone norm backward
    e.part = o; o += align64(part_max);
    e.partl[0] = -1;
    for (int i = 1; i < 5; ++i) {            // (sized for the finest plan, as part_max: a size must not depend on a knob)
        int Sd, rd;
        conv_wgrad_dma_plan(B * e.L[i], kGeom[i].k, &Sd, &rd, 512);
        e.partl[i] = o; o += align64((long)Sd * kC * kGeom[i].k * kC);
    }
    e.colpart = o; o += align64(col_max * 3 * kC);
    e.tmp = o; o += align64((long)kRowsSumGroups * 3 * kC);
    e.small = o; o += align64(5L * 3 * kC);
    e.colp[0] = e.tmpq[0] = -1;
    for (int i = 1; i < 5; ++i) {
        e.colp[i] = o; o += align64((long)cdiv(B * e.L[i], NB_ROWS) * 3 * kC);
        e.tmpq[i] = o; o += align64((long)kRowsSumGroups * 3 * kC);
    }
    e.conv0 = o; o += align64(cpc_conv0_backward_scratch_floats(B, Lw));
    e.bamax = o; o += 5 * kAmaxSlots;       // [i][slot] = partial max|dx_i| (i = 1..4), folded by the consumers (fold_amax)
    e.dyamax = o; o += 5 * kAmaxSlots;      // cleared together with bamax
    e.dxbound = o; o += 64;
    e.bwd_total = o;
    return true;
}

template <int BM>
static void launch_conv_fwd(const RowMap& am, const float* wp, int K, const float* bias,
                            const float* nw, const float* nb, float* y, float* xhat, float* rstd,
                            const float* x_amax, const float* w_amax, hipStream_t st, bool x_h2 = false,
                            const float* y_amax = nullptr) {
    if (x_h2 && g_small_pipe && BM < 128 && K % 64 == 0)
        hipLaunchKernelGGL((conv_fwd_kernel<BM, 2, true, true>), dim3(cdiv(am.M, BM)), dim3(ConvCfg<BM, 2, true, true>::Tile::NTHREADS),
                           0, st, am, wp, K, bias, nw, nb, y, xhat, rstd, x_amax, w_amax, y_amax);
    else if (x_h2)     // input in H2 storage (fp16-split modes): staged as it lies; y in H2 storage too when y_amax is given
        hipLaunchKernelGGL((conv_fwd_kernel<BM, 2, true>), dim3(cdiv(am.M, BM)), dim3(ConvCfg<BM, 2, true>::Tile::NTHREADS),
                           0, st, am, wp, K, bias, nw, nb, y, xhat, rstd, x_amax, w_amax, y_amax);
    else if (g_mfma_mode >= 2 && K % 32 == 0)
        hipLaunchKernelGGL((conv_fwd_kernel<BM, 2>), dim3(cdiv(am.M, BM)), dim3(ConvCfg<BM, 2>::Tile::NTHREADS),
                           0, st, am, wp, K, bias, nw, nb, y, xhat, rstd, x_amax, w_amax, y_amax);
    else if (g_mfma_mode != 0 && K % 32 == 0)
        hipLaunchKernelGGL((conv_fwd_kernel<BM, 1>), dim3(cdiv(am.M, BM)), dim3(ConvCfg<BM, 1>::Tile::NTHREADS),
                           0, st, am, wp, K, bias, nw, nb, y, xhat, rstd, x_amax, w_amax);
    else
        hipLaunchKernelGGL((conv_fwd_kernel<BM, 0>), dim3(cdiv(am.M, BM)), dim3(ConvCfg<BM, 0>::Tile::NTHREADS),
                           0, st, am, wp, K, bias, nw, nb, y, xhat, rstd, x_amax, w_amax);
}

template <int BM, bool FUSE>
static void launch_conv_dgrad(const RowMap& am, const float* wd, int s, int p, int Lin,
                              const float* xhat_prev, const float* y_prev, const float* rstd_prev,
                              const float* nw_prev, float* dprev, float* colpart, const float* dx_amax,
                              const float* w_amax, float* prev_amax, int amax_slots, hipStream_t st, bool dx_h2 = false) {
    if constexpr (!FUSE) {
        if (dx_h2 && g_small_pipe && BM < 128) {
            hipLaunchKernelGGL((conv_dgrad_kernel<BM, false, 2, true, true>), dim3(cdiv(am.M, BM), s),
                               dim3(ConvCfg<BM, 2, true, true>::Tile::NTHREADS), 0, st, am, wd, s, p, Lin, xhat_prev, y_prev,
                               rstd_prev, nw_prev, dprev, colpart, dx_amax, w_amax, prev_amax, amax_slots);
            return;
        }
        if (dx_h2) {   // dx in H2 storage scaled for the single bound *dx_amax (amax_slots == 1)
            hipLaunchKernelGGL((conv_dgrad_kernel<BM, false, 2, true>), dim3(cdiv(am.M, BM), s),
                               dim3(ConvCfg<BM, 2, true>::Tile::NTHREADS), 0, st, am, wd, s, p, Lin, xhat_prev, y_prev,
                               rstd_prev, nw_prev, dprev, colpart, dx_amax, w_amax, prev_amax, amax_slots);
            return;
        }
    }
    if (g_mfma_mode >= 2)
        hipLaunchKernelGGL((conv_dgrad_kernel<BM, FUSE, 2>), dim3(cdiv(am.M, BM), s),
                           dim3(ConvCfg<BM, 2>::Tile::NTHREADS), 0, st, am, wd, s, p, Lin, xhat_prev, y_prev,
                           rstd_prev, nw_prev, dprev, colpart, dx_amax, w_amax, prev_amax, amax_slots);
    else if (g_mfma_mode == 1)
        hipLaunchKernelGGL((conv_dgrad_kernel<BM, FUSE, 1>), dim3(cdiv(am.M, BM), s),
                           dim3(ConvCfg<BM, 1>::Tile::NTHREADS), 0, st, am, wd, s, p, Lin, xhat_prev, y_prev,
                           rstd_prev, nw_prev, dprev, colpart, dx_amax, w_amax, prev_amax, amax_slots);
    else
        hipLaunchKernelGGL((conv_dgrad_kernel<BM, FUSE, 0>), dim3(cdiv(am.M, BM), s),
                           dim3(ConvCfg<BM, 0>::Tile::NTHREADS), 0, st, am, wd, s, p, Lin, xhat_prev, y_prev,
                           rstd_prev, nw_prev, dprev, colpart, dx_amax, w_amax, prev_amax, amax_slots);
}

}  // namespace cpc

using namespace cpc;

static int weight_split() {
    return g_mfma_mode >= 2 ? 2 : ((g_mfma_mode == 1 && ConvCfg<128, 1>::kPreSplitW) ? 1 : 0);
}
static int conv_dgrad_core(const float* dx, const float* wd, int fuse, const float* xhat_prev, const float* y_prev,
                           const float* rstd_prev, const float* nw_prev, float* dprev, float* colpart, float* tmp,
                           float* small3, const float* dx_amax, float* dprev_amax, int B, int Lin, int k, int s, int p,
                           hipStream_t st, int amax_slots = 1, float* dprev_slots = nullptr, bool dx_h2 = false);

extern "C" int cpc_set_conv_tile(int bm) {
    CPC_RETURN_IF(bm != 0 && bm != 32 && bm != 64 && bm != 128, CPC_ERR_ARG);
    g_force_bm = bm;
    return 0;
}
extern "C" int cpc_set_dma_layer2(int on) {
    g_dma_layer2 = on ? 1 : 0;
    return 0;
}
extern "C" int cpc_set_conv_small_tile(int bm) {
    CPC_RETURN_IF(bm != 32 && bm != 64, CPC_ERR_ARG);
    g_small_bm = bm;
    return 0;
}
extern "C" int cpc_set_dgrad_nsplit(int min_wgs) {
    CPC_RETURN_IF(min_wgs < 0, CPC_ERR_ARG);
    g_dgrad_nsplit = min_wgs;
    return 0;
}
extern "C" int cpc_set_conv_small_pipe(int on) {
    g_small_pipe = on ? 1 : 0;
    return 0;
}
extern "C" int cpc_set_dma_tile(int bm) {
    CPC_RETURN_IF(bm != 0 && bm != 128 && bm != 256, CPC_ERR_ARG);
    g_dma_bm = bm;
    return 0;
}
// > 0: the forward of a short layer (H2 in) on 128 x 128 tiles, two workgroups per row tile with the ChannelNorm statistics
// exchanged between them (conv_fwd_nsplit_kernel), where that gives at least this many workgroups; 0 (default) = off.
// Measured at B = 64 inside the step (profiles/r5_ab_fwd_nsplit.txt): conv3 48.4 us against 52 on the 64-row tile, conv2 119.7
// against 100 on the 128 x 256 tile (it re-reads every input row for both halves and shares the chip with the criterion's index
// preparation); the step 2.805 against 2.806 ms -- no gain worth a cross-workgroup wait on the default path.
static int g_fwd_nsplit = 0;
static int g_fwd_nsplit_spin = 1 << 22;
extern "C" int cpc_set_fwd_nsplit(int min_wgs, int spin_limit) {
    CPC_RETURN_IF(min_wgs < 0, CPC_ERR_ARG);
    g_fwd_nsplit = min_wgs;
    g_fwd_nsplit_spin = spin_limit < 0 ? (1 << 22) : spin_limit;
    return 0;
}
namespace cpc {
int enc_error_flag_fetch(int clear, unsigned* out) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_enc_xch_timeout), sizeof(v)) != hipSuccess) return CPC_ERR_ARG;
    if (clear && v) {
        const unsigned zero = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_enc_xch_timeout), &zero, sizeof(zero)) != hipSuccess) return CPC_ERR_ARG;
    }
    *out = v;
    return 0;
}
}  // namespace cpc
static int g_tail_conv0_early = 2;      // cpc_set_tail_schedule (see cpc_encoder_forward); 2 since the end of round 6: 2.638 / 2.632 / 2.634 ->
                                        // 2.620 / 2.621 / 2.622 ms per step sustained (three alternations, profiles/r6_ab_tail_schedule.txt)
extern "C" int cpc_set_tail_schedule(int conv0_early) {
    if (conv0_early < 0 || conv0_early > 3) return CPC_ERR_ARG;
    g_tail_conv0_early = conv0_early;
    return 0;
}
// 0 (default): layer 1's weight gradient is released behind its data gradient; 1: together with dx1, i.e. beside that data gradient
static int g_wgrad1_early = 0;
extern "C" int cpc_set_wgrad1_early(int on) {
    g_wgrad1_early = on ? 1 : 0;
    return 0;
}

extern "C" int cpc_set_h2_dx(int on) {
    g_h2_dx = on ? 1 : 0;
    g_wgrad_dma = on == 2 ? 0 : 1;            // 2: H2 gradient, but its weight gradient on the register-staged TN tile (A/B)
    return 0;
}
extern "C" int cpc_set_h2_layers(int n) {
    CPC_RETURN_IF(n < 0 || n == 3 || n > 4, CPC_ERR_ARG);
    g_h2_layers = n;
    return 0;
}

// Weight re-layout for the forward GEMM: PyTorch (O,I,W) -> K-major rows wp[co][kk*C+ci]; in the default
// split-bf16 mode wp holds three bf16 planes.  wp must have room for 256*k*256*3/2 floats.
extern "C" int cpc_conv_weight_relayout(const float* w, float* wp, int k, void* stream) {
    CPC_RETURN_IF(!w || !wp || k <= 0, CPC_ERR_ARG);
    const long nw_elems = (long)kC * k * kC;
    hipStream_t st = (hipStream_t)stream;
    float* amax = wp + nw_elems;                     // spare floats behind the re-laid-out weight
    if (g_mfma_mode >= 2) {
        (void)hipMemsetAsync(amax, 0, sizeof(float), st);
        hipLaunchKernelGGL(absmax_kernel, dim3(64), dim3(256), 0, st, w, nw_elems, amax);
    }
    hipLaunchKernelGGL(permute_w_fwd_kernel, dim3(cdiv(nw_elems, 256)), dim3(256), 0, st, w, wp, k, weight_split(), amax);
    CPC_LAUNCH_CHECK();
    return 0;
}

// out = max(out, max|x|) (out must be initialised, e.g. to 0): an exact upper bound for scale_for_amax
extern "C" int cpc_absmax(const float* x, long n, float* out, void* stream) {
    CPC_RETURN_IF(!x || !out || n <= 0, CPC_ERR_ARG);
    hipLaunchKernelGGL(absmax_kernel, dim3(std::min<long>(cdiv(n, 4096), 1024)), dim3(256), 0, (hipStream_t)stream, x, n, out);
    CPC_LAUNCH_CHECK();
    return 0;
}

// The forward GEMM kernel alone, on a weight prepared by cpc_conv_weight_relayout (exactly one
// kernel launch: this is what bench.py times for the roofline figure).
// x_amax: device float, an upper bound of max|x| (read in the fp16-split mode only; see cpc_absmax).
// x_h2: x lies in H2 storage scaled for *x_amax; y_amax != NULL: y is written in H2 storage scaled for *y_amax (fp16-split modes)
static int conv_gemm_forward_impl(const float* x, const float* wp, const float* bias, const float* nw, const float* nb,
                                  float* y, float* xhat, float* rstd, const float* x_amax, int B, int Lin, int k, int s, int p,
                                  hipStream_t st, bool x_h2, const float* y_amax) {
    CPC_RETURN_IF(B <= 0 || Lin <= 0 || k != 2 * s || Lin + 2 * p < k, CPC_ERR_SHAPE);
    CPC_RETURN_IF(g_mfma_mode >= 2 && !x_amax, CPC_ERR_ARG);
    CPC_RETURN_IF((x_h2 || y_amax) && (g_mfma_mode < 2 || (k * kC) % 32 != 0), CPC_ERR_ARG);
    const int Lout = conv_out_len(Lin, k, s, p);
    const RowMap am = conv_rows(x, B, Lin, Lout, s, p);
    const int K = k * kC;
    const float* w_amax = wp + (long)kC * k * kC;
    switch (pick_bm(am.M)) {
        case 128: launch_conv_fwd<128>(am, wp, K, bias, nw, nb, y, xhat, rstd, x_amax, w_amax, st, x_h2, y_amax); break;
        case 64: launch_conv_fwd<64>(am, wp, K, bias, nw, nb, y, xhat, rstd, x_amax, w_amax, st, x_h2, y_amax); break;
        default: launch_conv_fwd<32>(am, wp, K, bias, nw, nb, y, xhat, rstd, x_amax, w_amax, st, x_h2, y_amax); break;
    }
    CPC_LAUNCH_CHECK();
    return 0;
}
extern "C" int cpc_conv_gemm_forward(const float* x, const float* wp, const float* bias, const float* nw,
                                     const float* nb, float* y, float* xhat, float* rstd, const float* x_amax,
                                     int B, int Lin, int k, int s, int p, void* stream) {
    return conv_gemm_forward_impl(x, wp, bias, nw, nb, y, xhat, rstd, x_amax, B, Lin, k, s, p, (hipStream_t)stream, false, nullptr);
}

// One conv layer forward (layers 1..4): x (B,Lin,C) -> y, xhat (B,Lout,C), rstd (B*Lout).
// wp is scratch for the permuted weight (256*k*256 floats).
extern "C" int cpc_conv_layer_forward(const float* x, const float* w, const float* bias,
                                      const float* nw, const float* nb, float* wp, float* y,
                                      float* xhat, float* rstd, int B, int Lin, int k, int s, int p,
                                      void* stream) {
    CPC_RETURN_IF(B <= 0 || Lin <= 0 || k != 2 * s || Lin + 2 * p < k, CPC_ERR_SHAPE);
    int rc = cpc_conv_weight_relayout(w, wp, k, stream);
    if (rc) return rc;
    float* x_amax = wp + (long)kC * k * kC + 1;          // second spare slot behind the weight
    (void)hipMemsetAsync(x_amax, 0, sizeof(float), (hipStream_t)stream);
    rc = cpc_absmax(x, (long)B * Lin * kC, x_amax, stream);
    if (rc) return rc;
    return cpc_conv_gemm_forward(x, wp, bias, nw, nb, y, xhat, rstd, x_amax, B, Lin, k, s, p, stream);
}

// ReLU'/ChannelNorm backward of a whole (M,256) activation: dy -> dx, plus
// small3 = [d norm.weight | d norm.bias | d conv.bias] (3*256 floats).
// colpart: cdiv(M,32)*768 floats, tmp: kRowsSumGroups*768 floats.
// dx_amax (may be NULL): device float that receives max(*dx_amax, max|dx|) -- initialise it to 0.
extern "C" int cpc_norm_backward(const float* dy, const float* xhat, const float* y,
                                 const float* rstd, const float* nw, float* dx, float* colpart,
                                 float* tmp, float* small3, float* dx_amax, int M, void* stream) {
    CPC_RETURN_IF(M <= 0, CPC_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = cdiv(M, NB_ROWS);
    hipLaunchKernelGGL(norm_bwd_kernel<0>, dim3(nblk), dim3(256), 0, st, dy, xhat, y, rstd, nw, nullptr, dx, colpart, M, dx_amax, 1);
    CPC_LAUNCH_CHECK();
    return rows_sum(colpart, nblk, 3 * kC, tmp, small3, st);
}

// dgrad of one conv layer (k = 2s).  dx: (B,Lout,C) pre-norm gradient of THIS layer.
// fuse != 0: also applies the previous layer's ReLU'/ChannelNorm backward and writes that
//   layer's pre-norm gradient to dprev (B,Lin,C) and its small3 gradients;
// fuse == 0: writes the gradient w.r.t. the previous layer's OUTPUT to dprev.
// dx_amax: device float, upper bound of max|dx| (mode 2 only; NULL = computed here with an extra pass over dx);
// dprev_amax (fuse only, may be NULL): receives max(*dprev_amax, max|dprev|).
extern "C" int cpc_conv_layer_dgrad(const float* dx, const float* w, float* wd, int fuse,
                                    const float* xhat_prev, const float* y_prev,
                                    const float* rstd_prev, const float* nw_prev, float* dprev,
                                    float* colpart, float* tmp, float* small3, const float* dx_amax,
                                    float* dprev_amax, int B, int Lin, int k, int s, int p, void* stream) {
    CPC_RETURN_IF(B <= 0 || Lin <= 0 || k != 2 * s || Lin + 2 * p < k, CPC_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const int Lout = conv_out_len(Lin, k, s, p);
    const long nw_elems = (long)kC * k * kC;
    float* w_amax = wd + nw_elems;                    // spare floats behind the re-laid-out weight
    if (g_mfma_mode >= 2) {
        (void)hipMemsetAsync(w_amax, 0, 2 * sizeof(float), st);
        hipLaunchKernelGGL(absmax_kernel, dim3(64), dim3(256), 0, st, w, nw_elems, w_amax);
    }
    hipLaunchKernelGGL(permute_w_dgrad_kernel, dim3(cdiv(nw_elems, 256)), dim3(256), 0, st, w, wd, s, weight_split(),
                       w_amax);
    if (g_mfma_mode >= 2) {
        if (!dx_amax) {
            hipLaunchKernelGGL(absmax_kernel, dim3(1024), dim3(256), 0, st, dx, (long)B * Lout * kC, w_amax + 1);
            dx_amax = w_amax + 1;
        }
    }
    const int rc = conv_dgrad_core(dx, wd, fuse, xhat_prev, y_prev, rstd_prev, nw_prev, dprev, colpart, tmp, small3, dx_amax,
                                   dprev_amax, B, Lin, k, s, p, st);
    if (rc == 0) zero_uncovered_rows(dprev, B, Lin, Lout, s, p, kC * 4, st);     // input steps no output window touches: gradient 0
    return rc;
}

// The dgrad GEMM on a weight already in the dgrad layout (max|w| behind it in the fp16-split mode).
static int conv_dgrad_core(const float* dx, const float* wd, int fuse, const float* xhat_prev, const float* y_prev,
                           const float* rstd_prev, const float* nw_prev, float* dprev, float* colpart, float* tmp,
                           float* small3, const float* dx_amax, float* dprev_amax, int B, int Lin, int k, int s, int p,
                           hipStream_t st, int amax_slots, float* dprev_slots, bool dx_h2) {
    // dprev_slots (plain dgrad only, may be NULL): kAmaxSlots floats that receive max|dprev| (atomicMax: zero them first)
    // dx_h2 (plain dgrad, fp16-split modes): dx lies in H2 storage scaled for the single bound *dx_amax (amax_slots == 1)
    CPC_RETURN_IF(dx_h2 && (fuse || g_mfma_mode < 2 || amax_slots != 1), CPC_ERR_ARG);
    const int Lout = conv_out_len(Lin, k, s, p);
    const float* w_amax = wd + (long)kC * k * kC;
    // 2-row windows [q-1, q] over dx, q in [0, Lout]
    RowMap am;
    am = dgrad_rows(dx, B, Lin, Lout, s, p);
    const int bm = pick_bm(am.M);
    const int nblk = cdiv(am.M, bm) * s;
    // H2 gradient below the 128-row regime: 128 x 128 tiles (conv_dgrad_nsplit_kernel) where they still give the chip >= 256 workgroups
    if (dx_h2 && g_dgrad_nsplit && bm < 128 && (long)cdiv(am.M, 128) * s * 2 >= g_dgrad_nsplit) {
        hipLaunchKernelGGL(conv_dgrad_nsplit_kernel, dim3(cdiv(am.M, 128), s, 2), dim3(DgradNsTile::NTHREADS), 0, st, am, wd, s, p, Lin,
                           dprev, dx_amax, w_amax, dprev_slots);
        CPC_LAUNCH_CHECK();
        return 0;
    }
    if (fuse) {
        switch (bm) {
            case 128: launch_conv_dgrad<128, true>(am, wd, s, p, Lin, xhat_prev, y_prev, rstd_prev, nw_prev, dprev, colpart, dx_amax, w_amax, dprev_amax, amax_slots, st); break;
            case 64: launch_conv_dgrad<64, true>(am, wd, s, p, Lin, xhat_prev, y_prev, rstd_prev, nw_prev, dprev, colpart, dx_amax, w_amax, dprev_amax, amax_slots, st); break;
            default: launch_conv_dgrad<32, true>(am, wd, s, p, Lin, xhat_prev, y_prev, rstd_prev, nw_prev, dprev, colpart, dx_amax, w_amax, dprev_amax, amax_slots, st); break;
        }
        CPC_LAUNCH_CHECK();
        return rows_sum(colpart, nblk, 3 * kC, tmp, small3, st);
    }
    switch (bm) {
        case 128: launch_conv_dgrad<128, false>(am, wd, s, p, Lin, nullptr, nullptr, nullptr, nullptr, dprev, nullptr, dx_amax, w_amax, dprev_slots, amax_slots, st, dx_h2); break;
        case 64: launch_conv_dgrad<64, false>(am, wd, s, p, Lin, nullptr, nullptr, nullptr, nullptr, dprev, nullptr, dx_amax, w_amax, dprev_slots, amax_slots, st, dx_h2); break;
        default: launch_conv_dgrad<32, false>(am, wd, s, p, Lin, nullptr, nullptr, nullptr, nullptr, dprev, nullptr, dx_amax, w_amax, dprev_slots, amax_slots, st, dx_h2); break;
    }
    CPC_LAUNCH_CHECK();
    return 0;
}

// wgrad of one conv layer: dW (256,256,k) = sum over rows of dx (B,Lout,C) (x) im2col(x).
// part: splits*256*k*256 floats of scratch.
// dx_amax, x_amax: device floats bounding max|dx| and max|x| (mode 2 only).
static int conv_layer_wgrad(const float* dx, const float* x, int x_h2, float* part, float* dW, const float* dx_amax,
                            const float* x_amax, int B, int Lin, int k, int s, int p, int splits, int rows_per_split,
                            void* stream, int amax_slots = 1);
extern "C" int cpc_conv_layer_wgrad(const float* dx, const float* x, float* part, float* dW,
                                    const float* dx_amax, const float* x_amax, int B,
                                    int Lin, int k, int s, int p, int splits, int rows_per_split,
                                    void* stream) {
    return conv_layer_wgrad(dx, x, 0, part, dW, dx_amax, x_amax, B, Lin, k, s, p, splits, rows_per_split, stream);
}
// x_h2: 1 = the layer's input activation is in H2 storage scaled by scale_for_amax(*x_amax) (fp16-split modes only);
// 2 = dx and x are both bf16 tensors (mode 4)
static int conv_layer_wgrad(const float* dx, const float* x, int x_h2, float* part, float* dW, const float* dx_amax,
                            const float* x_amax, int B, int Lin, int k, int s, int p, int splits, int rows_per_split,
                            void* stream, int amax_slots) {
    CPC_RETURN_IF((x_h2 == 1 || x_h2 == 3) && g_mfma_mode < 2, CPC_ERR_ARG);
    CPC_RETURN_IF(B <= 0 || Lin <= 0 || Lin + 2 * p < k || splits <= 0 || rows_per_split <= 0, CPC_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const int Lout = conv_out_len(Lin, k, s, p);
    const int K = k * kC;
    const RowMap dxm = plain_rows(dx, B * Lout, kC);
    const RowMap im = conv_rows(x, B, Lin, Lout, s, p);
    CPC_RETURN_IF((long)splits * rows_per_split < dxm.M, CPC_ERR_SHAPE);
    const dim3 grid(8 * 2 * (K / 128) * cdiv(splits, 8));
    CPC_RETURN_IF(g_mfma_mode >= 2 && x_h2 != 2 && (!dx_amax || !x_amax), CPC_ERR_ARG);
    if (x_h2 == 2)
        hipLaunchKernelGGL((conv_wgrad_kernel<4>), grid, dim3(256), 0, st, dxm, im, K, rows_per_split, splits, part, dx_amax, x_amax, amax_slots);
    else if (x_h2 == 3)          // x and dx in H2 storage; dx_amax = the bound dx was scaled for
        hipLaunchKernelGGL((conv_wgrad_kernel<5>), grid, dim3(256), 0, st, dxm, im, K, rows_per_split, splits, part, dx_amax, x_amax, amax_slots);
    else if (g_mfma_mode >= 2 && x_h2)
        hipLaunchKernelGGL((conv_wgrad_kernel<3>), grid, dim3(256), 0, st, dxm, im, K, rows_per_split, splits, part, dx_amax, x_amax, amax_slots);
    else if (g_mfma_mode >= 2)
        hipLaunchKernelGGL((conv_wgrad_kernel<2>), grid, dim3(256), 0, st, dxm, im, K, rows_per_split, splits, part, dx_amax, x_amax, amax_slots);
    else if (g_mfma_mode == 1)
        hipLaunchKernelGGL((conv_wgrad_kernel<1>), grid, dim3(256), 0, st, dxm, im, K, rows_per_split, splits, part, dx_amax, x_amax, amax_slots);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<0>), grid, dim3(256), 0, st, dxm, im, K, rows_per_split, splits, part, dx_amax, x_amax, amax_slots);
    const long total = (long)kC * k * kC;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, part, splits, k, dW);
    CPC_LAUNCH_CHECK();
    return 0;
}

// ---- composite encoder -------------------------------------------------------------
// sizes[0] = saved workspace floats, [1] = forward scratch floats, [2] = backward scratch floats,
// [3..7] = L0..L4, [8..11] = offsets of y0..y3, [12..15] = offsets of xhat1..4,
// [16..20] = offsets of rstd0..4, [21] = offset of mean0   (all in floats, into `saved`)
extern "C" int cpc_encoder_layout(int B, int L, long* sizes) {
    EncLayout e;
    CPC_RETURN_IF(B <= 0 || !enc_layout(B, L, e), CPC_ERR_SHAPE);
    sizes[0] = e.saved_total; sizes[1] = e.fwd_total; sizes[2] = e.bwd_total;
    for (int i = 0; i < 5; ++i) sizes[3 + i] = e.L[i];
    for (int i = 0; i < 4; ++i) sizes[8 + i] = e.y[i];
    for (int i = 1; i < 5; ++i) sizes[11 + i] = e.xhat[i];
    for (int i = 0; i < 5; ++i) sizes[16 + i] = e.rstd[i];
    sizes[21] = e.mean0;
    return 0;
}

// fp32 copy of the saved output of encoder layer `layer` (0..3), whatever its storage (mode 3 keeps layers 0 and 1 in H2
// form): dst (B, L_layer, 256).  For tests and debugging; call it in the mode the forward ran in.
extern "C" int cpc_encoder_saved_activation(const float* saved, int layer, float* dst, int B, int L, void* stream) {
    EncLayout e;
    CPC_RETURN_IF(B <= 0 || !enc_layout(B, L, e), CPC_ERR_SHAPE);
    CPC_RETURN_IF(!saved || !dst || layer < 0 || layer > 3, CPC_ERR_ARG);
    const long rows = (long)B * e.L[layer];
    if (e.bf16) return bf16_decode(saved + e.y[layer], dst, rows * kC, (hipStream_t)stream);
    if (act_h2(layer)) return cpc_h2_decode(saved + e.y[layer], dst, rows, saved + e.sbound + layer + 1, stream);
    if (hipMemcpyAsync(dst, saved + e.y[layer], rows * kC * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return CPC_ERR_ARG;
    return 0;
}

// train_step.hip: an event to record on the forward's stream right behind layer 0's launch (the composite step releases the
// criterion's index preparation there when asked to); per host thread, nullptr = none
namespace cpc {
static thread_local hipEvent_t t_after_conv0 = nullptr;
static thread_local int t_event_layer = 0;               // the event is recorded behind this layer's launch (0 or 1)
void enc_set_after_conv0_event(hipEvent_t ev) { t_after_conv0 = ev; t_event_layer = 0; }
void enc_set_forward_event(int layer, hipEvent_t ev) { t_after_conv0 = ev; t_event_layer = layer; }
// train_step.hip: a second stream (already ordered behind the start of the step) for the weight preparation of layers 1..4 --
// 36 us of small kernels that depend on the parameters only and that layer 0 does not need (but for the four ChannelNorm
// bounds, which then get a 4-workgroup launch of their own in front of it) -- and the event layer 1 waits for; per host thread
static thread_local hipStream_t t_prep_stream = nullptr;
static thread_local hipEvent_t t_prep_done = nullptr;
void enc_set_weight_prep_stream(hipStream_t st, hipEvent_t done) { t_prep_stream = st; t_prep_done = done; }
}  // namespace cpc

// Every weight-only quantity of layers 1..4 -- both GEMM layouts, max|w|, the bounds of the layers' inputs (the previous layer's
// ChannelNorm + ReLU output is bounded by its affine) -- in two launches; the backward finds the dgrad layouts and the bounds in
// `saved`.  mask: bit i (1..4) = layer i's layouts, bit 0 = the input bounds of layers 2..4 (+ the zero row the DMA kernels read
// for the conv's padding), bit 5 = the input bound of layer 1.  pst != st: the bounds by a 4-workgroup launch of their own on
// `st`, everything else on `pst`.
static int enc_prepare_weights(const EncLayout& e, const float* const* params, float* saved, float* scratch, int mask,
                               hipStream_t st, hipStream_t pst) {
    PrepArgs a;
    int nblk = 0;
    for (int i = 1; i < 5; ++i) {
        a.w[i - 1] = params[4 * i];
        a.wp[i - 1] = scratch + e.wp[i];
        a.wd[i - 1] = saved + e.swd[i];
        a.nw[i - 1] = params[4 * (i - 1) + 2];
        a.nb[i - 1] = params[4 * (i - 1) + 3];
        a.k[i - 1] = kGeom[i].k;
        a.dgrad_h2[i - 1] = e.dxh2[i] && e.dma[i];      // (the register-staged tiles keep the k-blocked weight layouts)
        a.fwd_h2[i - 1] = e.bf16 ? 2 : e.dma[i];        // 1: layer i runs on the DMA kernel (its K-tile-major H2 weight rows);
                                                        // 2: bf16 storage (both layouts in bf16 K-tile-major rows)
        const int per = ((mask >> i) & 1) ? cdiv((long)kC * kGeom[i].k * kC, 256) : 0;
        a.blk0[2 * (i - 1)] = nblk; nblk += per;
        a.blk0[2 * (i - 1) + 1] = nblk; nblk += per;
    }
    a.blk0[8] = nblk;
    a.bound = saved + e.sbound + 1;
    a.partial = scratch + e.famax;
    a.split = weight_split();
    a.mask = mask;
    const bool apart = pst != st || !(mask & 30);            // (bounds alone: the 4-workgroup launch)
    if (apart && (mask & 33)) hipLaunchKernelGGL(enc_prep_bounds_kernel, dim3(4), dim3(256), 0, st, a);
    if (apart) a.mask &= ~33;
    if (a.mask) hipLaunchKernelGGL(enc_prep_amax_kernel, dim3(kPrepParts, (a.mask & 33) ? 5 : 4), dim3(256), 0, pst, a);
    if (nblk > 0) hipLaunchKernelGGL(enc_prep_permute_kernel, dim3(nblk), dim3(256), 0, pst, a);
    CPC_LAUNCH_CHECK();
    // (the zero row the DMA kernels read for the conv's padding: with the full preparation only -- a partial one runs while the
    // previous step's last weight gradient may still be reading it, and zeros stay zeros)
    if (mask == 63 && g_mfma_mode >= 3 && hipMemsetAsync(saved + e.szero, 0, 64 * sizeof(float), pst) != hipSuccess) return CPC_ERR_ARG;
    // the exchange slots of the N-split forward: "empty" (0xFF) for every row of the coming forward -- with the full preparation, or
    // with the tail's share that carries bit 0 (both a whole forward ahead of the kernels that fill them)
    if ((mask & 1) && g_mfma_mode == 3 && hipMemsetAsync(scratch + e.xch[2], 0xFF, e.xch_total * sizeof(float), pst) != hipSuccess)
        return CPC_ERR_ARG;
    return 0;
}

// The preparation alone, for a caller that runs it ahead of the forward (cpc_train_step_tail: at the end of the previous step,
// right behind the optimiser's update, into the NEXT step's parity -- StepHooks -- and with layer 1's share on the stream its
// weight arrives on).  The forward that follows must be told (StepHooks::weights_ready).
extern "C" int cpc_encoder_prepare_weights(const float* const* params, float* saved, float* scratch, int B, int L, int mask,
                                           void* stream) {
    EncLayout e;
    CPC_RETURN_IF(B <= 0 || !enc_layout(B, L, e), CPC_ERR_SHAPE);
    CPC_RETURN_IF(!params || !saved || !scratch || mask <= 0 || mask > 63, CPC_ERR_ARG);
    return enc_prepare_weights(e, params, saved, scratch, mask, (hipStream_t)stream, (hipStream_t)stream);
}

// params: 20 pointers in the reference's state-dict order
//   conv{i}.weight, conv{i}.bias, batchNorm{i}.weight, batchNorm{i}.bias  for i = 0..4
extern "C" int cpc_encoder_forward(const float* wave, const float* const* params, float* saved,
                                   float* scratch, float* z, int B, int L, void* stream) {
    EncLayout e;
    CPC_RETURN_IF(B <= 0 || !enc_layout(B, L, e), CPC_ERR_SHAPE);
    // the weight preparation first, because in mode 3 conv0 already writes its output scaled by the bound of layer 1's input
    // (unless the previous step's tail has done it: StepHooks::weights_ready)
    hipStream_t st = (hipStream_t)stream;
    const StepHooks& hk = step_hooks();
    const bool apart = !hk.weights_ready && t_prep_stream && t_prep_done && t_prep_stream != st;
    hipStream_t pst = apart ? t_prep_stream : st;
    if (!hk.weights_ready) {
        const int rcp = enc_prepare_weights(e, params, saved, scratch, 63, st, pst);
        if (rcp) return rcp;
    }
    if (apart && hipEventRecord(t_prep_done, pst) != hipSuccess) return CPC_ERR_ARG;
    // (layer 0 reads none of it; layers 1..4 wait below -- for the preparation stream, or for the event behind layer 1's updated
    // weight when the previous step left its tail open)
    // Where the stream takes up the previous step's open tail: in front of layer 0 (default) or only in front of layer 1, with
    // layer 0 running UNDER the tail (cpc_set_tail_schedule(1); measured a loss: beside layer 1's weight gradient -- 194 VGPRs, two
    // waves per SIMD resident -- layer 0 gets one wave per SIMD instead of four and takes 107-131 us instead of 50, and the
    // small kernels that end the tail queue up behind its workgroups: 15 -> 66 us for the split reduction alone)
    auto wait_tail = [&]() {
        for (hipEvent_t w : hk.conv1_wait)
            if (w && hipStreamWaitEvent(st, w, 0) != hipSuccess) return false;
        return true;
    };
    auto join_prep = [&]() {
        if (g_tail_conv0_early && !wait_tail()) return false;
        return !apart || hipStreamWaitEvent(st, t_prep_done, 0) == hipSuccess;
    };
    if (!g_tail_conv0_early && !wait_tail()) return CPC_ERR_ARG;
    // 2 / 3: layer 0 starts behind layer 1's weight gradient itself -- behind its split reduction (2) or already behind its GEMM (3) --
    // and runs beside what is left of the tail, four short launches on the weight-gradient stream (the reduction, the optimiser's
    // update of conv1.weight, its maximum and its layouts: ~50 us with their gaps); layer 1 then waits for the tail as in 1
    if (g_tail_conv0_early >= 2 && hk.conv1_wait[0] != nullptr) {
        hipEvent_t* pool = stream_events(st);
        if (!pool || hipStreamWaitEvent(st, pool[g_tail_conv0_early == 3 ? kEvWgrad1Gemm : kEvWgrad1], 0) != hipSuccess) return CPC_ERR_ARG;
    }
    step_timer_mark(0, st);
    if (e.bf16) {
        // bf16-storage variant: y0..y3 and xhat1..4 as bf16 (half the activation bytes), weights rounded to bf16 by the
        // re-layout, one bf16 MFMA per product, fp32 accumulators and ChannelNorm statistics; z stays fp32
        int rc = conv0_forward_bf16(wave, params[0], params[1], params[2], params[3], saved + e.y[0], saved + e.mean0,
                                    saved + e.rstd[0], B, L, st);
        // In-step timing (cpc_set_step_timing): the side stream's release moves behind the markers, directly in front of layer 1's
        // launch -- where the untimed step has it relative to that launch.  With two marker packets between the release and the
        // launch the side stream's kernels get the CUs first and layer 1 measures 290 us instead of 235 in about half of the
        // timed steps (never in an untimed one: rocprofv3, 45 steps, max 254).
        const bool timing = step_hooks().timers != nullptr;
        if (!timing && !rc && t_after_conv0 && hipEventRecord(t_after_conv0, st) != hipSuccess) return CPC_ERR_ARG;
        step_timer_mark(1, st);
        if (!join_prep()) return CPC_ERR_ARG;
        step_timer_mark(7, st);
        if (timing && !rc && t_after_conv0 && hipEventRecord(t_after_conv0, st) != hipSuccess) return CPC_ERR_ARG;
        for (int i = 1; i < 5 && !rc; ++i) {
            rc = conv_fwd_dma_bf16(saved + e.y[i - 1], scratch + e.wp[i], params[4 * i + 1], params[4 * i + 2], params[4 * i + 3],
                                   i == 4 ? z : saved + e.y[i], i == 4, saved + e.xhat[i], saved + e.rstd[i], saved + e.szero,
                                   B, e.L[i - 1], kGeom[i].k, kGeom[i].s, kGeom[i].p, st);
            if (i == 1) step_timer_mark(2, st);
        }
        return rc;
    }
    int rc = cpc_conv0_forward_h2(wave, params[0], params[1], params[2], params[3], saved + e.y[0], saved + e.mean0,
                                  saved + e.rstd[0], act_h2(0) ? saved + e.sbound + 1 : nullptr, B, L, stream);
    if (rc) return rc;
    // In-step timing (cpc_set_step_timing): the side stream's release moves behind the markers, directly in front of layer 1's
    // launch -- where the untimed step has it relative to that launch.  With two marker packets between the release and the launch
    // the side stream's kernels get the CUs first and layer 1 measures 290 us instead of 235 in about half of the timed steps
    // (never in an untimed one: rocprofv3, 45 steps, max 254).
    const bool timing = step_hooks().timers != nullptr;
    const bool release0 = t_after_conv0 && t_event_layer == 0;
    if (!timing && release0 && hipEventRecord(t_after_conv0, st) != hipSuccess) return CPC_ERR_ARG;
    step_timer_mark(1, st);
    if (!join_prep()) return CPC_ERR_ARG;
    step_timer_mark(7, st);
    if (timing && release0 && hipEventRecord(t_after_conv0, st) != hipSuccess) return CPC_ERR_ARG;
    for (int i = 1; i < 5; ++i) {
        if (i == 2) step_timer_mark(2, st);
        if (i == 2 && t_after_conv0 && t_event_layer == 1 && hipEventRecord(t_after_conv0, st) != hipSuccess) return CPC_ERR_ARG;
        float* yo = i == 4 ? z : saved + e.y[i];
        if (e.dma[i]) {
            const long M = (long)B * e.L[i];
            rc = conv_fwd_dma(saved + e.y[i - 1], scratch + e.wp[i], params[4 * i + 1], params[4 * i + 2], params[4 * i + 3],
                              yo, i < 4 && act_h2(i), saved + e.xhat[i], saved + e.rstd[i], saved + e.sbound + i,
                              saved + e.sbound + i + 1, saved + e.szero, B, e.L[i - 1], kGeom[i].k, kGeom[i].s, kGeom[i].p,
                              g_dma_bm ? g_dma_bm : (M >= 256L * 200 ? 256 : 128), st);
        } else if (i >= 2 && act_h2(i - 1) && g_fwd_nsplit > 0 && g_mfma_mode == 3 && !g_force_bm && (kGeom[i].k * kC) % 64 == 0 &&
                   2L * cdiv((long)B * e.L[i], 128) >= g_fwd_nsplit) {
            // 128 x 128 tiles, two workgroups per row tile (1-D grid: partners 8 ids apart, one XCD)
            const RowMap am = conv_rows(saved + e.y[i - 1], B, e.L[i - 1], e.L[i], kGeom[i].s, kGeom[i].p);
            const int tiles = cdiv(am.M, 128);
            const float* wpi = scratch + e.wp[i];
            hipLaunchKernelGGL(conv_fwd_nsplit_kernel, dim3(16 * cdiv(tiles, 8)), dim3(DgradNsTile::NTHREADS), 0, st, am, wpi,
                               kGeom[i].k * kC, params[4 * i + 1], params[4 * i + 2], params[4 * i + 3], yo, saved + e.xhat[i],
                               saved + e.rstd[i], saved + e.sbound + i, wpi + (long)kC * kGeom[i].k * kC,
                               i < 4 && act_h2(i) ? saved + e.sbound + i + 1 : (const float*)nullptr,
                               reinterpret_cast<unsigned long long*>(scratch + e.xch[i]), g_fwd_nsplit_spin);
            CPC_LAUNCH_CHECK();
            rc = 0;
        } else {
            rc = conv_gemm_forward_impl(saved + e.y[i - 1], scratch + e.wp[i], params[4 * i + 1], params[4 * i + 2],
                                        params[4 * i + 3], yo, saved + e.xhat[i], saved + e.rstd[i], saved + e.sbound + i, B,
                                        e.L[i - 1], kGeom[i].k, kGeom[i].s, kGeom[i].p, st, act_h2(i - 1),
                                        i < 4 && act_h2(i) ? saved + e.sbound + i + 1 : nullptr);
        }
        if (rc) return rc;
    }
    return 0;
}

// grads: 20 output pointers in the same order as params (each overwritten).
static int encoder_backward_impl(const float* wave, const float* const* params, const float* saved, const float* z,
                                 const float* dz, float* scratch, float* const* grads, int B, int L, hipStream_t st,
                                 hipStream_t wst);

extern "C" int cpc_encoder_backward(const float* wave, const float* const* params,
                                    const float* saved, const float* z, const float* dz,
                                    float* scratch, float* const* grads, int B, int L, void* stream) {
    return encoder_backward_impl(wave, params, saved, z, dz, scratch, grads, B, L, (hipStream_t)stream,
                                 (hipStream_t)stream);
}

// The same with the four weight-gradient GEMMs (and their split reductions) on `wgrad_stream`: they are not on the
// dx chain (norm backward -> dgrad -> norm backward ...), whose streaming norm backwards and VALU-bound conv0
// backward leave the matrix pipes idle.  Ordering is internal (events); on return `stream` waits for `wgrad_stream`,
// so every output is safe to use on `stream` -- no host synchronisation.
extern "C" int cpc_encoder_backward_streams(const float* wave, const float* const* params, const float* saved,
                                            const float* z, const float* dz, float* scratch, float* const* grads,
                                            int B, int L, void* stream, void* wgrad_stream) {
    return encoder_backward_impl(wave, params, saved, z, dz, scratch, grads, B, L, (hipStream_t)stream,
                                 (hipStream_t)wgrad_stream);
}

static int encoder_backward_impl(const float* wave, const float* const* params, const float* saved, const float* z,
                                 const float* dz, float* scratch, float* const* grads, int B, int L, hipStream_t st,
                                 hipStream_t wst) {
    EncLayout e;
    CPC_RETURN_IF(B <= 0 || !enc_layout(B, L, e), CPC_ERR_SHAPE);
    void* stream = (void*)st;
    hipEvent_t* ev = wst != st ? stream_events(st) : nullptr;      // [0..4] used here
    CPC_RETURN_IF(wst != st && !ev, CPC_ERR_ARG);
    float* colpart = scratch + e.colpart;
    float* tmp = scratch + e.tmp;
    float* small = scratch + e.small;          // [5][3][256]
    // operand bounds of the fp16-split GEMMs: max|dx_i| is accumulated by the kernel that writes dx_i (integer
    // atomicMax on the float bits: exact, order-independent); the bounds of the layers' inputs and the dgrad weight
    // layouts come from the forward (saved)
    float* amax = scratch + e.bamax;
    float* dyamax = scratch + e.dyamax;
    float* dxbound = scratch + e.dxbound;
    const float* xbound = saved + e.sbound;
    (void)hipMemsetAsync(amax, 0, 2 * 5 * kAmaxSlots * sizeof(float), st);      // bamax and dyamax
    // top layer: ReLU'/norm backward of dz
    // the four stand-alone norm backwards leave their per-workgroup column partials in their own buffers; one batched
    // reduction at the end replaces eight small launches on the way
    RowsSumJob jobs[4];
    int njobs = 0;
    auto norm_bwd = [&](int layer, const float* dy, const float* yl, float* dxl) {
        const int M = B * e.L[layer], nblk = cdiv(M, NB_ROWS);
        // the ReLU mask is recomputed from xhat and the affine (bit-identical to the forward's): y is not read
        if (e.bf16 && layer == 4)         // bf16 storage: xhat and dx are bf16; the top layer's dy is autograd's fp32 dz
            hipLaunchKernelGGL((norm_bwd_kernel<2, false, true>), dim3(nblk), dim3(256), 0, st, dy, saved + e.xhat[layer], yl,
                               saved + e.rstd[layer], params[4 * layer + 2], params[4 * layer + 3], dxl, scratch + e.colp[layer],
                               M, (float*)nullptr, 1);
        else if (e.bf16)
            hipLaunchKernelGGL((norm_bwd_kernel<2, true, true>), dim3(nblk), dim3(256), 0, st, dy, saved + e.xhat[layer], yl,
                               saved + e.rstd[layer], params[4 * layer + 2], params[4 * layer + 3], dxl, scratch + e.colp[layer],
                               M, (float*)nullptr, 1);
        else if (e.dxh2[layer])           // dx in H2 storage, scaled for a bound derived from max|dy| (left by the dgrad above it)
            hipLaunchKernelGGL((norm_bwd_kernel<2, false, false, true>), dim3(nblk), dim3(256), 0, st, dy, saved + e.xhat[layer], yl,
                               saved + e.rstd[layer], params[4 * layer + 2], params[4 * layer + 3], dxl, scratch + e.colp[layer],
                               M, (float*)nullptr, 1, dyamax + layer * kAmaxSlots, kAmaxSlots, dxbound + layer);
        else
        hipLaunchKernelGGL(norm_bwd_kernel<2>, dim3(nblk), dim3(256), 0, st, dy, saved + e.xhat[layer], yl,
                           saved + e.rstd[layer], params[4 * layer + 2], params[4 * layer + 3], dxl, scratch + e.colp[layer], M,
                           amax + layer * kAmaxSlots, kAmaxSlots);
        jobs[njobs++] = RowsSumJob{scratch + e.colp[layer], nblk, 3 * kC, scratch + e.tmpq[layer], small + layer * 3 * kC};
    };
    // weight gradient of layer i on the wgrad stream: operand storages as the layout says; a dx in H2 storage comes with the
    // single bound it was scaled for instead of the kAmaxSlots partial maxima
    // Layers 2..4 with both operands in H2 storage leave their partials in own buffers and are reduced by ONE launch behind the
    // last of them (layer 2's): three reductions of 16-64 MB cost 114 us as three launches behind three short GEMMs.
    WgReduceJobs rj;
    int nrj = 0, rj_blocks = 0;
    auto wgrad = [&](int i, const float* xin) {
        if ((e.dxh2[i] || e.bf16) && g_wgrad_dma) {
            const bool batch = !e.bf16 && i >= 2;
            float* part = batch ? scratch + e.partl[i] : scratch + e.part;
            int S = 0;
            int rcw = e.bf16 ? conv_wgrad_dma_bf16(scratch + e.dx[i], xin, part, saved + e.szero, B, e.L[i - 1],
                                                   kGeom[i].k, kGeom[i].s, kGeom[i].p, &S, wst)
                             : conv_wgrad_dma(scratch + e.dx[i], xin, part, dxbound + i, xbound + i, saved + e.szero, B,
                                              e.L[i - 1], kGeom[i].k, kGeom[i].s, kGeom[i].p, &S, wst);
            if (rcw) return rcw;
            const long total = (long)kC * kGeom[i].k * kC;
            if (batch) {
                rj.part[nrj] = part; rj.dw[nrj] = grads[4 * i]; rj.S[nrj] = S; rj.k[nrj] = kGeom[i].k;
                rj.blk0[nrj] = rj_blocks;
                rj_blocks += cdiv(total, 256);
                rj.blk0[++nrj] = rj_blocks;
                if (i == 2) {
                    for (int q = nrj; q < 3; ++q) { rj.part[q] = part; rj.dw[q] = grads[4 * i]; rj.S[q] = 0; rj.k[q] = 1; rj.blk0[q + 1] = rj_blocks; }
                    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(rj_blocks), dim3(256), 0, wst, rj);
                }
                return 0;
            }
            if (i == 1 && ev && g_tail_conv0_early == 3 && hipEventRecord(ev[kEvWgrad1Gemm], wst) != hipSuccess) return CPC_ERR_ARG;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, wst, part, S, kGeom[i].k,
                               grads[4 * i]);
            return 0;
        }
        return conv_layer_wgrad(scratch + e.dx[i], xin, e.bf16 ? 2 : (e.dxh2[i] ? 3 : act_h2(i - 1)), scratch + e.part, grads[4 * i],
                                e.dxh2[i] ? dxbound + i : amax + i * kAmaxSlots, xbound + i, B, e.L[i - 1], kGeom[i].k, kGeom[i].s,
                                kGeom[i].p, e.wg_splits[i], e.wg_rows[i], (void*)wst, e.dxh2[i] ? 1 : kAmaxSlots);
    };
    // open tail (StepHooks): the batched column sums -- the bias / norm gradients of layers 1..4 -- on their own stream as soon
    // as the last stand-alone norm backward (layer 1's) is done, instead of at the end of the chain behind conv0's backward: the
    // caller's optimiser takes these parameters up beside the chain's last two kernels
    const StepHooks& hk = step_hooks();
    hipStream_t sums_st = (ev && hk.open_tail && !g_wgrad1_early && hk.sums_stream && hk.sums_stream != st) ? hk.sums_stream : nullptr;
    GradPtrs gp;
    for (int i = 1; i < 5; ++i) {        // layers 1..4: small[i] = [d norm.weight | d norm.bias | d conv.bias]
        gp.p[(i - 1) * 3 + 0] = grads[4 * i + 2];
        gp.p[(i - 1) * 3 + 1] = grads[4 * i + 3];
        gp.p[(i - 1) * 3 + 2] = grads[4 * i + 1];
    }
    auto sums = [&](hipStream_t s_) {
        const int r_ = rows_sum_multi(jobs, njobs, s_);
        if (r_) return r_;
        hipLaunchKernelGGL(small_to_grads_kernel, dim3(12), dim3(256), 0, s_, small, gp);
        return 0;
    };
    int rc = 0;
    if (e.dxh2[4] && !e.bf16) {          // the top layer's dy is the caller's dz: its maximum (the H2 scale of dx4 derives from it) is reduced here
        const float* xs[1] = {dz};
        const long ns[1] = {(long)B * e.L[4] * kC};
        rc = absmax_slots(xs, ns, 1, dyamax + 4 * kAmaxSlots, st);
        if (rc) return rc;
    }
    norm_bwd(4, dz, z, scratch + e.dx[4]);
    for (int i = 4; i >= 1; --i) {
        const float* xin = saved + e.y[i - 1];
        // dx_i and its bound are final: the weight gradient may start -- except layer 1's, which is as long as its
        // dgrad and only slows it down when both fight for the matrix pipes (0.66 + 0.44 ms together vs 0.29 + 0.30 alone);
        // it is released behind that dgrad and runs beside conv0's VALU-bound backward instead
        const bool late1 = ev && i == 1 && !g_wgrad1_early;
        if (late1 && hipEventRecord(ev[kEvWgradRest], wst) != hipSuccess) return CPC_ERR_ARG;   // all of wst but layer 1's weight gradient
        if (ev && !late1) {
            if (hipEventRecord(ev[i], st) != hipSuccess || hipStreamWaitEvent(wst, ev[i], 0) != hipSuccess)
                return CPC_ERR_ARG;
        }
        if (!late1)
        rc = wgrad(i, xin);
        if (rc) return rc;
        if (e.bf16) {
            // bf16 storage: DMA'd bf16 dgrad into a bf16 temporary (layer 1: straight into conv0's dy), then the norm backward
            float* tmpd = scratch + e.dy0;
            rc = conv_dgrad_dma_bf16(scratch + e.dx[i], saved + e.swd[i], tmpd, saved + e.szero, B, e.L[i - 1], kGeom[i].k,
                                     kGeom[i].s, kGeom[i].p, st);
            if (rc) return rc;
            zero_uncovered_rows(tmpd, B, e.L[i - 1], e.L[i], kGeom[i].s, kGeom[i].p, kC * 2, st);
            if (i >= 2) norm_bwd(i - 1, tmpd, xin, scratch + e.dx[i - 1]);
        } else if (i >= 2 && (e.dxh2[i] || e.dxh2[i - 1] || (g_unfuse_big && (g_unfuse_big == 2 || pick_bm(B * (e.L[i] + 1)) == 128)))) {
            // the fused ReLU'/ChannelNorm-backward epilogue is latency-bound (row-by-row reductions between the loads); a
            // plain dgrad into a temporary (dy0 is free until layer 1's dgrad) + the streaming norm backward is faster.
            // A layer below that keeps its gradient in H2 storage needs max|dy| from this kernel (slots).
            float* tmpd = scratch + e.dy0;
            float* slots = e.dxh2[i - 1] ? dyamax + (i - 1) * kAmaxSlots : nullptr;
            if (e.dxh2[i] && e.dma[i])
                rc = conv_dgrad_dma_h2(scratch + e.dx[i], saved + e.swd[i], tmpd, saved + e.szero, dxbound + i, slots, B,
                                       e.L[i - 1], kGeom[i].k, kGeom[i].s, kGeom[i].p, st);
            else if (e.dxh2[i])          // H2 gradient on the register-staged tile (staged as it lies)
                rc = conv_dgrad_core(scratch + e.dx[i], saved + e.swd[i], 0, nullptr, nullptr, nullptr, nullptr, tmpd, nullptr,
                                     nullptr, nullptr, dxbound + i, nullptr, B, e.L[i - 1], kGeom[i].k, kGeom[i].s,
                                     kGeom[i].p, st, 1, slots, true);
            else
                rc = conv_dgrad_core(scratch + e.dx[i], saved + e.swd[i], 0, nullptr, nullptr, nullptr, nullptr, tmpd, nullptr,
                                     nullptr, nullptr, amax + i * kAmaxSlots, nullptr, B, e.L[i - 1], kGeom[i].k, kGeom[i].s,
                                     kGeom[i].p, st, kAmaxSlots, slots);
            if (rc) return rc;
            zero_uncovered_rows(tmpd, B, e.L[i - 1], e.L[i], kGeom[i].s, kGeom[i].p, kC * 4, st);      // (none for k 4, s 2, p 1)
            norm_bwd(i - 1, tmpd, xin, scratch + e.dx[i - 1]);
        } else if (i >= 2) {
            rc = conv_dgrad_core(scratch + e.dx[i], saved + e.swd[i], 1, saved + e.xhat[i - 1], xin,
                                 saved + e.rstd[i - 1], params[4 * (i - 1) + 2], scratch + e.dx[i - 1], colpart, tmp,
                                 small + (i - 1) * 3 * kC, amax + i * kAmaxSlots, amax + (i - 1) * kAmaxSlots, B, e.L[i - 1], kGeom[i].k,
                                 kGeom[i].s, kGeom[i].p, st, kAmaxSlots);
        } else if (e.dxh2[1]) {
            rc = conv_dgrad_dma_h2(scratch + e.dx[1], saved + e.swd[1], scratch + e.dy0, saved + e.szero, dxbound + 1, nullptr, B,
                                   e.L[0], kGeom[1].k, kGeom[1].s, kGeom[1].p, st);
        } else {
            rc = conv_dgrad_core(scratch + e.dx[1], saved + e.swd[1], 0, nullptr, nullptr, nullptr, nullptr,
                                 scratch + e.dy0, nullptr, nullptr, nullptr, amax + kAmaxSlots, nullptr, B, e.L[0], kGeom[1].k,
                                 kGeom[1].s, kGeom[1].p, st, kAmaxSlots);
        }
        if (rc) return rc;
        // layer 0's output is 4 L1 + 3 steps long for some window lengths (e.g. 978 samples: 195 -> 48): its last step feeds no
        // window of layer 1 and gets no gradient from the phase GEMMs -- conv0's backward must read a zero row there
        if (i == 1 && !e.bf16) zero_uncovered_rows(scratch + e.dy0, B, e.L[0], e.L[1], kGeom[1].s, kGeom[1].p, kC * 4, st);
        if (i == 2 && sums_st) {         // (layer 1's norm backward has just been queued: every partial of the sums is final behind it)
            if (hipEventRecord(ev[kEvNorm1], st) != hipSuccess || hipStreamWaitEvent(sums_st, ev[kEvNorm1], 0) != hipSuccess)
                return CPC_ERR_ARG;
            rc = sums(sums_st);
            if (rc) return rc;
            if (hipEventRecord(ev[kEvSums], sums_st) != hipSuccess) return CPC_ERR_ARG;
        }
        if (late1) {
            if (hipEventRecord(ev[1], st) != hipSuccess || hipStreamWaitEvent(wst, ev[1], 0) != hipSuccess) return CPC_ERR_ARG;
            rc = wgrad(1, xin);
        }
        if (rc) return rc;
    }
    rc = conv0_backward(wave, params[0], params[1], params[2], params[3], saved + e.mean0, saved + e.rstd[0],
                        scratch + e.dy0, e.bf16, scratch + e.conv0, grads[0], grads[1], grads[2], grads[3], B, L, st);
    if (rc) return rc;
    if (!sums_st) {
        rc = sums(st);
        if (rc) return rc;
    }
    CPC_LAUNCH_CHECK();
    if (ev) {                                            // join: everything written on the weight-gradient stream
        const bool late1 = !g_wgrad1_early;
        if (hipEventRecord(ev[kEvWgrad1], wst) != hipSuccess) return CPC_ERR_ARG;
        if (!late1 && hipEventRecord(ev[kEvWgradRest], wst) != hipSuccess) return CPC_ERR_ARG;      // (no earlier point in this order)
        // open tail (the composite step at N = 1, StepHooks): `st` takes up everything but layer 1's weight gradient; the caller
        // orders what reads that gradient -- its optimiser update, on wst itself -- and the next step's layer 1 behind it
        const bool open = step_hooks().open_tail && late1;
        if (hipStreamWaitEvent(st, ev[open ? kEvWgradRest : kEvWgrad1], 0) != hipSuccess) return CPC_ERR_ARG;
    }
    return 0;
}
