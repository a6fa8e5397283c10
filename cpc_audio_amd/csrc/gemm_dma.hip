// Plain GEMMs on the conv layers' DMA-fed tile (dma_tile.h), for the feed-forward network of the transformer layer
// (cpc/transformers.py:86-101: lin1 256 -> 2048, ReLU, dropout, lin2 2048 -> 256 -- 80 % of the layer's FLOPs, six of its
// GEMMs forward + backward):
//   gemm_nt_dma_kernel   C[M, N]   = A[M, K] . B[N, K]^T      A in H2 storage (cpc_common.h: two fp16 pieces per element, 4 bytes,
//                                                             the power-of-two scale of a bound its producer knew), B prepared by
//                                                             gemm_weight_h2_kernel in the K-tile-major rows dma_gemm copies
//   gemm_tn_dma_kernel   C[N1, N2] = sum_m A[m, :]^T B[m, :]  both operands in H2 storage as they lie, fragments by the transposing
//                                                             LDS read (the conv weight gradient's kernel without the taps)
// The generic tiles of gemm.hip stage fp32 operands through registers and split them on the VALU on their way to LDS -- that
// staging, not the matrix pipe, bounds them (~180 TFLOP/s of algorithmic work at these shapes); here an
// operand piece is one global_load_lds per 1 KB and the loop is MFMAs and LDS reads.  Arithmetic as everywhere in mode >= 2:
// hh + hl + lh of the fp16 pieces on v_mfma_f32_32x32x16_f16, fp32 accumulators, the two power-of-two scales undone exactly.
#include "cpc_common.h"
#include "cpc_internal.h"
#include "philox.h"
#include "gemm_tile.h"
#include "dma_tile.h"

namespace cpc {

using GCfg = DmaCfg<256, 32, 2, 2>;

// >= 128 bytes of zeros: what rows past M read
static __device__ __attribute__((aligned(256))) unsigned char g_gemm_zero[256];

// ------------------------------------------------------------------ operand preparation
// B(n, k) = w[n * sn + k * sk] -> wq: 256-column tile n / 256 of the product's N, inside it row (k / 32) * 256 + n % 256 of 128
// bytes = the 32 contraction elements k & ~31 .. in H2 order (h2_byte_of), scaled for max|w| (`amax`: kAmaxSlots partial
// maxima).  l1 (or NULL): max over n of sum_k |B(n, k)| into kAmaxSlots slots (atomicMax; zeroed by the caller) -- the a-priori
// bound of a product's output: |A . B^T| <= max|A| * that.  One wave per row n; blockIdx.y: matrix of a group.
__global__ __launch_bounds__(256) void gemm_weight_h2_kernel(const float* __restrict__ w, long sn, long sk, int N, int K,
                                                             unsigned char* __restrict__ wq, const float* __restrict__ amax,
                                                             float* __restrict__ l1, long w_gs, long wq_gs, long amax_gs, long l1_gs) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const long g = blockIdx.y;
    w += g * w_gs; wq += g * wq_gs; amax += g * amax_gs;
    const float s = scale_for_amax(fold_amax(amax, kAmaxSlots));
    unsigned char* tile = wq + (long)(n >> 8) * K * 1024;
    float sum = 0.f;
    for (int k4 = 4 * lane; k4 < K && n < N; k4 += 256) {  // (rows past N -- wave-uniform -- do nothing, but meet the barrier below)
        float v[4];
        if (sk == 1) {
            const float4 q = *reinterpret_cast<const float4*>(w + (long)n * sn + k4);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = w[(long)n * sn + (long)(k4 + e) * sk];
        }
        sum += (fabsf(v[0]) + fabsf(v[1])) + (fabsf(v[2]) + fabsf(v[3]));
        unsigned char* row = tile + ((long)(k4 >> 5) * 256 + (n & 255)) * 128;
        h2_store4(row, k4 & 31, v[0], v[1], v[2], v[3], s);
    }
    if (l1 != nullptr) {                                  // (block-uniform) one atomic per workgroup: the largest of its four rows' sums
        __shared__ float rowsum[4];
        sum = wave_sum(sum);
        if (lane == 0) rowsum[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(reinterpret_cast<unsigned*>(l1 + g * l1_gs) + (blockIdx.x & (kAmaxSlots - 1)),
                      __float_as_uint(fmaxf(fmaxf(rowsum[0], rowsum[1]), fmaxf(rowsum[2], rowsum[3]))));
    }
}

// The same layout for a B that lies TRANSPOSED in memory (sn == 1: B(n, k) = w[k * sk + n], a data-gradient product's weight):
// lane = n, so the reads of a k are coalesced (the row-per-wave kernel above would read 4-byte pieces sk floats apart: 53 us for
// 2 MB); one wave writes the 128-byte rows of 64 n's for one 32-k block -- 8 KB contiguous.  grid (N / 64, K / 32, G).
__global__ __launch_bounds__(64) void gemm_weight_h2_t_kernel(const float* __restrict__ w, long sk, int N, int K,
                                                              unsigned char* __restrict__ wq, const float* __restrict__ amax,
                                                              long w_gs, long wq_gs, long amax_gs) {
    const long g = blockIdx.z;
    w += g * w_gs; wq += g * wq_gs;
    const float s = scale_for_amax(fold_amax(amax + g * amax_gs, kAmaxSlots));
    const int n = blockIdx.x * 64 + threadIdx.x, kb = blockIdx.y;
    if (n >= N) return;
    unsigned char* row = wq + (long)(n >> 8) * K * 1024 + ((long)kb * 256 + (n & 255)) * 128;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float* src = w + (long)(kb * 32 + 4 * q) * sk + n;
        h2_store4(row, 4 * q, src[0], src[sk], src[2 * sk], src[3 * sk], s);
    }
}
// l1[slot] = max over n of sum_k |w[k * sk + n]| (atomicMax; zeroed by the caller).  A workgroup takes 64 columns n (lane = n: the
// reads of a k are coalesced) and cuts K over its four waves, whose partial sums meet in LDS in a fixed order (one thread per n over
// all of K ran 96 workgroups of serial 2048-term sums: 41 us for 25 MB).  grid (N / 64, G).
__global__ __launch_bounds__(256) void col_l1_kernel(const float* __restrict__ w, long sk, int N, int K, float* __restrict__ l1,
                                                     long w_gs, long l1_gs) {
    __shared__ float part[4][64];
    const long g = blockIdx.y;
    w += g * w_gs;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int kq = (K + 3) / 4, k0 = wv * kq, k1 = min(K, k0 + kq);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int k = k0;
        for (; k + 3 < k1; k += 4) {
            s0 += fabsf(w[(long)k * sk + n]); s1 += fabsf(w[(long)(k + 1) * sk + n]);
            s2 += fabsf(w[(long)(k + 2) * sk + n]); s3 += fabsf(w[(long)(k + 3) * sk + n]);
        }
        for (; k < k1; ++k) s0 += fabsf(w[(long)k * sk + n]);
    }
    part[wv][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wv == 0) {
        const float m = wave_max((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(l1 + g * l1_gs) + (blockIdx.x & (kAmaxSlots - 1)), __float_as_uint(m));
    }
}

// fp32 rows of 256 -> H2 rows scaled for `bound` (kAmaxSlots partial maxima).  One wave per row; blockIdx.y: tensor of a group.
__global__ __launch_bounds__(256) void rows_to_h2_kernel(const float* __restrict__ x, unsigned char* __restrict__ xh,
                                                         const float* __restrict__ bound, long rows, long x_gs, long xh_gs,
                                                         long bound_gs) {
    const int lane = threadIdx.x & 63;
    const long g = blockIdx.y;
    const float s = scale_for_amax(fold_amax(bound + g * bound_gs, kAmaxSlots));
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4 v = *reinterpret_cast<const float4*>(x + g * x_gs + row * kC + 4 * lane);
    h2_store4(xh + g * xh_gs * 4 + row * (kC * 4), 4 * lane, v.x, v.y, v.z, v.w, s);
}

// ------------------------------------------------------------------ NT
// EPI 0: C (fp32, rows ldc floats apart) = A . B^T + bias; amax_out (or NULL): max|C| into kAmaxSlots slots (zeroed by the caller)
// EPI 2: the feed-forward ReLU's derivative fused (GemmEpilogue kind 2 of gemm.hip): v = (A . B^T) * scale where mask != 0, else 0,
//        written in H2 storage scaled for out_bound (a-priori: max|A| * l1 * scale, gemm_weight_h2_kernel); mask: one BIT per
//        element of C, N / 8 bytes per row (the hidden layer's [. != 0], relu_h2_kernel); colsum (or NULL): the tile's column sums,
//        colsum[row tile][N] -- the bias gradient of the layer in front, summed over the row tiles by rows_sum afterwards
struct NtDmaArgs {
    RowMap am;                               // rows of A in ELEMENTS of 4 bytes
    int m_base;                              // first row of this launch's tiles (a product may be cut into a 256-row and a 128-row launch)
    const unsigned char* wq; int K, N;
    const float* bias;
    float* C; long ldc;                      // EPI 2: the H2 tensor, ldc in elements
    const float* a_bound; const float* w_amax;
    float* amax_out;
    const unsigned char* mask; float scale; const float* out_bound; const float* out_l1; float* colsum;
    float* out_slots;                        // EPI 1 / 2: every slot = the bound the output was stored for (written by workgroup (0, 0))
    // EPI 1: C (H2) = dropout(relu(A . B^T + bias)), scaled for (max|A| * l1 + max|bias|) / (1 - p); bits: its [. != 0] mask
    float drop_p; unsigned long long seed; const float* bias_amax; unsigned char* bits; float* flag;
    long bias_amax_gs = 0, bits_gs = 0;
    long a_gs = 0, wq_gs = 0, bias_gs = 0, c_gs = 0, a_bound_gs = 0, w_amax_gs = 0, amax_gs = 0, mask_gs = 0, out_bound_gs = 0,
         out_l1_gs = 0, colsum_gs = 0;
};

template <int EPI, int BM = 256, int WR = 64>
__global__ __launch_bounds__((DmaCfg<BM, WR == 128 ? 16 : 32, WR == 128 ? 4 : 2, 2, WR>::NTHREADS)) void gemm_nt_dma_kernel(NtDmaArgs a) {
    using C = DmaCfg<BM, WR == 128 ? 16 : 32, WR == 128 ? 4 : 2, 2, WR>;
    static_assert(WR == 64 || EPI == 0, "the 128-row wave tile: plain epilogue only");
    constexpr int TM = C::TM, TN = C::TN;
    constexpr int kMaskBytes = EPI == 2 ? BM * 32 : 0;       // EPI 2: the tile's mask bits (BM rows x 256 columns), DMA'd up front
    __shared__ __attribute__((aligned(1024))) unsigned char smem[C::SMEM_BYTES + kMaskBytes];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long g = blockIdx.z;
    const int m0 = a.m_base + blockIdx.x * BM, n0 = blockIdx.y * 256;
    a.am.base += g * a.a_gs;
    const unsigned char* wq = a.wq + g * a.wq_gs + (long)blockIdx.y * a.K * 1024;
    if constexpr (EPI == 2) {
        // one DMA instruction per wave: rows 32 wave + lane / 2 of the tile, 16-byte half lane & 1 of their 32 mask bytes -- in
        // flight under the whole main loop (older than every request of dma_gemm: its counted waits cover it), read from LDS by the
        // epilogue behind dma_gemm's closing barrier (the 32 exposed global round trips per lane cost the epilogue ~4 us)
        static_assert(BM == 256, "the mask tile is dealt to eight waves");
        const int row = 32 * wave + (lane >> 1);
        const unsigned char* src = a.mask + g * a.mask_gs * 4 + (long)min(m0 + row, a.am.M - 1) * (a.N >> 3) + (n0 >> 3) + 16 * (lane & 1);
        dma16_to_lds(src, smem + C::SMEM_BYTES + wave * 1024);
    }
    f32x16 acc[TM][TN];
    dma_gemm<C>(acc, a.am, m0, wq, a.K, g_gemm_zero, 3, smem);
    const float sa = scale_for_amax(fold_amax(a.a_bound + g * a.a_bound_gs, kAmaxSlots));
    const float sw = scale_for_amax(fold_amax(a.w_amax + g * a.w_amax_gs, kAmaxSlots));
    const float inv = 1.0f / (sa * sw);                                   // powers of two: exact
    int col[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) col[tn] = n0 + dma_c_col(tn);
    if constexpr (EPI == 1) {
        // The feed-forward hidden layer straight out of lin1's accumulators (cpc/transformers.py:93,100): bias, ReLU, dropout by the
        // Philox blocks relu_kernel draws (registers 4 q .. 4 q + 3 of an accumulator tile are four consecutive rows of one column =
        // one block), the two fp16 pieces, and the bit mask [hid != 0] by wave ballots -- the 730 MB fp32 round trip of a separate
        // ReLU pass (0.31 ms per step at B = 64) is gone; the storage bound is a-priori: (max|y| * max_n sum_k |W1[n][k]| + max|b1|) / (1 - p).
        unsigned char* Ch = reinterpret_cast<unsigned char*>(a.C) + g * a.c_gs * 4;
        unsigned* bits = reinterpret_cast<unsigned*>(a.bits + g * a.bits_gs * 4);
        const float keep_scale = 1.0f / (1.0f - a.drop_p);
        const float ob = (fold_amax(a.a_bound + g * a.a_bound_gs, kAmaxSlots) * fold_amax(a.out_l1 + g * a.out_l1_gs, kAmaxSlots) +
                          fold_amax(a.bias_amax + g * a.bias_amax_gs, kAmaxSlots)) * keep_scale;
        const float so = scale_for_amax(ob);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < kAmaxSlots) {
            a.out_slots[g * a.out_bound_gs + threadIdx.x] = ob;
            if (threadIdx.x == 0) a.flag[g * a.out_bound_gs] = 1.0f;
        }
        const unsigned th = drop_threshold16(a.drop_p);
        const unsigned long long seed = a.seed + (unsigned long long)g;
        const bool odd = lane & 1;
        auto swap1 = [](unsigned v) __attribute__((always_inline)) {
            return __builtin_bit_cast(unsigned, dpp_mov<0xB1>(__builtin_bit_cast(float, v)));
        };
        float bv[TN];
        long cboff[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            bv[tn] = a.bias ? a.bias[g * a.bias_gs + col[tn]] : 0.f;
            cboff[tn] = h2_byte_of(col[tn] & ~1) + (odd ? 16 : 0);
        }
        const int dw0 = (n0 + (wave & 1) * 128) >> 5;                     // dword of this wave's first column tile in a mask row
        const int ndw = a.N >> 5;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int mq = m0 + dma_c_row(tm, 4 * q4);                // rows mq .. mq + 3 (a multiple of four)
                Philox4 draw[TN / 2];                                    // column tiles 2 u and 2 u + 1 (32 columns apart) share a block
                if (a.drop_p > 0.f) {
#pragma unroll
                    for (int u = 0; u < TN / 2; ++u) draw[u] = philox4x32_10(seed, 1u, ffn_drop_block(mq, col[2 * u]));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * q4 + j, m = mq + j;
                    const bool live = m < a.am.M;                         // uniform over each half-wave (one row)
                    unsigned word[TN];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        float v = fmaxf(fmaf(acc[tm][tn][r], inv, bv[tn]), 0.f);
                        if (a.drop_p > 0.f) v = ffn_drop_field(draw[tn >> 1], j, col[tn]) >= th ? v * keep_scale : 0.f;
                        _Float16 h, l;
                        h2_split(v, so, h, l);
                        const unsigned long long nz = __ballot(live && (float)h != 0.f);      // lanes 0-31: row m, 32-63: row m + 4
                        if ((lane & 31) == 0 && live) bits[(long)m * ndw + dw0 + tn] = (unsigned)(lane ? (nz >> 32) : nz);
                        const unsigned mine_h = __builtin_bit_cast(unsigned short, h), mine_l = __builtin_bit_cast(unsigned short, l);
                        const unsigned got = swap1(odd ? mine_h : mine_l);
                        word[tn] = odd ? (got | (mine_l << 16)) : (mine_h | (got << 16));
                    }
                    if (live) {
                        unsigned char* rowp = Ch + (long)m * a.ldc * 4;
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) *reinterpret_cast<unsigned*>(rowp + cboff[tn]) = word[tn];
                    }
                }
            }
    } else if constexpr (EPI == 0) {
        float* Cg = a.C + g * a.c_gs;
        float bv[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bv[tn] = a.bias ? a.bias[g * a.bias_gs + col[tn]] : 0.f;
        float cmax = 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + dma_c_row<C::WR>(tm, r);
                if (m < a.am.M) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const float v = fmaf(acc[tm][tn][r], inv, bv[tn]);
                        Cg[(long)m * a.ldc + col[tn]] = v;
                        cmax = fmaxf(cmax, fabsf(v));
                    }
                }
            }
        if (a.amax_out != nullptr) {                                      // block-uniform; one atomic per workgroup
            float* red = reinterpret_cast<float*>(smem);                  // (dma_gemm ended with a barrier: the stages are free)
            cmax = wave_max(cmax);
            if (lane == 0) red[wave] = cmax;
            __syncthreads();
            if (threadIdx.x == 0) {
                float m = 0.f;
                for (int w = 0; w < C::NW; ++w) m = fmaxf(m, red[w]);
                atomicMax(reinterpret_cast<unsigned*>(a.amax_out + g * a.amax_gs) + (blockIdx.x + 5u * blockIdx.y) % (unsigned)kAmaxSlots,
                          __float_as_uint(m));
            }
        }
    } else {
        unsigned char* Ch = reinterpret_cast<unsigned char*>(a.C) + g * a.c_gs * 4;
        // the bound the output is stored for: max|A| * max_n sum_k |B(n, k)| * scale (both factors as slots)
        const float ob = fold_amax(a.out_bound + g * a.out_bound_gs, kAmaxSlots) * fold_amax(a.out_l1 + g * a.out_l1_gs, kAmaxSlots) * a.scale;
        const float so = scale_for_amax(ob);
        if (a.out_slots != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < kAmaxSlots)
            a.out_slots[g * a.out_bound_gs + threadIdx.x] = ob;
        const float f = inv * a.scale;
        const bool odd = lane & 1;
        auto swap1 = [](unsigned v) __attribute__((always_inline)) {     // the neighbouring lane's value (quad_perm [1,0,3,2])
            return __builtin_bit_cast(unsigned, dpp_mov<0xB1>(__builtin_bit_cast(float, v)));
        };
        float csum[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) csum[tn] = 0.f;
        // mask: one bit per element; the 128 columns a wave's lanes hold of one row are 16 aligned bytes of the tile in LDS
        const unsigned char* mlds = smem + C::SMEM_BYTES + 16 * (wave & 1);     // [row][32 bytes]: this wave's 128 columns = one half
        const int bit = lane & 31;
        long cboff[TN];                                                   // byte of this lane's dword inside a row, per column tile
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) cboff[tn] = h2_byte_of(col[tn] & ~1) + (odd ? 16 : 0);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            uint4 mb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) mb[r] = *reinterpret_cast<const uint4*>(mlds + dma_c_row(tm, r) * 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + dma_c_row(tm, r);
                const bool live = m < a.am.M;                             // uniform over each half-wave (one row)
                const unsigned w4[4] = {mb[r].x, mb[r].y, mb[r].z, mb[r].w};
                unsigned word[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {                         // (the neighbour swap runs on every lane: convergent)
                    const float v = (live && ((w4[tn] >> bit) & 1u)) ? acc[tm][tn][r] * f : 0.f;
                    csum[tn] += v;
                    _Float16 h, l;
                    h2_split(v, so, h, l);
                    const unsigned mine_h = __builtin_bit_cast(unsigned short, h), mine_l = __builtin_bit_cast(unsigned short, l);
                    const unsigned got = swap1(odd ? mine_h : mine_l);    // even lane: the pair's h pieces, odd lane: its l pieces
                    word[tn] = odd ? (got | (mine_l << 16)) : (mine_h | (got << 16));
                }
                if (live) {                                               // ONE guard per row (a guard per store: 128 exec-mask round trips)
                    unsigned char* rowp = Ch + (long)m * a.ldc * 4;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) *reinterpret_cast<unsigned*>(rowp + cboff[tn]) = word[tn];
                }
            }
        }
        if (a.colsum != nullptr) {                                        // block-uniform; fixed order: lane halves, then the waves along M
            float (*cs)[256] = reinterpret_cast<float (*)[256]>(smem);    // [WAVES_M][256]
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const float s = csum[tn] + __shfl_xor(csum[tn], 32);
                if (lane < 32) cs[wave >> 1][dma_c_col(tn)] = s;
            }
            __syncthreads();
            if (threadIdx.x < 256) {
                const int c = threadIdx.x;
                const float s = (cs[0][c] + cs[1][c]) + (cs[2][c] + cs[3][c]);
                a.colsum[g * a.colsum_gs + (long)blockIdx.x * a.N + n0 + c] = s;
            }
        }
    }
}

// ------------------------------------------------------------------ TN
// part[z][N1][N2] = sum over the rows [z * rows_per_split, ...) of A[m, :]^T B[m, :]; 256 x 256 output tiles, 512 threads, four
// 16-row LDS stages with three in flight (conv_wgrad_dma_kernel<16, 4>'s loop).  1-D grid of 8 * T * ceil(S / 8): all T tiles of
// a row split run on XCD z % 8 (they read the same rows); blockIdx.y: problem of a group.
struct TnDmaArgs {
    const unsigned char* A; long lda;        // bytes between rows
    const unsigned char* B; long ldb;
    int M, N1, N2, rows_per_split, S;
    float* part;
    const float* a_bound; const float* b_bound;
    long a_gs = 0, b_gs = 0 /* bytes */, part_gs = 0, a_bound_gs = 0, b_bound_gs = 0;
};
constexpr int kTnPitch = 1024 + 64, kTnRows = 16, kTnStages = 4;
__global__ __launch_bounds__(512) void gemm_tn_dma_kernel(TnDmaArgs a) {
    constexpr int STAGE = 2 * kTnRows * kTnPitch;
    constexpr int RPW = kTnRows / 8;                               // rows (of each operand) a wave requests per stage
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kTnStages * STAGE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long g = blockIdx.y;
    const int t2 = a.N2 / 256, T = (a.N1 / 256) * t2;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = slot % T, z = (slot / T) * 8 + xcd;
    if (z >= a.S) return;                                          // block-uniform
    const int ta = tile / t2, tb = tile - ta * t2;
    const int mbeg = z * a.rows_per_split;
    const int mend = min(a.M, mbeg + a.rows_per_split);
    const int nch = (mend - mbeg + kTnRows - 1) / kTnRows;
    const unsigned char* zsrc = g_gemm_zero + (lane & 7) * 16;
    const unsigned char* Ab = a.A + g * a.a_gs + (long)ta * 1024 + 16 * lane;
    const unsigned char* Bb = a.B + g * a.b_gs + (long)tb * 1024 + 16 * lane;

    auto issue = [&](int ch, int stage) __attribute__((always_inline)) {
        unsigned char* as = smem + stage * STAGE;
        unsigned char* bs = as + kTnRows * kTnPitch;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = RPW * wave + r;
            const int m = mbeg + kTnRows * ch + row;               // wave-uniform
            const bool ok = m < mend;
            dma16_to_lds(ok ? Ab + (long)m * a.lda : zsrc, as + row * kTnPitch);
            dma16_to_lds(ok ? Bb + (long)m * a.ldb : zsrc, bs + row * kTnPitch);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // this lane inside its 16-lane group: source row (p >> 2), channel quad (p & 3) of the group's 16 channels (conv_dma.hip)
    const int G4 = lane >> 4, pq = lane & 15, q = pq & 3;
    const int lane_off = (pq >> 2) * kTnPitch + (2 * (G4 & 1) + (q >> 1)) * 32 + (q & 1) * 8 + 8 * (G4 >> 1) * kTnPitch;
    const int a_off = lane_off + wm * 64 * 4, b_off = lane_off + kTnRows * kTnPitch + wn * 128 * 4;

#pragma unroll
    for (int c0 = 0; c0 < kTnStages - 1; ++c0)
        if (c0 < nch) issue(c0, c0);
    for (int ch = 0; ch < nch; ++ch) {
        // stage ch has landed once at most the requests of the later stages are outstanding (2 * RPW per stage and wave)
        if (ch + 2 < nch) { CPC_WAIT_VMCNT(2 * 2 * RPW); }
        else if (ch + 1 < nch) { CPC_WAIT_VMCNT(2 * RPW); }
        else { CPC_WAIT_VMCNT(0); }
        __builtin_amdgcn_s_barrier();          // stage ch has landed for everybody; everybody is done with stage ch - 1
        if (ch + kTnStages - 1 < nch) issue(ch + kTnStages - 1, (ch + kTnStages - 1) % kTnStages);
        const unsigned char* st = smem + (ch % kTnStages) * STAGE;
        const unsigned char* sa = st + a_off, *sb = st + b_off;
        using SP = SplitPlanes<2>;
        s16x4 ah[2][2][2], bh[4][2][2];                             // [tile][piece][rows 0-3 / 4-7 of the 8-row operand]
#define CPC_TR_A(tm, pl) ah[tm][pl][0] = lds_read_tr16<16 * pl + tm * 128>(sa); \
                         ah[tm][pl][1] = lds_read_tr16<16 * pl + tm * 128 + 4 * kTnPitch>(sa)
#define CPC_TR_B(tn, pl) bh[tn][pl][0] = lds_read_tr16<16 * pl + tn * 128>(sb); \
                         bh[tn][pl][1] = lds_read_tr16<16 * pl + tn * 128 + 4 * kTnPitch>(sb)
        CPC_TR_A(0, 0); CPC_TR_A(1, 0); CPC_TR_B(0, 0); CPC_TR_B(1, 0); CPC_TR_B(2, 0); CPC_TR_B(3, 0);
        CPC_TR_A(0, 1); CPC_TR_A(1, 1); CPC_TR_B(0, 1); CPC_TR_B(1, 1); CPC_TR_B(2, 1); CPC_TR_B(3, 1);
#undef CPC_TR_A
#undef CPC_TR_B
        lds_wait_tr16(ah, bh);
        s16x8 af[2][2], bf[4][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) af[tm][pl] = __builtin_shufflevector(ah[tm][pl][0], ah[tm][pl][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) bf[tn][pl] = __builtin_shufflevector(bh[tn][pl][0], bh[tn][pl][1], 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int qq = 0; qq < SP::NPROD; ++qq)                      // small terms first (l*h, h*l, h*h)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = SP::mfma(af[tm][SP::pa(qq)], bf[tn][SP::pb(qq)], acc[tm][tn]);
    }
    const float inv = 1.0f / (scale_for_amax(fold_amax(a.a_bound + g * a.a_bound_gs, kAmaxSlots)) *
                              scale_for_amax(fold_amax(a.b_bound + g * a.b_bound_gs, kAmaxSlots)));      // powers of two: exact
    float* out = a.part + g * a.part_gs + (long)z * a.N1 * a.N2 + (long)ta * 256 * a.N2 + tb * 256;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) out[(long)co * a.N2 + wn * 128 + tn * 32 + (lane & 31)] = acc[tm][tn][r] * inv;
        }
}

__global__ __launch_bounds__(256) void tn_dma_reduce_kernel(const float* __restrict__ part, int S, long n, float* __restrict__ C,
                                                            long c_gs, long part_gs) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // one float4
    if (idx * 4 >= n) return;
    part += (long)blockIdx.y * part_gs;
    float4 s = reinterpret_cast<const float4*>(part)[idx];
    for (int z = 1; z < S; ++z) {
        const float4 v = reinterpret_cast<const float4*>(part + (long)z * n)[idx];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(C + (long)blockIdx.y * c_gs)[idx] = s;
}

// ------------------------------------------------------------------ host side
int g_gemm_dma = 1;        // cpc_set_gemm_dma: 0 off, 1 (default) where the launches fill the chip, 2 always (tests, emulator)
int g_gemm_relu_fused = 1; // cpc_set_gemm_dma(.. + 8 clears it): lin1's ReLU + dropout + H2 storage in its epilogue (else relu_h2_kernel behind it)
bool gemm_dma_relu_fused() { return g_gemm_relu_fused != 0; }
int g_gemm_tail_cus = 0;   // cpc_set_gemm_tail_cus: CU count the tail split plans for (0: the device's; tests)
int g_dma_wave_rows = 64;  // cpc_set_dma_wave_rows: 128 = the plain NT products' 256-row tiles as four 128 x 128 waves (dma_tile.h, W128 loop)
int g_gemm_tail_split = 1; // cpc_set_gemm_dma(.. + 4 clears it): a one-tile-wide NT product's last partial round as 128-row tiles

bool gemm_dma_wanted(int M, int G) {
    if (g_mfma_mode < 2 || g_gemm_dma == 0) return false;
    return g_gemm_dma == 2 || (long)cdiv(M, 256) * G >= 200;
}

int gemm_weight_h2(const float* w, long sn, long sk, int N, int K, float* wq, const float* amax, float* l1, int G, long w_gs,
                   long wq_gs, long amax_gs, long l1_gs, hipStream_t st) {
    if (N % 256 != 0 || K % 32 != 0) return CPC_ERR_SHAPE;
    if (sn == 1 && sk != 1) {
        hipLaunchKernelGGL(gemm_weight_h2_t_kernel, dim3(N / 64, K / 32, G), dim3(64), 0, st, w, sk, N, K,
                           reinterpret_cast<unsigned char*>(wq), amax, w_gs, wq_gs * 4, amax_gs);
        if (l1 != nullptr) hipLaunchKernelGGL(col_l1_kernel, dim3(cdiv(N, 64), G), dim3(256), 0, st, w, sk, N, K, l1, w_gs, l1_gs);
    } else
    hipLaunchKernelGGL(gemm_weight_h2_kernel, dim3(cdiv(N, 4), G), dim3(256), 0, st, w, sn, sk, N, K,
                       reinterpret_cast<unsigned char*>(wq), amax, l1, w_gs, wq_gs * 4, amax_gs, l1_gs);
    CPC_LAUNCH_CHECK();
    return 0;
}

int rows_to_h2(const float* x, float* xh, const float* bound, long rows, int G, long x_gs, long xh_gs, long bound_gs, hipStream_t st) {
    hipLaunchKernelGGL(rows_to_h2_kernel, dim3(cdiv(rows, 4), G), dim3(256), 0, st, x, reinterpret_cast<unsigned char*>(xh), bound, rows,
                       x_gs, xh_gs, bound_gs);
    CPC_LAUNCH_CHECK();
    return 0;
}

static bool nt_shape_ok(int M, int N, int K) {
    return M > 0 && N % 256 == 0 && K % 256 == 0 && ((K >> 8) & ((K >> 8) - 1)) == 0;      // K = 256 * 2^j (dma_gemm's K walk)
}

// the plain product's launches: one workgroup per CU (128 KB of LDS): T tiles on C CUs take ceil(T / C) rounds, and a last round
// that is a third full costs a whole one -- 348 tiles of the K = 2048 products of the predictors' group on 256 CUs run two rounds
// for 1.36 rounds of work.  Such a product is cut by rows: as many 256-row tiles per problem as fill whole rounds, the rest as
// 128-row tiles (half the time each): 252 + 192 workgroups = 1 + 0.55 rounds.  (The wide-N products of the feed-forward network have
// > 10 rounds and are left alone: `wide_ok` is the criterion's heads' product, 29 x 12 tiles.)
static int launch_nt_plain(NtDmaArgs& a, int M, int N, int G, bool wide_ok, hipStream_t st) {
    a.m_base = 0;
    int dev = 0, cus = 0;
    const int per_row = (N / 256) * G;                                 // workgroups per row tile
    const long tiles = (long)cdiv(M, 256) * per_row;
    if (g_gemm_tail_cus > 0) cus = g_gemm_tail_cus;
    else if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    if (g_gemm_tail_split && (N == 256 || wide_ok) && cus > 0 && tiles > cus && tiles % cus != 0 && tiles % cus < (3 * cus) / 4) {
        const int n256 = (int)((tiles / cus) * cus / per_row);         // 256-row tiles per problem in the first launch
        if (n256 > 0 && n256 < cdiv(M, 256)) {
            if (g_dma_wave_rows == 128) hipLaunchKernelGGL((gemm_nt_dma_kernel<0, 256, 128>), dim3(n256, N / 256, G), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((gemm_nt_dma_kernel<0, 256>), dim3(n256, N / 256, G), dim3(DmaCfg<256, 32, 2, 2>::NTHREADS), 0, st, a);
            a.m_base = n256 * 256;
            hipLaunchKernelGGL((gemm_nt_dma_kernel<0, 128>), dim3(cdiv(M - a.m_base, 128), N / 256, G), dim3(DmaCfg<128, 32, 2, 2>::NTHREADS), 0, st, a);
            CPC_LAUNCH_CHECK();
            return 0;
        }
    }
    if (g_dma_wave_rows == 128) hipLaunchKernelGGL((gemm_nt_dma_kernel<0, 256, 128>), dim3(cdiv(M, 256), N / 256, G), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_nt_dma_kernel<0, 256>), dim3(cdiv(M, 256), N / 256, G), dim3(GCfg::NTHREADS), 0, st, a);
    CPC_LAUNCH_CHECK();
    return 0;
}

int gemm_nt_dma(const float* a_h2, int lda, const float* wq, const float* bias, float* C, long ldc, int M, int N, int K,
                const float* a_bound, const float* w_amax, float* amax_out, int G, long a_gs, long wq_gs, long bias_gs, long c_gs,
                long a_bound_gs, long w_amax_gs, long amax_gs, hipStream_t st) {
    if (!nt_shape_ok(M, N, K)) return CPC_ERR_SHAPE;
    NtDmaArgs a{};
    a.am = plain_rows(a_h2, M, lda);
    a.wq = reinterpret_cast<const unsigned char*>(wq); a.K = K; a.N = N;
    a.bias = bias; a.C = C; a.ldc = ldc; a.a_bound = a_bound; a.w_amax = w_amax; a.amax_out = amax_out;
    a.a_gs = a_gs; a.wq_gs = wq_gs * 4; a.bias_gs = bias_gs; a.c_gs = c_gs; a.a_bound_gs = a_bound_gs; a.w_amax_gs = w_amax_gs;
    a.amax_gs = amax_gs;
    return launch_nt_plain(a, M, N, G, false, st);
}

// One product whose A rows are a RowMap over an H2 tensor (the criterion's heads: the rows (b, t < W) of the context c), C fp32
int gemm_nt_dma_rows(const RowMap& am_h2, const float* wq, float* C, long ldc, int N, int K, const float* a_bound, const float* w_amax,
                     hipStream_t st) {
    if (!nt_shape_ok(am_h2.M, N, K)) return CPC_ERR_SHAPE;
    NtDmaArgs a{};
    a.am = am_h2;
    a.wq = reinterpret_cast<const unsigned char*>(wq); a.K = K; a.N = N;
    a.C = C; a.ldc = ldc; a.a_bound = a_bound; a.w_amax = w_amax;
    return launch_nt_plain(a, am_h2.M, N, 1, true, st);
}

int gemm_nt_dma_masked(const float* a_h2, int lda, const float* wq, float* c_h2, long ldc, const float* mask_h2, float scale, int M,
                       int N, int K, const float* a_bound, const float* w_amax, const float* w_l1, float* colsum, float* out_slots, int G,
                       long a_gs, long wq_gs, long c_gs, long mask_gs, long a_bound_gs, long w_amax_gs, long w_l1_gs, long colsum_gs,
                       hipStream_t st) {
    if (!nt_shape_ok(M, N, K)) return CPC_ERR_SHAPE;
    NtDmaArgs a{};
    a.am = plain_rows(a_h2, M, lda);
    a.wq = reinterpret_cast<const unsigned char*>(wq); a.K = K; a.N = N;
    a.C = c_h2; a.ldc = ldc; a.a_bound = a_bound; a.w_amax = w_amax;
    a.mask = reinterpret_cast<const unsigned char*>(mask_h2); a.scale = scale; a.out_bound = a_bound; a.out_l1 = w_l1; a.colsum = colsum;
    a.out_slots = out_slots;
    a.a_gs = a_gs; a.wq_gs = wq_gs * 4; a.c_gs = c_gs; a.mask_gs = mask_gs; a.a_bound_gs = a_bound_gs; a.w_amax_gs = w_amax_gs;
    a.out_bound_gs = a_bound_gs; a.out_l1_gs = w_l1_gs; a.colsum_gs = colsum_gs;
    a.m_base = 0;
    hipLaunchKernelGGL((gemm_nt_dma_kernel<2, 256>), dim3(cdiv(M, 256), N / 256, G), dim3(GCfg::NTHREADS), 0, st, a);
    CPC_LAUNCH_CHECK();
    return 0;
}

int gemm_nt_dma_relu(const float* a_h2, int lda, const float* wq, const float* bias, float* c_h2, long ldc, float* bits, float drop_p,
                     unsigned long long seed, int M, int N, int K, const float* a_bound, const float* w_amax, const float* w_l1,
                     const float* bias_amax, float* out_slots, float* flag, int G, long a_gs, long wq_gs, long bias_gs, long c_gs,
                     long bits_gs, long a_bound_gs, long w_gs, long out_gs, hipStream_t st) {
    if (!nt_shape_ok(M, N, K) || N != kFfnWidth) return CPC_ERR_SHAPE;
    NtDmaArgs a{};
    a.am = plain_rows(a_h2, M, lda);
    a.m_base = 0;
    a.wq = reinterpret_cast<const unsigned char*>(wq); a.K = K; a.N = N;
    a.bias = bias; a.C = c_h2; a.ldc = ldc; a.a_bound = a_bound; a.w_amax = w_amax; a.out_l1 = w_l1; a.bias_amax = bias_amax;
    a.bits = reinterpret_cast<unsigned char*>(bits); a.drop_p = drop_p; a.seed = seed; a.out_slots = out_slots; a.flag = flag;
    a.a_gs = a_gs; a.wq_gs = wq_gs * 4; a.bias_gs = bias_gs; a.c_gs = c_gs; a.bits_gs = bits_gs; a.a_bound_gs = a_bound_gs;
    a.w_amax_gs = w_gs; a.out_l1_gs = w_gs; a.bias_amax_gs = w_gs; a.out_bound_gs = out_gs;
    hipLaunchKernelGGL((gemm_nt_dma_kernel<1, 256>), dim3(cdiv(M, 256), N / 256, G), dim3(GCfg::NTHREADS), 0, st, a);
    CPC_LAUNCH_CHECK();
    return 0;
}

// row splits: ~512 workgroups over the T tiles of the G problems, a multiple of 8 splits (one XCD each), whole 64-row blocks, at
// least 512 rows per split (every split costs a partial tile written and read again, and a prologue)
void gemm_tn_dma_plan(int M, int N1, int N2, int G, int* splits, int* rows) {
    const long T = (long)(N1 / 256) * (N2 / 256) * (G > 0 ? G : 1);
    int S = cdiv(512, T);
    S = cdiv(S, 8) * 8;
    int r = cdiv(cdiv(M, S), 64) * 64;
    if (r < 512) r = 512;
    *rows = r;
    *splits = cdiv(M, r);
}
long gemm_tn_dma_part_floats(int M, int N1, int N2, int G) {
    int S, rows;
    gemm_tn_dma_plan(M, N1, N2, G, &S, &rows);
    return (long)S * N1 * N2;
}

int gemm_tn_dma(const float* a_h2, int lda, int N1, const float* b_h2, int ldb, int N2, int M, float* part, float* C,
                const float* a_bound, const float* b_bound, int G, long a_gs, long b_gs, long part_gs, long c_gs, long a_bound_gs,
                long b_bound_gs, hipStream_t st) {
    if (M <= 0 || N1 % 256 != 0 || N2 % 256 != 0) return CPC_ERR_SHAPE;
    TnDmaArgs a{};
    a.A = reinterpret_cast<const unsigned char*>(a_h2); a.lda = 4L * lda;
    a.B = reinterpret_cast<const unsigned char*>(b_h2); a.ldb = 4L * ldb;
    a.M = M; a.N1 = N1; a.N2 = N2;
    gemm_tn_dma_plan(M, N1, N2, G, &a.S, &a.rows_per_split);
    a.part = part; a.a_bound = a_bound; a.b_bound = b_bound;
    a.a_gs = a_gs * 4; a.b_gs = b_gs * 4; a.part_gs = part_gs; a.a_bound_gs = a_bound_gs; a.b_bound_gs = b_bound_gs;
    const int T = (N1 / 256) * (N2 / 256);
    hipLaunchKernelGGL(gemm_tn_dma_kernel, dim3(8 * T * cdiv(a.S, 8), G), dim3(512), 0, st, a);
    const long n = (long)N1 * N2;
    hipLaunchKernelGGL(tn_dma_reduce_kernel, dim3(cdiv(n / 4, 256), G), dim3(256), 0, st, part, a.S, n, C, c_gs, part_gs);
    CPC_LAUNCH_CHECK();
    return 0;
}

}  // namespace cpc

// 0: the transformer layer's feed-forward GEMMs stay on the register-staged tiles of gemm.hip; 1 (default): on the DMA-fed tiles
// of this file where a call's launches fill the chip (the K predictors as a group); 2: always (tests).  Results agree to summation
// order (same pieces, same products).
extern "C" int cpc_set_gemm_dma(int mode) {
    if (mode < 0 || mode > 14 || (mode & 3) == 3) return CPC_ERR_ARG;
    cpc::g_gemm_dma = mode & 3;
    cpc::g_gemm_tail_split = (mode & 4) ? 0 : 1;      // + 4: no 128-row tail launch (A/B)
    cpc::g_gemm_relu_fused = (mode & 8) ? 0 : 1;      // + 8: the ReLU / dropout pass behind lin1 instead of in its epilogue (A/B, tests)
    return 0;
}

// Wave tile of the DMA-fed plain NT products' 256-row tiles: 64 (default: eight waves of 64 x 128) or 128 (four waves of 128 x 128,
// one per SIMD -- dma_tile.h, the W128 loop).  Same products in the same order: the same bits.
extern "C" int cpc_set_dma_wave_rows(int rows) {
    if (rows != 64 && rows != 128) return CPC_ERR_ARG;
    cpc::g_dma_wave_rows = rows;
    return 0;
}

// CU count the tail split of the DMA-fed NT products plans for: 0 (default) = the device's; tests set a small one to walk the
// two-launch form (256-row tiles, then 128-row tiles) at sizes the emulator can run.
extern "C" int cpc_set_gemm_tail_cus(int cus) {
    if (cus < 0) return CPC_ERR_ARG;
    cpc::g_gemm_tail_cus = cus;
    return 0;
}
