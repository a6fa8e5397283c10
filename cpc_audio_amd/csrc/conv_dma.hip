// Conv1d(256 -> 256, k = 2s) + bias + ChannelNorm + ReLU as an implicit NT GEMM whose BOTH operands go global -> LDS by
// DMA (global_load_lds_dwordx4): encoder layers 1 and 2 in the default mode (cpc_set_mfma_mode(3)).
//
// Reference: cpc/model.py:85-88 (conv1: k8 s4 p2, conv2: k4 s2 p1), :50-58 (ChannelNorm), :101-102 (relu(norm(conv))).
//
// Why a second forward kernel.  conv_fwd_kernel<128,2> (enc_conv.hip) stages fp32 activations through VGPRs, splits them
// into two fp16 pieces on the VALU and stores them to LDS: its cost is the sum of MFMA + split/store + global-load time
// (DESIGN.md section 4.2), i.e. the staging path, not the matrix pipe, sets its speed.  Here
//   * the activation arrives already split ("H2" storage, cpc_common.h: the producing epilogue -- conv0's, or this
//     kernel's -- writes the two fp16 pieces in the 4 bytes of every element, scaled by the power of two that its
//     ChannelNorm bound fixes a priori), and the weight re-layout kernel writes the same format tile by tile, so a
//     K-tile of both operands is a set of 1 KB pieces that 64 lanes copy with ONE instruction each: no VGPR staging, no
//     VALU, no ds_write in the main loop;
//   * tiles are BM x 256 x 32(k) with 64 x 128 wave tiles (8 accumulator tiles of 32 x 32 per wave): 12 ds_read_b128
//     per 24 MFMAs, 62 B/clk of LDS reads per CU at the full MFMA rate (the 64 x 64 wave tiles of the older kernel need
//     85 + 32 B/clk of the 128 available);
//   * two LDS stages of (BM + 256) x 128 B (32 k), one in flight while the other is multiplied; counted vmcnt + one raw
//     s_barrier per stage (cpc_set_dma_pipeline: the other schedules that were measured, and what they showed -- neither
//     the L2 -> LDS path (27 TB/s chip-wide in tools/probe_dma_bw.hip, the kernel draws 5) nor the bytes in flight limit
//     this kernel; at the full clock it is the issue structure (a DMA piece costs its wave ~70 clocks among MFMAs and
//     100-300 next to LDS reads), inside the train step the power budget);
//   * LDS rows are 64 B with the four 16-byte pieces XOR-swizzled by ((row >> 2) & 3) (128 B / eight pieces /
//     ((row >> 1) & 7) in the two-stage variant): the lanes of every ds_read_b128 service group then hit 16 distinct
//     16-byte slots (conflict-free).  The swizzle is applied on the GLOBAL side
//     (the LDS side of a DMA is lane-linear), which costs nothing: 8 consecutive lanes still cover one 128-byte line.
// Arithmetic is that of mode 2: x*y accumulated as hh + hl + lh of the fp16 pieces on v_mfma_f32_32x32x16_f16, fp32
// accumulators, power-of-two operand scales undone exactly; ChannelNorm on the accumulators.
#include "cpc_common.h"
#include "cpc_internal.h"
#include "gemm_tile.h"
#include "dma_tile.h"

namespace cpc {

// ---- the same product with every input row fetched ONCE (forward, k = 2s, H2 operands, 256-row tiles inside one sequence)
// Tap j of output row t and tap j + s of row t - 1 are the same input row: dma_gemm walks the taps in pairs so that the
// second fetch hits L2, but it still brings the row to LDS twice.  Here the unit is a PAIR (taps j and j + s, 32 channels):
// the 257 input rows (t0 + r) s + j - p, r = 0..256, as 128-byte rows stay in LDS for two half-stages; half 0 multiplies rows
// r by tap j's weight tile, half 1 rows r + 1 by tap j + s's.  A and W rotate separately: two A buffers of 264 rows, two W
// buffers of one tap each (34 + 2 x 32 KB per 64 k instead of 2 x 64 KB).  What this buys is less the bytes than the time the
// activation rows -- the operand that comes from HBM -- get to arrive: a pair's rows are requested two half-stages before they
// are needed (dma_gemm: one), the weights (L2 hits) one.  Requires the tile's rows to be consecutive steps of ONE sequence
// (Lout % 256 == 0), so that "row r + 1" is the next step of the same sequence.
constexpr int kPairARows = 264;                              // 33 DMA pieces of 8 rows; rows 0..256 are used
constexpr int kPairA = kPairARows * 128, kPairW = kC * 128, kPairSmem = 2 * kPairA + 2 * kPairW;
__device__ __forceinline__ void dma_gemm_pair(f32x16 (&acc)[2][4], const RowMap& am, int m0, const unsigned char* __restrict__ wq,
                                              int K, const unsigned char* __restrict__ zeros, int rot_step, unsigned char* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int taps = K >> kCLog2, s = taps >> 1;
    const int npair = s * (kC / 32);                         // (tap pair, 32-channel chunk)
    const int rot = (int)((blockIdx.x * (unsigned)rot_step) % (unsigned)npair);
    const int b = m0 / am.R, t0 = m0 - b * am.R;             // the tile lies inside sequence b
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(am.base) + (long)b * am.bstride * 4;
    // lane l of a 1 KB piece: row l / 8 of its 8 rows, LDS slot l % 8, which holds global piece (l % 8) ^ ((row >> 1) & 7)
    const int prow = lane >> 3, pslot = lane & 7;
    const unsigned char* zsrc = zeros + pslot * 16;
    // A pieces 0..32: wave w copies w, w + 8, w + 16, w + 24; wave 0 also piece 32 (row 256), FIRST, so that "the four
    // youngest may be in flight" means the same thing for every wave
    int a_tau0[5], a_goff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int piece = i == 0 ? 32 : wave + 8 * (i - 1), row = 8 * piece + prow;
        a_tau0[i] = row <= kC ? (t0 + row) * am.tmul + am.tadd : -(1 << 30);
        a_goff[i] = (pslot ^ ((row >> 1) & 7)) * 16;
    }
    const unsigned char* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (wave * 4 + i) + prow;
        w_src[i] = wq + (long)row * 128 + (pslot ^ ((row >> 1) & 7)) * 16;
    }
    unsigned char* const a_lds = smem;
    unsigned char* const w_lds = smem + 2 * kPairA;
    auto pair_of = [&](int pi, int& j, int& c) __attribute__((always_inline)) {
        int q = pi + rot;
        q = q >= npair ? q - npair : q;
        j = q % s;
        c = q / s;
    };
    auto issue_a = [&](int pi) __attribute__((always_inline)) {
        int j, c;
        pair_of(pi, j, c);
        unsigned char* as = a_lds + (pi & 1) * kPairA;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (i > 0 || wave == 0) {                        // wave-uniform
                const int piece = i == 0 ? 32 : wave + 8 * (i - 1);
                const int tau = a_tau0[i] + j;
                const bool ok = (unsigned)tau < (unsigned)am.Lin;
                dma16_to_lds(ok ? xb + (long)tau * (kC * 4) + c * 128 + a_goff[i] : zsrc, as + piece * 1024);
            }
    };
    auto issue_w = [&](int pi, int half) __attribute__((always_inline)) {
        int j, c;
        pair_of(pi, j, c);
        const long koff = (long)((j + half * s) * (kC / 32) + c) * (kC * 128);
        unsigned char* ws = w_lds + half * kPairW + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16_to_lds(w_src[i] + koff, ws + i * 1024);
    };
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int l31 = lane & 31, kg = lane >> 5;
    const int swb = (l31 >> 1) & 7;
    const int a_row = wm * 64 + l31, b_row = wn * 128 + l31;
    auto multiply = [&](const unsigned char* As, const unsigned char* Ws, int half) __attribute__((always_inline)) {
        using SP = SplitPlanes<2>;
        const int swa = ((l31 + half) >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s16x8 af[2][2], bf[4][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const int slot = 4 * ks + 2 * kg + pl;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    af[tm][pl] = *reinterpret_cast<const s16x8*>(As + (a_row + 32 * tm + half) * 128 + ((slot ^ swa) * 16));
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    bf[tn][pl] = *reinterpret_cast<const s16x8*>(Ws + (b_row + 32 * tn) * 128 + ((slot ^ swb) * 16));
            }
#pragma unroll
            for (int q = 0; q < SP::NPROD; ++q)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc[tm][tn] = SP::mfma(af[tm][SP::pa(q)], bf[tn][SP::pb(q)], acc[tm][tn]);
        }
    };
    issue_w(0, 0);
    issue_a(0);
    for (int pi = 0; pi < npair; ++pi) {
        const bool more = pi + 1 < npair;
        const unsigned char* As = a_lds + (pi & 1) * kPairA;
        CPC_WAIT_VMCNT(0);                      // this pair's rows (requested two half-stages ago) and tap j's weights
        __builtin_amdgcn_s_barrier();           // ... of every wave; and everybody is done with the previous pair
        issue_w(pi, 1);
        if (more) issue_a(pi + 1);
        multiply(As, w_lds, 0);
        if (more) { CPC_WAIT_VMCNT(4); }        // tap j + s's weights; the next pair's rows stay in flight
        else { CPC_WAIT_VMCNT(0); }
        __builtin_amdgcn_s_barrier();
        if (more) issue_w(pi + 1, 0);
        multiply(As, w_lds + kPairW, 1);
    }
    __syncthreads();                            // the buffers are free (the epilogue reuses them)
}

// Storage of an epilogue's outputs
constexpr int kStoreF32 = 0, kStoreH2 = 1, kStoreBf16 = 2;

// Forward conv layer.  am: im2col rows of the input activation (H2 for NP = 2, bf16 for NP = 1); wq: weight in K-tile-major
// rows (permute_w_h2 / permute_w_bf16), for NP = 2 max|w| behind it.  y is written as `ykind` says (H2: scaled by
// scale_for_amax(*y_amax)), xhat as `xkind` (fp32 or bf16), rstd fp32.
// zeros: >= 128 bytes of zeros (the rows of the conv's zero padding and of the ragged last tile read them).
template <int BM, int BKE, int NST, int NP, int WALK = 0, int WR = 64>      // WALK 1: dma_gemm_pair; 2, 3, 4: ping-pong slots with 0, 2, 4 pieces issued among the MFMAs
__global__ __launch_bounds__((DmaCfg<BM, BKE, NST, NP, WR>::NTHREADS)) void conv_fwd_dma_kernel(
    RowMap am, const unsigned char* __restrict__ wq, int K, const float* __restrict__ bias,
    const float* __restrict__ nw, const float* __restrict__ nb, void* __restrict__ y, int ykind,
    void* __restrict__ xhat, int xkind, float* __restrict__ rstd_out, const float* __restrict__ x_amax,
    const float* __restrict__ w_amax, const float* __restrict__ y_amax, const unsigned char* __restrict__ zeros,
    int rot_step) {
    using C = DmaCfg<BM, BKE, NST, NP, WR>;
    constexpr int TM = C::TM, TN = C::TN;
    // ONE LDS object: a second one makes the compiler drain the DMA queue (vmcnt(0)) before every ds_read of the loop
    constexpr bool PAIR = WALK == 1;
    static_assert(!PAIR || (BM == 256 && NP == 2), "the pair walk is built for 256-row H2 tiles");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[PAIR ? kPairSmem : C::SMEM_BYTES];
    const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) % C::WAVES_N;
    const int m0 = blockIdx.x * BM;
    f32x16 acc[TM][TN];
    if constexpr (PAIR) dma_gemm_pair(acc, am, m0, wq, K, zeros, rot_step, smem);
    else dma_gemm<C, (WALK == 0 || WALK == 5) ? -1 : 2 * (WALK - 2), WALK == 5>(acc, am, m0, wq, K, zeros, rot_step, smem);

    // ---- epilogue: undo the operand scales, bias, ChannelNorm (two passes over the accumulators), ReLU
    float inv = 1.0f;
    if constexpr (NP == 2) inv = 1.0f / (scale_for_amax(*x_amax) * scale_for_amax(*w_amax));      // powers of two: exact
    float (*red)[2] = reinterpret_cast<float (*)[2]>(smem);         // [BM][2]: one partial per column half (wave wn)
    int col[TN];
    float gw[TN], gb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        col[tn] = dma_c_col(tn);
        const float bc = bias[col[tn]];
        gw[tn] = nw[col[tn]];
        gb[tn] = nb[col[tn]];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = fmaf(acc[tm][tn][r], inv, bc);
    }
    float mean[TM][16];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = (acc[tm][0][r] + acc[tm][1][r]) + (acc[tm][2][r] + acc[tm][3][r]);
            v = half_wave_sum(v);
            if ((lane & 31) == 0) red[dma_c_row<C::WR>(tm, r)][wn] = v;
        }
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = dma_c_row<C::WR>(tm, r);
            mean[tm][r] = (red[row][0] + red[row][1]) * (1.0f / kC);
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
            if ((lane & 31) == 0) red[dma_c_row<C::WR>(tm, r)][wn] = v;
        }
    __syncthreads();
    const float sy = ykind == kStoreH2 ? scale_for_amax(*y_amax) : 1.0f;
    const bool odd = lane & 1;
    auto swap1 = [](unsigned v) __attribute__((always_inline)) {     // the neighbouring lane's value (quad_perm [1,0,3,2])
        return __builtin_bit_cast(unsigned, dpp_mov<0xB1>(__builtin_bit_cast(float, v)));
    };
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = dma_c_row<C::WR>(tm, r);
            const float var = (red[row][0] + red[row][1]) * (1.0f / (kC - 1));
            const float rs = 1.0f / sqrtf(var + kNormEps);
            const int m = m0 + row;
            const bool live = m < am.M;                              // uniform over each half-wave (one row)
            if (live && wn == 0 && (lane & 31) == 0) rstd_out[m] = rs;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const float xh = (acc[tm][tn][r] - mean[tm][r]) * rs;
                const float yv = fmaxf(relu_in(fmaf(xh, gw[tn], gb[tn])), 0.f);
                const int c0 = col[tn] & ~1;
                // Neighbouring lanes hold neighbouring channels.  16-bit storages are written as one dword per lane pair
                // and tensor: the exchanges are unconditional (convergent), the stores are guarded.
                if (xkind == kStoreF32) {
                    if (live) __builtin_nontemporal_store(xh, reinterpret_cast<float*>(xhat) + (long)m * kC + col[tn]);
                }
                if (ykind == kStoreF32) {
                    if (live) reinterpret_cast<float*>(y)[(long)m * kC + col[tn]] = yv;
                }
                if (ykind == kStoreH2) {
                    // the even lane stores the pair's h pieces, the odd lane the l pieces
                    _Float16 h, l;
                    h2_split(yv, sy, h, l);
                    const unsigned mine_h = __builtin_bit_cast(unsigned short, h), mine_l = __builtin_bit_cast(unsigned short, l);
                    const unsigned got = swap1(odd ? mine_h : mine_l);
                    const unsigned word = odd ? (got | (mine_l << 16)) : (mine_h | (got << 16));
                    if (live)
                        *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(y) + (long)m * (kC * 4) + h2_byte_of(c0) +
                                                     (odd ? 16 : 0)) = word;
                }
                if (ykind == kStoreBf16 || xkind == kStoreBf16) {
                    // both 16-bit: the even lane stores the pair's y, the odd lane the pair's xhat; only xhat 16-bit (the
                    // last layer, whose y = z stays fp32): the odd lane stores it
                    const unsigned yb = bf16_rne(yv), xb = bf16_rne(xh);
                    const unsigned got = swap1(odd ? yb : xb);       // even gets the odd lane's y, odd the even lane's xhat
                    if (live) {
                        if (!odd && ykind == kStoreBf16)
                            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(y) + (long)m * kC + c0) = yb | (got << 16);
                        if (odd && xkind == kStoreBf16)
                            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(xhat) + (long)m * kC + c0) = got | (xb << 16);
                    }
                }
            }
        }
}

// Data gradient of a conv layer (k = 2s) as `s` phase GEMMs with K = 512 (enc_conv.hip, conv_dgrad_kernel): phase r of
// input step tau = q*s + r - p receives dx[q-1] . W[:,:,r+s] + dx[q] . W[:,:,r], a contiguous 2-row window of the
// (B, Lout, 256) gradient.  am: those windows (rows q = 0..Lout of every batch item); wd: the phase's weight in K-tile-major
// rows, phases 256*512*ESZ bytes apart; dprev (B, Lin, 256): gradient w.r.t. the previous layer's output.
//   NP = 1: dx, wd and dprev are bf16 (the bf16-storage variant).
//   NP = 2: dx is H2 storage scaled by scale_for_amax(*dx_bound) (the norm backward that wrote it chose the bound, enc_conv.hip),
//           wd H2 rows (permute_w_dgrad_h2_elem) with max|w| in *w_amax; dprev is fp32, and amax_out (or NULL) receives
//           max|dprev| spread over kAmaxSlots addresses (fold_amax).
template <int BM, int BKE, int NST, int NP, int WR = 64>
__global__ __launch_bounds__((DmaCfg<BM, BKE, NST, NP, WR>::NTHREADS)) void conv_dgrad_dma_kernel(
    RowMap am, const unsigned char* __restrict__ wd, int s, int p, int Lin, void* __restrict__ dprev,
    const unsigned char* __restrict__ zeros, int rot_step, const float* __restrict__ dx_bound,
    const float* __restrict__ w_amax, float* __restrict__ amax_out) {
    using C = DmaCfg<BM, BKE, NST, NP, WR>;
    constexpr int TM = C::TM, TN = C::TN;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[C::SMEM_BYTES];
    const int lane = threadIdx.x & 63;
    // 1-D grid of 8 * s * ceil(tiles / 8): workgroup b runs on XCD b % 8, and the s phases of one row tile read the same
    // rows of dx -- they are given to ONE XCD, 8 ids apart (adjacent in dispatch order), so that its L2 fetches those rows
    // once instead of s times (PMC, layer 1: 589 MB per launch with the phases on blockIdx.y, of which 4 x 67 MB dx)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ph = slot % s, tile = (slot / s) * 8 + xcd;
    const int m0 = tile * BM;
    if (m0 >= am.M) return;                                        // block-uniform
    // Exact rows (dgrad_rows_exact, am.R == Lout): phase ph covers q = q0 .. q0 + Lout - 1, the only rows whose tau is inside
    // [0, Lin) when Lin == s * Lout -- B * Lout rows per phase instead of B * (Lout + 1), i.e. no ragged last tile.
    const int q0 = (am.R == am.Lin && ph < p) ? 1 : 0;
    am.off += q0 * kC;
    am.tadd += q0;
    f32x16 acc[TM][TN];
    dma_gemm<C>(acc, am, m0, wd + (long)ph * (kC * 2 * kC * C::ESZ), 2 * kC, zeros, rot_step, smem);
    const bool odd = lane & 1;
    float inv = 1.0f, amax = 0.f;
    if constexpr (NP == 2) inv = 1.0f / (scale_for_amax(*dx_bound) * scale_for_amax(*w_amax));      // powers of two: exact
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + dma_c_row<C::WR>(tm, r);
            long o = -1;
            if (m < am.M) {
                const int b = m / am.R, q = m - b * am.R + q0;
                const int tau = q * s + ph - p;
                if ((unsigned)tau < (unsigned)Lin) o = (long)b * Lin + tau;
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                if constexpr (NP == 2) {
                    const float v = acc[tm][tn][r] * inv;
                    if (o >= 0) {
                        reinterpret_cast<float*>(dprev)[o * kC + dma_c_col(tn)] = v;
                        amax = fmaxf(amax, fabsf(v));
                    }
                } else {
                    const unsigned mine = bf16_rne(acc[tm][tn][r]);
                    const unsigned got = __builtin_bit_cast(unsigned, dpp_mov<0xB1>(__builtin_bit_cast(float, mine)));
                    if (o >= 0 && !odd)
                        *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(dprev) + o * kC + (dma_c_col(tn) & ~1)) = mine | (got << 16);
                }
            }
        }
    if constexpr (NP == 2) {
        if (amax_out != nullptr) {
            amax = wave_max(amax);
            if (lane == 0)
                atomicMax(reinterpret_cast<unsigned*>(amax_out + (int)(blockIdx.x % (unsigned)kAmaxSlots)),
                          __float_as_uint(amax));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient of a conv layer whose gradient dx AND input x are kept in H2 storage (enc_conv.hip, EncLayout::dxh2):
//   dW[co][tap*256 + ci] = sum_m dx[m][co] * x[b, t*s + tap - p][ci]        (m = (b, t), zero outside [0, Lin))
// The contraction runs over ROWS of both operands, so an MFMA fragment wants 8 consecutive rows of one channel -- the
// transpose of how the tensors lie in memory.  conv_wgrad_kernel (TnTileX3) transposes while it stages through registers;
// here both operands go global -> LDS by DMA exactly as they lie (one 1 KB row per wave instruction, 32 rows of dx and the 32
// matching rows of x per stage, two stages) and the fragments are fetched with ds_read_b64_tr_b16, the LDS read that hands
// each lane of a 16-lane group one COLUMN of a 4 x 16 block of halves (semantics measured in tools/probe_tr16.hip): lane p of
// a group points at row p >> 2, channels 4 (p & 3).. of the group's 16 channels, and receives four consecutive rows of its
// own channel; two such reads make the 8-row operand of v_mfma_f32_32x32x16_f16.  No VGPR staging, no VALU, no ds_write.
//   * workgroup = 512 threads = 8 waves (4 x 2), tile 256 co x 256 ci (one tap), wave tile 64 x 128 (8 accumulator tiles);
//   * 1-D grid of 8 * k * ceil(S / 8): all k taps of a row split z run on XCD z % 8 (they read the same dx rows and
//     overlapping x rows); part[z][co][K] partials are summed by wgrad_reduce_kernel in a fixed order;
//   * LDS rows are 1024 + 64 bytes apart: the four rows a 16-lane group reads then start 16 banks apart.
// Arithmetic: hh + hl + lh of the fp16 pieces, fp32 accumulators, the two power-of-two scales undone exactly.
constexpr int kWgPitch = 1024 + 64;
constexpr int kWgRows = 32;
constexpr int kWgStage = 2 * kWgRows * kWgPitch;
// ROWS x NST: contraction rows per LDS stage x stages.  32 x 2 (round 2): one stage in flight while the other is multiplied --
// and the DMA queue drains completely (vmcnt(0)) before the next stage is requested: rocprofv3 shows 4-4.6 us per 32-row stage
// whatever the grid size (64, 128 or 256 workgroups), against 1.3 us of MFMAs: the loop waits for the round trip of its own 64 KB.
// 16 x 4: the same 136 KB of LDS as four 16-row stages, three of them in flight (counted vmcnt waits), one barrier per 16 rows;
// rows stay whole 1 KB DMA pieces either way (the stage dimension is the row count, not k).
template <int ROWS, int NST>
__global__ __launch_bounds__(512) void conv_wgrad_dma_kernel(
    const unsigned char* __restrict__ dx, const unsigned char* __restrict__ x, int B, int Lin, int Lout, int k, int s, int p,
    int rows_per_split, int S, float* __restrict__ part, const float* __restrict__ dx_bound, const float* __restrict__ x_bound,
    const unsigned char* __restrict__ zeros) {
    static_assert(ROWS * NST == 64 && (ROWS == 16 || ROWS == 32), "136 KB of LDS: 32 x 2 or 16 x 4");
    constexpr int STAGE = 2 * ROWS * kWgPitch;
    constexpr int RPW = ROWS / 8;                                  // rows (of each operand) a wave requests per stage
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tap = slot % k, z = (slot / k) * 8 + xcd;
    if (z >= S) return;                                            // block-uniform
    const int M = B * Lout, K = k * kC;
    const int mbeg = z * rows_per_split;
    const int mend = min(M, mbeg + rows_per_split);
    const int nch = (mend - mbeg + ROWS - 1) / ROWS;
    const unsigned char* zsrc = zeros + (lane & 7) * 16;

    auto issue = [&](int ch, int stage) __attribute__((always_inline)) {
        unsigned char* as = smem + stage * STAGE;
        unsigned char* bs = as + ROWS * kWgPitch;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = RPW * wave + r;
            const int m = mbeg + ROWS * ch + row;                  // wave-uniform
            const unsigned char* a_src = zsrc;
            const unsigned char* b_src = zsrc;
            if (m < mend) {
                a_src = dx + (long)m * (kC * 4) + 16 * lane;
                const int b = m / Lout, t = m - b * Lout;
                const int pos = t * s + tap - p;
                if ((unsigned)pos < (unsigned)Lin) b_src = x + ((long)b * Lin + pos) * (kC * 4) + 16 * lane;
            }
            dma16_to_lds(a_src, as + row * kWgPitch);
            dma16_to_lds(b_src, bs + row * kWgPitch);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // this lane inside its 16-lane group: source row (p >> 2), channel quad (p & 3) of the group's 16 channels
    const int G = lane >> 4, pq = lane & 15, q = pq & 3;
    const int lane_off = (pq >> 2) * kWgPitch + (2 * (G & 1) + (q >> 1)) * 32 + (q & 1) * 8 + 8 * (G >> 1) * kWgPitch;
    const int a_off = lane_off + wm * 64 * 4, b_off = lane_off + ROWS * kWgPitch + wn * 128 * 4;

#pragma unroll
    for (int c0 = 0; c0 < NST - 1; ++c0)
        if (c0 < nch) issue(c0, c0);
    for (int ch = 0; ch < nch; ++ch) {
        // stage ch has landed once at most the requests of the later stages are outstanding (2 * RPW per stage and wave)
        if constexpr (NST == 2) {
            CPC_WAIT_VMCNT(0);
        } else {
            if (ch + 2 < nch) CPC_WAIT_VMCNT(2 * 2 * RPW);
            else if (ch + 1 < nch) CPC_WAIT_VMCNT(2 * RPW);
            else CPC_WAIT_VMCNT(0);
        }
        __builtin_amdgcn_s_barrier();          // stage ch has landed for everybody; everybody is done with stage ch - 1
        if (ch + NST - 1 < nch) issue(ch + NST - 1, (ch + NST - 1) % NST);
        const unsigned char* st = smem + (ch % NST) * STAGE;
        const unsigned char* sa = st + a_off, *sb = st + b_off;
        auto kstep = [&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
            using SP = SplitPlanes<2>;
            s16x4 ah[2][2][2], bh[4][2][2];                         // [tile][piece][rows 0-3 / 4-7 of the 8-row operand]
#define CPC_TR_A(tm, pl) ah[tm][pl][0] = lds_read_tr16<16 * ks * kWgPitch + 16 * pl + tm * 128>(sa); \
                         ah[tm][pl][1] = lds_read_tr16<16 * ks * kWgPitch + 16 * pl + tm * 128 + 4 * kWgPitch>(sa)
#define CPC_TR_B(tn, pl) bh[tn][pl][0] = lds_read_tr16<16 * ks * kWgPitch + 16 * pl + tn * 128>(sb); \
                         bh[tn][pl][1] = lds_read_tr16<16 * ks * kWgPitch + 16 * pl + tn * 128 + 4 * kWgPitch>(sb)
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
            for (int qq = 0; qq < SP::NPROD; ++qq)                  // small terms first (l*h, h*l, h*h)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc[tm][tn] = SP::mfma(af[tm][SP::pa(qq)], bf[tn][SP::pb(qq)], acc[tm][tn]);
        };
        kstep(std::integral_constant<int, 0>());
        if constexpr (ROWS == 32) kstep(std::integral_constant<int, 1>());
    }
    const float inv = 1.0f / (scale_for_amax(*dx_bound) * scale_for_amax(*x_bound));      // powers of two: exact
    float* out = part + (long)z * kC * K + tap * kC;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) out[(long)co * K + wn * 128 + tn * 32 + (lane & 31)] = acc[tm][tn][r] * inv;
        }
}

// The same for the bf16-storage variant (cpc_set_mfma_mode(4)): dx and x are bf16 tensors (512-byte rows), one
// v_mfma_f32_32x32x16_bf16 per product.  One DMA instruction moves row m of BOTH operands (lanes 0..31 the 512 bytes of dx,
// lanes 32..63 those of x) into one 1 KB LDS row [dx | x]; rows 1024 + 64 bytes apart as above.
constexpr int kWgRowsB = 64;                       // contraction rows per stage (64 KB + padding, as the H2 kernel)
constexpr int kWgStageB = kWgRowsB * kWgPitch;
__global__ __launch_bounds__(512) void conv_wgrad_dma_bf16_kernel(
    const unsigned char* __restrict__ dx, const unsigned char* __restrict__ x, int B, int Lin, int Lout, int k, int s, int p,
    int rows_per_split, int S, float* __restrict__ part, const unsigned char* __restrict__ zeros) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * kWgStageB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tap = slot % k, z = (slot / k) * 8 + xcd;
    if (z >= S) return;                                            // block-uniform
    const int M = B * Lout, K = k * kC;
    const int mbeg = z * rows_per_split;
    const int mend = min(M, mbeg + rows_per_split);
    const int nch = (mend - mbeg + kWgRowsB - 1) / kWgRowsB;
    const unsigned char* zsrc = zeros + (lane & 7) * 16;
    const bool xhalf = lane >= 32;                                 // this lane copies x (else dx)
    const int lbyte = 16 * (lane & 31);

    auto issue = [&](int ch, int stage) __attribute__((always_inline)) {
        unsigned char* st = smem + stage * kWgStageB;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = 8 * wave + r;
            const int m = mbeg + kWgRowsB * ch + row;              // wave-uniform
            const unsigned char* src = zsrc;
            if (m < mend) {
                const int b = m / Lout, t = m - b * Lout;
                const int pos = t * s + tap - p;
                if (!xhalf) src = dx + (long)m * (kC * 2) + lbyte;
                else if ((unsigned)pos < (unsigned)Lin) src = x + ((long)b * Lin + pos) * (kC * 2) + lbyte;
            }
            dma16_to_lds(src, st + row * kWgPitch);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int G = lane >> 4, pq = lane & 15;
    const int lane_off = ((pq >> 2) + 8 * (G >> 1)) * kWgPitch + (16 * (G & 1) + 4 * (pq & 3)) * 2;
    const int a_off = lane_off + wm * 64 * 2, b_off = lane_off + 512 + wn * 128 * 2;

    if (nch > 0) issue(0, 0);
    for (int ch = 0; ch < nch; ++ch) {
        CPC_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (ch + 1 < nch) issue(ch + 1, (ch + 1) & 1);
        const unsigned char* st = smem + (ch & 1) * kWgStageB;
        const unsigned char* sa = st + a_off, *sb = st + b_off;
        auto kstep = [&](auto KS) __attribute__((always_inline)) {          // (asm reads: see lds_read_tr16)
            constexpr int ks = decltype(KS)::value;
            s16x4 ah[2][2], bh[4][2];
#define CPC_TR_A(tm) ah[tm][0] = lds_read_tr16<16 * ks * kWgPitch + tm * 64>(sa); ah[tm][1] = lds_read_tr16<16 * ks * kWgPitch + tm * 64 + 4 * kWgPitch>(sa)
#define CPC_TR_B(tn) bh[tn][0] = lds_read_tr16<16 * ks * kWgPitch + tn * 64>(sb); bh[tn][1] = lds_read_tr16<16 * ks * kWgPitch + tn * 64 + 4 * kWgPitch>(sb)
            CPC_TR_A(0); CPC_TR_A(1); CPC_TR_B(0); CPC_TR_B(1); CPC_TR_B(2); CPC_TR_B(3);
#undef CPC_TR_A
#undef CPC_TR_B
            lds_wait_tr16(ah, bh);
            s16x8 af[2], bf[4];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) af[tm] = __builtin_shufflevector(ah[tm][0], ah[tm][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) bf[tn] = __builtin_shufflevector(bh[tn][0], bh[tn][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[tm]),
                                                                          __builtin_bit_cast(bf16x8, bf[tn]), acc[tm][tn], 0, 0, 0);
        };
        kstep(std::integral_constant<int, 0>());
        kstep(std::integral_constant<int, 1>());
        kstep(std::integral_constant<int, 2>());
        kstep(std::integral_constant<int, 3>());
    }
    float* out = part + (long)z * kC * K + tap * kC;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) out[(long)co * K + wn * 128 + tn * 32 + (lane & 31)] = acc[tm][tn][r];
        }
}

// row splits of the DMA weight gradient: >= 512 workgroups over the k taps, a multiple of 8 splits (one XCD each), whole
// 32-row stages
static int g_wgrad_dma_wgs = 256;     // cpc_set_wgrad_dma_groups (<= 512: the partial buffer is sized for 512).  One workgroup per CU
                                      // (its LDS fills one): measured in the step at B = 64, 512 / 384 / 256 workgroups: 3.44 / 3.49 / 3.42 ms
static int g_wgrad_dma_stages = 4;     // LDS stages of the H2 weight-gradient kernel: 2 x 32 rows or 4 x 16 rows (cpc_set_wgrad_dma_stages)
static int g_wgrad_dma_min_rows = 512; // fewest rows a split walks (cpc_set_wgrad_dma_min_rows): every split costs a 256 KB partial tile
                                      // written and read again, and a prologue; the short layers get fewer splits instead of short ones
void conv_wgrad_dma_plan(int M, int k, int* splits, int* rows, int wgs) {
    int S = cdiv(wgs > 0 ? wgs : g_wgrad_dma_wgs, k);
    S = cdiv(S, 8) * 8;
    int r = cdiv(cdiv(M, S), kWgRowsB) * kWgRowsB;        // whole stages of either kernel (32 / 64 rows)
    if (r < 2 * kWgRowsB) r = 2 * kWgRowsB;
    if (wgs <= 0 && r < g_wgrad_dma_min_rows) r = g_wgrad_dma_min_rows;
    *rows = r;
    *splits = cdiv(M, r);
}

// dx (B*Lout, 256) and x (B, Lin, 256) in H2 storage scaled for *dx_bound / *x_bound; part: splits * 256 * k * 256 floats
// (conv_wgrad_dma_plan); the caller reduces the partials (wgrad_reduce_kernel).
int conv_wgrad_dma(const void* dx_h2, const void* x_h2, float* part, const float* dx_bound, const float* x_bound,
                   const float* zeros, int B, int Lin, int k, int s, int p, int* splits_out, hipStream_t st) {
    const int Lout = conv_out_len(Lin, k, s, p);
    int S, rows;
    conv_wgrad_dma_plan(B * Lout, k, &S, &rows, 0);
    if (g_wgrad_dma_stages == 4)
        hipLaunchKernelGGL((conv_wgrad_dma_kernel<16, 4>), dim3(8 * k * cdiv(S, 8)), dim3(512), 0, st,
                           reinterpret_cast<const unsigned char*>(dx_h2), reinterpret_cast<const unsigned char*>(x_h2), B, Lin, Lout,
                           k, s, p, rows, S, part, dx_bound, x_bound, reinterpret_cast<const unsigned char*>(zeros));
    else
        hipLaunchKernelGGL((conv_wgrad_dma_kernel<32, 2>), dim3(8 * k * cdiv(S, 8)), dim3(512), 0, st,
                           reinterpret_cast<const unsigned char*>(dx_h2), reinterpret_cast<const unsigned char*>(x_h2), B, Lin, Lout,
                           k, s, p, rows, S, part, dx_bound, x_bound, reinterpret_cast<const unsigned char*>(zeros));
    CPC_LAUNCH_CHECK();
    *splits_out = S;
    return 0;
}

// bf16 tensors (mode 4); see conv_wgrad_dma
int conv_wgrad_dma_bf16(const void* dx, const void* x, float* part, const float* zeros, int B, int Lin, int k, int s, int p,
                        int* splits_out, hipStream_t st) {
    const int Lout = conv_out_len(Lin, k, s, p);
    int S, rows;
    conv_wgrad_dma_plan(B * Lout, k, &S, &rows, 0);
    hipLaunchKernelGGL(conv_wgrad_dma_bf16_kernel, dim3(8 * k * cdiv(S, 8)), dim3(512), 0, st,
                       reinterpret_cast<const unsigned char*>(dx), reinterpret_cast<const unsigned char*>(x), B, Lin, Lout, k, s, p,
                       rows, S, part, reinterpret_cast<const unsigned char*>(zeros));
    CPC_LAUNCH_CHECK();
    *splits_out = S;
    return 0;
}

__global__ __launch_bounds__(256) void permute_w_h2_kernel(const float* __restrict__ w, unsigned char* __restrict__ wq,
                                                           int k, const float* __restrict__ amax) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < (long)kC * k * kC) permute_w_h2_elem(w, wq, k, *amax, idx);
}

// H2 -> fp32 (tests, debugging): dst[i] = (h + l) / scale_for_amax(*amax) for n_rows rows of 256 channels
__global__ __launch_bounds__(256) void h2_decode_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst,
                                                        long n_rows, const float* __restrict__ amax) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;             // one thread per 4 channels
    if (i >= n_rows * (kC / 4)) return;
    const long row = i / (kC / 4);
    const int c = (int)(i - row * (kC / 4)) * 4;
    uint2 hp, lp;
    h2_load4_raw(src + row * (kC * 4), c, hp, lp);
    const float inv = 1.0f / scale_for_amax(*amax);
    float4 o;
    o.x = h2_join((unsigned short)(hp.x & 0xFFFF), (unsigned short)(lp.x & 0xFFFF), inv);
    o.y = h2_join((unsigned short)(hp.x >> 16), (unsigned short)(lp.x >> 16), inv);
    o.z = h2_join((unsigned short)(hp.y & 0xFFFF), (unsigned short)(lp.y & 0xFFFF), inv);
    o.w = h2_join((unsigned short)(hp.y >> 16), (unsigned short)(lp.y >> 16), inv);
    *reinterpret_cast<float4*>(dst + row * kC + c) = o;
}
// fp32 -> H2 (tests, and callers that hand an fp32 activation to the DMA kernel)
__global__ __launch_bounds__(256) void h2_encode_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst,
                                                        long n_rows, const float* __restrict__ amax) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * (kC / 4)) return;
    const long row = i / (kC / 4);
    const int c = (int)(i - row * (kC / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + row * kC + c);
    h2_store4(dst + row * (kC * 4), c, v.x, v.y, v.z, v.w, scale_for_amax(*amax));
}

static int g_dma_rot = 5;     // K-walk rotation step between neighbouring workgroups (0: all walk in lockstep); measured on
                              // layer 1 at B = 64: 0.224 / 0.217 / 0.209 ms for steps 0 / 1 / 5 (two 32-k stages)
static int g_dgrad_wave_rows = 64;   // cpc_set_dma_pipeline(.. + 8): the 256-row data-gradient tiles as four 128 x 128 waves (dma_tile.h, W128 loop)
static int g_dma_pipe = 2;    // main-loop schedule of the forward kernel on 256-row tiles (cpc_set_dma_pipeline), layer 1 at B = 64,
                              // clocks per 32 k of contraction and CU with zero operands (= at the full clock; floor 3072 = the
                              // MFMAs of two waves per SIMD), tools/bench_conv_k.py:
                              //   1: two 32-k stages                                   6130
                              //   2: tap-pair walk (dma_gemm_pair; where k = 2s and Lout % 256 == 0, else 1; default)  5160
                              //   0: four 16-k stages, three in flight                 slower than 1 (64-byte rows, twice the barriers)
                              //   3, 4, 5: ping-pong slots with 0 / 2 / 4 DMA pieces among the MFMAs   5600-6400
                              //   6: skewed slots (16 + 8 MFMAs per slot and wave)     6400
                              //   7: four 128 x 128 waves, ONE per SIMD, four 16-k stages (dma_tile.h, the W128 loop; round 6)
                              // Inside the train step all of them take 205-210 us (the kernel runs against
                              // the chip's power budget there, random operands cost 35-40 % over zeros in every schedule).

int conv_fwd_dma(const float* x_h2, const float* wq, const float* bias, const float* nw, const float* nb, float* y,
                 int y_h2, float* xhat, float* rstd, const float* x_amax, const float* y_amax, const float* zeros, int B,
                 int Lin, int k, int s, int p, int bm, hipStream_t st) {
    const int Lout = conv_out_len(Lin, k, s, p);
    const RowMap am = conv_rows(x_h2, B, Lin, Lout, s, p);
    const int K = k * kC;
    const int taps = K >> kCLog2;
    if (K % 128 != 0 || (taps & (taps - 1)) != 0) return CPC_ERR_SHAPE;
    const unsigned char* wqb = reinterpret_cast<const unsigned char*>(wq);
    const float* w_amax = wq + (long)kC * k * kC;
    const unsigned char* zb = reinterpret_cast<const unsigned char*>(zeros);
    const int ykind = y_h2 ? kStoreH2 : kStoreF32;
#define CPC_LAUNCH_DMA(BM_, BKE_, NST_)                                                                                       \
    hipLaunchKernelGGL((conv_fwd_dma_kernel<BM_, BKE_, NST_, 2>), dim3(cdiv(am.M, BM_)), dim3(DmaCfg<BM_, BKE_, NST_, 2>::NTHREADS), \
                       0, st, am, wqb, K, bias, nw, nb, (void*)y, ykind, (void*)xhat, kStoreF32, rstd, x_amax, w_amax, y_amax, zb,   \
                       g_dma_rot)
    if (bm == 256 && g_dma_pipe == 2 && k == 2 * s && Lout % 256 == 0)          // every input row once (dma_gemm_pair)
        hipLaunchKernelGGL((conv_fwd_dma_kernel<256, 32, 2, 2, 1>), dim3(cdiv(am.M, 256)), dim3(DmaCfg<256, 32, 2, 2>::NTHREADS), 0,
                           st, am, wqb, K, bias, nw, nb, (void*)y, ykind, (void*)xhat, kStoreF32, rstd, x_amax, w_amax, y_amax, zb,
                           g_dma_rot);
#define CPC_LAUNCH_PP(WALK_)                                                                                                       \
    hipLaunchKernelGGL((conv_fwd_dma_kernel<256, 16, 4, 2, WALK_>), dim3(cdiv(am.M, 256)), dim3(DmaCfg<256, 16, 4, 2>::NTHREADS), 0, \
                       st, am, wqb, K, bias, nw, nb, (void*)y, ykind, (void*)xhat, kStoreF32, rstd, x_amax, w_amax, y_amax, zb,   \
                       g_dma_rot)
    else if (bm == 256 && g_dma_pipe == 3) CPC_LAUNCH_PP(2);                    // ping-pong slots
    else if (bm == 256 && g_dma_pipe == 4) CPC_LAUNCH_PP(3);
    else if (bm == 256 && g_dma_pipe == 5) CPC_LAUNCH_PP(4);
    else if (bm == 256 && g_dma_pipe == 6) CPC_LAUNCH_PP(5);                    // skewed slots
#undef CPC_LAUNCH_PP
    else if (bm == 256 && g_dma_pipe == 7)                                      // four 128 x 128 waves, one per SIMD (dma_tile.h)
        hipLaunchKernelGGL((conv_fwd_dma_kernel<256, 16, 4, 2, 0, 128>), dim3(cdiv(am.M, 256)), dim3(256), 0, st, am, wqb, K, bias, nw, nb,
                           (void*)y, ykind, (void*)xhat, kStoreF32, rstd, x_amax, w_amax, y_amax, zb, g_dma_rot);
    else if (bm == 256 && g_dma_pipe == 0) CPC_LAUNCH_DMA(256, 16, 4);
    else if (bm == 256) CPC_LAUNCH_DMA(256, 32, 2);
    else if (g_dma_pipe == 0) CPC_LAUNCH_DMA(128, 16, 4);
    else CPC_LAUNCH_DMA(128, 32, 2);
#undef CPC_LAUNCH_DMA
    CPC_LAUNCH_CHECK();
    return 0;
}

// ---- bf16-storage variant (cpc_set_mfma_mode(4)): activations, saved xhat and gradients as bf16, weights rounded to bf16
// by the re-layout, one v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate and fp32 ChannelNorm statistics.
// rows per workgroup: 256 when that still gives ~one workgroup per CU, else 128, else 64
static int bf16_bm(long M) { return M >= 256L * 200 ? 256 : (M >= 128L * 200 ? 128 : 64); }

// x, y (y_f32 == 0), xhat: bf16 (B, L, 256); y_f32 != 0: y is the fp32 encoder output z.  wq: permute_w_bf16(..., dgrad = 0).
int conv_fwd_dma_bf16(const void* x, const void* wq, const float* bias, const float* nw, const float* nb, void* y, int y_f32,
                      void* xhat, float* rstd, const float* zeros, int B, int Lin, int k, int s, int p, hipStream_t st) {
    const int Lout = conv_out_len(Lin, k, s, p);
    const RowMap am = conv_rows(reinterpret_cast<const float*>(x), B, Lin, Lout, s, p);    // strides in elements
    const int K = k * kC;
    const int taps = K >> kCLog2;
    if (K % 128 != 0 || (taps & (taps - 1)) != 0) return CPC_ERR_SHAPE;
    const unsigned char* wqb = reinterpret_cast<const unsigned char*>(wq);
    const unsigned char* zb = reinterpret_cast<const unsigned char*>(zeros);
    const int ykind = y_f32 ? kStoreF32 : kStoreBf16;
#define CPC_LAUNCH_DMA(BM_)                                                                                                    \
    hipLaunchKernelGGL((conv_fwd_dma_kernel<BM_, 64, 2, 1>), dim3(cdiv(am.M, BM_)), dim3(DmaCfg<BM_, 64, 2, 1>::NTHREADS), 0, st,  \
                       am, wqb, K, bias, nw, nb, y, ykind, xhat, kStoreBf16, rstd, (const float*)nullptr, (const float*)nullptr, \
                       (const float*)nullptr, zb, g_dma_rot)
    switch (bf16_bm(am.M)) {
        case 256: CPC_LAUNCH_DMA(256); break;
        case 128: CPC_LAUNCH_DMA(128); break;
        default: CPC_LAUNCH_DMA(64); break;
    }
#undef CPC_LAUNCH_DMA
    CPC_LAUNCH_CHECK();
    return 0;
}

// dx (B, Lout, 256) bf16 -> dprev (B, Lin, 256) bf16: gradient w.r.t. the previous layer's output.  wd: permute_w_bf16(..., dgrad = 1).
int conv_dgrad_dma_bf16(const void* dx, const void* wd, void* dprev, const float* zeros, int B, int Lin, int k, int s, int p,
                        hipStream_t st) {
    if (k != 2 * s) return CPC_ERR_SHAPE;
    const int Lout = conv_out_len(Lin, k, s, p);
    RowMap am;                                           // 2-row windows [q-1, q] over dx, q in [0, Lout] (enc_conv.hip)
    am = dgrad_rows(reinterpret_cast<const float*>(dx), B, Lin, Lout, s, p);
    const unsigned char* wdb = reinterpret_cast<const unsigned char*>(wd);
    const unsigned char* zb = reinterpret_cast<const unsigned char*>(zeros);
#define CPC_LAUNCH_DMA(BM_)                                                                                                    \
    hipLaunchKernelGGL((conv_dgrad_dma_kernel<BM_, 64, 2, 1>), dim3(8 * s * cdiv(cdiv(am.M, BM_), 8)), dim3(DmaCfg<BM_, 64, 2, 1>::NTHREADS), 0, \
                       st, am, wdb, s, p, Lin, dprev, zb, 0, (const float*)nullptr, (const float*)nullptr, (float*)nullptr)
    switch (bf16_bm((long)am.M * s)) {
        case 256: CPC_LAUNCH_DMA(256); break;
        case 128: CPC_LAUNCH_DMA(128); break;
        default: CPC_LAUNCH_DMA(64); break;
    }
#undef CPC_LAUNCH_DMA
    CPC_LAUNCH_CHECK();
    return 0;
}

// The fp32-accurate data gradient on the DMA kernel: dx (B, Lout, 256) in H2 storage scaled for *dx_bound, wd = s phases of H2
// weight rows with max|w| behind them (enc_prep_permute_kernel) -> dprev (B, Lin, 256) fp32; amax_out: see the kernel.
int conv_dgrad_dma_h2(const void* dx_h2, const float* wd, float* dprev, const float* zeros, const float* dx_bound,
                      float* amax_out, int B, int Lin, int k, int s, int p, hipStream_t st) {
    if (k != 2 * s) return CPC_ERR_SHAPE;
    const int Lout = conv_out_len(Lin, k, s, p);
    RowMap am;                                           // 2-row windows [q-1, q] over dx, q in [0, Lout] (enc_conv.hip)
    am = dgrad_rows(reinterpret_cast<const float*>(dx_h2), B, Lin, Lout, s, p);
    const unsigned char* wdb = reinterpret_cast<const unsigned char*>(wd);
    const float* w_amax = wd + (long)kC * k * kC;
    const unsigned char* zb = reinterpret_cast<const unsigned char*>(zeros);
#define CPC_LAUNCH_DMA(BM_)                                                                                                    \
    hipLaunchKernelGGL((conv_dgrad_dma_kernel<BM_, 32, 2, 2>), dim3(8 * s * cdiv(cdiv(am.M, BM_), 8)), dim3(DmaCfg<BM_, 32, 2, 2>::NTHREADS), 0, \
                       st, am, wdb, s, p, Lin, (void*)dprev, zb, g_dma_rot, dx_bound, w_amax, amax_out)
    if ((long)am.M * s >= 256L * 200 && g_dgrad_wave_rows == 128)
        hipLaunchKernelGGL((conv_dgrad_dma_kernel<256, 16, 4, 2, 128>), dim3(8 * s * cdiv(cdiv(am.M, 256), 8)), dim3(256), 0, st, am, wdb, s, p,
                           Lin, (void*)dprev, zb, g_dma_rot, dx_bound, w_amax, amax_out);
    else if ((long)am.M * s >= 256L * 200) CPC_LAUNCH_DMA(256);
    else CPC_LAUNCH_DMA(128);
#undef CPC_LAUNCH_DMA
    CPC_LAUNCH_CHECK();
    return 0;
}

// bf16 <-> fp32 rows of 256 channels (tests, debugging)
__global__ __launch_bounds__(256) void bf16_decode_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = bf16_val(src[i]);
}
int bf16_decode(const void* src, float* dst, long n, hipStream_t st) {
    hipLaunchKernelGGL(bf16_decode_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, reinterpret_cast<const unsigned short*>(src), dst, n);
    CPC_LAUNCH_CHECK();
    return 0;
}

int permute_w_h2(const float* w, float* wq, int k, const float* amax, hipStream_t st) {
    const long n = (long)kC * k * kC;
    hipLaunchKernelGGL(permute_w_h2_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, w, reinterpret_cast<unsigned char*>(wq), k, amax);
    CPC_LAUNCH_CHECK();
    return 0;
}

}  // namespace cpc

using namespace cpc;

// Workgroups the DMA weight gradient aims at (row splits = this / taps, rounded up to a multiple of 8); 128..512
extern "C" int cpc_set_wgrad_dma_groups(int wgs) {
    if (wgs < 64 || wgs > 512) return CPC_ERR_ARG;
    g_wgrad_dma_wgs = wgs;
    return 0;
}
extern "C" int cpc_set_wgrad_dma_stages(int n) {
    if (n != 2 && n != 4) return CPC_ERR_ARG;
    g_wgrad_dma_stages = n;
    return 0;
}
extern "C" int cpc_set_wgrad_dma_min_rows(int rows) {
    if (rows < 128 || rows > 8192 || rows % 64 != 0) return CPC_ERR_ARG;
    g_wgrad_dma_min_rows = rows;
    return 0;
}
extern "C" int cpc_set_dma_rotation(int step) {
    CPC_RETURN_IF(step < 0, CPC_ERR_ARG);
    g_dma_rot = step;
    return 0;
}
#ifdef CPC_DMA_TIMING
extern "C" int cpc_debug_dma_stamps(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(cpc::g_dma_stamps), sizeof(unsigned long long) * 96) == hipSuccess ? 0 : CPC_ERR_ARG;
}
#endif
extern "C" int cpc_set_dma_pipeline(int variant) {   // 2: as 1, with the pair walk (dma_gemm_pair) where the shape allows; 3: ping-pong slots
    CPC_RETURN_IF(variant < 0 || (variant & 7) > 7 || variant > 15, CPC_ERR_ARG);      // + 8: the data gradient's 256-row tiles as four 128 x 128 waves
    g_dma_pipe = variant & 7;
    g_dgrad_wave_rows = (variant & 8) ? 128 : 64;
    return 0;
}

extern "C" int cpc_h2_decode(const void* src, float* dst, long n_rows, const float* amax, void* stream) {
    CPC_RETURN_IF(!src || !dst || !amax || n_rows <= 0, CPC_ERR_ARG);
    hipLaunchKernelGGL(h2_decode_kernel, dim3(cdiv(n_rows * (kC / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned char*>(src), dst, n_rows, amax);
    CPC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cpc_h2_encode(const float* src, void* dst, long n_rows, const float* amax, void* stream) {
    CPC_RETURN_IF(!src || !dst || !amax || n_rows <= 0, CPC_ERR_ARG);
    hipLaunchKernelGGL(h2_encode_kernel, dim3(cdiv(n_rows * (kC / 4), 256)), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<unsigned char*>(dst), n_rows, amax);
    CPC_LAUNCH_CHECK();
    return 0;
}

// Weight re-layout for cpc_conv_gemm_forward_h2: wq must hold 256*k*256 + 64 floats (max|w| is kept behind the tiles).
extern "C" int cpc_conv_weight_relayout_h2(const float* w, float* wq, int k, void* stream) {
    CPC_RETURN_IF(!w || !wq || k <= 0, CPC_ERR_ARG);
    const long n = (long)kC * k * kC;
    hipStream_t st = (hipStream_t)stream;
    float* amax = wq + n;
    (void)hipMemsetAsync(amax, 0, sizeof(float), st);
    int rc = cpc_absmax(w, n, amax, stream);
    if (rc) return rc;
    return permute_w_h2(w, wq, k, amax, st);
}

// The DMA forward GEMM alone (one launch): x in H2 storage scaled by scale_for_amax(*x_amax); y in H2 (y_amax != NULL,
// scaled by scale_for_amax(*y_amax), which must bound |y|) or fp32 (y_amax == NULL); xhat, rstd fp32.
// zeros: 32 floats of zeros.  bm: 128 or 256 rows per workgroup (0: chosen by problem size).
extern "C" int cpc_conv_gemm_forward_h2(const void* x_h2, const float* wq, const float* bias, const float* nw,
                                        const float* nb, void* y, float* xhat, float* rstd, const float* x_amax,
                                        const float* y_amax, const float* zeros, int B, int Lin, int k, int s, int p,
                                        int bm, void* stream) {
    CPC_RETURN_IF(B <= 0 || Lin <= 0 || k != 2 * s || Lin + 2 * p < k, CPC_ERR_SHAPE);
    CPC_RETURN_IF(!x_h2 || !wq || !y || !xhat || !rstd || !x_amax || !zeros, CPC_ERR_ARG);
    CPC_RETURN_IF(bm != 0 && bm != 128 && bm != 256, CPC_ERR_ARG);
    const long M = (long)B * conv_out_len(Lin, k, s, p);
    if (bm == 0) bm = M >= 256L * 200 ? 256 : 128;
    return conv_fwd_dma(reinterpret_cast<const float*>(x_h2), wq, bias, nw, nb, reinterpret_cast<float*>(y), y_amax != nullptr,
                        xhat, rstd, x_amax, y_amax, zeros, B, Lin, k, s, p, bm, (hipStream_t)stream);
}
