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
//   * four LDS stages of (BM + 256) x 64 B (16 k), three of them in flight or landed ahead of the one being multiplied:
//     what limits this kernel is the per-CU global -> LDS rate, which is latency-bound (~10 B/clk per CU with one 64 KB
//     stage in flight, far below L2's bandwidth), so the bytes kept in flight are what counts; counted vmcnt (the two
//     younger stages stay in flight) + one raw s_barrier per stage.  (cpc_set_dma_pipeline(1): two 32-k stages.)
//   * LDS rows are 64 B with the four 16-byte pieces XOR-swizzled by ((row >> 2) & 3) (128 B / eight pieces /
//     ((row >> 1) & 7) in the two-stage variant): the lanes of every ds_read_b128 service group then hit 16 distinct
//     16-byte slots (conflict-free).  The swizzle is applied on the GLOBAL side
//     (the LDS side of a DMA is lane-linear), which costs nothing: 8 consecutive lanes still cover one 128-byte line.
// Arithmetic is that of mode 2: x*y accumulated as hh + hl + lh of the fp16 pieces on v_mfma_f32_32x32x16_f16, fp32
// accumulators, power-of-two operand scales undone exactly; ChannelNorm on the accumulators.
#include "cpc_common.h"
#include "cpc_internal.h"
#include "gemm_tile.h"

namespace cpc {

// BKE: contraction elements per LDS stage (16 or 32), NST: LDS stages.  NST - 1 stages are in flight or being read at
// any time; what sets the speed of this kernel is the per-CU global -> LDS rate (~10 B/clk measured: latency-bound, far
// below L2's bandwidth), so the number of bytes kept in flight matters more than the size of a stage.
template <int BM, int BKE_ = 16, int NST_ = 4>
struct DmaCfg {
    static constexpr int BN = kC;
    static constexpr int BKE = BKE_, NST = NST_;
    static constexpr int ROWB = BKE * 4;               // bytes per row and stage (h + l pieces)
    static constexpr int PPR = ROWB / 16;              // 16-byte pieces per row (4 or 8)
    static constexpr int RPP = 1024 / ROWB;            // rows per 1 KB DMA piece (16 or 8)
    static constexpr int SWSH = PPR == 8 ? 1 : 2;      // swizzle: piece ^= (row >> SWSH) & (PPR - 1)
    static constexpr int KS = BKE / 16;                // MFMA k-steps per stage
    static constexpr int WAVES_N = 2, WAVES_M = BM / 64, NW = WAVES_M * WAVES_N;
    static constexpr int NTHREADS = 64 * NW;
    static constexpr int TM = 2, TN = 4;               // 32 x 32 accumulator tiles per wave (64 x 128)
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    static constexpr int A_PER = (BM / RPP) / NW, B_PER = (BN / RPP) / NW;    // 1 KB DMA pieces per wave and stage
    static constexpr int NPS = A_PER + B_PER;          // DMA instructions per wave and stage (vmcnt bookkeeping)
    static constexpr int SMEM_BYTES = NST * STAGE;
    static_assert(BKE == 16 || BKE == 32, "one or two MFMA k-steps per stage");
    static_assert(NST >= 2 && NST <= 4, "2..4 stages");
    static_assert((BM / RPP) % NW == 0 && (BN / RPP) % NW == 0, "whole pieces per wave");
    static_assert(SMEM_BYTES >= BM * 2 * 4 * 2, "the row-statistics exchange reuses the stage buffers");
    static_assert(NPS * (NST - 2) < 64, "vmcnt is a 6-bit counter");
};

// row of the C tile held in accumulator register `reg` of tile tm / column held by this lane for tile tn
template <int BM>
__device__ __forceinline__ int dma_c_row(int tm, int reg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave / DmaCfg<BM>::WAVES_N) * 64 + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
template <int BM>
__device__ __forceinline__ int dma_c_col(int tn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave % DmaCfg<BM>::WAVES_N) * 128 + tn * 32 + (lane & 31);
}

// am: im2col rows of the H2 activation (element strides; an element is 4 bytes in either storage).
// wq: weight in K-tile-major H2 rows [K/32][256][128 B] (permute_w_h2), max|w| behind it.
// y_h2 != 0: y is written in H2 storage scaled by scale_for_amax(*y_amax) (the next layer's operand); else fp32.
// zeros: >= 128 bytes of zeros (the rows of the conv's zero padding and of the ragged last tile read them).
template <int BM, int BKE, int NST>
__global__ __launch_bounds__((DmaCfg<BM, BKE, NST>::NTHREADS)) void conv_fwd_dma_kernel(
    RowMap am, const unsigned char* __restrict__ wq, int K, const float* __restrict__ bias,
    const float* __restrict__ nw, const float* __restrict__ nb, float* __restrict__ y, int y_h2,
    float* __restrict__ xhat, float* __restrict__ rstd_out, const float* __restrict__ x_amax,
    const float* __restrict__ w_amax, const float* __restrict__ y_amax, const unsigned char* __restrict__ zeros,
    int rot_step) {
    using C = DmaCfg<BM, BKE, NST>;
    constexpr int TM = C::TM, TN = C::TN;
    // ONE LDS object: a second one makes the compiler drain the DMA queue (vmcnt(0)) before every ds_read of the loop
    __shared__ __attribute__((aligned(1024))) unsigned char smem[C::SMEM_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
    const int m0 = blockIdx.x * BM;
    const int nkt = K / C::BKE;
    const int taps = K >> kCLog2;                                  // power of two for every caller (8, 4)
    const int tshift = 31 - __builtin_clz(taps);
    const int rot = (int)((blockIdx.x * (unsigned)rot_step) % (unsigned)nkt);

    // ---- per-lane DMA sources.  A piece = RPP rows x ROWB bytes (1 KB); lane l of the piece covers row (l / PPR), LDS
    // slot (l % PPR), which holds global piece (l % PPR) ^ ((row >> SWSH) & (PPR - 1)) of that row.
    const unsigned char* a_src[C::A_PER];
    int a_tau0[C::A_PER];
    const unsigned char* b_src[C::B_PER];
#pragma unroll
    for (int i = 0; i < C::A_PER; ++i) {
        const int row = (wave * C::A_PER + i) * C::RPP + lane / C::PPR;
        const int piece = (lane % C::PPR) ^ ((row >> C::SWSH) & (C::PPR - 1));
        const RowRef rr = resolve_row(am, m0 + row, am.M);
        a_src[i] = reinterpret_cast<const unsigned char*>(rr.ptr) + piece * 16;
        a_tau0[i] = rr.tau0;
    }
#pragma unroll
    for (int i = 0; i < C::B_PER; ++i) {
        const int row = (wave * C::B_PER + i) * C::RPP + lane / C::PPR;
        const int piece = (lane % C::PPR) ^ ((row >> C::SWSH) & (C::PPR - 1));
        b_src[i] = wq + (long)row * 128 + piece * 16;       // global weight rows are 128 B (32 k) whatever the stage depth
    }
    const unsigned char* zsrc = zeros + (lane % C::PPR) * 16;
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
        int q = kt + rot;
        q = q >= nkt ? q - nkt : q;
        q = (q & (taps - 1)) * (nkt >> tshift) + (q >> tshift);            // tap-fastest walk (gemm_tile.h, tshift)
        const int tap = (q * C::BKE) >> kCLog2;
        const int k0 = q * C::BKE;
        const long koff = (long)k0 * 4;                                    // bytes into an A row
        const long boff = (long)(k0 >> 5) * (C::BN * 128) + (k0 & 31) * 4; // weight: [k / 32][256 rows][128 B]
        unsigned char* as = smem + stage * C::STAGE + (wave * C::A_PER) * 1024;
        unsigned char* bs = smem + stage * C::STAGE + C::A_BYTES + (wave * C::B_PER) * 1024;
#pragma unroll
        for (int i = 0; i < C::A_PER; ++i) {
            const bool ok = (unsigned)(a_tau0[i] + tap) < (unsigned)am.Lin;
            dma16_to_lds(ok ? a_src[i] + koff : zsrc, as + i * 1024);
        }
#pragma unroll
        for (int i = 0; i < C::B_PER; ++i) dma16_to_lds(b_src[i] + boff, bs + i * 1024);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int sw = ((lane & 31) >> C::SWSH) & (C::PPR - 1), kg = lane >> 5;
    const int a_row0 = (wm * 64 + (lane & 31)) * C::ROWB, b_row0 = (wn * 128 + (lane & 31)) * C::ROWB;
    using SP = SplitPlanes<2>;

    // NST - 1 stages ahead: at the top of iteration kt the stages kt .. kt + NST - 2 have been issued; stage kt must have
    // landed (vmcnt leaves the NST - 2 younger ones in flight), the barrier publishes it to the other waves and retires
    // everybody's reads of stage kt - 1, whose buffer the DMA of stage kt + NST - 1 then overwrites.
#pragma unroll
    for (int j = 0; j < C::NST - 1; ++j)
        if (j < nkt) issue(j, j);
    int stage = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const int younger = min(C::NST - 2, nkt - 1 - kt);      // issued after stage kt and allowed to be in flight (uniform)
        if (younger >= 2) { CPC_WAIT_VMCNT(2 * C::NPS); }
        else if (younger == 1) { CPC_WAIT_VMCNT(C::NPS); }
        else { CPC_WAIT_VMCNT(0); }
        __builtin_amdgcn_s_barrier();
        if (kt + C::NST - 1 < nkt) {
            int st = stage + C::NST - 1;
            st = st >= C::NST ? st - C::NST : st;
            issue(kt + C::NST - 1, st);
        }
        const unsigned char* As = smem + stage * C::STAGE;
        const unsigned char* Bs = As + C::A_BYTES;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            s16x8 af[TM][2], bf[TN][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const int off = ((4 * ks + 2 * kg + pl) ^ sw) * 16;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    af[tm][pl] = *reinterpret_cast<const s16x8*>(As + a_row0 + tm * 32 * C::ROWB + off);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    bf[tn][pl] = *reinterpret_cast<const s16x8*>(Bs + b_row0 + tn * 32 * C::ROWB + off);
            }
#pragma unroll
            for (int q = 0; q < SP::NPROD; ++q)                     // small terms first (l*h, h*l, h*h)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = SP::mfma(af[tm][SP::pa(q)], bf[tn][SP::pb(q)], acc[tm][tn]);
        }
        stage = stage + 1 == C::NST ? 0 : stage + 1;
    }
    __syncthreads();                            // the stage buffers are free: reused for the row statistics below

    // ---- epilogue: undo the operand scales, bias, ChannelNorm (two passes over the accumulators), ReLU
    const float sa = scale_for_amax(*x_amax), sb = scale_for_amax(*w_amax);
    const float inv = 1.0f / (sa * sb);                             // powers of two: exact
    float (*red)[2] = reinterpret_cast<float (*)[2]>(smem);         // [BM][2]: one partial per column half (wave wn)
    int col[TN];
    float gw[TN], gb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        col[tn] = dma_c_col<BM>(tn);
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
            if ((lane & 31) == 0) red[dma_c_row<BM>(tm, r)][wn] = v;
        }
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = dma_c_row<BM>(tm, r);
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
            if ((lane & 31) == 0) red[dma_c_row<BM>(tm, r)][wn] = v;
        }
    __syncthreads();
    const float sy = y_h2 ? scale_for_amax(*y_amax) : 1.0f;
    const bool odd = lane & 1;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = dma_c_row<BM>(tm, r);
            const float var = (red[row][0] + red[row][1]) * (1.0f / (kC - 1));
            const float rs = 1.0f / sqrtf(var + kNormEps);
            const int m = m0 + row;
            const bool live = m < am.M;                              // uniform over each half-wave (one row)
            if (live && wn == 0 && (lane & 31) == 0) rstd_out[m] = rs;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const float xh = (acc[tm][tn][r] - mean[tm][r]) * rs;
                const float yv = fmaxf(fmaf(xh, gw[tn], gb[tn]), 0.f);
                if (live) __builtin_nontemporal_store(xh, xhat + (long)m * kC + col[tn]);   // read again only in backward
                if (!y_h2) {
                    if (live) y[(long)m * kC + col[tn]] = yv;
                } else {
                    // H2: neighbouring lanes hold neighbouring channels; the even lane stores the pair's h pieces, the
                    // odd lane the l pieces (one dword store per lane, as for fp32).  The exchange is unconditional.
                    _Float16 h, l;
                    h2_split(yv, sy, h, l);
                    const unsigned mine_h = __builtin_bit_cast(unsigned short, h), mine_l = __builtin_bit_cast(unsigned short, l);
                    const unsigned got = __builtin_bit_cast(unsigned, dpp_mov<0xB1>(__builtin_bit_cast(float, odd ? mine_h : mine_l)));
                    const unsigned word = odd ? (got | (mine_l << 16)) : (mine_h | (got << 16));
                    if (live) {
                        const int c0 = col[tn] & ~1;
                        unsigned char* dst = reinterpret_cast<unsigned char*>(y + (long)m * kC) + h2_byte_of(c0) + (odd ? 16 : 0);
                        *reinterpret_cast<unsigned*>(dst) = word;
                    }
                }
            }
        }
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
static int g_dma_pipe = 1;    // 1: two 32-k stages (default); 0: four 16-k stages, three in flight -- measured slower on layer 1 at
                              // B = 64 (0.223-0.244 vs 0.196-0.206 ms): 64-byte row segments and twice the barriers cost more
                              // than the deeper prefetch buys

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
#define CPC_LAUNCH_DMA(BM_, BKE_, NST_)                                                                                       \
    hipLaunchKernelGGL((conv_fwd_dma_kernel<BM_, BKE_, NST_>), dim3(cdiv(am.M, BM_)), dim3(DmaCfg<BM_, BKE_, NST_>::NTHREADS), 0, \
                       st, am, wqb, K, bias, nw, nb, y, y_h2, xhat, rstd, x_amax, w_amax, y_amax, zb, g_dma_rot)
    if (bm == 256 && g_dma_pipe == 0) CPC_LAUNCH_DMA(256, 16, 4);
    else if (bm == 256) CPC_LAUNCH_DMA(256, 32, 2);
    else if (g_dma_pipe == 0) CPC_LAUNCH_DMA(128, 16, 4);
    else CPC_LAUNCH_DMA(128, 32, 2);
#undef CPC_LAUNCH_DMA
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

extern "C" int cpc_set_dma_rotation(int step) {
    CPC_RETURN_IF(step < 0, CPC_ERR_ARG);
    g_dma_rot = step;
    return 0;
}
extern "C" int cpc_set_dma_pipeline(int variant) {
    CPC_RETURN_IF(variant != 0 && variant != 1, CPC_ERR_ARG);
    g_dma_pipe = variant;
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
