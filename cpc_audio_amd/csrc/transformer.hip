// Transformer layer of the reference (cpc/transformers.py), forward + backward, d_model 256, 8 heads of 32,
// d_ff 2048, sequence length S <= 128 (128 as the auto-regressive network, --arMode transformer; 116 as a
// prediction network, --rnnMode transformer).  BASELINE.json config 4.  Forward only (inference, no dropout) also for
// 128 < S <= 512: a layer built for a longer window, e.g. the 400 frames of a 64000-sample feature-extraction chunk.
//
//   q,k,v = x Wq^T, x Wk^T, x Wv^T                         cpc/transformers.py:60-63, 82-84 (bias-free)
//   score[i,j] = (q_i.k_j + q_i.P[:, S-1-(i-j)]) / sqrt(32), j <= i      :37-48 (relative positions through the
//                                                          "z trick": P = Krelpos (32,S) indexed by the distance)
//   o = softmax(score) v;  y = LN(x + o Wo^T);  out = LN(y + lin2(relu(lin1(y))))      :49, :85, :97-100, :109-110
// Dropout (0.1, hard-coded in the reference) is not applied: parity is defined for eval / dropout 0.
//
// Structure: the seven projections are the library's NT/TN GEMMs (split-bf16 or f32 MFMA, gemm.hip); this
// file adds the per-head attention kernels (one workgroup per (sequence, head): Q, K, V, P staged in LDS, all
// products on v_mfma_f32_32x32x2_f32, softmax on the accumulator registers, causal tiles skipped), the fused
// residual + LayerNorm kernels and ReLU.  The relative-position term is one more small GEMM E = Q.P whose
// result is read back skewed (E[i][S-1-i+j]); backward un-skews dScore on the fly as an MFMA operand.
#include "cpc_common.h"
#include "cpc_internal.h"
#include "philox.h"

namespace cpc {

constexpr int kTH = 8;             // heads
constexpr int kDk = 32;            // head width
constexpr int kDff = 2048;
constexpr int kSmax = 128;         // padded sequence length of the attention tiles
constexpr int kLdH = kDk + 1;      // LDS row pitch of the (S, 32) operands
constexpr int kLdS = kSmax + 1;    // LDS row pitch of (., S) operands
constexpr float kLnEps = 1e-5f;    // nn.LayerNorm default, transformers.py:107-108

__device__ __forceinline__ int c_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Reductions over the 32 lanes sharing lane >> 5 (one row of a 32 x 32 accumulator tile): four DPP steps inside each row of 16
// lanes, the two rows of a half joined through v_readlane -- a butterfly of five ds_bpermute per reduction, two reductions per
// softmax row, was a dependent chain of LDS round trips.
__device__ __forceinline__ float half_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));       // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));       // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));      // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));      // row_mirror
    const float lo = fmaxf(lane_value(v, 0), lane_value(v, 16)), hi = fmaxf(lane_value(v, 32), lane_value(v, 48));
    return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ float half_sum(float v) { return half_wave_sum(v); }

// Stage the (S, 32) head slice `col0` of a (B*S, ld) matrix into LDS rows of pitch kLdH, zero-padded to 128 rows; 256 threads,
// four 16-byte pieces each.  Two halves so that a kernel can request all its slices before it waits for the first: the
// loads are unconditional on a clamped row (a predicated load is a branch and a full vmcnt(0) each -- as a loop of
// load -> store per piece, staging four slices was 28 round trips to L2 one after the other, half of the kernel's time).
template <int NT> struct HeadPieces { float4 v[1024 / NT]; };     // NT threads per workgroup
template <int NT>
__device__ __forceinline__ HeadPieces<NT> load_head(const float* __restrict__ src, long row0, int ld, int col0, int S) {
    HeadPieces<NT> p;
#pragma unroll
    for (int u = 0; u < 1024 / NT; ++u) {
        const int e = threadIdx.x + NT * u, i = e >> 3, c4 = (e & 7) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + (row0 + min(i, S - 1)) * ld + col0 + c4);
        p.v[u] = i < S ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return p;
}
template <int NT>
__device__ __forceinline__ void store_head(float* dst, const HeadPieces<NT>& p) {
#pragma unroll
    for (int u = 0; u < 1024 / NT; ++u) {
        const int e = threadIdx.x + NT * u, i = e >> 3, c4 = (e & 7) * 4;
        float* d = dst + i * kLdH + c4;
        d[0] = p.v[u].x; d[1] = p.v[u].y; d[2] = p.v[u].z; d[3] = p.v[u].w;
    }
}
// Krelpos (32, S) -> LDS rows of pitch kLdS, zero beyond S (or everywhere without relative positions)
template <int NT>
__device__ __forceinline__ void stage_relpos(float* dst, const float* __restrict__ P, int S) {
    if (P != nullptr && (S & 3) == 0) {                       // block-uniform
        float4 v[1024 / NT];
#pragma unroll
        for (int u = 0; u < 1024 / NT; ++u) {
            const int e = threadIdx.x + NT * u, d = e >> 5, c4 = (e & 31) * 4;
            v[u] = *reinterpret_cast<const float4*>(P + d * S + min(c4, S - 4));
            if (c4 >= S) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 1024 / NT; ++u) {
            const int e = threadIdx.x + NT * u, d = e >> 5, c4 = (e & 31) * 4;
            float* q = dst + d * kLdS + c4;
            q[0] = v[u].x; q[1] = v[u].y; q[2] = v[u].z; q[3] = v[u].w;
        }
        return;
    }
    for (int e = threadIdx.x; e < kDk * kSmax; e += NT) {
        const int d = e >> 7, c = e & (kSmax - 1);
        dst[d * kLdS + c] = (P != nullptr && c < S) ? P[d * S + c] : 0.f;
    }
}

// G layers of one shape run in lock-step, one launch per kernel for all of them (blockIdx.y / .z = layer): the K transformer
// predictors of the criterion (cpc/criterion/criterion.py:82-88) are K such layers on the same input.  Layer g works at the
// given pointers + g * stride (floats): `saved` / `scratch` are G workspaces of tf_layout's sizes back to back, parameters and
// their gradients are stacked per kind (par[i]: stride of params[i] and grads[i]; 0 when G == 1).
struct TfStrides {
    long saved = 0, scratch = 0;
    long par[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

// Operand bounds for the GEMMs (GemmBounds, cpc_internal.h: with max|A| and max|B| known a product runs on the fp16 pipe with
// two-piece operands, 3 MFMAs instead of the 6 of three bf16 pieces).  Every tensor a GEMM of the layer reads is written by one
// of the kernels below, which leaves max|.| of what it wrote in kAmaxSlots slots (zeroed by the host before; integer atomicMax
// on the bits of non-negative floats; slot by workgroup, so that no address takes more than a few hundred atomics).  The layers
// Every layer of a group keeps its own bounds (in its own copy of the workspace: `amax` arrives with the layer's offset), so a
// layer computes the same bits inside a group as alone.
__device__ __forceinline__ void publish_amax(float* __restrict__ amax, float m) {        // every lane of the wave must call
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && amax != nullptr)
        atomicMax(reinterpret_cast<unsigned*>(amax) + blockIdx.x % (unsigned)kAmaxSlots, __float_as_uint(m));
}
// one atomic per workgroup of 256 threads (atomics of different XCDs on one address take ~1 us each: a kernel must not issue
// more than a few dozen per slot); every thread of the workgroup must call
__device__ __forceinline__ void publish_amax_block(float* __restrict__ amax, float m) {
    __shared__ float wmax[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0 && amax != nullptr)
        atomicMax(reinterpret_cast<unsigned*>(amax) + blockIdx.x % (unsigned)kAmaxSlots,
                  __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}
__device__ __forceinline__ float amax4(float m, const float4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// Layer blockIdx.x's bound bookkeeping at the start of a forward (nz = 0: backward) pass: out = max(a, b, c) slot by slot (the
// stacked [Wq; Wk; Wv] of the backward), and the nz slots from `zero` on cleared for the kernels' atomicMax.
__global__ __launch_bounds__(kAmaxSlots) void bounds_begin_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                                  const float* __restrict__ b, const float* __restrict__ c,
                                                                  float* __restrict__ zero, int nz, long gs) {
    const long g = (long)blockIdx.x * gs;
    if (out != nullptr) out[g + threadIdx.x] = fmaxf(a[g + threadIdx.x], fmaxf(b[g + threadIdx.x], c[g + threadIdx.x]));
    for (int i = threadIdx.x; i < nz; i += kAmaxSlots) zero[g + i] = 0.f;
}

// ------------------------------------------------------------------ dropout (cpc/transformers.py:18,50 and :93,100)
// The reference applies nn.Dropout(0.1) to the attention probabilities and to the feed-forward hidden layer in training
// mode.  Here the keep decision of an element is a pure function of (seed, site, element index) -- Philox4x32-10, the
// counter-based generator torch uses on GPUs -- so the backward regenerates it instead of reading a saved mask, and a test
// can ask for exactly the mask a layer call used (cpc_dropout_keep_mask).
//   site 0: attention probability (b*8 + head, i, j) at flat index ((b*8 + head)*S + i)*S + j
//   site 1: hidden activation (row, col): block (row >> 2)*2048 + col, word row & 3 (philox.h)
// Site 0 (attention probabilities): the four words of one Philox block belong to the four rows 4u .. 4u+3 of one column --
// block index ((b*8 + head) * ceil(S/4) + (i >> 2)) * S + j, word i & 3 -- because that is what one lane of an MFMA
// accumulator holds (c_row: registers 4m .. 4m+3 are four consecutive rows of one column): a lane draws 16 blocks for its 64
// probabilities instead of 64 (a block is ~120 integer instructions; drawn one per element they were most of the attention
// kernels' time in training mode).
__device__ __forceinline__ unsigned long long attn_drop_block(int bh, int S, int i, int j) {
    return ((unsigned long long)bh * ((S + 3) >> 2) + (i >> 2)) * S + j;
}
// keep bits of the 16 accumulator rows of one lane in tile column j: bit r <-> row 32 w + c_row(r, lane)
__device__ __forceinline__ unsigned attn_keep_bits(unsigned long long seed, int bh, int S, int w, int lane, int j, unsigned th) {
    unsigned bits = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const Philox4 q = philox4x32_10(seed, 0u, attn_drop_block(bh, S, 32 * w + 8 * m + 4 * (lane >> 5), j));
        bits |= ((q.x >= th ? 1u : 0u) | (q.y >= th ? 2u : 0u) | (q.z >= th ? 4u : 0u) | (q.w >= th ? 8u : 0u)) << (4 * m);
    }
    return bits;
}

// ------------------------------------------------------------------ attention forward
// grid = B * 8, 256 threads; wave w owns query rows 32w .. 32w+31.
// qkv (B*S, 768) = [q | k | v]; P = Krelpos (32, S) or NULL; o (B*S, 256); A (B*8, S, S) saved for backward.
// drop_p > 0 (training): the probabilities that multiply V are A * keep / (1 - p); A itself (pre-dropout) is what is saved.
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                       float* __restrict__ o, float* __restrict__ A, int S, float drop_p,
                                                       unsigned long long seed, TfStrides gs, float* __restrict__ o_amax) {
    __shared__ float lds[3 * kSmax * kLdH + kDk * kLdS + 4 * 32 * kLdS];     // 133 KB of the CU's 160 KB
    {                                                 // layer blockIdx.y of a group (TfStrides); its dropout stream: seed + layer
        const long g = blockIdx.y;
        qkv += g * gs.saved; o += g * gs.saved; A += g * gs.saved;
        if (o_amax != nullptr) o_amax += g * gs.saved;
        if (P != nullptr) P += g * gs.par[4];
        seed += (unsigned long long)g;
    }
    float* Qs = lds;
    float* Ks = Qs + kSmax * kLdH;
    float* Vs = Ks + kSmax * kLdH;
    float* Ps = Vs + kSmax * kLdH;
    float* Ws = Ps + kDk * kLdS;                     // [4 waves][32][kLdS]: E = Q.P, then the probabilities
    const int bh = blockIdx.x, b = bh / kTH, h = bh % kTH;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const long row0 = (long)b * S;
    {
        const HeadPieces<256> pq = load_head<256>(qkv, row0, 3 * kC, h * kDk, S), pk = load_head<256>(qkv, row0, 3 * kC, kC + h * kDk, S),
                              pv = load_head<256>(qkv, row0, 3 * kC, 2 * kC + h * kDk, S);
        stage_relpos<256>(Ps, P, S);
        store_head(Qs, pq); store_head(Ks, pk); store_head(Vs, pv);
    }
    __syncthreads();
    if (32 * w >= S) return;                          // wave-uniform: no query rows here (no barrier follows)

    float* Ww = Ws + w * 32 * kLdS;
    const float* qrow = Qs + (32 * w + l31) * kLdH;
    if (P != nullptr) {                               // E[i][c] = q_i . P[:, c]
#pragma unroll 1
        for (int ct = 0; ct < 4; ++ct) {
            f32x16 e;
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < kDk / 2; ++kk) {
                const int k = 2 * kk + khalf;
                e = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[k], Ps[k * kLdS + ct * 32 + l31], e, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) Ww[c_row(r, lane) * kLdS + ct * 32 + l31] = e[r];
        }
    }
    __builtin_amdgcn_wave_barrier();

    const float scale = 0.17677669529663687f;         // 1 / sqrt(32)
    f32x16 sc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[ct][r] = -INFINITY;
        if (ct <= w) {                                // tiles right of the diagonal are entirely in the future
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* krow = Ks + (ct * 32 + l31) * kLdH;
#pragma unroll
            for (int kk = 0; kk < kDk / 2; ++kk) {
                const int k = 2 * kk + khalf;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[k], krow[k], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = c_row(r, lane), i = 32 * w + il, j = ct * 32 + l31;
                if (j <= i && i < S) {
                    const float rel = P != nullptr ? Ww[il * kLdS + (S - 1 - i + j)] : 0.f;
                    sc[ct][r] = (acc[r] + rel) * scale;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();                  // every lane has read E; the buffer now takes the probabilities

    unsigned keep[4] = {0u, 0u, 0u, 0u};
    if (drop_p > 0.f) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            if (ct <= w) keep[ct] = attn_keep_bits(seed, bh, S, w, lane, ct * 32 + l31, drop_threshold(drop_p));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = c_row(r, lane), i = 32 * w + il;
        float m = fmaxf(fmaxf(sc[0][r], sc[1][r]), fmaxf(sc[2][r], sc[3][r]));
        m = half_max(m);
        float p[4], s = 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            p[ct] = (i < S && sc[ct][r] > -INFINITY) ? expf(sc[ct][r] - m) : 0.f;
            s += p[ct];
        }
        s = half_sum(s);
        const float inv = i < S ? 1.0f / s : 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const float a = p[ct] * inv;
            const int j = ct * 32 + l31;
            float kept = a;
            if (drop_p > 0.f && i < S && j <= i) kept = ((keep[ct] >> r) & 1u) ? a * (1.0f / (1.0f - drop_p)) : 0.f;
            Ww[il * kLdS + j] = kept;
            if (i < S && j < S) A[((long)bh * S + i) * S + j] = a;
        }
    }
    __builtin_amdgcn_wave_barrier();

    f32x16 ov;                                        // o_w = A_w (32 x 32(w+1)) . V
#pragma unroll
    for (int r = 0; r < 16; ++r) ov[r] = 0.f;
    const float* arow = Ww + l31 * kLdS;
    for (int kk = 0; kk < 16 * (w + 1); ++kk) {
        const int k = 2 * kk + khalf;
        ov = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[k], Vs[k * kLdH + l31], ov, 0, 0, 0);
    }
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = 32 * w + c_row(r, lane);
        if (i < S) {
            o[(row0 + i) * kC + h * kDk + l31] = ov[r];
            m = fmaxf(m, fabsf(ov[r]));
        }
    }
    publish_amax(o_amax, m);
}

constexpr int kSlong = 512;
constexpr int kLdE = 64 + 1;       // per-wave 32 x 64 work tile: E, then the tile's probabilities
// ------------------------------------------------------------------ attention forward, two workgroups per CU (round 6)
// attn_fwd_kernel keeps Q, K, V, Krelpos and a 32 x 128 work tile per wave in LDS: 133 KB, ONE workgroup of four waves per CU --
// one wave per SIMD, which cannot issue back to back, and nothing to run while it waits for its loads or streams the
// probabilities out (per workgroup ~28 us for ~5 us of MFMAs, rocprofv3 at B = 64).  This form needs 67 KB: Q lives in
// registers (a lane's 16 operand values of its own row), Krelpos is read from L2 as the MFMA's B operand (16 KB per layer), and
// the relative-position term E = Q . P is formed per 32 x 32 score tile from the 63 distance columns that tile can see
// (attn_fwd_long_kernel's scheme) in a 32 x 64 work tile per wave, which then takes the tile's probabilities for the product
// with V.  Two workgroups share a CU: one's loads, softmax, Philox draws and probability stores run in the other's MFMA time.
// Same operands into the same MFMA chains in the same order as attn_fwd_kernel (E per column, the scores, the product with V
// tile by tile): BIT-IDENTICAL o and A (tests/test_gpu_transformer.py); 96 more MFMAs for the last wave (E twice per tile).
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                           float* __restrict__ o, float* __restrict__ A, int S, float drop_p,
                                                           unsigned long long seed, TfStrides gs, float* __restrict__ o_amax) {
    __shared__ float lds[2 * kSmax * kLdH + 4 * 32 * kLdE];            // 67 KB
    {
        const long g = blockIdx.y;
        qkv += g * gs.saved; o += g * gs.saved; A += g * gs.saved;
        if (o_amax != nullptr) o_amax += g * gs.saved;
        if (P != nullptr) P += g * gs.par[4];
        seed += (unsigned long long)g;
    }
    float* Ks = lds;
    float* Vs = Ks + kSmax * kLdH;
    const int bh = blockIdx.x, b = bh / kTH, h = bh % kTH;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const long row0 = (long)b * S;
    float* Ww = Vs + kSmax * kLdH + w * 32 * kLdE;
    const int i0 = 32 * w;
    // this lane's operand values of query row i0 + l31: q[2 kk + khalf], kk = 0..15 (the row is 128 contiguous bytes; rows past
    // S: the last row, never used unmasked)
    float qreg[16];
    {
        const float* qp = qkv + (row0 + min(i0 + l31, S - 1)) * (3 * kC) + h * kDk;
        float4 qv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) qv[u] = *reinterpret_cast<const float4*>(qp + 4 * u);
        const HeadPieces<256> pk = load_head<256>(qkv, row0, 3 * kC, kC + h * kDk, S), pv = load_head<256>(qkv, row0, 3 * kC, 2 * kC + h * kDk, S);
        store_head(Ks, pk); store_head(Vs, pv);
#pragma unroll
        for (int u = 0; u < 8; ++u) {                 // elements 4u .. 4u+3: k = 4u + {0,1,2,3} -> kk = 2u, 2u + 1 for either parity
            qreg[2 * u] = khalf ? qv[u].y : qv[u].x;
            qreg[2 * u + 1] = khalf ? qv[u].w : qv[u].z;
        }
    }
    __syncthreads();
    if (i0 >= S) return;                              // wave-uniform: no query rows here (no barrier follows)

    const float scale = 0.17677669529663687f;         // 1 / sqrt(32)
    f32x16 sc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[ct][r] = -INFINITY;
        if (ct <= w) {                                // tiles right of the diagonal are entirely in the future
            const int j0 = 32 * ct;
            if (P != nullptr) {
                // E[il][cc] = q_(i0 + il) . P[:, cbase + cc]; score (il, jl) needs distance column S - 1 - i + j = cbase + 31 - il + jl
                const int cbase = S - 1 - (i0 + 31) + j0;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int cc = min(max(cbase + 32 * u + l31, 0), S - 1);      // (clamped columns belong to masked entries)
                    float pcol[16];
#pragma unroll
                    for (int kk = 0; kk < kDk / 2; ++kk) pcol[kk] = P[(long)(2 * kk + khalf) * S + cc];
                    f32x16 e;
#pragma unroll
                    for (int r = 0; r < 16; ++r) e[r] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < kDk / 2; ++kk) e = __builtin_amdgcn_mfma_f32_32x32x2f32(qreg[kk], pcol[kk], e, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) Ww[c_row(r, lane) * kLdE + 32 * u + l31] = e[r];
                }
                __builtin_amdgcn_wave_barrier();
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* krow = Ks + (j0 + l31) * kLdH;
#pragma unroll
            for (int kk = 0; kk < kDk / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qreg[kk], krow[2 * kk + khalf], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = c_row(r, lane), i = i0 + il, j = j0 + l31;
                if (j <= i && i < S) {
                    const float rel = P != nullptr ? Ww[il * kLdE + 31 - il + l31] : 0.f;
                    sc[ct][r] = (acc[r] + rel) * scale;
                }
            }
            __builtin_amdgcn_wave_barrier();          // every lane has read E: the tile is free for the next one
        }
    }

    unsigned keep[4] = {0u, 0u, 0u, 0u};
    if (drop_p > 0.f) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            if (ct <= w) keep[ct] = attn_keep_bits(seed, bh, S, w, lane, ct * 32 + l31, drop_threshold(drop_p));
    }
    // softmax over the row (registers), A out; the kept probabilities stay in sc for the product with V
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = c_row(r, lane), i = i0 + il;
        float m = fmaxf(fmaxf(sc[0][r], sc[1][r]), fmaxf(sc[2][r], sc[3][r]));
        m = half_max(m);
        float p[4], sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            p[ct] = (i < S && sc[ct][r] > -INFINITY) ? expf(sc[ct][r] - m) : 0.f;
            sum += p[ct];
        }
        sum = half_sum(sum);
        const float inv = i < S ? 1.0f / sum : 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const float a = p[ct] * inv;
            const int j = ct * 32 + l31;
            float kept = a;
            if (drop_p > 0.f && i < S && j <= i) kept = ((keep[ct] >> r) & 1u) ? a * (1.0f / (1.0f - drop_p)) : 0.f;
            sc[ct][r] = kept;
            if (i < S && j < S) A[((long)bh * S + i) * S + j] = a;
        }
    }

    f32x16 ov;                                        // o_w = A_w (32 x 32(w+1)) . V, key tile by key tile through the work tile
#pragma unroll
    for (int r = 0; r < 16; ++r) ov[r] = 0.f;
    const float* arow = Ww + l31 * kLdE;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        if (ct <= w) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Ww[c_row(r, lane) * kLdE + l31] = sc[ct][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int k = 2 * kk + khalf;
                ov = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[k], Vs[(32 * ct + k) * kLdH + l31], ov, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();          // (the next tile's probabilities overwrite these)
        }
    }
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + c_row(r, lane);
        if (i < S) {
            o[(row0 + i) * kC + h * kDk + l31] = ov[r];
            m = fmaxf(m, fabsf(ov[r]));
        }
    }
    publish_amax(o_amax, m);
}

// ------------------------------------------------------------------ attention forward, 128 < S <= 512 (inference)
// A layer built for a 64000-sample feature-extraction window (cpc/feature_loader.py:247-266 feeds 400 frames; cpc/transformers.py
// :22-49 at sizeSeq = 400) does not fit attn_fwd_kernel's one-tile-per-sequence LDS image.  Forward only, no dropout, no saved
// probabilities: grid = (B * 8, ceil(S / 128), G); a workgroup takes 128 query rows (wave w: rows 32 w ..) and walks the key blocks
// up to its own diagonal with a running softmax (row maximum, denominator and the output tile rescaled as the maximum moves).
// Per 32-key tile: scores Q.K^T (16 MFMAs), the relative-position term E = Q . P[:, c0 .. c0 + 63] for the 63 distances the tile
// can see (32 MFMAs, P's columns straight from L2 as the B operand -- Krelpos is 51 KB at S = 400), read back skewed as in the
// short kernel, then probabilities . V (16 MFMAs); every product on v_mfma_f32_32x32x2_f32.
__global__ __launch_bounds__(256) void attn_fwd_long_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                            float* __restrict__ o, int S, TfStrides gs, float* __restrict__ o_amax) {
    __shared__ float lds[3 * kSmax * kLdH + 4 * 32 * kLdE];           // 84 KB
    {
        const long g = blockIdx.z;
        qkv += g * gs.saved; o += g * gs.saved;
        if (o_amax != nullptr) o_amax += g * gs.saved;
        if (P != nullptr) P += g * gs.par[4];
    }
    float* Qs = lds;
    float* Ks = Qs + kSmax * kLdH;
    float* Vs = Ks + kSmax * kLdH;
    const int bh = blockIdx.x, b = bh / kTH, h = bh % kTH, qb = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const long row0 = (long)b * S;
    const int q0 = 128 * qb, i0 = q0 + 32 * w;        // this wave's first query row
    float* Ww = Vs + kSmax * kLdH + w * 32 * kLdE;
    store_head(Qs, load_head<256>(qkv, row0 + q0, 3 * kC, h * kDk, S - q0));
    const float* qrow = Qs + (32 * w + l31) * kLdH;
    const float scale = 0.17677669529663687f;         // 1 / sqrt(32)
    float mrow[16], lrow[16];
    f32x16 ov;
#pragma unroll
    for (int r = 0; r < 16; ++r) { mrow[r] = -INFINITY; lrow[r] = 0.f; ov[r] = 0.f; }
    for (int kb = 0; kb <= qb; ++kb) {
        const int k0 = 128 * kb;
        const HeadPieces<256> pk = load_head<256>(qkv, row0 + k0, 3 * kC, kC + h * kDk, S - k0),
                              pv = load_head<256>(qkv, row0 + k0, 3 * kC, 2 * kC + h * kDk, S - k0);
        __syncthreads();                              // every wave is done with the previous block (and Q is staged)
        store_head(Ks, pk); store_head(Vs, pv);
        __syncthreads();
        if (i0 >= S) continue;                        // wave-uniform: no query rows here (the barriers above are still met)
        for (int ct = 0; ct < 4; ++ct) {
            const int j0 = k0 + 32 * ct;
            if (j0 > i0 + 31 || j0 >= S) break;       // wave-uniform: entirely in the future / past the sequence
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* krow = Ks + (ct * 32 + l31) * kLdH;
#pragma unroll
            for (int kk = 0; kk < kDk / 2; ++kk) {
                const int k = 2 * kk + khalf;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[k], krow[k], acc, 0, 0, 0);
            }
            if (P != nullptr) {
                // E[il][cc] = q_(i0 + il) . P[:, cbase + cc]; score (il, jl) needs distance column S - 1 - i + j = cbase + 31 - il + jl
                const int cbase = S - 1 - (i0 + 31) + j0;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int cc = min(max(cbase + 32 * u + l31, 0), S - 1);      // (clamped columns belong to masked entries)
                    f32x16 e;
#pragma unroll
                    for (int r = 0; r < 16; ++r) e[r] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < kDk / 2; ++kk) {
                        const int k = 2 * kk + khalf;
                        e = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[k], P[(long)k * S + cc], e, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) Ww[c_row(r, lane) * kLdE + 32 * u + l31] = e[r];
                }
                __builtin_amdgcn_wave_barrier();
            }
            float sc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = c_row(r, lane), i = i0 + il, j = j0 + l31;
                const float rel = P != nullptr ? Ww[il * kLdE + 31 - il + l31] : 0.f;
                sc[r] = (j <= i && i < S) ? (acc[r] + rel) * scale : -INFINITY;
            }
            __builtin_amdgcn_wave_barrier();          // every lane has read E: the tile now takes the probabilities
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = c_row(r, lane);
                const float mn = fmaxf(mrow[r], half_max(sc[r]));
                const float pr = sc[r] > -INFINITY ? expf(sc[r] - mn) : 0.f;
                const float alpha = mrow[r] > -INFINITY ? expf(mrow[r] - mn) : 1.0f;      // (nothing accumulated yet: lrow, ov are 0)
                lrow[r] = lrow[r] * alpha + half_sum(pr);
                ov[r] *= alpha;
                mrow[r] = mn;
                Ww[il * kLdE + l31] = pr;
            }
            __builtin_amdgcn_wave_barrier();
            const float* arow = Ww + l31 * kLdE;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int k = 2 * kk + khalf;
                ov = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[k], Vs[(32 * ct + k) * kLdH + l31], ov, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();          // (the next tile's E overwrites the probabilities)
        }
    }
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + c_row(r, lane);
        if (i < S) {
            const float v = ov[r] / lrow[r];          // (row i always sees key i: lrow > 0)
            o[(row0 + i) * kC + h * kDk + l31] = v;
            m = fmaxf(m, fabsf(v));
        }
    }
    publish_amax(o_amax, m);
}

// ------------------------------------------------------------------ attention backward
// dqkv (B*S, 768) = [dq | dk | dv];  dPpart (B*8, 32, S): per-workgroup partial of dKrelpos (reduced afterwards).
__global__ __launch_bounds__(512) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                       const float* __restrict__ o, const float* __restrict__ A,
                                                       const float* __restrict__ dO, float* __restrict__ dqkv,
                                                       float* __restrict__ dPpart, int S, float drop_p,
                                                       unsigned long long seed, TfStrides gs, float* __restrict__ dqkv_amax) {
    __shared__ float lds[4 * kSmax * kLdH + kDk * kLdS + kSmax * kLdS];      // 150 KB
    {
        const long g = blockIdx.y;
        qkv += g * gs.saved; o += g * gs.saved; A += g * gs.saved;
        dO += g * gs.scratch; dqkv += g * gs.scratch; dPpart += g * gs.scratch;
        if (dqkv_amax != nullptr) dqkv_amax += g * gs.scratch;
        if (P != nullptr) P += g * gs.par[4];
        seed += (unsigned long long)g;
    }
    float* Qs = lds;
    float* Ks = Qs + kSmax * kLdH;
    float* Vs = Ks + kSmax * kLdH;
    float* Gs = Vs + kSmax * kLdH;                    // dO
    float* Ps = Gs + kSmax * kLdH;
    float* Ds = Ps + kDk * kLdS;                      // dScore [128][kLdS]
    __shared__ float rdot[kSmax];
    const int bh = blockIdx.x, b = bh / kTH, h = bh % kTH;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;          // eight waves: tasks are dealt out below
    const int l31 = lane & 31, khalf = lane >> 5;
    const long row0 = (long)b * S;
    float gmax = 0.f;                                 // max|dq|, |dk|, |dv| of what this thread stores
    {
        const HeadPieces<512> pq = load_head<512>(qkv, row0, 3 * kC, h * kDk, S), pk = load_head<512>(qkv, row0, 3 * kC, kC + h * kDk, S),
                              pv = load_head<512>(qkv, row0, 3 * kC, 2 * kC + h * kDk, S), pg = load_head<512>(dO, row0, kC, h * kDk, S);
        stage_relpos<512>(Ps, P, S);
        store_head(Qs, pq); store_head(Ks, pk); store_head(Vs, pv); store_head(Gs, pg);
    }
    if (threadIdx.x < kSmax) {                        // rowdot_i = dO_i . o_i = sum_j dA_ij A_ij
        const int i = threadIdx.x;
        float s = 0.f;
        if (i < S) {
            const float* g = dO + (row0 + i) * kC + h * kDk;
            const float* ov = o + (row0 + i) * kC + h * kDk;
#pragma unroll
            for (int d = 0; d < kDk; ++d) s = fmaf(g[d], ov[d], s);
        }
        rdot[i] = s;
    }
    __syncthreads();

    // ---- phase 1: the ten 32 x 32 tiles (row block w, column block ct <= w) of dScore, dealt to the eight waves (tiles above
    // the diagonal are never read).  One workgroup per CU is all the LDS allows; with four waves -- one per SIMD, which cannot
    // issue back to back -- the kernel spent its time in instruction issue, so the same tasks now run on two waves per SIMD.
    const float scale = 0.17677669529663687f;
    const float* Abh = A + (long)bh * S * S;
#pragma unroll 1
    for (int q = wv; q < 10; q += 8) {
        const int w = q >= 6 ? 3 : (q >= 3 ? 2 : (q >= 1 ? 1 : 0)), ct = q - w * (w + 1) / 2;
        const float* grow = Gs + (32 * w + l31) * kLdH;
        f32x16 acc;
        float ap[16];                                 // this tile's probabilities: requested (unconditionally, on clamped
#pragma unroll                                        // indices) before the products they will meet
        for (int r = 0; r < 16; ++r) {
            acc[r] = 0.f;
            ap[r] = Abh[(long)min(32 * w + c_row(r, lane), S - 1) * S + min(ct * 32 + l31, S - 1)];
        }
        unsigned keep = 0u;
        if (drop_p > 0.f) keep = attn_keep_bits(seed, bh, S, w, lane, ct * 32 + l31, drop_threshold(drop_p));
        const float* vrow = Vs + (ct * 32 + l31) * kLdH;
#pragma unroll
        for (int kk = 0; kk < kDk / 2; ++kk) {
            const int k = 2 * kk + khalf;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(grow[k], vrow[k], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * w + c_row(r, lane), j = ct * 32 + l31;
            float ds = 0.f;
            if (i < S && j <= i) {
                // through the dropout: dA_ij = (dO_i . v_j) * keep_ij / (1 - p); rowdot_i = dO_i . o_i holds as it is
                float da = acc[r];
                if (drop_p > 0.f) da = ((keep >> r) & 1u) ? da * (1.0f / (1.0f - drop_p)) : 0.f;
                ds = ap[r] * (da - rdot[i]) * scale;
            }
            Ds[i * kLdS + j] = ds;
        }
    }
    __syncthreads();

    // ---- phase 2: three kinds of task per 32-row / 32-column block w
    auto dq_task = [&](int w) __attribute__((always_inline)) {
        // dq_i = sum_j dS_ij k_j + sum_c dE_ic P[:, c],  dE_ic = dS[i][c - (S-1) + i]     (query rows of block w)
        const int jlim = 32 * (w + 1);                // causal: these queries see keys < jlim
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int il = 32 * w + l31;
        const float* drow = Ds + il * kLdS;
        for (int kk = 0; kk < jlim / 2; ++kk) {
            const int j = 2 * kk + khalf;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(drow[j], Ks[j * kLdH + l31], acc, 0, 0, 0);
        }
        if (P != nullptr) {
            for (int kk = 0; kk < kSmax / 2; ++kk) {
                const int c = 2 * kk + khalf;
                const int j = c - (S - 1) + il;
                const float de = (j >= 0 && j <= il && il < S) ? drow[j] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(de, Ps[l31 * kLdS + c], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * w + c_row(r, lane);
            if (i < S) {
                dqkv[(row0 + i) * 3 * kC + h * kDk + l31] = acc[r];
                gmax = fmaxf(gmax, fabsf(acc[r]));
            }
        }
    };
    auto dkdv_task = [&](int w) __attribute__((always_inline)) {
        // dk_j = sum_{i >= j} dS_ij q_i;  dv_j = sum_{i >= j} A_ij dO_i   (key rows j of block w)
        f32x16 ak, av;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ak[r] = 0.f; av[r] = 0.f; }
        const int jl = 32 * w + l31;
        const float* acol = Abh + min(jl, S - 1);
        for (int kb = 16 * w; kb < kSmax / 2; kb += 8) {      // eight probabilities in flight per lane (they are L2 hits: the
            float a8[8];                                      // workgroup read them in phase 1)
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] = acol[(long)min(2 * (kb + u) + khalf, S - 1) * S];
            // rows i = 2 (kb + u) + khalf of column jl: four Philox blocks of four rows each, and the two lanes l31 / l31 + 32
            // need the same four (one the even words, one the odd): each draws two and they swap the 8 decisions
            unsigned keep = 0u;
            if (drop_p > 0.f) {
                const unsigned th = drop_threshold(drop_p);
                unsigned mine = 0u;                                 // bit 4 mm + word, blocks m = 2 khalf + mm
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const Philox4 q = philox4x32_10(seed, 0u, attn_drop_block(bh, S, 2 * kb + 4 * (2 * khalf + mm), jl));
                    mine |= ((q.x >= th ? 1u : 0u) | (q.y >= th ? 2u : 0u) | (q.z >= th ? 4u : 0u) | (q.w >= th ? 8u : 0u)) << (4 * mm);
                }
                const unsigned theirs = (unsigned)__shfl_xor((int)mine, 32);
                const unsigned all = khalf ? (theirs | (mine << 8)) : (mine | (theirs << 8));     // bit 4 m + word, m = 0..3
#pragma unroll
                for (int m = 0; m < 4; ++m)                         // bit u <-> row 2 (kb + u) + khalf: words khalf, 2 + khalf
                    keep |= (((all >> (4 * m + khalf)) & 1u) | (((all >> (4 * m + 2 + khalf)) & 1u) << 1)) << (2 * m);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = 2 * (kb + u) + khalf;
                ak = __builtin_amdgcn_mfma_f32_32x32x2f32(Ds[i * kLdS + jl], Qs[i * kLdH + l31], ak, 0, 0, 0);
                float a = (i < S && jl < S) ? a8[u] : 0.f;
                if (drop_p > 0.f && i < S && jl <= i)        // dv sees the probabilities that multiplied V
                    a = ((keep >> u) & 1u) ? a * (1.0f / (1.0f - drop_p)) : 0.f;
                av = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Gs[i * kLdH + l31], av, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * w + c_row(r, lane);
            if (j < S) {
                dqkv[(row0 + j) * 3 * kC + kC + h * kDk + l31] = ak[r];
                dqkv[(row0 + j) * 3 * kC + 2 * kC + h * kDk + l31] = av[r];
                gmax = fmaxf(gmax, fmaxf(fabsf(ak[r]), fabsf(av[r])));
            }
        }
    };
    auto dp_task = [&](int w) __attribute__((always_inline)) {
        // dP[d][c] partial = sum_i q_i[d] dE_ic, columns c of block w
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int c = 32 * w + l31;
        for (int kk = 0; kk < kSmax / 2; ++kk) {
            const int i = 2 * kk + khalf;
            const int j = c - (S - 1) + i;
            const float de = (j >= 0 && j <= i && i < S) ? Ds[i * kLdS + j] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[i * kLdH + l31], de, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = c_row(r, lane);
            if (c < S) dPpart[((long)bh * kDk + d) * S + c] = acc[r];
        }
    };
    // in MFMAs (relative positions): dq 16 (w + 1) + 64, dk/dv 32 (4 - w), dP 64 -- 96 to 144 per wave
    const bool rel = P != nullptr;                    // block-uniform
    switch (wv) {
        case 0: dq_task(3); break;
        case 1: dkdv_task(0); break;
        case 2: dq_task(2); break;
        case 3: dq_task(1); dkdv_task(3); break;
        case 4: dkdv_task(1); break;
        case 5: dq_task(0); if (rel) dp_task(0); break;
        case 6: dkdv_task(2); if (rel) dp_task(1); break;
        default: if (rel) { dp_task(2); dp_task(3); } break;
    }
    publish_amax(dqkv_amax, gmax);
}

// ------------------------------------------------------------------ residual + LayerNorm
// out = LN(a + b) * w + bias, one wavefront per 256-wide row, kLnFwdRows rows per workgroup; xhat and rstd are kept for backward.
constexpr int kLnFwdRows = 16;
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, float* __restrict__ xhat,
                                                         float* __restrict__ rstd, int M, long a_gs, long b_gs, long w_gs,
                                                         long out_gs, int out_ld, long xh_gs, float* __restrict__ out_amax) {
    const int lane = threadIdx.x & 63;
    {                                                 // layer blockIdx.y of a group: rows of `out` are out_ld floats apart
        const long g = blockIdx.y;
        a += g * a_gs; b += g * b_gs; w += g * w_gs; bias += g * w_gs; out += g * out_gs; xhat += g * xh_gs; rstd += g * xh_gs;
        if (out_amax != nullptr) out_amax += g * xh_gs;
    }
    const float4 vw = *reinterpret_cast<const float4*>(w + 4 * lane);
    const float4 vbi = *reinterpret_cast<const float4*>(bias + 4 * lane);
    float m = 0.f;
    for (int it = 0; it < kLnFwdRows / 4; ++it) {
        const long row = (long)blockIdx.x * kLnFwdRows + it * 4 + (threadIdx.x >> 6);
        if (row >= M) break;                          // wave-uniform
        const float4 va = *reinterpret_cast<const float4*>(a + row * kC + 4 * lane);
        const float4 vb = *reinterpret_cast<const float4*>(b + row * kC + 4 * lane);
        float x[4] = {va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w};
        const float mu = wave_sum((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / kC);
        float d[4], q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { d[e] = x[e] - mu; q = fmaf(d[e], d[e], q); }
        const float var = wave_sum(q) * (1.0f / kC);                   // biased, as nn.LayerNorm
        const float rs = 1.0f / sqrtf(var + kLnEps);
        float4 xh, y;
        xh.x = d[0] * rs; xh.y = d[1] * rs; xh.z = d[2] * rs; xh.w = d[3] * rs;
        y.x = xh.x * vw.x + vbi.x; y.y = xh.y * vw.y + vbi.y; y.z = xh.z * vw.z + vbi.z; y.w = xh.w * vw.w + vbi.w;
        *reinterpret_cast<float4*>(xhat + row * kC + 4 * lane) = xh;
        *reinterpret_cast<float4*>(out + row * out_ld + 4 * lane) = y;
        if (lane == 0) rstd[row] = rs;
        m = amax4(m, y);
    }
    publish_amax_block(out_amax, m);
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w  (gradient w.r.t. the SUM a + b);
// `add` (may be NULL) is added to dx.  Per-workgroup partials of dw = sum dy*xhat and db = sum dy go to
// part[block][512] (32 rows per workgroup), reduced in a fixed order by rows_sum.
constexpr int kLnRowsPerBlock = 32;
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                     const float* __restrict__ rstd, const float* __restrict__ w,
                                                     const float* __restrict__ add, float* __restrict__ dx,
                                                     float* __restrict__ part, int M, long dy_gs, int dy_ld, long xh_gs,
                                                     long w_gs, long dx_gs, long part_gs, float* __restrict__ dx_amax) {
    __shared__ float red[4][2][kC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    {                                                 // layer blockIdx.y of a group: rows of dy are dy_ld floats apart
        const long g = blockIdx.y;
        dy += g * dy_gs; xhat += g * xh_gs; rstd += g * xh_gs; w += g * w_gs; dx += g * dx_gs; part += g * part_gs;
        if (add != nullptr) add += g * dx_gs;
        if (dx_amax != nullptr) dx_amax += g * dx_gs;
    }
    const float4 vw = *reinterpret_cast<const float4*>(w + 4 * lane);
    float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f}, dmax = 0.f;
    for (int it = 0; it < kLnRowsPerBlock / 4; ++it) {
        const long row = (long)blockIdx.x * kLnRowsPerBlock + it * 4 + wv;
        if (row >= M) break;                                       // wave-uniform
        const float4 g4 = *reinterpret_cast<const float4*>(dy + row * dy_ld + 4 * lane);
        const float4 x4 = *reinterpret_cast<const float4*>(xhat + row * kC + 4 * lane);
        const float gy[4] = {g4.x, g4.y, g4.z, g4.w}, xh[4] = {x4.x, x4.y, x4.z, x4.w};
        const float ww[4] = {vw.x, vw.y, vw.z, vw.w};
        float g[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            g[e] = gy[e] * ww[e];
            s1 += g[e];
            s2 = fmaf(g[e], xh[e], s2);
            aw[e] = fmaf(gy[e], xh[e], aw[e]);
            ab[e] += gy[e];
        }
        const float c1 = wave_sum(s1) * (1.0f / kC), c2 = wave_sum(s2) * (1.0f / kC);
        const float rs = rstd[row];
        float4 r;
        r.x = rs * (g[0] - c1 - xh[0] * c2); r.y = rs * (g[1] - c1 - xh[1] * c2);
        r.z = rs * (g[2] - c1 - xh[2] * c2); r.w = rs * (g[3] - c1 - xh[3] * c2);
        if (add != nullptr) {
            const float4 a4 = *reinterpret_cast<const float4*>(add + row * kC + 4 * lane);
            r.x += a4.x; r.y += a4.y; r.z += a4.z; r.w += a4.w;
        }
        *reinterpret_cast<float4*>(dx + row * kC + 4 * lane) = r;
        dmax = amax4(dmax, r);
    }
    publish_amax_block(dx_amax, dmax);
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wv][0][4 * lane + e] = aw[e]; red[wv][1][4 * lane + e] = ab[e]; }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * kC; e += 256) {
        const int which = e >> 8, c = e & (kC - 1);
        part[(long)blockIdx.x * 2 * kC + e] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// x = relu(x), then (drop_p > 0) the hidden layer's dropout: x *= keep / (1 - p)   (transformers.py:93,100)
// (the forward runs this as the epilogue of the lin1 GEMM where that GEMM is on the wide fp16-piece tile -- gemm.hip,
// GemmEpilogue kind 1 -- and as this kernel otherwise: same arithmetic, same masks)
// x (M, 2048); a thread takes one float4 column of four consecutive rows, whose 16 keep decisions are four Philox blocks; a
// workgroup covers kReluIters * 256 such pieces (grid: cdiv(cdiv(M, 4) * 512, 256 * kReluIters)) and publishes one max|x|.
constexpr int kReluIters = 4;
__global__ __launch_bounds__(256) void relu_kernel(float* __restrict__ x, int M, float drop_p, unsigned long long seed,
                                                   long x_gs, float* __restrict__ x_amax) {
    x += (long)blockIdx.y * x_gs;                     // layer blockIdx.y of a group, dropout stream seed + layer
    if (x_amax != nullptr) x_amax += (long)blockIdx.y * x_gs;
    seed += (unsigned long long)blockIdx.y;
    const unsigned th = drop_threshold16(drop_p);
    const float sc = 1.0f / (1.0f - drop_p);
    const long npiece = (long)((M + 3) >> 2) * (kDff / 4);
    float m = 0.f;
    for (int it = 0; it < kReluIters; ++it) {
        const long e = ((long)blockIdx.x * kReluIters + it) * 256 + threadIdx.x;
        if (e >= npiece) break;
        const long row0 = (e / (kDff / 4)) * 4;
        const int c4 = (int)(e % (kDff / 4)) * 4;
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)                   // rows past M: re-read the last row (unconditional loads), never stored
            v[q] = *reinterpret_cast<const float4*>(x + min(row0 + q, (long)M - 1) * kDff + c4);
        unsigned keep = 0xFFFFu;                      // bit 4 * column + row
        if (drop_p > 0.f) {
            keep = 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const Philox4 r = philox4x32_10(seed, 1u, ffn_drop_block(row0, c4 + c));
                keep |= ((ffn_drop_field(r, 0, c4 + c) >= th ? 1u : 0u) | (ffn_drop_field(r, 1, c4 + c) >= th ? 2u : 0u) |
                         (ffn_drop_field(r, 2, c4 + c) >= th ? 4u : 0u) | (ffn_drop_field(r, 3, c4 + c) >= th ? 8u : 0u)) << (4 * c);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* f = reinterpret_cast<float*>(&v[q]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float u = fmaxf(f[c], 0.f);
                if (drop_p > 0.f) u = ((keep >> (4 * c + q)) & 1u) ? u * sc : 0.f;
                f[c] = u;
            }
            if (row0 + q < M) {
                *reinterpret_cast<float4*>(x + (row0 + q) * kDff + c4) = v[q];
                m = amax4(m, v[q]);
            }
        }
    }
    publish_amax_block(x_amax, m);
}
// The same pass for the DMA-fed feed-forward GEMMs (gemm_dma.hip): x (M, 2048) fp32 = lin1's output is replaced IN PLACE by the
// hidden layer in H2 storage (two fp16 pieces per element, the 32 bytes of eight channels hold [8 h | 8 l]) -- what lin2 and the
// weight gradient of lin2 copy into LDS as it lies.  Scaled for bound = max|lin1 output| / (1 - p) (lin_amax: the slots lin1's
// epilogue left), which block 0 writes into every slot of hid_bound for the consumers; flag[0] = 1 marks the storage
// (cpc_transformer_hidden).  A thread takes EIGHT channels of four consecutive rows -- its own 32-byte groups, so the in-place
// rewrite races with nobody -- and draws the same Philox blocks as relu_kernel: the same masks.
__global__ __launch_bounds__(256) void relu_h2_kernel(float* __restrict__ x, int M, float drop_p, unsigned long long seed, long x_gs,
                                                      const float* __restrict__ lin_amax, float* __restrict__ hid_bound,
                                                      float* __restrict__ flag, unsigned char* __restrict__ bits) {
    // bits: [hid != 0] as one bit per element, row-major, 256 bytes per row (bit c & 7 of byte c >> 3) -- what the backward's
    // ReLU-derivative epilogue reads instead of the 730 MB tensor (gemm_nt_dma_kernel<2>: ONE 16-byte load per accumulator row)
    x += (long)blockIdx.y * x_gs;
    bits += (long)blockIdx.y * x_gs * 4;
    lin_amax += (long)blockIdx.y * x_gs; hid_bound += (long)blockIdx.y * x_gs; flag += (long)blockIdx.y * x_gs;
    seed += (unsigned long long)blockIdx.y;
    const unsigned th = drop_threshold16(drop_p);
    const float sc = 1.0f / (1.0f - drop_p);
    const float bound = fold_amax(lin_amax, kAmaxSlots) * sc;
    const float s = scale_for_amax(bound);
    if (blockIdx.x == 0 && threadIdx.x < kAmaxSlots) {
        hid_bound[threadIdx.x] = bound;
        if (threadIdx.x == 0) flag[0] = 1.0f;
    }
    const long npiece = (long)((M + 3) >> 2) * (kDff / 8);
    for (int it = 0; it < kReluIters; ++it) {
        const long e = ((long)blockIdx.x * kReluIters + it) * 256 + threadIdx.x;
        if (e >= npiece) break;
        const long row0 = (e / (kDff / 8)) * 4;
        const int c8 = (int)(e % (kDff / 8)) * 8;
        float4 v[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                 // rows past M: re-read the last row (unconditional loads), never stored
            const float* r = x + min(row0 + q, (long)M - 1) * kDff + c8;
            v[q][0] = *reinterpret_cast<const float4*>(r);
            v[q][1] = *reinterpret_cast<const float4*>(r + 4);
        }
        unsigned keep = 0xFFFFFFFFu;                  // bit 4 * column + row
        if (drop_p > 0.f) {
            keep = 0u;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const Philox4 r = philox4x32_10(seed, 1u, ffn_drop_block(row0, c8 + c));
                keep |= ((ffn_drop_field(r, 0, c8 + c) >= th ? 1u : 0u) | (ffn_drop_field(r, 1, c8 + c) >= th ? 2u : 0u) |
                         (ffn_drop_field(r, 2, c8 + c) >= th ? 4u : 0u) | (ffn_drop_field(r, 3, c8 + c) >= th ? 8u : 0u)) << (4 * c);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float u[8];
            const float* f0 = reinterpret_cast<const float*>(&v[q][0]);
            const float* f1 = reinterpret_cast<const float*>(&v[q][1]);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float t = fmaxf(c < 4 ? f0[c] : f1[c - 4], 0.f);
                if (drop_p > 0.f) t = ((keep >> (4 * c + q)) & 1u) ? t * sc : 0.f;
                u[c] = t;
            }
            if (row0 + q < M) {
                unsigned char* row = reinterpret_cast<unsigned char*>(x + (row0 + q) * kDff);
                h2_store4(row, c8, u[0], u[1], u[2], u[3], s);
                h2_store4(row, c8 + 4, u[4], u[5], u[6], u[7], s);
                unsigned b = 0u;                  // (an element whose high piece rounds to zero -- below 2^-39 of the bound -- counts as cut)
#pragma unroll
                for (int c = 0; c < 8; ++c) b |= ((float)(_Float16)(u[c] * s) != 0.f ? 1u : 0u) << c;
                bits[(row0 + q) * (kDff / 8) + (c8 >> 3)] = (unsigned char)b;
            }
        }
    }
}
// the hidden layer back as fp32 (tests, inspection): out[m][c] = (h + l) / scale_for_amax(bound)
__global__ __launch_bounds__(256) void hid_h2_decode_kernel(const unsigned char* __restrict__ hid, const float* __restrict__ bound,
                                                            float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const float inv = 1.0f / scale_for_amax(fold_amax(bound, kAmaxSlots));
    if (i >= n) return;
    const long row = i / kDff;
    const int c = (int)(i - row * kDff);
    const unsigned char* p = hid + row * (kDff * 4) + h2_byte_of(c);
    out[i] = h2_join(*reinterpret_cast<const unsigned short*>(p), *reinterpret_cast<const unsigned short*>(p + 16), inv);
}
// g *= (y > 0) * scale, y the SAVED hidden layer: it is zero where the ReLU cut or the dropout dropped, so the product of
// the two derivatives is scale = 1 / (1 - p) exactly where y > 0
constexpr int kReluBwdIters = 16;
__global__ __launch_bounds__(256) void relu_bwd_kernel(float* __restrict__ g, const float* __restrict__ y, long n4, float scale,
                                                       long g_gs, long y_gs, float* __restrict__ g_amax) {
    g += (long)blockIdx.y * g_gs;
    y += (long)blockIdx.y * y_gs;
    if (g_amax != nullptr) g_amax += (long)blockIdx.y * g_gs;
    float m = 0.f;
#pragma unroll 4
    for (int it = 0; it < kReluBwdIters; ++it) {
        const long i = ((long)blockIdx.x * kReluBwdIters + it) * 256 + threadIdx.x;
        if (i >= n4) break;
        float4 v = reinterpret_cast<float4*>(g)[i];
        const float4 a = reinterpret_cast<const float4*>(y)[i];
        v.x = a.x > 0.f ? v.x * scale : 0.f; v.y = a.y > 0.f ? v.y * scale : 0.f;
        v.z = a.z > 0.f ? v.z * scale : 0.f; v.w = a.w > 0.f ? v.w * scale : 0.f;
        reinterpret_cast<float4*>(g)[i] = v;
        m = amax4(m, v);
    }
    publish_amax_block(g_amax, m);
}
// out[i] = keep_i / (1 - p) of site `site` (tests: the mask a layer call with this seed applied)
// site 0: out (BH, S, S), element (bh, i, j) -- attn_drop_block; site 1: out (rows, 2048) -- ffn_drop_block
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, long n, int site, int S, float drop_p,
                                                           unsigned long long seed) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned th = drop_threshold(drop_p);
    const float sc = 1.0f / (1.0f - drop_p);
    Philox4 r;
    int word;
    if (site == 0) {
        const int bh = (int)(i / ((long)S * S)), rem = (int)(i - (long)bh * S * S), row = rem / S, col = rem - row * S;
        r = philox4x32_10(seed, 0u, attn_drop_block(bh, S, row, col));
        word = row & 3;
    } else {
        const long row = i / kFfnWidth;
        const int col = (int)(i - row * kFfnWidth);
        r = philox4x32_10(seed, 1u, ffn_drop_block(row, col));
        out[i] = ffn_drop_field(r, (int)(row & 3), col) >= drop_threshold16(drop_p) ? sc : 0.f;
        return;
    }
    out[i] = philox_word(r, word) >= th ? sc : 0.f;
}
// dst[0:n] = src[0:n] for G (dst, src) pairs dst_gs / src_gs floats apart (blockIdx.y)
__global__ __launch_bounds__(256) void gcopy_kernel(float* __restrict__ dst, const float* __restrict__ src, long n, long dst_gs,
                                                    long src_gs) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[(long)blockIdx.y * dst_gs + i] = src[(long)blockIdx.y * src_gs + i];
}
static void gcopy(float* dst, const float* src, long n, int G, long dst_gs, long src_gs, hipStream_t st) {
    hipLaunchKernelGGL(gcopy_kernel, dim3(cdiv(n, 256), G), dim3(256), 0, st, dst, src, n, dst_gs, src_gs);
}
// out[0:n4*4] = sum_g a[g * a_gs + .]: the gradients of G layers that shared one input
__global__ __launch_bounds__(256) void gsum_kernel(float* __restrict__ out, const float* __restrict__ a, long n4, int G, long a_gs) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<const float4*>(a)[i];
    for (int g = 1; g < G; ++g) {
        const float4 u = reinterpret_cast<const float4*>(a + (long)g * a_gs)[i];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    reinterpret_cast<float4*>(out)[i] = v;
}
__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ a, const float* __restrict__ b, long n4, long a_gs, long b_gs) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    a += (long)blockIdx.y * a_gs;
    b += (long)blockIdx.y * b_gs;
    float4 v = reinterpret_cast<float4*>(a)[i];
    const float4 u = reinterpret_cast<const float4*>(b)[i];
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    reinterpret_cast<float4*>(a)[i] = v;
}

// ------------------------------------------------------------------ host side
int g_attn_fwd = 1;        // cpc_set_attn_fwd: 1 = attn_fwd2_kernel (67 KB of LDS, two workgroups per CU), 0 = attn_fwd_kernel (133 KB)
struct TfLayout {
    long qkv, A, o, xhat1, rstd1, y, hid, xhat2, rstd2, bounds, yh, hbits, saved_total;   // saved for backward (yh: y in H2 storage, DMA-fed GEMMs)
    long wq1, wq2, fwd_total;                                              // forward scratch: one (M,256) buffer + lin1 / lin2's weights as the DMA tiles read them
    long ds2, dhid, dyb, ds1, dob, dqkv, w2t, w1t, wot, wqkv, wqkvt, part, lnpart, tmp, dppart, bbounds, colpart, ds2h, bwd_total;
};
// GEMM operand bounds (publish_amax), kAmaxSlots floats each.  Forward's live in `saved` (the backward reads them again; a
// group keeps them in layer 0's copy), the gradients' in the backward scratch.
enum { kBX = 0, kBWq, kBWk, kBWv, kBWo, kBW1, kBW2, kBWqkv, kBO, kBY, kBHid, kBLin1, kBFlag, kBW1L1, kBB1, kTfBounds };   // (kBO on: zeroed per forward)
enum { kBDs2 = 0, kBDh, kBDs1, kBDqkv, kBW2L1, kTfBwdBounds };

static bool tf_layout(int B, int S, TfLayout& t) {
    if (B <= 0 || S <= 0 || S > kSlong) return false;
    const long M = (long)B * S;
    long o = 0;
    t.qkv = o; o += align64l(M * 3 * kC);
    t.A = o; o += S > kSmax ? 0 : align64l((long)B * kTH * S * S);     // (longer sequences: forward only, attn_fwd_long_kernel)
    t.o = o; o += align64l(M * kC);
    t.xhat1 = o; o += align64l(M * kC);
    t.rstd1 = o; o += align64l(M);
    t.y = o; o += align64l(M * kC);
    t.hid = o; o += align64l(M * kDff);
    t.xhat2 = o; o += align64l(M * kC);
    t.rstd2 = o; o += align64l(M);
    t.bounds = o; o += (long)kTfBounds * kAmaxSlots;
    t.yh = o; o += align64l(M * kC);
    t.hbits = o; o += align64l(M * (kDff / 32));                       // [hid != 0], one bit per element
    t.saved_total = o;
    t.wq1 = align64l(M * kC);
    t.wq2 = t.wq1 + (long)kDff * kC;
    t.fwd_total = t.wq2 + (long)kDff * kC;
    o = 0;
    t.ds2 = o; o += align64l(M * kC);
    t.dhid = o; o += align64l(M * kDff);
    t.dyb = o; o += align64l(M * kC);
    t.ds1 = o; o += align64l(M * kC);
    t.dob = o; o += align64l(M * kC);
    t.dqkv = o; o += align64l(M * 3 * kC);
    t.w2t = o; o += (long)kDff * kC;
    t.w1t = o; o += (long)kDff * kC;
    t.wot = o; o += (long)kC * kC;
    t.wqkv = o; o += 3L * kC * kC;
    t.wqkvt = o; o += 3L * kC * kC;
    t.part = o; o += align64l(std::max(tn_gemm_part_floats((int)M, kDff, kC), gemm_tn_dma_part_floats((int)M, kDff, kC, 1)));
    const long nblk = cdiv(M, kLnRowsPerBlock);
    t.lnpart = o; o += align64l(nblk * 2 * kC);
    t.tmp = o; o += align64l((long)kRowsSumGroups * (kDk * S > kDff ? kDk * S : kDff));
    t.dppart = o; o += align64l((long)B * kTH * kDk * S);
    t.bbounds = o; o += (long)kTfBwdBounds * kAmaxSlots;
    t.colpart = o; o += align64l((long)cdiv(M, 128) * kDff);          // column sums of dh per 128-row tile (GemmEpilogue::colsum)
    t.ds2h = o; o += align64l(M * kC);                                // ds2 in H2 storage (DMA-fed GEMMs)
    t.bwd_total = o;
    return true;
}

// Host-side description of a group call: the strides the kernels need (TfStrides) + those of the tensors the caller owns.
struct TfGroup {
    int G = 1;
    TfStrides ks;                 // saved / scratch / parameter strides
    long x = 0;                   // input of layer g at x + g * x (0: all layers read the same input)
    long out = 0; int out_ld = kC;   // forward output of layer g at out + g * out, rows out_ld floats apart
    long dy = 0; int dy_ld = kC;     // backward input, same addressing
    long dx = 0;                  // backward output of layer g at dx + g * dx (dense (M,256))
};

// strides of a group of G stacked layers (cpc_transformer_group_*)
static TfGroup tf_group(const TfLayout& t, int G, int S) {
    TfGroup tg;
    tg.G = G;
    tg.ks.saved = t.saved_total;
    tg.ks.scratch = 0;                   // set by the caller's direction: forward / backward workspaces differ in size
    const long numel[13] = {(long)kC * kC, (long)kC * kC, (long)kC * kC, (long)kC * kC, (long)kDk * S, kC, kC,
                            (long)kDff * kC, kDff, (long)kC * kDff, kC, kC, kC};
    for (int i = 0; i < 13; ++i) tg.ks.par[i] = numel[i];
    return tg;
}

// Bounds are kept for a single layer and for a group whose layers share the input and stack their parameters (what the two
// entry points build); anything else keeps the GEMMs on three bf16 pieces.
static bool tf_bounded(const TfGroup& tg) { return tg.G == 1 || (tg.x == 0 && tg.ks.par[0] == (long)kC * kC); }

static int tf_forward(const TfGroup& tg, const float* x, const float* const* params, float* saved, float* scratch, float* out,
                      int B, int S, float p, unsigned long long seed, hipStream_t st) {
    TfLayout t;
    if (!tf_layout(B, S, t)) return CPC_ERR_SHAPE;
    if (S > kSmax && p > 0.f) return CPC_ERR_SHAPE;   // beyond 128 steps: inference only (no dropout masks, nothing saved for a backward)
    const int M = B * S, G = tg.G;
    const long sv = tg.ks.saved, sc = tg.ks.scratch;
    const long* ps = tg.ks.par;
    const float *Wo = params[0], *Wk = params[1], *Wq = params[2], *Wv = params[3], *P = params[4];
    float* qkv = saved + t.qkv;
    const RowMap xm = plain_rows(x, M, kC);
    auto grp = [&](long a, long b, long bias, long c) { GemmGroup g; g.G = G; g.a = a; g.b = b; g.bias = bias; g.c = c; return g; };
    int rc;
    // operand bounds: the input (shared by the layers of a group: kept in layer 0's workspace) and the six weight matrices by
    // reduction, the intermediates by the kernels that write them
    float* bnd = saved + t.bounds;
    const bool bounded = tf_bounded(tg);
    if (bounded) {
        const float* ax[1] = {x};
        const long an[1] = {(long)M * kC};
        if ((rc = absmax_slots(ax, an, 1, bnd + kBX * kAmaxSlots, st))) return rc;
        const float* wx[4] = {Wq, Wk, Wv, Wo};                    // in the order of kBWq .. kBWo
        const long wn[4] = {(long)kC * kC, (long)kC * kC, (long)kC * kC, (long)kC * kC}, wg[4] = {ps[2], ps[1], ps[3], ps[0]};
        if ((rc = absmax_group(wx, wn, wg, 4, G, bnd + kBWq * kAmaxSlots, sv, st))) return rc;
        const float* fx[2] = {params[7], params[9]};
        const long fn[2] = {(long)kDff * kC, (long)kDff * kC}, fg[2] = {ps[7], ps[9]};
        if ((rc = absmax_group(fx, fn, fg, 2, G, bnd + kBW1 * kAmaxSlots, sv, st))) return rc;
        hipLaunchKernelGGL(bounds_begin_kernel, dim3(G), dim3(kAmaxSlots), 0, st, bnd + kBWqkv * kAmaxSlots, bnd + kBWq * kAmaxSlots,
                           bnd + kBWk * kAmaxSlots, bnd + kBWv * kAmaxSlots, bnd + kBO * kAmaxSlots, (kTfBounds - kBO) * kAmaxSlots, sv);
    }
    auto gbnd = [&](int ia, int ib) {
        GemmBounds g;
        if (bounded) {
            g.a = bnd + ia * kAmaxSlots; g.b = bnd + ib * kAmaxSlots; g.a_slots = g.b_slots = kAmaxSlots;
            g.a_gs = ia == kBX ? 0 : sv; g.b_gs = sv;
        }
        return g;
    };
    auto slot = [&](int i) { return bounded ? bnd + i * kAmaxSlots : (float*)nullptr; };
    if ((rc = nt_gemm(xm, Wq, kC, nullptr, qkv, 3 * kC, kC, kC, st, 0, 0, gbnd(kBX, kBWq), grp(tg.x, ps[2], 0, sv)))) return rc;
    if ((rc = nt_gemm(xm, Wk, kC, nullptr, qkv + kC, 3 * kC, kC, kC, st, 0, 0, gbnd(kBX, kBWk), grp(tg.x, ps[1], 0, sv)))) return rc;
    if ((rc = nt_gemm(xm, Wv, kC, nullptr, qkv + 2 * kC, 3 * kC, kC, kC, st, 0, 0, gbnd(kBX, kBWv), grp(tg.x, ps[3], 0, sv)))) return rc;
    if (S > kSmax)       // (inference: checked by the caller -- no dropout, no backward)
        hipLaunchKernelGGL(attn_fwd_long_kernel, dim3(B * kTH, cdiv(S, kSmax), G), dim3(256), 0, st, qkv, P, saved + t.o, S, tg.ks,
                           slot(kBO));
    else if (g_attn_fwd == 1)
        hipLaunchKernelGGL(attn_fwd2_kernel, dim3(B * kTH, G), dim3(256), 0, st, qkv, P, saved + t.o, saved + t.A, S, p, seed, tg.ks,
                           slot(kBO));
    else
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * kTH, G), dim3(256), 0, st, qkv, P, saved + t.o, saved + t.A, S, p, seed, tg.ks,
                       slot(kBO));
    CPC_LAUNCH_CHECK();
    float* att = scratch;
    if ((rc = nt_gemm(plain_rows(saved + t.o, M, kC), Wo, kC, nullptr, att, kC, kC, kC, st, 0, 0, gbnd(kBO, kBWo), grp(sv, ps[0], 0, sc))))
        return rc;
    hipLaunchKernelGGL(add_ln_fwd_kernel, dim3(cdiv(M, kLnFwdRows), G), dim3(256), 0, st, x, att, params[5], params[6],
                       saved + t.y, saved + t.xhat1, saved + t.rstd1, M, tg.x, sc, ps[5], sv, kC, sv, slot(kBY));
    float* ff = scratch;
    // The feed-forward network on the DMA-fed tiles (gemm_dma.hip) where the call's launches fill the chip -- the K predictors as a
    // group: y -> H2 (its bound: the slots add_ln_fwd left), lin1 / lin2's weights -> the K-tile-major H2 rows of the tile, lin1 ->
    // fp32 + max|.|, ReLU + dropout -> the hidden layer in H2 storage IN PLACE of it, lin2.  The backward takes the same branch
    // (tf_backward asks gemm_dma_wanted with the same arguments; cpc_set_gemm_dma must not change between the two).
    if (bounded && gemm_dma_wanted(M, G)) {
        if ((rc = rows_to_h2(saved + t.y, saved + t.yh, slot(kBY), M, G, sv, sv, sv, st))) return rc;
        const bool fuse_relu = gemm_dma_relu_fused();
        if ((rc = gemm_weight_h2(params[7], kC, 1, kDff, kC, scratch + t.wq1, slot(kBW1), fuse_relu ? slot(kBW1L1) : nullptr, G, ps[7], sc, sv,
                                 sv, st))) return rc;
        if ((rc = gemm_weight_h2(params[9], kDff, 1, kC, kDff, scratch + t.wq2, slot(kBW2), nullptr, G, ps[9], sc, sv, 0, st))) return rc;
        if (fuse_relu) {
            // bias, ReLU, dropout, the two fp16 pieces and the mask bits in lin1's epilogue (gemm_nt_dma_kernel<1>)
            const float* bx[1] = {params[8]};
            const long bn[1] = {kDff}, bg[1] = {ps[8]};
            if ((rc = absmax_group(bx, bn, bg, 1, G, slot(kBB1), sv, st))) return rc;
            if ((rc = gemm_nt_dma_relu(saved + t.yh, kC, scratch + t.wq1, params[8], saved + t.hid, kDff, saved + t.hbits, p, seed, M, kDff, kC,
                                       slot(kBY), slot(kBW1), slot(kBW1L1), slot(kBB1), slot(kBHid), slot(kBFlag), G, sv, sc, ps[8], sv, sv,
                                       sv, sv, sv, st))) return rc;
        } else {
        if ((rc = gemm_nt_dma(saved + t.yh, kC, scratch + t.wq1, params[8], saved + t.hid, kDff, M, kDff, kC, slot(kBY), slot(kBW1),
                              slot(kBLin1), G, sv, sc, ps[8], sv, sv, sv, sv, st))) return rc;
        hipLaunchKernelGGL(relu_h2_kernel, dim3(cdiv((long)cdiv(M, 4) * (kDff / 8), 256 * kReluIters), G), dim3(256), 0, st,
                           saved + t.hid, M, p, seed, sv, slot(kBLin1), slot(kBHid), slot(kBFlag),
                           reinterpret_cast<unsigned char*>(saved + t.hbits));
        CPC_LAUNCH_CHECK();
        }
        if ((rc = gemm_nt_dma(saved + t.hid, kDff, scratch + t.wq2, params[10], ff, kC, M, kC, kDff, slot(kBHid), slot(kBW2), nullptr,
                              G, sv, sc, ps[10], sc, sv, sv, 0, st))) return rc;
    } else {
    // hid = dropout(relu(y W1^T + b1)): as the GEMM's epilogue where it runs on the tile that has one, else a pass behind it
    // (with dropout the elementwise kernel draws its Philox blocks at eight waves per SIMD; in the epilogue of a tile that runs
    // two per SIMD the same draws cost more than the pass they save: 698 vs 377 + 284 us per group at B = 64)
    if (bounded && p == 0.f && nt_gemm_fuses(M, kDff, kC, kDff, gbnd(kBY, kBW1), grp(sv, ps[7], ps[8], sv))) {
        GemmEpilogue ep;
        ep.kind = 1; ep.drop_p = p; ep.seed = seed; ep.amax = slot(kBHid); ep.amax_gs = sv;
        if ((rc = nt_gemm_fused(plain_rows(saved + t.y, M, kC), params[7], kC, params[8], saved + t.hid, kDff, kDff, kC, st,
                                gbnd(kBY, kBW1), grp(sv, ps[7], ps[8], sv), ep))) return rc;
    } else {
    if ((rc = nt_gemm(plain_rows(saved + t.y, M, kC), params[7], kC, params[8], saved + t.hid, kDff, kDff, kC, st, 0, 0,
                      gbnd(kBY, kBW1), grp(sv, ps[7], ps[8], sv)))) return rc;
    hipLaunchKernelGGL(relu_kernel, dim3(cdiv((long)cdiv(M, 4) * (kDff / 4), 256 * kReluIters), G), dim3(256), 0, st,
                       saved + t.hid, M, p, seed, sv, slot(kBHid));
    }
    if ((rc = nt_gemm(plain_rows(saved + t.hid, M, kDff), params[9], kDff, params[10], ff, kC, kC, kDff, st, 0, 0, gbnd(kBHid, kBW2),
                      grp(sv, ps[9], ps[10], sc)))) return rc;
    }
    hipLaunchKernelGGL(add_ln_fwd_kernel, dim3(cdiv(M, kLnFwdRows), G), dim3(256), 0, st, saved + t.y, ff, params[11], params[12],
                       out, saved + t.xhat2, saved + t.rstd2, M, sv, sc, ps[11], tg.out, tg.out_ld, sv, (float*)nullptr);
    CPC_LAUNCH_CHECK();
    return 0;
}

static int tf_backward(const TfGroup& tg, const float* x, const float* const* params, const float* saved, const float* dy,
                       float* scratch, float* dx, float* const* grads, int B, int S, float p, unsigned long long seed,
                       hipStream_t st) {
    TfLayout t;
    if (!tf_layout(B, S, t) || S > kSmax) return CPC_ERR_SHAPE;      // (sequences beyond 128 steps: forward only)
    const int M = B * S, G = tg.G;
    const long sv = tg.ks.saved, sc = tg.ks.scratch;
    const long* ps = tg.ks.par;
    const long n4 = (long)M * kC / 4;
    const int nblk = cdiv(M, kLnRowsPerBlock);
    const float *Wo = params[0], *Wk = params[1], *Wq = params[2], *Wv = params[3], *P = params[4];
    const float *W1 = params[7], *W2 = params[9];
    float *ds2 = scratch + t.ds2, *dhid = scratch + t.dhid, *dyb = scratch + t.dyb, *ds1 = scratch + t.ds1;
    float *dob = scratch + t.dob, *dqkv = scratch + t.dqkv, *part = scratch + t.part, *lnpart = scratch + t.lnpart;
    float *tmp = scratch + t.tmp;
    auto grp = [&](long a, long b, long c) { GemmGroup g; g.G = G; g.a = a; g.b = b; g.c = c; g.part = sc; return g; };
    int rc;
    const float* bnd = saved + t.bounds;              // the forward's operand bounds (tf_forward)
    float* bb = scratch + t.bbounds;                  // the gradients', left by the kernels that write them
    const bool bounded = tf_bounded(tg);
    if (bounded)
        hipLaunchKernelGGL(bounds_begin_kernel, dim3(G), dim3(kAmaxSlots), 0, st, (float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, bb, kTfBwdBounds * kAmaxSlots, sc);
    auto gbnd = [&](int ig, int iw) {                 // a: gradient bound ig, b: forward bound iw
        GemmBounds g;
        if (bounded) {
            g.a = bb + ig * kAmaxSlots; g.b = bnd + iw * kAmaxSlots; g.a_slots = g.b_slots = kAmaxSlots;
            g.a_gs = sc; g.b_gs = iw == kBX ? 0 : sv;
        }
        return g;
    };
    auto slot = [&](int i) { return bounded ? bb + i * kAmaxSlots : (float*)nullptr; };
    // out = LN2(y + ff)
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(nblk, G), dim3(256), 0, st, dy, saved + t.xhat2, saved + t.rstd2, params[11],
                       (const float*)nullptr, ds2, lnpart, M, tg.dy, tg.dy_ld, sv, ps[11], sc, sc, slot(kBDs2));
    if ((rc = rows_sum(lnpart, nblk, 2 * kC, tmp, dhid, st, G, sc, sc, sc))) return rc;          // dhid as a 512-float staging area
    gcopy(grads[11], dhid, kC, G, ps[11], sc, st);
    gcopy(grads[12], dhid + kC, kC, G, ps[12], sc, st);
    if (bounded && gemm_dma_wanted(M, G)) {
        // the feed-forward network's backward on the DMA-fed tiles (the forward left hid and y in H2 storage, tf_forward)
        float* ds2h = scratch + t.ds2h;
        if ((rc = rows_to_h2(ds2, ds2h, slot(kBDs2), M, G, sc, sc, sc, st))) return rc;
        // dW2 (256, 2048) = ds2^T . hid
        if ((rc = gemm_tn_dma(ds2h, kC, kC, saved + t.hid, kDff, kDff, M, part, grads[9], slot(kBDs2), bnd + kBHid * kAmaxSlots, G, sc, sv,
                              sc, ps[9], sc, sv, st))) return rc;
        if ((rc = rows_sum(ds2, M, kC, tmp, grads[10], st, G, sc, sc, ps[10]))) return rc;
        // dh = (ds2 W2) * [hid > 0] / (1 - p), straight into H2 storage: B(n, k) = W2[k][n]; its bound is a-priori,
        // max|ds2| * max_n sum_k |W2[k][n]| / (1 - p) -- the l1 slots the weight layout kernel leaves
        if ((rc = gemm_weight_h2(W2, 1, kDff, kDff, kC, scratch + t.w2t, bnd + kBW2 * kAmaxSlots, slot(kBW2L1), G, ps[9], sc, sv, sc, st)))
            return rc;
        if ((rc = gemm_nt_dma_masked(ds2h, kC, scratch + t.w2t, dhid, kDff, saved + t.hbits, 1.0f / (1.0f - p), M, kDff, kC, slot(kBDs2),
                                     bnd + kBW2 * kAmaxSlots, slot(kBW2L1), scratch + t.colpart, slot(kBDh), G, sc, sc, sc, sv, sc, sv, sc,
                                     sc, st))) return rc;
        // dW1 (2048, 256) = dh^T . y;  db1 = column sums of dh (per 256-row tile from the epilogue)
        if ((rc = gemm_tn_dma(dhid, kDff, kDff, saved + t.yh, kC, kC, M, part, grads[7], slot(kBDh), bnd + kBY * kAmaxSlots, G, sc, sv, sc,
                              ps[7], sc, sv, st))) return rc;
        if ((rc = rows_sum(scratch + t.colpart, cdiv(M, 256), kDff, tmp, grads[8], st, G, sc, sc, ps[8]))) return rc;
        // dy_ff = dh . W1: B(n, k) = W1[k][n]
        if ((rc = gemm_weight_h2(W1, 1, kC, kC, kDff, scratch + t.w1t, bnd + kBW1 * kAmaxSlots, nullptr, G, ps[7], sc, sv, 0, st))) return rc;
        if ((rc = gemm_nt_dma(dhid, kDff, scratch + t.w1t, nullptr, dyb, kC, M, kC, kDff, slot(kBDh), bnd + kBW1 * kAmaxSlots, nullptr, G,
                              sc, sc, 0, sc, sc, sv, 0, st))) return rc;
    } else {
    // ff = hid W2^T + b2
    const RowMap ds2m = plain_rows(ds2, M, kC), hidm = plain_rows(saved + t.hid, M, kDff);
    if ((rc = tn_gemm(ds2m, kC, hidm, kDff, part, grads[9], 0, st, gbnd(kBDs2, kBHid), grp(sc, sv, ps[9])))) return rc;   // dW2 (256,2048)
    if ((rc = rows_sum(ds2, M, kC, tmp, grads[10], st, G, sc, sc, ps[10]))) return rc;
    if ((rc = transpose(W2, scratch + t.w2t, kC, kDff, st, G, ps[9], sc))) return rc;           // (256,2048) -> (2048,256)
    // dh = (ds2 W2) * [hid > 0] / (1 - p): hid is zero where the ReLU cut or the dropout dropped
    bool db1_from_tiles = false;
    if (bounded && nt_gemm_fuses(M, kDff, kC, kDff, gbnd(kBDs2, kBW2), grp(sc, sc, sc))) {
        GemmEpilogue ep;
        ep.kind = 2; ep.mask = saved + t.hid; ep.mask_gs = sv; ep.scale = 1.0f / (1.0f - p); ep.amax = slot(kBDh); ep.amax_gs = sc;
        ep.colsum = scratch + t.colpart; ep.colsum_gs = sc;            // db1 = column sums of dh, per row tile here
        db1_from_tiles = true;
        if ((rc = nt_gemm_fused(ds2m, scratch + t.w2t, kC, nullptr, dhid, kDff, kDff, kC, st, gbnd(kBDs2, kBW2), grp(sc, sc, sc), ep)))
            return rc;
    } else {
    if ((rc = nt_gemm(ds2m, scratch + t.w2t, kC, nullptr, dhid, kDff, kDff, kC, st, 0, 0, gbnd(kBDs2, kBW2), grp(sc, sc, sc)))) return rc;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(cdiv((long)M * kDff / 4, 256 * kReluBwdIters), G), dim3(256), 0, st, dhid, saved + t.hid,
                       (long)M * kDff / 4, 1.0f / (1.0f - p), sc, sv, slot(kBDh));
    }
    // hid = relu(y W1^T + b1)
    const RowMap dhm = plain_rows(dhid, M, kDff), ym = plain_rows(saved + t.y, M, kC);
    if ((rc = tn_gemm(dhm, kDff, ym, kC, part, grads[7], 0, st, gbnd(kBDh, kBY), grp(sc, sv, ps[7])))) return rc;      // dW1 (2048,256)
    if (db1_from_tiles) rc = rows_sum(scratch + t.colpart, cdiv(M, 128), kDff, tmp, grads[8], st, G, sc, sc, ps[8]);
    else rc = rows_sum(dhid, M, kDff, tmp, grads[8], st, G, sc, sc, ps[8]);
    if (rc) return rc;
    if ((rc = transpose(W1, scratch + t.w1t, kDff, kC, st, G, ps[7], sc))) return rc;           // (2048,256) -> (256,2048)
    if ((rc = nt_gemm(dhm, scratch + t.w1t, kDff, nullptr, dyb, kC, kC, kDff, st, 0, 0, gbnd(kBDh, kBW1), grp(sc, sc, sc)))) return rc;
    }
    hipLaunchKernelGGL(add_kernel, dim3(cdiv(n4, 256), G), dim3(256), 0, st, dyb, ds2, n4, sc, sc);   // dy_total = ds2 + dhid W1
    // y = LN1(x + att)
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(nblk, G), dim3(256), 0, st, dyb, saved + t.xhat1, saved + t.rstd1, params[5],
                       (const float*)nullptr, ds1, lnpart, M, sc, kC, sv, ps[5], sc, sc, slot(kBDs1));
    if ((rc = rows_sum(lnpart, nblk, 2 * kC, tmp, dhid, st, G, sc, sc, sc))) return rc;
    gcopy(grads[5], dhid, kC, G, ps[5], sc, st);
    gcopy(grads[6], dhid + kC, kC, G, ps[6], sc, st);
    // att = o Wo^T
    const RowMap ds1m = plain_rows(ds1, M, kC);
    if ((rc = tn_gemm(ds1m, kC, plain_rows(saved + t.o, M, kC), kC, part, grads[0], 0, st, gbnd(kBDs1, kBO), grp(sc, sv, ps[0])))) return rc;
    if ((rc = transpose(Wo, scratch + t.wot, kC, kC, st, G, ps[0], sc))) return rc;
    if ((rc = nt_gemm(ds1m, scratch + t.wot, kC, nullptr, dob, kC, kC, kC, st, 0, 0, gbnd(kBDs1, kBWo), grp(sc, sc, sc)))) return rc;
    // attention
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * kTH, G), dim3(512), 0, st, saved + t.qkv, P, saved + t.o,
                       saved + t.A, dob, dqkv, scratch + t.dppart, S, p, seed, tg.ks, slot(kBDqkv));
    CPC_LAUNCH_CHECK();
    if (P != nullptr && (rc = rows_sum(scratch + t.dppart, B * kTH, kDk * S, tmp, grads[4], st, G, sc, sc, ps[4]))) return rc;
    // projections
    const RowMap xm = plain_rows(x, M, kC);
    if ((rc = tn_gemm(plain_rows(dqkv, M, 3 * kC), kC, xm, kC, part, grads[2], 0, st, gbnd(kBDqkv, kBX), grp(sc, tg.x, ps[2])))) return rc;            // dWq
    if ((rc = tn_gemm(plain_rows(dqkv + kC, M, 3 * kC), kC, xm, kC, part, grads[1], 0, st, gbnd(kBDqkv, kBX), grp(sc, tg.x, ps[1])))) return rc;       // dWk
    if ((rc = tn_gemm(plain_rows(dqkv + 2 * kC, M, 3 * kC), kC, xm, kC, part, grads[3], 0, st, gbnd(kBDqkv, kBX), grp(sc, tg.x, ps[3])))) return rc;   // dWv
    float* wqkv = scratch + t.wqkv;                                               // [Wq; Wk; Wv] (768,256)
    gcopy(wqkv, Wq, (long)kC * kC, G, sc, ps[2], st);
    gcopy(wqkv + kC * kC, Wk, (long)kC * kC, G, sc, ps[1], st);
    gcopy(wqkv + 2 * kC * kC, Wv, (long)kC * kC, G, sc, ps[3], st);
    if ((rc = transpose(wqkv, scratch + t.wqkvt, 3 * kC, kC, st, G, sc, sc))) return rc;     // -> (256,768)
    // dx of layer g: into the caller's buffer when every layer has its own, else into ds2 (free by now) for the sum below
    float* dxg = (G > 1 && tg.dx == 0) ? ds2 : dx;
    const long dxg_gs = (G > 1 && tg.dx == 0) ? sc : tg.dx;
    if ((rc = nt_gemm(plain_rows(dqkv, M, 3 * kC), scratch + t.wqkvt, 3 * kC, nullptr, dxg, kC, kC, 3 * kC, st, 0, 0,
                      gbnd(kBDqkv, kBWqkv), grp(sc, sc, dxg_gs)))) return rc;
    hipLaunchKernelGGL(add_kernel, dim3(cdiv(n4, 256), G), dim3(256), 0, st, dxg, ds1, n4, dxg_gs, sc);   // + the residual branch
    if (dxg != dx)                                                                // layers that shared x: dx = sum over them
        hipLaunchKernelGGL(gsum_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, dx, dxg, n4, G, dxg_gs);
    CPC_LAUNCH_CHECK();
    return 0;
}

}  // namespace cpc

using namespace cpc;

// sizes[0] = saved floats, [1] = forward scratch floats, [2] = backward scratch floats; [3..7] = offsets inside
// `saved` of qkv (B*S,768), A (B*8,S,S), o (B*S,256), y (B*S,256), hid (B*S,2048) (for tests / inspection)
extern "C" int cpc_transformer_layout(int B, int S, long* sizes) {
    TfLayout t;
    CPC_RETURN_IF(!tf_layout(B, S, t) || !sizes, CPC_ERR_SHAPE);
    sizes[0] = t.saved_total; sizes[1] = t.fwd_total; sizes[2] = t.bwd_total;
    sizes[3] = t.qkv; sizes[4] = t.A; sizes[5] = t.o; sizes[6] = t.y; sizes[7] = t.hid;
    return 0;
}

// params (13 pointers, state-dict order of the reference's TransformerLayer): multihead.Wo, Wk, Wq, Wv (256,256),
// multihead.Att.Krelpos (32,S) or NULL (abspos), ln_multihead.weight/bias, ffnetwork.lin1.weight (2048,256)/bias,
// lin2.weight (256,2048)/bias, ln_ffnetwork.weight/bias.   x, out: (B,S,256).
extern "C" int cpc_transformer_layer_forward(const float* x, const float* const* params, float* saved, float* scratch,
                                             float* out, int B, int S, void* stream) {
    return cpc_transformer_layer_forward_dropout(x, params, saved, scratch, out, B, S, 0.f, 0ull, stream);
}

// The same in training mode with dropout probability p on the attention probabilities and on the feed-forward hidden
// layer (cpc/transformers.py:18,50,93,100; the reference hard-codes p = 0.1).  The masks are a function of `seed`
// (Philox4x32-10 over the element index, see the kernels): pass the same seed to the backward call.
extern "C" int cpc_transformer_layer_forward_dropout(const float* x, const float* const* params, float* saved,
                                                     float* scratch, float* out, int B, int S, float p,
                                                     unsigned long long seed, void* stream) {
    CPC_RETURN_IF(!(p >= 0.f && p < 1.f), CPC_ERR_ARG);
    CPC_RETURN_IF(!x || !params || !saved || !scratch || !out, CPC_ERR_ARG);
    return tf_forward(TfGroup(), x, params, saved, scratch, out, B, S, p, seed, (hipStream_t)stream);
}

// dy (B,S,256) -> dx (B,S,256) and grads[13] (same order as params; grads[4] ignored when params[4] is NULL).
extern "C" int cpc_transformer_layer_backward(const float* x, const float* const* params, const float* saved,
                                              const float* dy, float* scratch, float* dx, float* const* grads,
                                              int B, int S, void* stream) {
    return cpc_transformer_layer_backward_dropout(x, params, saved, dy, scratch, dx, grads, B, S, 0.f, 0ull, stream);
}

// Backward of a cpc_transformer_layer_forward_dropout call made with the same p and seed.
extern "C" int cpc_transformer_layer_backward_dropout(const float* x, const float* const* params, const float* saved,
                                                      const float* dy, float* scratch, float* dx, float* const* grads,
                                                      int B, int S, float p, unsigned long long seed, void* stream) {
    CPC_RETURN_IF(!(p >= 0.f && p < 1.f), CPC_ERR_ARG);
    CPC_RETURN_IF(!x || !params || !saved || !dy || !scratch || !dx || !grads, CPC_ERR_ARG);
    return tf_backward(TfGroup(), x, params, saved, dy, scratch, dx, grads, B, S, p, seed, (hipStream_t)stream);
}

// G transformer layers of one shape on ONE input x (B,S,256) in lock-step -- the K predictors of the criterion in
// --rnnMode transformer (cpc/criterion/criterion.py:82-88, :97-118): every kernel of the layer is launched once for all of
// them instead of G times (G = 12 at B = 64: 17 -> 8 ms per train step, mostly because a 7424-row GEMM alone cannot fill the
// chip).  params[i] / grads[i]: the G tensors of kind i stacked (layer g at + g * numel); saved / scratch: G workspaces of
// cpc_transformer_layout's sizes back to back.  out / dy: (B*S, G*256), layer g at columns g*256.. (the layout the score
// kernels read); dx (B,S,256) = the SUM of the layers' input gradients.  Layer g's dropout masks are those of a single-layer
// call with seed + g.
extern "C" int cpc_transformer_group_forward(const float* x, const float* const* params, float* saved, float* scratch,
                                             float* out, int B, int S, int G, float p, unsigned long long seed, void* stream) {
    CPC_RETURN_IF(!(p >= 0.f && p < 1.f) || G <= 0 || G > 64, CPC_ERR_ARG);
    CPC_RETURN_IF(!x || !params || !saved || !scratch || !out, CPC_ERR_ARG);
    TfLayout t;
    CPC_RETURN_IF(!tf_layout(B, S, t), CPC_ERR_SHAPE);
    TfGroup tg = tf_group(t, G, S);
    tg.ks.scratch = t.fwd_total;
    tg.out = kC; tg.out_ld = G * kC;
    return tf_forward(tg, x, params, saved, scratch, out, B, S, p, seed, (hipStream_t)stream);
}

extern "C" int cpc_transformer_group_backward(const float* x, const float* const* params, const float* saved, const float* dy,
                                              float* scratch, float* dx, float* const* grads, int B, int S, int G, float p,
                                              unsigned long long seed, void* stream) {
    CPC_RETURN_IF(!(p >= 0.f && p < 1.f) || G <= 0 || G > 64, CPC_ERR_ARG);
    CPC_RETURN_IF(!x || !params || !saved || !dy || !scratch || !dx || !grads, CPC_ERR_ARG);
    TfLayout t;
    CPC_RETURN_IF(!tf_layout(B, S, t), CPC_ERR_SHAPE);
    TfGroup tg = tf_group(t, G, S);
    tg.ks.scratch = t.bwd_total;
    tg.dy = kC; tg.dy_ld = G * kC;
    return tf_backward(tg, x, params, saved, dy, scratch, dx, grads, B, S, p, seed, (hipStream_t)stream);
}

// The hidden layer (B*S, 2048) of the forward call that filled `saved` (one layer's workspace), as fp32 whatever its storage: the
// DMA-fed feed-forward path keeps it in H2 storage (tf_forward).  Tests / inspection; synchronises `stream`.
extern "C" int cpc_transformer_hidden(const float* saved, float* out, int B, int S, void* stream) {
    TfLayout t;
    CPC_RETURN_IF(!tf_layout(B, S, t), CPC_ERR_SHAPE);
    CPC_RETURN_IF(!saved || !out, CPC_ERR_ARG);
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)B * S * kDff;
    float flag = 0.f;
    if (hipMemcpyAsync(&flag, saved + t.bounds + kBFlag * kAmaxSlots, sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return CPC_ERR_ARG;
    if (flag == 1.0f)
        hipLaunchKernelGGL(hid_h2_decode_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(saved + t.hid),
                           saved + t.bounds + kBHid * kAmaxSlots, out, n);
    else if (hipMemcpyAsync(out, saved + t.hid, sizeof(float) * n, hipMemcpyDeviceToDevice, st) != hipSuccess) return CPC_ERR_ARG;
    CPC_LAUNCH_CHECK();
    return 0;
}

// Forward attention kernel of the transformer layer (S <= 128): 1 (default) the two-workgroups-per-CU form, 0 the one-tile-in-LDS
// form of rounds 1-5.  Bit-identical outputs and saved probabilities.
extern "C" int cpc_set_attn_fwd(int variant) {
    CPC_RETURN_IF(variant != 0 && variant != 1, CPC_ERR_ARG);
    cpc::g_attn_fwd = variant;
    return 0;
}

// Test helper: out[i] = keep_i / (1 - p) for element i of dropout site `site` (0: attention probabilities, flat
// ((b*8 + head)*S + i)*S + j, n a multiple of S*S; 1: feed-forward hidden layer, flat row*2048 + col, S ignored) under `seed`
// -- the mask a cpc_transformer_layer_forward_dropout call with that seed applies.
extern "C" int cpc_dropout_keep_mask(float* out, long n, int site, int S, float p, unsigned long long seed, void* stream) {
    CPC_RETURN_IF(!out || n <= 0 || (site != 0 && site != 1) || !(p >= 0.f && p < 1.f), CPC_ERR_ARG);
    CPC_RETURN_IF(site == 0 && (S <= 0 || S > kSmax || n % ((long)S * S) != 0), CPC_ERR_SHAPE);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, n, site, S, p, seed);
    CPC_LAUNCH_CHECK();
    return 0;
}
