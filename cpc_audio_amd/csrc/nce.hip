// InfoNCE criterion: K linear prediction heads, shared negatives, batched dot-product scoring,
// per-head cross entropy against class 0 (the positive) and arg-max accuracy.
//
// Reference: cpc/criterion/criterion.py
//   :90-91,108   pred_k = W_k c_t                      (nn.Linear, bias-free)
//   :174-219     sampleClean: negatives z[ext[b,n,t]] drawn ONCE, shared by all K heads;
//                positive for head k is z[b, t+k]
//   :115-116     score = mean over the 256 features of pred_k * candidate   (dot / 256)
//   :248-257     CE(target 0) averaged over the B*W rows; acc = [argmax == 0] averaged
//
// The reference materialises (B, 1+N, W, 256) candidates per head (11.8 GB at B = 64).
// Here nothing of that size exists: for every (b,t) one wavefront forms the dense
// (heads x negatives) score matrix  P[16 x 256] . Neg^T[256 x N]  on the matrix pipe
// (v_mfma_f32_16x16x4_f32; the 12 heads are padded to the 16-row tile), gathering the
// negative rows of z straight from L2 / Infinity Cache into MFMA B-fragments (z for the
// whole batch is 8.4 MB), and folds the log-softmax online.  Only the (B*W, K, 1+N)
// logits and K log-sum-exps per row are kept for the backward pass.
//
// Backward per (b,t):
//   dPred[16 x 256] = dS[16 x N] . Neg[N x 256]        (gather again, MFMA; dS is kept, 64 bytes per candidate)
// then the head GEMMs:  dc = dPred . W,  dW_k = dPred_k^T . c   (gemm.hip).
// dz -- a scatter of dS[slot,k] * pred_k(window of slot) over random destination rows -- is re-associated through c for
// the linear heads (pred_k = W_k c):
//   dz[j] = sum_k W_k . G[j,k,:],   G[j,k,:] = sum over the slots landing on row j of dS[slot,k] * c[window(slot)]
// i.e. per destination row a gather-GEMM  G_j[16 x 256] = dS_j^T[16 x n_j] . C_j[n_j x 256]  over the destination-sorted
// slot list (1 KB rows of c from L2 / Infinity Cache: the forward kernel's shape mirrored), then ONE dense GEMM
// (B*S x K*256) . (K*256 x 256) with the stacked heads.  HBM: ~100 MB of G + 66 MB of dS at B = 64, no atomics.
// Predictions of a foreign network (cpc_nce_scores_*) are not linear in c: their dz goes through per-candidate gradient
// rows V (1 GB at B = 64) and a destination-sorted gather.
#include <algorithm>
#include "cpc_common.h"
#include "cpc_internal.h"
#include "gemm_tile.h"

namespace cpc {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// ------------------------------------------------------------------ forward scores
// Sixteen 1 KB rows -> this lane's fragments of them as an MFMA operand whose non-contracted index is the ROW (lane & 15 =
// row, lane >> 4 = k group):  out[ii] = floats 16 ii + 4 kq .. + 3 of row i.  Read in that layout directly, the 64 lanes of a
// load touch 16 rows x four 16-byte pieces and the coalescer -- which merges adjacent lanes only -- makes 64 cache accesses of
// it: PMC showed nce_fwd_kernel bound by the vector L1's tag lookups (84 M accesses per launch at B = 64, 330 k clocks per CU of
// a 390 k-clock kernel; the backward kernels, whose gathered rows are the operand with the CHANNEL in the low lane bits, make 19
// per instruction).  So the rows are loaded the coalesced way -- lane (c = lane & 15, r4 = lane >> 4) reads piece c of rows
// 4 q + r4: 256 contiguous bytes per row group, 16 accesses per instruction -- and the 16 x 16 transpose of 16-byte pieces goes
// through a 4 KB LDS tile per wave, one 256-byte column block (four fragments) at a time.  Piece (row, chunk) sits in slot
// row * 16 + (chunk ^ row): writes cover whole 256-byte rows, reads take 16 distinct slots per 16 lanes -- no bank conflicts.
struct Gather16 {
    float4 v[4][4];                                   // [q][g]: piece c of row 4 q + r4, column block g
    __device__ __forceinline__ void issue(const float* const (&rowp)[4]) {     // rowp[q]: row 4 q + r4, + 4 c floats
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) v[q][g] = ld4(rowp[q] + 64 * g);
    }
    // fragments 4 g .. 4 g + 3 (ii = 4 g + e) of row i; convergent: the whole wave calls it
    __device__ __forceinline__ void block(int g, float4* tile, float4 (&out)[4]) const {
        const int lane = threadIdx.x & 63, c = lane & 15, r4 = lane >> 4;
        __builtin_amdgcn_wave_barrier();                // the previous block's reads are done
#pragma unroll
        for (int q = 0; q < 4; ++q) tile[(4 * q + r4) * 16 + (c ^ (4 * q + r4))] = v[q][g];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = tile[c * 16 + ((4 * e + r4) ^ c)];        // here c plays the row i, r4 the k group
    }
};

// one wavefront per (b,t) row; 4 rows per block.  pred: [BW][K*C]; ext: [BW][N] row ids into z.
//
__global__ __launch_bounds__(256, 3) void nce_fwd_kernel(
    const float* __restrict__ pred, const float* __restrict__ z, const int* __restrict__ ext,
    float* __restrict__ logits, float* __restrict__ lse_out, float* __restrict__ rowstat, int BW, int W,
    int S, int K, int N, unsigned* __restrict__ ticket, int Nv, int koff) {
    __shared__ float4 tiles[4][256];
    const int lane = threadIdx.x & 63;
    const int bt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0u;      // nce_reduce_finalize_kernel, the next launch on this stream
    if (bt >= BW) return;                           // whole wave leaves together
    float4* tile = tiles[threadIdx.x >> 6];
    const int b = bt / W, t = bt - b * W;
    const int i = lane & 15, kq = lane >> 4;        // i: head (and candidate row of the A operand); kq: k group
    const bool hv = i < K;
    const float inv = 1.0f / kC;
    Gather16 gt;
    const float* rowp[4];

    float4 pa[16];                                  // pred[head i][16 ii + 4 kq ..]
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int head = 4 * q + kq;
            rowp[q] = pred + ((long)bt * K + (head < K ? head : 0)) * kC + 4 * i;
        }
        gt.issue(rowp);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 o[4];
            gt.block(g, tile, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) pa[4 * g + e] = hv ? o[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // positives: head h <-> z[b, t+h+1].  They go through the SAME MFMA chain as the negatives
    // (a 16-row tile whose row j is head j's positive row; the diagonal is kept), so a
    // negative that happens to be the positive row scores bit-identically and the arg-max tie
    // resolves to class 0 exactly as in the reference (criterion.py:253).
    auto score_tile = [&]() __attribute__((always_inline)) {      // gt holds 16 rows of z: scores of row i against the heads
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 zf[4];
            gt.block(g, tile, zf);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(zf[e], jj), f4c(pa[4 * g + e], jj), acc, 0, 0, 0);
        }
        return acc;
    };
    float posl;
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int head = 4 * q + kq;
            rowp[q] = z + ((long)b * S + t + koff + (head < K ? head : 0) + 1) * kC + 4 * i;
        }
        gt.issue(rowp);
        const f32x4 acc = score_tile();
        // acc[r] on lane (i, q) = score(positive row of head 4q+r, head i); the diagonal sits on lane (i, i >> 2), reg i & 3
        const float mine = (i & 3) == 0 ? acc[0] : (i & 3) == 1 ? acc[1] : (i & 3) == 2 ? acc[2] : acc[3];
        posl = __shfl(mine, i + 16 * (i >> 2)) * inv;
    }
    float M = posl;                                 // reference of the softmax weights, common to the four lanes of a head
    float ssum = kq == 0 ? 1.0f : 0.0f;             // the positive enters the sum once (exp(posl - M) = 1)
    float mneg = -3.0e38f;
    for (int nt = 0; nt < N / 16; ++nt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rowp[q] = z + (long)ext[(long)bt * N + nt * 16 + 4 * q + kq] * kC + 4 * i;
        gt.issue(rowp);
        const f32x4 acc = score_tile();
        // acc[r] = score of head i against negative nt*16 + 4 kq + r (padding candidates -- index >= Nv -- weigh nothing)
        float l[4], lmax = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            l[r] = nt * 16 + 4 * kq + r < Nv ? acc[r] * inv : -3.0e38f;
            if (hv) logits[((long)bt * K + i) * (N + 1) + 1 + nt * 16 + 4 * kq + r] = l[r];
            lmax = fmaxf(lmax, l[r]);
        }
        mneg = fmaxf(mneg, lmax);
        if (__any(lmax - M > 40.0f)) {              // (wave-uniform) move the reference: rare
            float tm = fmaxf(lmax, __shfl_xor(lmax, 16));
            tm = fmaxf(tm, __shfl_xor(tm, 32));
            const float Mn = fmaxf(M, tm), alpha = expf(M - Mn);
            ssum *= alpha;
            M = Mn;
        }
        float pw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pw[r] = expf(l[r] - M);
            ssum += pw[r];
        }
    }
    // fold the four k groups of a head
    ssum += __shfl_xor(ssum, 16);
    ssum += __shfl_xor(ssum, 32);
    mneg = fmaxf(mneg, __shfl_xor(mneg, 16));
    mneg = fmaxf(mneg, __shfl_xor(mneg, 32));
    const float lse = M + logf(ssum);
    if (kq == 0 && hv) {
        logits[((long)bt * K + i) * (N + 1)] = posl;
        lse_out[(long)bt * K + i] = lse;
        rowstat[(long)bt * 2 * K + i] = lse - posl;                       // CE(target 0)
        rowstat[(long)bt * 2 * K + K + i] = posl >= mneg ? 1.f : 0.f;     // argmax == 0
    }
}

// ------------------------------------------------------------------ forward scores + the prediction gradient, ONE gather pass
// The backward needs dPred[head][:] = sum_n dS[head][n] * cand[n][:] with dS = g_head * (softmax - [n = 0]) -- the same 1 KB
// candidate rows the scores were just formed from (1.06 GB of gathers at B = 64, of which 625 MB miss the 4 MB L2 and come out of
// the Infinity Cache: nce_bwd_dpred_kernel spent 170 us of the main stream on fetching them a second time).  The softmax weights
// of a window are only known after its last candidate, but the weighted row sum can be carried along unnormalised, exactly as
// the denominator is (the "online softmax" of the scores above):
//     U[head][:] = sum_n exp(l[head][n] - M[head]) * neg[n][:]         (rescaled by exp(M - M') whenever the reference moves)
//     T[head][:] = exp(M - lse) * U[head][:] + (p0 - 1) * pos[head][:]  = d loss_head / d pred_head   for a UNIT upstream gradient
// so this kernel does in one pass over the gathered rows what nce_fwd_kernel + nce_bwd_dpred_kernel did in two: the rows a
// lane group has just gathered for the transposition ARE the B operand of the second product (candidate = contraction index,
// channel in the low lane bits), and the weights exp(l - M) come out of the first product in the A-operand layout -- once
// lane group r4 gathers candidates 4 r4 + q instead of 4 q + r4 -- so the second product costs 64 more MFMAs per 16-candidate
// tile and no data movement.  The per-head upstream gradients g_head (gloss / (B W C)) are applied by the consumers: the dc GEMM
// reads the stacked head weights pre-multiplied by g_head, the heads' weight gradient is scaled per 256-row block on its way
// out of the split reduction, the dz path multiplies the softmax rows this kernel leaves per candidate slot.  T replaces dPred.
struct Gather16R {                                      // Gather16 with lane group r4 holding rows 4 r4 + q
    float4 v[4][4];                                   // [q][g]: piece c of row 4 r4 + q, column block g
    __device__ __forceinline__ void issue(const float* const (&rowp)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) v[q][g] = ld4(rowp[q] + 64 * g);
    }
    __device__ __forceinline__ void block(int g, float4* tile, float4 (&out)[4]) const {
        const int lane = threadIdx.x & 63, c = lane & 15, r4 = lane >> 4;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) tile[(4 * r4 + q) * 16 + (c ^ (4 * r4 + q))] = v[q][g];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = tile[c * 16 + ((4 * e + r4) ^ c)];
    }
};

// as nce_fwd_kernel, plus tpred [BW][K*C] = T and its max|.| into tamax (kAmaxSlots slots, zeroed by the caller)
__global__ __launch_bounds__(256, 2) void nce_fwd_fused_kernel(
    const float* __restrict__ pred, const float* __restrict__ z, const int* __restrict__ ext,
    float* __restrict__ logits, float* __restrict__ lse_out, float* __restrict__ rowstat, float* __restrict__ tpred,
    float* __restrict__ tamax, float* __restrict__ ps, int BW, int W, int S, int K, int N, unsigned* __restrict__ ticket,
    int Nv, int koff) {
    __shared__ float4 tiles[4][256];
    const int lane = threadIdx.x & 63;
    const int bt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0u;
    if (bt >= BW) return;
    float4* tile = tiles[threadIdx.x >> 6];
    const int b = bt / W, t = bt - b * W;
    const int i = lane & 15, kq = lane >> 4;
    const bool hv = i < K;
    const float inv = 1.0f / kC;
    Gather16R gt;
    const float* rowp[4];

    float4 pa[16];                                  // pred[head i][16 ii + 4 kq ..]
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int head = 4 * kq + q;
            rowp[q] = pred + ((long)bt * K + (head < K ? head : 0)) * kC + 4 * i;
        }
        gt.issue(rowp);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 o[4];
            gt.block(g, tile, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) pa[4 * g + e] = hv ? o[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    auto score_tile = [&]() __attribute__((always_inline)) {      // gt holds 16 rows of z: scores of row i against the heads
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 zf[4];
            gt.block(g, tile, zf);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(zf[e], jj), f4c(pa[4 * g + e], jj), acc, 0, 0, 0);
        }
        return acc;
    };
    float posl;
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int head = 4 * kq + q;
            rowp[q] = z + ((long)b * S + t + koff + (head < K ? head : 0) + 1) * kC + 4 * i;
        }
        gt.issue(rowp);
        const f32x4 acc = score_tile();
        // acc[r] on lane (i, q) = score(positive row of head 4q+r, head i); the diagonal sits on lane (i, i >> 2), reg i & 3
        const float mine = (i & 3) == 0 ? acc[0] : (i & 3) == 1 ? acc[1] : (i & 3) == 2 ? acc[2] : acc[3];
        posl = __shfl(mine, i + 16 * (i >> 2)) * inv;
    }
    float M = posl;
    float ssum = kq == 0 ? 1.0f : 0.0f;
    float mneg = -3.0e38f;
    f32x4 U[16];                                     // U[4 g + e][r]: head 4 kq + r, channel 64 g + 4 i + e
#pragma unroll
    for (int q = 0; q < 16; ++q) U[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int nt = 0; nt < N / 16; ++nt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rowp[q] = z + (long)ext[(long)bt * N + nt * 16 + 4 * kq + q] * kC + 4 * i;
        gt.issue(rowp);
        const f32x4 acc = score_tile();
        // acc[r] = score of head i against negative nt*16 + 4 kq + r (padding candidates -- index >= Nv -- weigh nothing)
        float l[4], lmax = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            l[r] = nt * 16 + 4 * kq + r < Nv ? acc[r] * inv : -3.0e38f;
            if (hv) logits[((long)bt * K + i) * (N + 1) + 1 + nt * 16 + 4 * kq + r] = l[r];
            lmax = fmaxf(lmax, l[r]);
        }
        mneg = fmaxf(mneg, lmax);
        if (__any(lmax - M > 40.0f)) {              // (wave-uniform) move the reference: rare
            float tm = fmaxf(lmax, __shfl_xor(lmax, 16));
            tm = fmaxf(tm, __shfl_xor(tm, 32));
            const float Mn = fmaxf(M, tm), alpha = expf(M - Mn);
            ssum *= alpha;
            M = Mn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {           // head 4 kq + r's factor lives on lane 4 kq + r (any lane group)
                const float ar = __shfl(alpha, 4 * kq + r);
#pragma unroll
                for (int q = 0; q < 16; ++q) U[q][r] *= ar;
            }
        }
        float pw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pw[r] = hv ? expf(l[r] - M) : 0.f;      // (padding heads: exactly 0 -- their logits alias head 0's row)
            ssum += hv ? pw[r] : expf(l[r] - M);    // (their sums are never read; kept finite either way)
        }
        // second product: U[head][channel] += sum over this tile's candidates of exp(l - M) * row.  Call q contracts the four
        // candidates 4 r4 + q (one per lane group): A = pw[q] on lane (head i, r4 = kq), B = the row lane group r4 gathered
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    U[4 * g + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[q], f4c(gt.v[q][g], e), U[4 * g + e], 0, 0, 0);
    }
    ssum += __shfl_xor(ssum, 16);
    ssum += __shfl_xor(ssum, 32);
    mneg = fmaxf(mneg, __shfl_xor(mneg, 16));
    mneg = fmaxf(mneg, __shfl_xor(mneg, 32));
    const float lse = M + logf(ssum);
    if (kq == 0 && hv) {
        logits[((long)bt * K + i) * (N + 1)] = posl;
        lse_out[(long)bt * K + i] = lse;
        rowstat[(long)bt * 2 * K + i] = lse - posl;
        rowstat[(long)bt * 2 * K + K + i] = posl >= mneg ? 1.f : 0.f;
    }
    // ---- T = exp(M - lse) * U + (p0 - 1) * pos;  C layout of U: head = 4 kq + r, channel = 64 g + 4 i + e
    const float fac = expf(M - lse), p0m1 = expf(posl - lse) - 1.0f;       // per head i (equal over the lane groups)
    float amax = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int head = 4 * kq + r;
        const float fr = __shfl(fac, head), dr = __shfl(p0m1, head);       // convergent: before the guard
        if (head < K) {
            const float* zp = z + ((long)b * S + t + koff + head + 1) * kC + 4 * i;
            float* op = tpred + ((long)bt * K + head) * kC + 4 * i;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 zv = ld4(zp + 64 * g);
                float4 o;
                o.x = fmaf(dr, zv.x, fr * U[4 * g + 0][r]);
                o.y = fmaf(dr, zv.y, fr * U[4 * g + 1][r]);
                o.z = fmaf(dr, zv.z, fr * U[4 * g + 2][r]);
                o.w = fmaf(dr, zv.w, fr * U[4 * g + 3][r]);
                *reinterpret_cast<float4*>(op + 64 * g) = o;
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
        }
    }
    amax = wave_max(amax);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(tamax + (bt & (kAmaxSlots - 1))), __float_as_uint(amax));
    // ---- the softmax itself, one 64-byte row of 16 heads per candidate slot bt * (N + K) + j (j < N: exp(l - lse); j >= N: the
    // positive of head j - N, p0 - 1 on its own head, 0 elsewhere): what the re-associated dz path (nce_bwd_g_kernel) contracts
    // with the rows of c, times the heads' upstream gradients.  The logits of this window are re-read (just written, 6 KB).
    float* prow = ps + (long)bt * (N + K) * 16 + i;
    const float* lrow = logits + ((long)bt * K + (hv ? i : 0)) * (N + 1) + 1 + 4 * kq;
    for (int nt = 0; nt < N / 16; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = hv ? expf(lrow[nt * 16 + r] - lse) : 0.f;
            prow[(long)(nt * 16 + 4 * kq + r) * 16] = pv;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int head = 4 * kq + r;
        const float dr = __shfl(p0m1, head);
        if (head < K) prow[(long)(N + head) * 16] = i == head ? dr : 0.f;
    }
}

// ------------------------------------------------------------------ the same pass on the 16-bit matrix pipe (round 6)
// nce_fwd_fused_kernel spends 128 exact-f32 MFMAs (4096 matrix-pipe cycles) per 16-candidate tile and holds every gathered row
// in registers (252 VGPRs).  Here both products run on fp16 pieces -- x s = h + l, hh + hl + lh, the arithmetic of the conv
// layers (gemm_tile.h) -- from a gather source kept in H2 storage (cpc_common.h: the same 1 KB rows, two fp16 pieces per
// channel): zh, a copy of z made by nce_rows_to_h2_kernel.  A tile's sixteen rows go global -> LDS by DMA (global_load_lds_dwordx4,
// one row per wave instruction, no VGPRs), and both products read their fragments from there:
//   * scores (contraction over channels): A = the rows as they lie -- lane (row i, k group kq) reads the 16-byte h and l pieces
//     of channels 32 s + 8 kq .. -- against P (the window's K predictions, split once per window, 64 VGPRs);
//     24 v_mfma_f32_16x16x32_f16 per tile;
//   * weighted row sum (contraction over CANDIDATES): the B operand wants four candidates of one channel per lane -- the
//     transpose of how the rows lie -- which ds_read_b64_tr_b16 hands out directly (lane q of a 16-lane group points at row
//     4 kq + (q >> 2), channels 4 (q & 3).., and receives rows 4 kq .. + 3 of channel q); A = the softmax weights exp(l - M) of
//     (head i, candidates 4 kq ..) exactly where the first product's accumulator left them; 48 v_mfma_f32_16x16x16_f16 per tile.
// LDS image of a tile: 16-byte piece p (0..63) of row r at slot p ^ swz(r), swz(r) = (r & 1) | (r & 6) << 1 -- the DMA lane that
// fills slot L fetches piece L ^ swz(r); found by search over linear swizzles: both read patterns then touch every bank once
// per LDS cycle (ds_read_b128's 16-lane service groups, the transposing read's 32-lane halves; MI355X_MICROARCH.md, LDS).
// The weights: exp(l - M) <= e^4 with the reference M moved whenever a logit exceeds it by 4 (f32 kernel: 40), times 1024 -> fp16
// range; the sum of weights is >= 1 at all times (the positive, or the element that moved M), so a piece that underflows (2^-35
// absolute) is 2^-35 of the softmax's denominator.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4t __attribute__((ext_vector_type(4)));
constexpr int kRowBytes = kC * 4;                  // an fp32 row and an H2 row alike
constexpr int kTileBytes = 16 * kRowBytes;
constexpr float kPwScale = 1024.0f;                // the softmax weights are split as fp16 pieces of 1024 exp(l - M) (<= 55.9 k)
constexpr float kMoveRef = 4.0f;                   // the reference M moves when a logit exceeds it by this much
__host__ __device__ constexpr int tile_swz(int r) { return (r & 1) | ((r & 6) << 1); }

// sixteen 1 KB rows `rows[r]` (wave-uniform) of `base` -> the wave's LDS tile
// (`tile` must be wave-uniform in a way the compiler can see -- derived from a readfirstlane -- so that the sixteen LDS bases are
// scalar additions; the global side is a uniform 64-bit row address + a 32-bit lane offset, eight distinct ones per wave)
__device__ __forceinline__ void tile_dma16(const unsigned char* __restrict__ base, const int (&rows)[16], unsigned char* tile) {
    const unsigned lane = threadIdx.x & 63;
    __builtin_amdgcn_wave_barrier();                 // every lane's reads of the tile's previous contents are issued (convergent)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned char* rowp = base + (long)rows[r] * kRowBytes;
        dma16_to_lds(rowp + 16u * (lane ^ (unsigned)tile_swz(r)), tile + r * kRowBytes);
    }
}
__device__ __forceinline__ f16x4 tile_read_tr(const unsigned char* p) {
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4t __attribute__((address_space(3)))*)(p)));
}
// Lane addresses into a tile.  p1(c): this lane's 16-byte piece c + 2 kq of row i (product 1's fragment / an fp32 row's float4;
// c = 8 s + hl).  p2(c): the 8 bytes this lane points the transposing read at for pieces c + 2 ((q & 3) >> 1) of row
// 4 kq + (q >> 2), q = i (c = 4 ct + hl).  Slot = c ^ L with L < 16, so c's bits 4, 5 are an immediate offset and its low bits
// select one of 4 (c & 9) / 8 (c & 13) precomputed addresses: c must be a compile-time constant.
struct TileAddr {
    const unsigned char* a1[4];
    const unsigned char* a2[8];
    __device__ __forceinline__ TileAddr(const unsigned char* tile, int i, int kq) {
        const int L1 = (2 * kq) ^ tile_swz(i);
        const int rho = 4 * kq + (i >> 2);
        const int L2 = (2 * ((i & 3) >> 1)) ^ tile_swz(rho);
#pragma unroll
        for (int m = 0; m < 4; ++m) a1[m] = tile + i * kRowBytes + 16 * (((m & 1) | ((m & 2) << 2)) ^ L1);
#pragma unroll
        for (int m = 0; m < 8; ++m) a2[m] = tile + rho * kRowBytes + 8 * (i & 1) + 16 * (((m & 1) | ((m & 6) << 1)) ^ L2);
    }
    __device__ __forceinline__ const unsigned char* p1(int c) const { return a1[(c & 1) | ((c & 8) >> 2)] + 16 * (c & 0x30); }
    __device__ __forceinline__ const unsigned char* p2(int c) const { return a2[(c & 1) | ((c & 12) >> 1)] + 16 * (c & 0x30); }
};
struct F16Pair { f16x8 h, l; };
__device__ __forceinline__ F16Pair split8(const float4& u, const float4& v, float s) {
    const float x[8] = {u.x * s, u.y * s, u.z * s, u.w * s, v.x * s, v.y * s, v.z * s, v.w * s};
    F16Pair f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hi = (_Float16)x[e];
        f.h[e] = hi;
        f.l[e] = (_Float16)(x[e] - (float)hi);
    }
    return f;
}
__device__ __forceinline__ void split4(const float (&v)[4], float s, f16x4& h, f16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e] * s;
        const _Float16 hi = (_Float16)x;
        h[e] = hi;
        l[e] = (_Float16)(x - (float)hi);
    }
}

// fp32 rows -> H2 rows scaled for the bound in `bound` (kAmaxSlots partial maxima): the gather source of the kernels below.
// One wave per row; plain stores (the copy is gathered from right away: it should stay in L2).
__global__ __launch_bounds__(256) void nce_rows_to_h2_kernel(const float* __restrict__ x, unsigned char* __restrict__ xh,
                                                             const float* __restrict__ bound, long rows) {
    const int lane = threadIdx.x & 63;
    const float s = scale_for_amax(fold_amax(bound, kAmaxSlots));
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4 v = ld4(x + row * kC + 4 * lane);
    _Float16 h0, h1, h2, h3, l0, l1, l2, l3;
    h2_split(v.x, s, h0, l0); h2_split(v.y, s, h1, l1); h2_split(v.z, s, h2, l2); h2_split(v.w, s, h3, l3);
    const unsigned hw0 = __builtin_bit_cast(unsigned, f16x2{h0, h1}), hw1 = __builtin_bit_cast(unsigned, f16x2{h2, h3});
    const unsigned lw0 = __builtin_bit_cast(unsigned, f16x2{l0, l1}), lw1 = __builtin_bit_cast(unsigned, f16x2{l2, l3});
    const bool odd = lane & 1;                       // (h2_store_row_nt's neighbour swap, with a plain store)
    const unsigned r0 = __builtin_bit_cast(unsigned, dpp_mov<0xB1>(__builtin_bit_cast(float, odd ? hw0 : lw0)));
    const unsigned r1 = __builtin_bit_cast(unsigned, dpp_mov<0xB1>(__builtin_bit_cast(float, odd ? hw1 : lw1)));
    f32x4 o;
    o.x = __builtin_bit_cast(float, odd ? r0 : hw0);
    o.y = __builtin_bit_cast(float, odd ? r1 : hw1);
    o.z = __builtin_bit_cast(float, odd ? lw0 : r0);
    o.w = __builtin_bit_cast(float, odd ? lw1 : r1);
    *reinterpret_cast<f32x4*>(xh + row * kRowBytes + 16 * lane) = o;
}

// The softmax rows per candidate slot (ps: what the dz path contracts with c, nce_fwd_fused_kernel's layout) from the saved logits
// and log-sum-exps, as a launch of its own: nothing of the forward's chain reads them, so a caller with a second stream runs this
// beside the backward's first GEMMs instead of inside the scoring kernel (20 us of its 175 at B = 64).  One wave per window;
// lane L takes candidate L / 4 of a 16-candidate tile and heads 4 (L % 4) .. + 3: four strided loads, one 16-byte store -- a
// tile's sixteen 64-byte rows leave as ONE contiguous 1 KB per wave instruction.
__global__ __launch_bounds__(256) void nce_softmax_rows_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                               float* __restrict__ ps, int BW, int K, int N) {
    const int lane = threadIdx.x & 63;
    const int bt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bt >= BW) return;
    const int cand = lane >> 2, h0 = 4 * (lane & 3);
    float ls[4];
    const float* lp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int head = h0 + e < K ? h0 + e : 0;
        ls[e] = lse[(long)bt * K + head];
        lp[e] = logits + ((long)bt * K + head) * (N + 1);
    }
    float* prow = ps + (long)bt * (N + K) * 16 + 4 * lane;
    for (int nt0 = 0; nt0 < N / 16; nt0 += 4) {
        float lv[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) lv[q][e] = nt0 + q < N / 16 ? lp[e][1 + (nt0 + q) * 16 + cand] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (nt0 + q < N / 16) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = h0 + e < K ? expf(lv[q][e] - ls[e]) : 0.f;
                __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(prow + (long)(nt0 + q) * 256));
            }
        }
    }
    // the positives: slot N + head carries p0 - 1 on its own head, 0 elsewhere
    if (cand < K) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = h0 + e == cand ? expf(lp[e][0] - ls[e]) - 1.0f : 0.f;
        *reinterpret_cast<f32x4*>(ps + ((long)bt * (N + K) + N + cand) * 16 + h0) = o;
    }
}

// as nce_fwd_fused_kernel; zh / zbound: the H2 copy of z and the bound it was scaled for.  The grid may be smaller than the
// number of window quads: a workgroup then walks windows 4 blockIdx.x + wave, + 4 gridDim.x, ... (cpc_set_nce_grid)
__global__ __launch_bounds__(256, 2) void nce_fwd_h2_kernel(
    const float* __restrict__ pred, const float* __restrict__ z, const unsigned char* __restrict__ zh,
    const float* __restrict__ zbound, const int* __restrict__ ext, float* __restrict__ logits, float* __restrict__ lse_out,
    float* __restrict__ rowstat, float* __restrict__ tpred, float* __restrict__ tamax, float* __restrict__ ps, int BW, int W,
    int S, int K, int N, unsigned* __restrict__ ticket, int Nv, int koff, int dbg) {
    __shared__ __attribute__((aligned(16))) unsigned char tiles[4][kTileBytes];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0u;
    unsigned char* tile = tiles[wv];
    const int i = lane & 15, kq = lane >> 4;
    const bool hv = i < K;
    // product 1's fragment of row i: pieces 8 s + 2 kq (+ 1) -> slot (8 s + hl) ^ L1;  the transposing read of product 2: lane
    // q = i of group kq points at row 4 kq + (q >> 2), pieces 4 ct + 2 ((q & 3) >> 1) (+ 1), half q & 1 -> slot (4 ct + hl) ^ L2
    // (the swizzle touches bits 0..3 of the slot only: the bits of the compile-time part above them are an immediate offset, the
    //  rest gives 4 / 8 distinct lane addresses -- kept in registers instead of one address per read)
    const TileAddr ta(tile, i, kq);
    const float sz = scale_for_amax(fold_amax(zbound, kAmaxSlots));
    float amax_t = 0.f;
    for (int btv = blockIdx.x * 4 + wv; btv < BW; btv += gridDim.x * 4) {
        const int bt = __builtin_amdgcn_readfirstlane(btv);
        const int b = bt / W, t = bt - b * W;
        int rows[16];
        // ---- P: the window's predictions, through the tile, split once
        F16Pair pp[8];
        float inv;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) rows[r] = bt * K + (r < K ? r : 0);
            tile_dma16(reinterpret_cast<const unsigned char*>(pred), rows, tile);
            CPC_WAIT_VMCNT(0);
            __builtin_amdgcn_wave_barrier();
            float m = 0.f;                                          // (two passes over the tile: the scale needs the window's maximum)
#pragma unroll
            for (int q = 0; q < 16; ++q) {                          // q = 2 s + half: channels 32 s + 8 kq + 4 half ..
                const float4 v = *reinterpret_cast<const float4*>(ta.p1((q >> 1) * 8 + (q & 1)));
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
            const float sp = scale_for_amax(wave_max(hv ? m : 0.f));
            inv = 1.0f / (kC * sz * sp);                            // (powers of two: exact)
            const float spv = hv ? sp : 0.f;                        // (padding heads alias head 0's row: exactly 0)
#pragma unroll
            for (int sx = 0; sx < 8; ++sx)
                pp[sx] = split8(*reinterpret_cast<const float4*>(ta.p1(8 * sx)), *reinterpret_cast<const float4*>(ta.p1(8 * sx + 1)), spv);
        }
        auto score_tile = [&]() __attribute__((always_inline)) {   // the tile holds 16 rows of zh: scores of row 4 kq + r against head i
            f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll
            for (int sx = 0; sx < 8; ++sx) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(ta.p1(8 * sx));
                const f16x8 al = *reinterpret_cast<const f16x8*>(ta.p1(8 * sx + 1));
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, pp[sx].h, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, pp[sx].l, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, pp[sx].h, a2, 0, 0, 0);
            }
            f32x4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = (a0[r] + (a1[r] + a2[r])) * inv;
            return acc;
        };
        // ---- positives: head h <-> z[b, t + h + 1], through the SAME chain as the negatives (a negative that happens to be the
        // positive row scores bit-identically: the arg-max tie resolves to class 0 as in the reference, criterion.py:253)
        float posl;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) rows[r] = b * S + t + koff + (r < K ? r : 0) + 1;
            tile_dma16(zh, rows, tile);
            CPC_WAIT_VMCNT(0);
            __builtin_amdgcn_wave_barrier();
            const f32x4 acc = score_tile();
            const float mine = (i & 3) == 0 ? acc[0] : (i & 3) == 1 ? acc[1] : (i & 3) == 2 ? acc[2] : acc[3];
            posl = __shfl(mine, i + 16 * (i >> 2));
        }
        float M = posl;
        float ssum = kq == 0 ? 1.0f : 0.0f;
        float mneg = -3.0e38f;
        f32x4 U[16];                                     // U[ct][r]: head 4 kq + r, channel 16 ct + i
#pragma unroll
        for (int q = 0; q < 16; ++q) U[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int nt = 0; nt < N / 16; ++nt) {
            const int* e = ext + (long)bt * N + nt * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) rows[r] = e[r];
            tile_dma16(zh, rows, tile);
            CPC_WAIT_VMCNT(0);
            __builtin_amdgcn_wave_barrier();
            const f32x4 acc = score_tile();
            float l[4], lmax = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                l[r] = nt * 16 + 4 * kq + r < Nv ? acc[r] : -3.0e38f;
                if (hv && !(dbg & 8)) logits[((long)bt * K + i) * (N + 1) + 1 + nt * 16 + 4 * kq + r] = l[r];
                lmax = fmaxf(lmax, l[r]);
            }
            mneg = fmaxf(mneg, lmax);
            if (__any(lmax - M > kMoveRef)) {              // (wave-uniform) move the reference
                float tm = fmaxf(lmax, __shfl_xor(lmax, 16));
                tm = fmaxf(tm, __shfl_xor(tm, 32));
                const float Mn = fmaxf(M, tm), alpha = expf(M - Mn);
                ssum *= alpha;
                M = Mn;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ar = __shfl(alpha, 4 * kq + r);
#pragma unroll
                    for (int q = 0; q < 16; ++q) U[q][r] *= ar;
                }
            }
            float pw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pw[r] = hv ? expf(l[r] - M) : 0.f;
                ssum += hv ? pw[r] : expf(l[r] - M);
            }
            f16x4 wh, wl;
            split4(pw, kPwScale, wh, wl);
            if (!(dbg & 4))
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {                // four channel tiles at a time: 8 transposing reads, 12 MFMAs
                f16x4 bh[4], bl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ct = 4 * c4 + u;
                    bh[u] = tile_read_tr(ta.p2(4 * ct));
                    bl[u] = tile_read_tr(ta.p2(4 * ct + 1));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) U[4 * c4 + u] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh, bh[u], U[4 * c4 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) U[4 * c4 + u] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh, bl[u], U[4 * c4 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) U[4 * c4 + u] = __builtin_amdgcn_mfma_f32_16x16x16f16(wl, bh[u], U[4 * c4 + u], 0, 0, 0);
            }
        }
        ssum += __shfl_xor(ssum, 16);
        ssum += __shfl_xor(ssum, 32);
        mneg = fmaxf(mneg, __shfl_xor(mneg, 16));
        mneg = fmaxf(mneg, __shfl_xor(mneg, 32));
        const float lse = M + logf(ssum);
        if (kq == 0 && hv) {
            logits[((long)bt * K + i) * (N + 1)] = posl;
            lse_out[(long)bt * K + i] = lse;
            rowstat[(long)bt * 2 * K + i] = lse - posl;
            rowstat[(long)bt * 2 * K + K + i] = posl >= mneg ? 1.f : 0.f;
        }
        // ---- T = exp(M - lse) * U + (p0 - 1) * pos, out through the tile so that a head's 1 KB row leaves as one store instruction
        const float fac = expf(M - lse) / (kPwScale * sz), p0m1 = expf(posl - lse) - 1.0f;       // per head i
        __builtin_amdgcn_wave_barrier();                 // (the last tile's transposing reads are done: in order with the writes below)
        float* tf = reinterpret_cast<float*>(tile);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float fr = __shfl(fac, 4 * kq + r);
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) tf[(4 * kq + r) * kC + 16 * ct + i] = fr * U[ct][r];
        }
        __builtin_amdgcn_wave_barrier();
        if (!(dbg & 2)) {
            // (every positive row requested before the first is used: one trip to L2 per window instead of one per head -- U's
            //  registers are free by now)
            for (int h0 = 0; h0 < K; h0 += 8) {
                float4 zv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (h0 + q < K) zv[q] = ld4(z + ((long)b * S + t + koff + h0 + q + 1) * kC + 4 * lane);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (h0 + q < K) {                     // (wave-uniform)
                        const float dr = __shfl(p0m1, h0 + q);
                        const float4 u = *reinterpret_cast<const float4*>(tf + (h0 + q) * kC + 4 * lane);
                        f32x4 o;
                        o.x = fmaf(dr, zv[q].x, u.x); o.y = fmaf(dr, zv[q].y, u.y);
                        o.z = fmaf(dr, zv[q].z, u.z); o.w = fmaf(dr, zv[q].w, u.w);
                        *reinterpret_cast<f32x4*>(tpred + ((long)bt * K + h0 + q) * kC + 4 * lane) = o;
                        amax_t = fmaxf(fmaxf(amax_t, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                    }
                }
            }
        }
        // ---- the softmax rows per candidate slot (nce_fwd_fused_kernel's layout), from the logits just written
        float* prow = ps + (long)bt * (N + K) * 16 + i;
        const float* lrow = logits + ((long)bt * K + (hv ? i : 0)) * (N + 1) + 1 + 4 * kq;
        for (int nt0 = 0; nt0 < ((dbg & 1) ? 0 : N / 16); nt0 += 4) {       // (four tiles' logits requested before the first exp)
            float lv[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) lv[q][r] = nt0 + q < N / 16 ? lrow[(nt0 + q) * 16 + r] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (nt0 + q < N / 16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) prow[(long)((nt0 + q) * 16 + 4 * kq + r) * 16] = hv ? expf(lv[q][r] - lse) : 0.f;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int head = 4 * kq + r;
            const float dr = __shfl(p0m1, head);
            if (head < K) prow[(long)(N + head) * 16] = i == head ? dr : 0.f;
        }
        __builtin_amdgcn_wave_barrier();                 // (the tile is refilled by the next window's DMA)
    }
    amax_t = wave_max(amax_t);
    if (lane == 0 && amax_t > 0.f)
        atomicMax(reinterpret_cast<unsigned*>(tamax + ((blockIdx.x * 4 + wv) & (kAmaxSlots - 1))), __float_as_uint(amax_t));
}

// wallT_g[i][k*256 + o] = g_k * wall[(k*256 + o)*256 + i]: the stacked head weights transposed AND pre-multiplied by the heads'
// upstream gradients -- the B operand of dc = T . wallT_g^T (the one-pass criterion keeps T, the unit-gradient dPred)
__global__ __launch_bounds__(256) void nce_wallT_scaled_kernel(const float* __restrict__ wall, const float* __restrict__ gscale,
                                                               float* __restrict__ out, int K) {
    __shared__ float tile[32][33];
    const int R = K * kC;                               // rows of wall
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int q = ty; q < 32; q += 8) {
        const int r = r0 + q;
        tile[q][tx] = r < R ? wall[(long)r * kC + c0 + tx] * gscale[r >> kCLog2] : 0.f;
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int r = r0 + tx;
        if (r < R) out[(long)(c0 + q) * R + r] = tile[tx][q];
    }
}

// Column sums of rowstat [nrows][n <= 32] -> losses / accuracies, in ONE launch: block g sums its rows_per_group rows into
// tmp[g] (the arithmetic and order of rows_sum_kernel), takes a ticket, and the block that draws the last ticket folds the
// groups (again rows_sum_kernel's order) and scales.  `ticket` was cleared by the kernel that produced rowstat.  Replaces two
// rows_sum launches + nce_finalize_kernel on the step's critical path (three dependent 6 us launches).
__global__ __launch_bounds__(256) void nce_reduce_finalize_kernel(const float* __restrict__ rowstat, int nrows, int n,
                                                                 int rows_per_group, float* __restrict__ tmp,
                                                                 unsigned* __restrict__ ticket, float* __restrict__ losses,
                                                                 float* __restrict__ acc, int K, float inv_rows) {
    __shared__ float red[8][33];
    __shared__ unsigned drawn;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    auto fold = [&](auto load, int r0, int r1) __attribute__((always_inline)) {      // rows r0..r1-1 of one column
        float s0 = 0.f, s1 = 0.f;
        if (tx < n) {
            int r = r0 + ty;
            for (; r + 8 < r1; r += 16) {
                s0 += load(r);
                s1 += load(r + 8);
            }
            if (r < r1) s0 += load(r);
        }
        red[ty][tx] = s0 + s1;
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += red[q][tx];
        __syncthreads();
        return s;
    };
    const int r0 = blockIdx.x * rows_per_group;
    const float mine = fold([&](int r) { return rowstat[(long)r * n + tx]; }, r0, min(nrows, r0 + rows_per_group));
    if (ty == 0 && tx < n) tmp[(long)blockIdx.x * n + tx] = mine;
    __threadfence();                                    // my group's sums are visible device-wide before my ticket is
    __syncthreads();
    if (threadIdx.x == 0) drawn = atomicAdd(ticket, 1u);
    __syncthreads();
    if (drawn != gridDim.x - 1) return;
    __threadfence();
    // the groups' sums, all loads in flight before the first addition (dependent L2 round trips are what this block would
    // otherwise spend its time on); at most kRowsSumGroups / 8 = 16 per thread
    float v[kRowsSumGroups / 8];
#pragma unroll
    for (int j = 0; j < kRowsSumGroups / 8; ++j) {
        const int g = ty + 8 * j;
        v[j] = (tx < n && g < (int)gridDim.x) ? __hip_atomic_load(tmp + (long)g * n + tx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    }
    const float total = fold([&](int g) { return v[(g - ty) >> 3]; }, 0, (int)gridDim.x);
    if (ty == 0 && tx < n) {
        if (tx < K) losses[tx] = total * inv_rows;
        else if (tx < 2 * K) acc[tx - K] = total * inv_rows;
    }
}

// gscale[k] = dL/dloss_k / (B*W) / C
// fwd_bounds (linear heads; NULL otherwise): the forward's max|c|, max|wall| slots.  Then also gscale[17] = max|c|,
// gscale[18] = max|wall| (operand bounds of the backward GEMMs, GemmBounds), the 64 max|dPred| slots at gscale[64..127]
// are cleared for nce_bwd_dpred_kernel and the 64 max|G| slots at gscale[128..191] for nce_bwd_g_kernel.  (An a-priori bound -- |dPred| <= 2 max|gscale| max|z| -- needs no slots but is far
// too loose once the softmax is confident: operands 2^-20 of their bound lose the low fp16 piece.)
// Blocks 1.. (if any) zero dc_tail: the last S - W steps of every sequence of dc [B][S][256], which predict nothing and get no
// gradient from the GEMM that writes the other rows (this replaces a memset of the whole tensor on the critical path).
__global__ __launch_bounds__(64) void nce_gscale_kernel(const float* __restrict__ gloss, float* __restrict__ gscale,
                                                        int K, float f, const float* __restrict__ fwd_bounds,
                                                        float* __restrict__ dc_tail, int B, int S, int W) {
    if (blockIdx.x > 0) {
        const int per = (S - W) * (kC / 4);                           // float4 per sequence
        for (int i = (blockIdx.x - 1) * 64 + threadIdx.x; i < B * per; i += (gridDim.x - 1) * 64) {
            const int b = i / per, r = i - b * per;
            reinterpret_cast<float4*>(dc_tail + ((long)b * S + W) * kC)[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const int k = threadIdx.x;
    const float gk = k < K ? gloss[k] * f : 0.f;
    if (k < K) gscale[k] = gk;
    if (fwd_bounds != nullptr) {
        const float cm = wave_max(fwd_bounds[k]), wm = wave_max(fwd_bounds[kAmaxSlots + k]), gm = wave_max(fabsf(gk));
        if (k == 0) { gscale[17] = cm; gscale[18] = wm; gscale[19] = wm * gm; }    // [19]: bound of the heads' weights * g_k (fused)
        gscale[64 + k] = 0.f;
        gscale[128 + k] = 0.f;
    }
}

// ------------------------------------------------------------------ backward: dPred
// dS (optional): the score gradients themselves, one 64-byte row of 16 heads per candidate slot bt * (N + K) + j
// (j < N: negative j, all K heads; j >= N: positive of head j - N, the other heads 0) -- what the re-associated dz path
// (nce_bwd_g_kernel) contracts with the rows of c.  A lane of the loop below holds ONE head's gradient for four candidates,
// a dS row is 16 heads of one candidate: each 16-candidate tile is transposed through a per-wave 1 KB LDS tile and leaves as
// one contiguous 1 KB store per wave (written lane by lane in 4-byte pieces the same 66 MB cost the kernel 56 us at B = 64).
constexpr int kDsTile = 256 + 48;       // 16 candidates x 16 heads; the four lane groups' rows start 16 floats further apart
                                        // each, so that a transposing write hits every LDS bank exactly twice
__global__ __launch_bounds__(256, 3) void nce_bwd_dpred_kernel(
    const float* __restrict__ z, const int* __restrict__ ext, const float* __restrict__ logits,
    const float* __restrict__ lse, const float* __restrict__ gscale, float* __restrict__ dpred, int BW,
    int W, int S, int K, int N, float* __restrict__ amax_slots, float* __restrict__ dS, int koff) {
    __shared__ float ds_tile[4][kDsTile];
    const int lane = threadIdx.x & 63;
    const int bt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bt >= BW) return;
    const int b = bt / W, t = bt - b * W;
    float amax = 0.f;
    const int i = lane & 15, kq = lane >> 4;
    float* tile = ds_tile[threadIdx.x >> 6];
    const bool hv = i < K;
    const float gs = hv ? gscale[i] : 0.f;
    const float ls = hv ? lse[(long)bt * K + i] : 0.f;
    const float* lp = logits + ((long)bt * K + (hv ? i : 0)) * (N + 1) + 1 + 4 * kq;
    const int* ep = ext + (long)bt * N + 4 * kq;
    f32x4 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ii = 0; ii < N / 16; ++ii) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float a = hv ? gs * expf(lp[16 * ii + jj] - ls) : 0.f;   // d score[head i][n]
            const int row = ep[16 * ii + jj];                            // n = 16 ii + 4 kq + jj
            if (dS != nullptr) tile[(4 * kq + jj) * 16 + 16 * kq + i] = a;
            const float* zr = z + (long)row * kC + 4 * i;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 bv = ld4(zr + 64 * u);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[u * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, f4c(bv, e), acc[u * 4 + e], 0, 0, 0);
            }
        }
        if (dS != nullptr) {            // (wave-uniform) lane l leaves with heads 4 (l & 3).. of candidate 16 ii + (l >> 2)
            __builtin_amdgcn_wave_barrier();
            const int nl = lane >> 2;
            const float4 v = *reinterpret_cast<const float4*>(tile + nl * 16 + 16 * (nl >> 2) + 4 * (lane & 3));
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<float4*>(dS + ((long)bt * (N + K) + 16 * ii) * 16 + 4 * lane) = v;
        }
    }
    // C layout: head = 4kq + r, channel = 64u + 4i + e
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int head = 4 * kq + r;
        if (head < K) {
            const float d0 = gscale[head] * (expf(logits[((long)bt * K + head) * (N + 1)] - lse[(long)bt * K + head]) - 1.0f);
            if (dS != nullptr) dS[((long)bt * (N + K) + N + head) * 16 + i] = i == head ? d0 : 0.f;
            const float* zp = z + ((long)b * S + t + koff + head + 1) * kC + 4 * i;
            float* op = dpred + ((long)bt * K + head) * kC + 4 * i;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 zv = ld4(zp + 64 * u);
                float4 o;
                o.x = fmaf(d0, zv.x, acc[u * 4 + 0][r]);
                o.y = fmaf(d0, zv.y, acc[u * 4 + 1][r]);
                o.z = fmaf(d0, zv.z, acc[u * 4 + 2][r]);
                o.w = fmaf(d0, zv.w, acc[u * 4 + 3][r]);
                *reinterpret_cast<float4*>(op + 64 * u) = o;
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
        }
    }
    if (amax_slots != nullptr) {             // max|dPred| for the GEMMs that read it: one atomic per wave, 64 addresses
        amax = wave_max(amax);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(amax_slots + (bt & (kAmaxSlots - 1))), __float_as_uint(amax));
    }
}

// ------------------------------------------------------------------ backward: dz, linear heads (re-associated through c)
// Wcat[o][k*256 + i] = W_k[o][i] = wall[(k*256 + o)*256 + i]: the B operand (N = 256 rows o, contraction k*256 + i) of
// dz = G . Wcat^T.  One 1 KB row per wave.
__global__ __launch_bounds__(256) void nce_wcat_kernel(const float* __restrict__ wall, float* __restrict__ wcat, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);          // (k, o)
    if (row >= K * kC) return;
    const int k = row >> kCLog2, o = row & (kC - 1);
    *reinterpret_cast<float4*>(wcat + ((long)o * K + k) * kC + 4 * lane) = ld4(wall + (long)row * kC + 4 * lane);
}

// G[j][k][:] = sum over the candidate slots s that land on row j of z (perm[row_ptr[j] .. row_ptr[j+1]), cpc_nce_prepare)
// of dS[s][k] * c[window of s]: one wavefront (= one 64-thread workgroup) per destination row,
//   G_j[16 heads x 256] = dS_j^T[16 x n_j] . C_j[n_j x 256]        on v_mfma_f32_16x16x4_f32,
// four slots per MFMA step (lane group kq takes slot 4 q + kq: its 16 lanes read that slot's 64-byte dS row as the A
// operand and 4 x 256 bytes of its c row as B operands), nce_bwd_dpred_kernel's loop with the roles of windows and
// destination rows exchanged.  The slot list of a row arrives in arbitrary order (nce_fill_kernel places slots with an
// integer atomic cursor); it is rank-sorted in LDS first, so the summation order is the ascending slot order whatever the
// fill order was: results are bit-reproducible.  amax_slots: 64 partial maxima of |G| (cleared by nce_gscale_kernel), the
// operand bound of the GEMM that follows.
constexpr int GATHER_MAX_SORT = 1024;
__global__ __launch_bounds__(64, 4) void nce_bwd_g_kernel(const float* __restrict__ c, const float* __restrict__ dS,
                                                       const int* __restrict__ perm, const int* __restrict__ row_ptr,
                                                       float* __restrict__ G, int W, int S, int K, int NK,
                                                       float* __restrict__ amax_slots, const float* __restrict__ gscale = nullptr) {
    // gscale (one-pass criterion): dS holds the softmax rows of a unit upstream gradient; head i's gradient multiplies them here
    __shared__ int raw[GATHER_MAX_SORT];          // the slot list as filled; after the sort: row of c per sorted slot
    __shared__ int sorted[GATHER_MAX_SORT];
    const int lane = threadIdx.x;
    const int j = blockIdx.x;
    const int beg = row_ptr[j], len = row_ptr[j + 1] - beg;
    const bool do_sort = len <= GATHER_MAX_SORT;
    auto crow_of = [&](int slot) __attribute__((always_inline)) {     // slot -> window (b,t) -> row b*S + t of c
        const int bt = slot / NK, b = bt / W;
        return b * S + (bt - b * W);
    };
    if (do_sort) {
        for (int q = lane; q < len; q += 64) raw[q] = perm[beg + q];
        __syncthreads();
        for (int q = lane; q < len; q += 64) {
            const int e = raw[q];
            int rank = 0;
            for (int v = 0; v < len; ++v) rank += raw[v] < e ? 1 : 0;      // slots are unique
            sorted[rank] = e;
        }
        __syncthreads();
        for (int q = lane; q < len; q += 64) raw[q] = crow_of(sorted[q]);
        __syncthreads();
    }
    const int i = lane & 15, kq = lane >> 4;
    const float gsi = gscale == nullptr ? 1.0f : (i < K ? gscale[i] : 0.f);
    f32x4 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < len; p0 += 16) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int p = p0 + 4 * jj + kq;
            const bool ok = p < len;
            const int pc = ok ? p : 0;                                   // a valid slot: its row is read and multiplied by 0
            const int slot = do_sort ? sorted[pc] : perm[beg + pc];
            const int crow = do_sort ? raw[pc] : crow_of(slot);
            const float a = ok ? gsi * dS[(long)slot * 16 + i] : 0.f;    // d score[head i][slot]
            const float* cr = c + (long)crow * kC + 4 * i;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 bv = ld4(cr + 64 * u);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[u * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, f4c(bv, e), acc[u * 4 + e], 0, 0, 0);
            }
        }
    }
    // C layout: head = 4kq + r, channel = 64u + 4i + e
    float amax = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int head = 4 * kq + r;
        if (head < K) {
            float* op = G + ((long)j * K + head) * kC + 4 * i;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 o;
                o.x = acc[u * 4 + 0][r]; o.y = acc[u * 4 + 1][r]; o.z = acc[u * 4 + 2][r]; o.w = acc[u * 4 + 3][r];
                *reinterpret_cast<float4*>(op + 64 * u) = o;
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
        }
    }
    if (amax_slots != nullptr) {
        amax = wave_max(amax);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(amax_slots + (j & (kAmaxSlots - 1))), __float_as_uint(amax));
    }
}

// The same gather-GEMM on fp16 pieces (round 6): the rows of c come from an H2 copy (ch, scaled for the bound of |c| the forward
// left: nce_rows_to_h2_kernel) by DMA into a 16 KB LDS tile, sixteen sorted slots at a time; the contraction runs over the
// SLOTS, so the B operand is fetched with the transposing LDS read exactly as the forward kernel's weighted row sum
// (nce_fwd_h2_kernel: same tile image, same swizzle); A = g_head * softmax row of (head i, slots 4 kq .. + 3), split with the
// power-of-two scale of max|g| (softmax rows are in [-1, 1]).  48 v_mfma_f32_16x16x16_f16 per 16 slots where the f32 kernel
// issues 256 v_mfma_f32_16x16x4_f32.  G leaves through the tile: one 1 KB store instruction per head.
constexpr int GATHER_H2_SORT = 512;                 // slots per destination row that are sorted (more: fill order -- never at N = 128)
__global__ __launch_bounds__(64, 2) void nce_bwd_g_h2_kernel(const unsigned char* __restrict__ ch, const float* __restrict__ cbound,
                                                             const float* __restrict__ dS, const int* __restrict__ perm,
                                                             const int* __restrict__ row_ptr, float* __restrict__ G, int W, int S,
                                                             int K, int NK, float* __restrict__ amax_slots,
                                                             const float* __restrict__ gscale) {
    __shared__ int raw[GATHER_H2_SORT];           // the slot list as filled; after the sort: row of c per sorted slot
    __shared__ int sorted[GATHER_H2_SORT];
    __shared__ __attribute__((aligned(16))) unsigned char tile[kTileBytes];
    const int lane = threadIdx.x;
    const int j = blockIdx.x;
    const int beg = row_ptr[j], len = row_ptr[j + 1] - beg;
    const bool do_sort = len <= GATHER_H2_SORT;
    auto crow_of = [&](int slot) __attribute__((always_inline)) {
        const int bt = slot / NK, b = bt / W;
        return b * S + (bt - b * W);
    };
    if (do_sort) {
        for (int q = lane; q < len; q += 64) raw[q] = perm[beg + q];
        __syncthreads();
        for (int q = lane; q < len; q += 64) {
            const int e = raw[q];
            int rank = 0;
            for (int v = 0; v < len; ++v) rank += raw[v] < e ? 1 : 0;      // slots are unique
            sorted[rank] = e;
        }
        __syncthreads();
        for (int q = lane; q < len; q += 64) raw[q] = crow_of(sorted[q]);
        __syncthreads();
    }
    const int i = lane & 15, kq = lane >> 4;
    const float gsi = gscale == nullptr ? 1.0f : (i < K ? gscale[i] : 0.f);
    const float sa = scale_for_amax(gscale == nullptr ? 1.0f : wave_max(fabsf(gsi)));
    const float sc = scale_for_amax(fold_amax(cbound, kAmaxSlots));
    const TileAddr ta(tile, i, kq);
    f32x4 acc[16];                                // acc[ct][r]: head 4 kq + r, channel 16 ct + i
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // One tile = sixteen sorted slots.  Its rows are requested (DMA) and its weights loaded while the previous tile's 48 MFMAs run
    // from REGISTERS: as soon as a tile has landed, its 32 transposed fragments are read out of LDS and the next request goes out.
    auto request = [&](int p0, float (&a)[4]) __attribute__((always_inline)) {
        const int pl = p0 + (lane & 15) < len ? p0 + (lane & 15) : 0;                  // (past the end: a valid row, weight 0)
        const int mine = do_sort ? raw[pl] : crow_of(perm[beg + pl]);                  // lane r (and r + 16 ..): row of slot p0 + r
        int rows[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rows[r] = __builtin_amdgcn_readlane(mine, r);
        tile_dma16(ch, rows, tile);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = p0 + 4 * kq + r;
            const bool ok = p < len;
            const int slot = do_sort ? sorted[ok ? p : 0] : perm[beg + (ok ? p : 0)];
            a[r] = ok ? gsi * dS[(long)slot * 16 + i] : 0.f;                           // d score[head i][slot]
        }
    };
    float a[4];
    if (len > 0) request(0, a);
    for (int p0 = 0; p0 < len; p0 += 16) {
        CPC_WAIT_VMCNT(0);
        __builtin_amdgcn_wave_barrier();
        f16x4 wh, wl;
        split4(a, sa, wh, wl);
        f16x4 bh[16], bl[16];
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            bh[ct] = tile_read_tr(ta.p2(4 * ct));
            bl[ct] = tile_read_tr(ta.p2(4 * ct + 1));
        }
        CPC_WAIT_LGKMCNT0();                             // (the fragments are in registers: the tile may be overwritten)
        if (p0 + 16 < len) request(p0 + 16, a);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[4 * c4 + u] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh, bh[4 * c4 + u], acc[4 * c4 + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[4 * c4 + u] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh, bl[4 * c4 + u], acc[4 * c4 + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[4 * c4 + u] = __builtin_amdgcn_mfma_f32_16x16x16f16(wl, bh[4 * c4 + u], acc[4 * c4 + u], 0, 0, 0);
        }
    }
    const float inv = 1.0f / (sa * sc);
    __builtin_amdgcn_wave_barrier();
    float* tf = reinterpret_cast<float*>(tile);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) tf[(4 * kq + r) * kC + 16 * ct + i] = acc[ct][r] * inv;
    __builtin_amdgcn_wave_barrier();
    float amax = 0.f;
    for (int head = 0; head < K; ++head) {
        const float4 u = *reinterpret_cast<const float4*>(tf + head * kC + 4 * lane);
        *reinterpret_cast<float4*>(G + ((long)j * K + head) * kC + 4 * lane) = u;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(u.x), fabsf(u.y))), fmaxf(fabsf(u.z), fabsf(u.w)));
    }
    if (amax_slots != nullptr) {
        amax = wave_max(amax);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(amax_slots + (j & (kAmaxSlots - 1))), __float_as_uint(amax));
    }
}

// ------------------------------------------------------------------ backward: dz, any prediction network
// Every candidate (negative n or positive k of window (b,t)) contributes one 256-float row to
// dz[its row of z].  Destinations are random, so instead of 32k float atomics per window the
// contributions are first written as rows of V (slot = bt*(N+K) + candidate, plain coalesced
// 16-byte stores straight from the MFMA accumulators), then reduced per destination row by
// nce_gather_rows_kernel through a destination-sorted slot list: no atomics, no memset,
// bit-reproducible.
__global__ __launch_bounds__(256) void nce_bwd_dz_rows_kernel(
    const float* __restrict__ pred, const float* __restrict__ logits, const float* __restrict__ lse,
    const float* __restrict__ gscale, float* __restrict__ V, int BW, int K, int N) {
    const int lane = threadIdx.x & 63;
    const int bt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bt >= BW) return;
    const int i = lane & 15, kq = lane >> 4;
    // B operand: P[head 4s+kq][channels 64u + 4i + e]
    float4 pb[4][4];
    float gs[4], ls[4];
    const float* lp[4];
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) {
        const int head = 4 * sx + kq;
        const bool hv = head < K;
        gs[sx] = hv ? gscale[head] : 0.f;
        ls[sx] = hv ? lse[(long)bt * K + head] : 0.f;
        lp[sx] = logits + ((long)bt * K + (hv ? head : 0)) * (N + 1);
        const float* pp = pred + ((long)bt * K + (hv ? head : 0)) * kC + 4 * i;
#pragma unroll
        for (int u = 0; u < 4; ++u) pb[sx][u] = hv ? ld4(pp + 64 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float* vrow = V + (long)bt * (N + K) * kC + 4 * i;
    for (int nt = 0; nt < N / 16; ++nt) {
        f32x4 acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            // d score[head 4sx+kq][n = nt*16+i]; exactly 0 for the padding heads (0 * exp(.) would be NaN once a logit of
            // head 0, whose row they alias, exceeds 88)
            const float a = 4 * sx + kq < K ? gs[sx] * expf(lp[sx][1 + nt * 16 + i] - ls[sx]) : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[u * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, f4c(pb[sx][u], e), acc[u * 4 + e], 0, 0, 0);
        }
        // C layout: negative n = nt*16 + 4kq + r, channel = 64u + 4i + e
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* dst = vrow + (long)(nt * 16 + 4 * kq + r) * kC;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 o;
                o.x = acc[u * 4 + 0][r]; o.y = acc[u * 4 + 1][r]; o.z = acc[u * 4 + 2][r]; o.w = acc[u * 4 + 3][r];
                // streamed: 1 GB written once and read once by the gather (non-temporal: 4.62 vs 4.66 ms per step)
                __builtin_nontemporal_store(o.x, dst + 64 * u); __builtin_nontemporal_store(o.y, dst + 64 * u + 1);
                __builtin_nontemporal_store(o.z, dst + 64 * u + 2); __builtin_nontemporal_store(o.w, dst + 64 * u + 3);
            }
        }
    }
    // positives: candidate slot N + head carries d score[head][pos] * P[head]
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) {
        const int head = 4 * sx + kq;
        if (head < K) {
            const float d0 = gs[sx] * (expf(lp[sx][0] - ls[sx]) - 1.0f);
            float* dst = vrow + (long)(N + head) * kC;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 o;
                o.x = d0 * pb[sx][u].x; o.y = d0 * pb[sx][u].y; o.z = d0 * pb[sx][u].z; o.w = d0 * pb[sx][u].w;
                *reinterpret_cast<float4*>(dst + 64 * u) = o;
            }
        }
    }
}

// dz[j] = sum of V rows whose destination is j:  slots perm[row_ptr[j] .. row_ptr[j+1]).
// One wavefront (= one 64-thread workgroup) per destination row, 4 channels per lane.  The slot list of
// a row arrives in arbitrary order (nce_fill_kernel places slots with an integer atomic cursor); it is
// rank-sorted in LDS first, so the floating-point summation order is the ascending slot order whatever
// the fill order was: results are bit-reproducible and equal to a stable sort by destination.
__global__ __launch_bounds__(64) void nce_gather_rows_kernel(const float* __restrict__ V,
                                                             const int* __restrict__ perm,
                                                             const int* __restrict__ row_ptr,
                                                             float* __restrict__ dz, int nrows) {
    __shared__ int raw[GATHER_MAX_SORT];
    __shared__ int sorted[GATHER_MAX_SORT];
    const int lane = threadIdx.x;
    const int j = blockIdx.x;
    const int beg = row_ptr[j], len = row_ptr[j + 1] - beg;
    const bool do_sort = len <= GATHER_MAX_SORT;
    if (do_sort) {
        for (int i = lane; i < len; i += 64) raw[i] = perm[beg + i];
        __syncthreads();
        for (int i = lane; i < len; i += 64) {
            const int e = raw[i];
            int rank = 0;
            for (int q = 0; q < len; ++q) rank += raw[q] < e ? 1 : 0;      // slots are unique
            sorted[rank] = e;
        }
        __syncthreads();
    }
    const int* list = do_sort ? sorted : perm + beg;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    int p = 0;
    auto ldv = [&](const float* q) __attribute__((always_inline)) {      // V is read exactly once: non-temporal (4.61 vs 4.66 ms)
        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
        return make_float4(t.x, t.y, t.z, t.w);
    };
    for (; p + 4 <= len; p += 4) {
        const float4 v0 = ldv(V + (long)list[p] * kC + 4 * lane);
        const float4 v1 = ldv(V + (long)list[p + 1] * kC + 4 * lane);
        const float4 v2 = ldv(V + (long)list[p + 2] * kC + 4 * lane);
        const float4 v3 = ldv(V + (long)list[p + 3] * kC + 4 * lane);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; p < len; ++p) {
        const float4 v0 = ldv(V + (long)list[p] * kC + 4 * lane);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    float4 o;
    o.x = (a0.x + a1.x) + (a2.x + a3.x);
    o.y = (a0.y + a1.y) + (a2.y + a3.y);
    o.z = (a0.z + a1.z) + (a2.z + a3.z);
    o.w = (a0.w + a1.w) + (a2.w + a3.w);
    *reinterpret_cast<float4*>(dz + (long)j * kC + 4 * lane) = o;
}

// ------------------------------------------------------------------ negative-index preparation
// Turns the two draws of sampleClean (criterion.py:181-189; int64, flat (b,n,t) order) into what the
// kernels consume: ext[(b*W+t)*N + n] = ((seqIdx + t) mod S) + batchIdx*S (criterion.py:191-199), and the
// destination-sorted candidate slot list (perm, row_ptr) used by the backward gather.
// Set (bit 0) when a caller-supplied draw is outside its range (batchIdx in [0,B), seqIdx in [0,S)); the offending index
// is clamped so that nothing is read or counted out of bounds.  Read through cpc_device_error_flags() (capi.hip).
static __device__ unsigned g_nce_bad_index = 0;

// One wavefront per window (b,t): its N negative rows from the two draws (criterion.py:191-199), written to ext in ASCENDING
// order (ties in draw order).  The criterion is invariant under a permutation of a window's negatives -- the softmax, the
// arg-max test and every gradient sum over them -- and with sorted lists the waves of the scoring kernels, which walk their
// lists in step, gather from a narrow band of z at any moment: it stays in the 4 MB L2 of an XCD instead of coming from
// Infinity Cache (measured at B = 64: scoring kernel 130 -> 116 us, its backward twin 185 -> 153 us).
constexpr int kSortMax = 1024;       // negatives per window that are sorted (more: left in draw order)
// N: negatives per window as drawn; Np >= N: the row pitch of ext -- entries N .. Np-1 are padding (row b*S + t, a valid row; the
// scoring kernels mask them by POSITION, so the sort below covers the drawn negatives only).
__global__ __launch_bounds__(256) void nce_rows_kernel(const long* __restrict__ batchIdx, const long* __restrict__ seqIdx,
                                                       int* __restrict__ ext_, int B, int S, int W, int N, int Np) {
    __shared__ int rows[4][kSortMax];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // (a wave per window; with a capped grid -- cpc_set_index_prep_groups -- a wave walks several)
    for (int bt = blockIdx.x * 4 + wv; bt < B * W; bt += gridDim.x * 4) {
    const int b = bt / W, t = bt - b * W;
    int* __restrict__ ext = ext_ + (long)bt * (Np - N);  // (rows below are addressed with pitch N: shift by the padding so far)
    for (int j = N + lane; j < Np; j += 64) ext[(long)bt * N + j] = b * S + t;
    const bool sort = N <= kSortMax;
    for (int j = lane; j < N; j += 64) {
        const long flat = ((long)b * N + j) * W + t;
        long si = seqIdx[flat], bi = batchIdx[flat];
        if (si < 0 || si >= S || bi < 0 || bi >= B) {
            atomicOr(&g_nce_bad_index, 1u);
            si = si < 0 ? 0 : (si >= S ? S - 1 : si);
            bi = bi < 0 ? 0 : (bi >= B ? B - 1 : bi);
        }
        const int d = (int)((si + t) % S) + (int)bi * S;
        if (sort) rows[wv][j] = d; else ext[(long)bt * N + j] = d;
    }
    if (!sort) continue;
    // bitonic sort of the window's rows in the wave's LDS tile, padded to a power of two with a key above every row (equal rows
    // are indistinguishable, so any sorting network gives the list a stable rank sort gives): 28 compare-exchange stages of one
    // pair per lane at N = 128, where ranking every element against every other took 2 x 128 LDS reads per lane
    int P = 64;
    while (P < N) P <<= 1;
    for (int j = N + lane; j < P; j += 64) rows[wv][j] = 0x7fffffff;
    __builtin_amdgcn_wave_barrier();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int pr = lane; pr < P / 2; pr += 64) {
                const int i = ((pr / j) * 2 * j) + (pr % j), q = i + j;      // the pair (i, i ^ j), bit j of i clear
                const int a = rows[wv][i], c = rows[wv][q];
                const bool up = (i & k) == 0;
                if ((a > c) == up) { rows[wv][i] = c; rows[wv][q] = a; }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    for (int j = lane; j < N; j += 64) ext[(long)bt * N + j] = rows[wv][j];
    __builtin_amdgcn_wave_barrier();                     // (the next window's rows overwrite the tile)
    }
}

__global__ __launch_bounds__(256) void nce_index_kernel(const int* __restrict__ ext, int* __restrict__ dest,
                                                        int* __restrict__ count, int B, int S, int W, int K,
                                                        int N, int koff) {
    const long total = (long)B * W * (N + K);
    for (long slot = (long)blockIdx.x * 256 + threadIdx.x; slot < total; slot += (long)gridDim.x * 256) {
        const int bt = (int)(slot / (N + K)), j = (int)(slot - (long)bt * (N + K));
        const int b = bt / W, t = bt - b * W;
        int d;
        if (j < N) {
            d = ext[(long)bt * N + j];
        } else {
            d = b * S + t + koff + (j - N) + 1;         // positive of head koff + j-N (criterion.py:210-215)
        }
        dest[slot] = d;
        atomicAdd(&count[d], 1);
    }
}

// exclusive scan of count[0..n) -> row_ptr[0..n], cursor[0..n) = row_ptr[0..n)   (single workgroup)
__global__ __launch_bounds__(1024) void nce_scan_kernel(const int* __restrict__ count, int* __restrict__ row_ptr,
                                                        int* __restrict__ cursor, int n) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = tid * per, hi = min(n, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += count[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan of the partials
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;                            // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) {
        row_ptr[i] = run;
        cursor[i] = run;
        run += count[i];
    }
    if (tid == 1023) row_ptr[n] = part[1023];
}

__global__ __launch_bounds__(256) void nce_fill_kernel(const int* __restrict__ dest, int* __restrict__ cursor,
                                                       int* __restrict__ perm, long total) {
    for (long slot = (long)blockIdx.x * 256 + threadIdx.x; slot < total; slot += (long)gridDim.x * 256) {
        const int pos = atomicAdd(&cursor[dest[slot]], 1);
        perm[pos] = (int)slot;
    }
}

// ------------------------------------------------------------------ host side
int g_index_prep_groups = -1;  // cpc_set_index_prep_groups: -1 (default) = at most one workgroup per CU and launch, 0 = a workgroup per
                               // 4 windows / 256 slots, n > 0 = at most n.  The preparation runs beside conv1 / conv2 with ~0.8 ms in
                               // hand: one resident workgroup per CU walking its share costs the conv layers less than thousands of
                               // short ones queueing for their CUs (same-box A/B at B = 64: 2.822 vs 2.837 ms per step sustained;
                               // 128 / 192: 2.830, 384: 2.845, 512: 2.855, 64: 2.99, 16: 3.94 -- then the lists are late)
int g_nce_fused = 2;       // cpc_set_nce_fused: 1 the one-pass criterion (nce_fwd_fused_kernel: scores and the unit-gradient
                           // dPred from ONE gather pass; linear heads), 0 the two-pass kernels (nce_fwd_kernel + nce_bwd_dpred_kernel),
                           // 2 (default since round 6) the one-pass criterion with the scoring kernel on fp16 pieces (nce_fwd_h2_kernel: an H2 copy of z as
                           // gather source, DMA'd tiles, transposing LDS reads), 3 = 2 + the dz path's gather-GEMM likewise
                           // (nce_bwd_g_h2_kernel, from an H2 copy of c)
int g_nce_heads_dma = 1;   // cpc_set_nce_heads_dma: the prediction product pred = c . wall^T (criterion.py:108-116: the K linear heads, one NT GEMM
                           // with N = K * 256, K = 256) on the DMA-fed tile of gemm_dma.hip -- c as an H2 copy (its bound is the GRU's |h| < 1 or
                           // the measured max|c|), the stacked weights re-laid once per step beside the bounds -- instead of the
                           // register-staged generic tile (80 us at B = 64: 0.18 of the fp16-piece roof)
int g_nce_dbg = 0;         // cpc_set_nce_debug: measurement switches of nce_fwd_h2_kernel (tools/time_nce.py) -- bit 0 no softmax-row
                           // pass, 1 no T epilogue, 2 no weighted row sum, 3 no logits stores; results are then WRONG
int g_nce_rows_apart = 1;  // cpc_set_nce_rows_apart: the softmax rows of the fp16-piece scoring kernel by a launch of their own (on the
                           // loss reduction's stream) instead of inside it
int g_nce_grid = -1;       // cpc_set_nce_grid: workgroups of nce_fwd_h2_kernel -- 0 one per four windows, -1 (default since the end of round 6:
                           // -10 us per step sustained at B = 64, three alternations, profiles/r6_sweep_knobs.txt) two per CU (each
                           // walking windows 4 wg + wave, + 4 grid, ...: all waves sweep the sorted candidate lists in step), n > 0

struct NceLayout {
    int W, BW;
    int koff;       // first head of this call within the criterion's heads (cpc_nce_head_group; 0 unless K > 16 is walked in groups)
    int N, Nv;      // negatives per window as the kernels walk them (a multiple of 16) / as drawn (criterion.py:176-189): the
                    // candidates Nv .. N-1 of every window are padding -- a valid row of z, their logit forced to -3e38, so that
                    // they weigh exactly 0 in the softmax, the arg-max and every gradient
    long pred, logits, lse, bounds, tpred, ps, zh, chf, wqh, saved_total;
    long rowstat, tmp, sums, fwd_total;
    long dpred, wallT, part, gscale, V, dS, G, wcat, part_dz, ch, bwd_total;
};

constexpr int kDzSplits = 4;      // K-walk splits the dz GEMM's partial buffer is sized for (SplitK)

// Head groups (cpc_nce_head_group): the score tiles hold 16 heads per wavefront, so a criterion with more prediction steps is
// walked 16 heads at a time -- every cpc_nce_* call of the calling thread then works on heads k0 .. k0 + K - 1 of k_total: its
// windows are the W = S - k_total of the whole criterion and head k's positive is z[b, t + k0 + k + 1] (criterion.py:210-215).
// The groups share nothing but the inputs: losses / accuracies are per head, the gradients add.
static thread_local int g_head_off = 0, g_head_total = 0;
static thread_local bool g_zh_ready = false;     // cpc_nce_prepare_z ran for the calling thread's next forward
static thread_local bool g_wqh_ready = false;    // cpc_nce_bounds laid the head weights out for the calling thread's next forward

static bool nce_layout(int B, int S, int K, int N, NceLayout& n) {
    const int Ktot = g_head_total > 0 ? g_head_total : K;
    if (B <= 0 || K <= 0 || K > 16 || g_head_off + K > Ktot || S <= Ktot || N <= 0) return false;
    n.Nv = N;
    N = (N + 15) & ~15;                      // (every size below in padded candidates)
    n.N = N;
    n.koff = g_head_total > 0 ? g_head_off : 0;
    n.W = S - Ktot;
    n.BW = B * n.W;
    long o = 0;
    n.pred = o; o += align64l((long)n.BW * K * kC);
    n.logits = o; o += align64l((long)n.BW * K * (N + 1));
    n.lse = o; o += align64l((long)n.BW * K);
    n.bounds = o; o += 4 * kAmaxSlots;       // max|c|, max|wall| as 64 partial maxima each (linear heads only); max|T| slots (fused);
                                             // max|z| (fp16-piece kernels: the scale of zh)
    n.tpred = o; o += align64l((long)n.BW * K * kC);      // T: d loss_k / d pred_k for a unit upstream gradient (one-pass criterion)
    n.ps = o; o += align64l((long)n.BW * (N + K) * 16);   // ... and the softmax rows per candidate slot (the dz path's dS / g_k)
    n.zh = o; o += align64l((long)B * S * kC);            // z in H2 storage: the gather source of nce_fwd_h2_kernel
    n.chf = o; o += align64l((long)B * S * kC);           // c in H2 storage and the stacked head weights in the K-tile-major H2 rows of
    n.wqh = o; o += align64l((long)K * kC * kC + 64);     // gemm_dma.hip: the operands of the prediction product on the DMA-fed tile
    n.saved_total = o;
    o = 0;
    n.rowstat = o; o += align64l((long)n.BW * 2 * K);
    n.tmp = o; o += align64l((long)kRowsSumGroups * 2 * K);
    n.sums = o; o += 64;
    n.fwd_total = o;
    o = 0;
    n.dpred = o; o += align64l((long)n.BW * K * kC);
    n.wallT = o; o += (long)kC * K * kC;
    n.part = o; o += align64l(tn_gemm_part_floats(n.BW, K * kC, kC));
    n.gscale = o; o += 192;                  // [0..15] gscale, [17..18] bounds, [32..47] the V path's copy, [64..127] max|dPred| slots,
                                             // [128..191] max|G| slots
    // the two dz paths never run in one call: the linear heads' dS / G / Wcat / split partials share the space of the
    // candidate-row buffer V of the foreign-prediction path
    n.V = o;
    const long v_floats = align64l((long)n.BW * (N + K) * kC);
    long q = o;
    n.dS = q; q += align64l((long)n.BW * (N + K) * 16);
    n.G = q; q += align64l((long)B * S * K * kC);
    n.wcat = q; q += (long)kC * K * kC;
    n.part_dz = q; q += align64l((long)kDzSplits * B * S * kC);
    n.ch = q; q += align64l((long)B * S * kC);             // c in H2 storage: the gather source of nce_bwd_g_h2_kernel
    n.bwd_total = std::max(o + v_floats, q);
    return true;
}

static RowMap window_rows(const float* c, int B, int S, int W) {     // rows (b, t < W) of a (B,S,256) tensor
    RowMap r;
    r.base = c; r.R = W; r.bstride = (long)S * kC; r.rstride = kC; r.off = 0;
    r.tmul = 0; r.tadd = 0; r.Lin = 0x7fffffff; r.M = B * W;
    return r;
}

// scores, log-softmax, per-head loss / accuracy from given predictions
static int nce_fused(int) { return g_nce_fused; }
// the prediction product on the DMA-fed tile?  (fp16-piece arithmetic; the generic tile otherwise)
static bool nce_heads_dma(int K) { return g_nce_heads_dma && g_mfma_mode >= 2 && K > 0; }
// wall^T for dc = dPred . wall: plain, or -- one-pass criterion, whose dPred is the unit-gradient T -- pre-multiplied by the heads'
// upstream gradients (scratch + gscale must hold them: nce_gscale_kernel on this stream or one it has waited for)
static int nce_wallT(const float* wall, float* scratch, const NceLayout& n, int K, int N, hipStream_t st) {
    if (nce_fused(N) == 0) return transpose(wall, scratch + n.wallT, K * kC, kC, st);
    hipLaunchKernelGGL(nce_wallT_scaled_kernel, dim3(kC / 32, cdiv(K * kC, 32)), dim3(256), 0, st, wall, scratch + n.gscale,
                       scratch + n.wallT, K);
    CPC_LAUNCH_CHECK();
    return 0;
}
// fp32 rows -> H2 rows scaled for `bound` (kAmaxSlots partial maxima, on `st` or a stream it has waited for)
static int nce_rows_to_h2(const float* x, float* xh, const float* bound, long rows, hipStream_t st) {
    hipLaunchKernelGGL(nce_rows_to_h2_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, reinterpret_cast<unsigned char*>(xh), bound, rows);
    CPC_LAUNCH_CHECK();
    return 0;
}
// zh and its bound for the fp16-piece scoring kernel: max|z| into the fourth slot array of `saved`, then the H2 copy
static int nce_prepare_zh(const NceLayout& n, const float* z, float* saved, int B, int S, hipStream_t st) {
    const float* xs[1] = {z};
    const long ns[1] = {(long)B * S * kC};
    int rc = absmax_slots(xs, ns, 1, saved + n.bounds + 3 * kAmaxSlots, st);
    if (rc) return rc;
    return nce_rows_to_h2(z, saved + n.zh, saved + n.bounds + 3 * kAmaxSlots, (long)B * S, st);
}
static int nce_scores_forward(const NceLayout& n, const float* pred, const float* z, const int* ext, float* saved,
                              float* scratch, float* losses, float* acc, int S, int K, int N, hipStream_t st,
                              hipStream_t fin = nullptr, int fused = 0, bool want_rows = true) {
    // fin (nullptr: st): the stream the loss / accuracy reduction runs on.  Nothing of the backward reads its results (the
    // score gradients come from the saved logits), so a caller that joins `fin` later takes 15 us off its critical path.
    // fused: the one-pass kernel, which also leaves T (unit-gradient dPred) and max|T| in `saved` (slots zeroed by the caller)
    unsigned* ticket = reinterpret_cast<unsigned*>(scratch + n.sums + 32);
    step_timer_mark(8, st);
    if (fused >= 2) {
        // (zh: nce_prepare_zh, queued by the caller on this stream)
        long wgs = cdiv(n.BW, 4);
        if (g_nce_grid != 0) {
            long cap = g_nce_grid;
            if (cap < 0) {
                int dev = 0, cus = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
                cap = 2L * cus;
            }
            if (cap > 0) wgs = std::min(wgs, cap);
        }
        hipLaunchKernelGGL(nce_fwd_h2_kernel, dim3((unsigned)wgs), dim3(256), 0, st, pred, z,
                           reinterpret_cast<const unsigned char*>(saved + n.zh), saved + n.bounds + 3 * kAmaxSlots, ext, saved + n.logits,
                           saved + n.lse, scratch + n.rowstat, saved + n.tpred, saved + n.bounds + 2 * kAmaxSlots, saved + n.ps,
                           n.BW, n.W, S, K, N, ticket, n.Nv, n.koff, g_nce_dbg | ((g_nce_rows_apart || !want_rows) ? 1 : 0));
    } else if (fused)
        hipLaunchKernelGGL(nce_fwd_fused_kernel, dim3(cdiv(n.BW, 4)), dim3(256), 0, st, pred, z, ext, saved + n.logits,
                           saved + n.lse, scratch + n.rowstat, saved + n.tpred, saved + n.bounds + 2 * kAmaxSlots, saved + n.ps,
                           n.BW, n.W, S, K, N, ticket, n.Nv, n.koff);
    else
    hipLaunchKernelGGL(nce_fwd_kernel, dim3(cdiv(n.BW, 4)), dim3(256), 0, st, pred, z, ext, saved + n.logits,
                       saved + n.lse, scratch + n.rowstat, n.BW, n.W, S, K, N, ticket, n.Nv, n.koff);
    step_timer_mark(9, st);
    CPC_LAUNCH_CHECK();
    if (fin != nullptr && fin != st) {
        hipEvent_t* ev = stream_events(st);
        CPC_RETURN_IF(!ev, CPC_ERR_ARG);
        CPC_RETURN_IF(hipEventRecord(ev[10], st) != hipSuccess || hipStreamWaitEvent(fin, ev[10], 0) != hipSuccess, CPC_ERR_ARG);
        st = fin;
    }
    if (fused >= 2 && want_rows && g_nce_rows_apart && !(g_nce_dbg & 1)) {
        // the softmax rows, on the stream the reduction runs on (fin: off the forward's chain where the caller has one)
        hipLaunchKernelGGL(nce_softmax_rows_kernel, dim3(cdiv(n.BW, 4)), dim3(256), 0, st, saved + n.logits, saved + n.lse, saved + n.ps,
                           n.BW, K, N);
        CPC_LAUNCH_CHECK();
    }
    {                                                   // 2 K <= 32 columns (nce_layout): rows_sum's groups and order of additions
        int groups = n.BW > 64 ? kRowsSumGroups : 1;
        const int rpg = cdiv(n.BW, groups);
        groups = cdiv(n.BW, rpg);
        hipLaunchKernelGGL(nce_reduce_finalize_kernel, dim3(groups), dim3(256), 0, st, scratch + n.rowstat, n.BW, 2 * K, rpg,
                           scratch + n.tmp, ticket, losses, acc, K, 1.0f / (float)n.BW);
    }
    CPC_LAUNCH_CHECK();
    return 0;
}

// dz for predictions of any network = scatter of the per-candidate gradient rows, as candidate rows V + destination-sorted
// gather.  Needs only saved / gloss / perm / row_ptr: with its own copy of gscale it is independent of the dPred kernel.
static int nce_dz_rows_path(const NceLayout& n, const float* pred, const float* saved, const float* gloss, const int* perm,
                            const int* row_ptr, float* scratch, float* gscale_dz, float* dz, int B, int S, int K, int N,
                            hipStream_t st, bool own_gscale) {
    const float* logits = saved + n.logits, *lse = saved + n.lse;
    if (own_gscale)
        hipLaunchKernelGGL(nce_gscale_kernel, dim3(1), dim3(64), 0, st, gloss, gscale_dz, K, 1.0f / ((float)n.BW * (float)kC),
                           (const float*)nullptr, (float*)nullptr, 0, 0, 0);
    hipLaunchKernelGGL(nce_bwd_dz_rows_kernel, dim3(cdiv(n.BW, 4)), dim3(256), 0, st, pred, logits, lse, gscale_dz,
                       scratch + n.V, n.BW, K, N);
    hipLaunchKernelGGL(nce_gather_rows_kernel, dim3(B * S), dim3(64), 0, st, scratch + n.V, perm, row_ptr, dz, B * S);
    CPC_LAUNCH_CHECK();
    return 0;
}

// dz for the linear heads, from the dS rows nce_bwd_dpred_kernel left in scratch (and the max|wall| bound / cleared
// max|G| slots nce_gscale_kernel left there): Wcat, the per-destination gather-GEMM G, then dz = G . Wcat^T on the
// split-K wide tile.  Reads nothing of V's size: ~100 MB of G + 66 MB of dS at B = 64.
static int nce_dz_linear_path(const NceLayout& n, const float* c, const float* wall, const int* perm, const int* row_ptr,
                              float* scratch, float* dz, int B, int S, int K, int N, hipStream_t st,
                              const float* saved_for_ds = nullptr) {
    float* G = scratch + n.G, *wcat = scratch + n.wcat;
    float* bnd = scratch + n.gscale;
    const bool h2 = g_mfma_mode >= 2;
    hipLaunchKernelGGL(nce_wcat_kernel, dim3(cdiv(K * kC, 4)), dim3(256), 0, st, wall, wcat, K);
    if (saved_for_ds != nullptr && nce_fused(N) == 3) {     // ... on fp16 pieces, from an H2 copy of c scaled for the forward's bound of |c|
        int rc = nce_rows_to_h2(c, scratch + n.ch, saved_for_ds + n.bounds, (long)B * S, st);
        if (rc) return rc;
        hipLaunchKernelGGL(nce_bwd_g_h2_kernel, dim3(B * S), dim3(64), 0, st, reinterpret_cast<const unsigned char*>(scratch + n.ch),
                           saved_for_ds + n.bounds, saved_for_ds + n.ps, perm, row_ptr, G, n.W, S, K, N + K,
                           h2 ? bnd + 128 : (float*)nullptr, (const float*)(scratch + n.gscale));
    } else if (saved_for_ds != nullptr)           // one-pass criterion: the forward left the softmax rows; the heads' gradients apply here
        hipLaunchKernelGGL(nce_bwd_g_kernel, dim3(B * S), dim3(64), 0, st, c, saved_for_ds + n.ps, perm, row_ptr, G, n.W, S, K, N + K,
                           h2 ? bnd + 128 : (float*)nullptr, (const float*)(scratch + n.gscale));
    else
    hipLaunchKernelGGL(nce_bwd_g_kernel, dim3(B * S), dim3(64), 0, st, c, scratch + n.dS, perm, row_ptr, G, n.W, S, K, N + K,
                       h2 ? bnd + 128 : (float*)nullptr, (const float*)nullptr);
    CPC_LAUNCH_CHECK();
    GemmBounds gb;
    if (h2) { gb.a = bnd + 128; gb.a_slots = kAmaxSlots; gb.b = bnd + 18; }
    SplitK sk;
    sk.part = scratch + n.part_dz;
    sk.floats = (long)kDzSplits * B * S * kC;
    return nt_gemm(plain_rows(G, B * S, K * kC), wcat, K * kC, nullptr, dz, kC, kC, K * kC, st, 0, 0, gb, GemmGroup(), sk);
}

// dpred[bw][k][:] = gscale[k] * T[bw][k][:]: the one-pass kernel's unit-gradient dPred with the heads' upstream gradients applied
// (predictions of a foreign network: its backward wants dPred itself).  One float4 per thread.
__global__ __launch_bounds__(256) void nce_scale_tpred_kernel(const float* __restrict__ T, const float* __restrict__ gscale,
                                                              float* __restrict__ dpred, long n4, int K) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float g = gscale[(int)((i / (kC / 4)) % K)];
    float4 v = reinterpret_cast<const float4*>(T)[i];
    v.x *= g; v.y *= g; v.z *= g; v.w *= g;
    reinterpret_cast<float4*>(dpred)[i] = v;
}

// dPred (and, for the linear heads, the dS rows of the re-associated dz path) from the upstream per-head gradients.
static int nce_scores_backward(const NceLayout& n, const float* z, const int* ext, const float* saved, const float* gloss,
                               float* scratch, float* dpred, int B, int S, int K, int N, hipStream_t st,
                               const float* fwd_bounds = nullptr, float* dc_tail = nullptr, float* dS = nullptr,
                               bool gscale_done = false) {
    const float* logits = saved + n.logits, *lse = saved + n.lse;
    float* gscale = scratch + n.gscale;
    const float gs = 1.0f / ((float)n.BW * (float)kC);
    if (!gscale_done)      // (else: cpc_nce_backward_prepare ran it, on a stream this one has waited for)
    hipLaunchKernelGGL(nce_gscale_kernel, dim3(dc_tail ? 1 + 128 : 1), dim3(64), 0, st, gloss, gscale, K, gs, fwd_bounds, dc_tail, B,
                       S, n.W);
    hipLaunchKernelGGL(nce_bwd_dpred_kernel, dim3(cdiv(n.BW, 4)), dim3(256), 0, st, z, ext, logits, lse, gscale, dpred, n.BW,
                       n.W, S, K, N, fwd_bounds ? gscale + 64 : (float*)nullptr, dS, n.koff);
    CPC_LAUNCH_CHECK();
    return 0;
}

// bit 1 of cpc_device_error_flags(): cpc_nce_prepare saw an out-of-range negative index
int nce_error_flag_fetch(int clear, unsigned* out) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_nce_bad_index), sizeof(v)) != hipSuccess) return CPC_ERR_ARG;
    if (clear && v) {
        const unsigned zero = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_nce_bad_index), &zero, sizeof(zero)) != hipSuccess) return CPC_ERR_ARG;
    }
    *out = v;
    return 0;
}

}  // namespace cpc

using namespace cpc;

extern "C" int cpc_nce_layout(int B, int S, int K, int N, long* sizes) {
    NceLayout n;
    CPC_RETURN_IF(!nce_layout(B, S, K, N, n), CPC_ERR_SHAPE);
    N = n.N;                                 // (padded; n.Nv = as drawn)
    sizes[0] = n.saved_total; sizes[1] = n.fwd_total; sizes[2] = n.bwd_total;
    sizes[3] = n.pred; sizes[4] = n.logits; sizes[5] = n.lse;
    return 0;
}

// Negatives per window as the kernels lay them out: N rounded up to the MFMA tile (16).  ext is (B*W, padded), logits
// (B*W, K, 1 + padded), the slot lists count padded + K candidates per window; callers size their buffers with it.
extern "C" int cpc_nce_padded_negatives(int N) { return N <= 0 ? 0 : (N + 15) & ~15; }

// batchIdx, seqIdx: the two int64 draws of sampleClean, B*N*W each, flat in (b,n,t) order.
// Outputs: ext (B*W*N int32: the rows of each window's negatives, ASCENDING -- see nce_rows_kernel), perm (B*W*(N+K) int32),
// row_ptr (B*S+1 int32); work: B*W*(N+K) + 2*B*S + 2 ints.
extern "C" int cpc_nce_prepare(const long* batchIdx, const long* seqIdx, int* ext, int* perm, int* row_ptr,
                               int* work, int B, int S, int K, int N, void* stream) {
    NceLayout n;
    CPC_RETURN_IF(!nce_layout(B, S, K, N, n), CPC_ERR_SHAPE);
    N = n.N;                                 // (padded; n.Nv = as drawn)
    CPC_RETURN_IF(!batchIdx || !seqIdx || !ext || !perm || !row_ptr || !work, CPC_ERR_ARG);
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)n.BW * (N + K);
    const int rows = B * S;
    int* dest = work;
    int* count = work + total;
    int* cursor = count + rows + 1;
    (void)hipMemsetAsync(count, 0, sizeof(int) * (rows + 1), st);
    // (cpc_set_index_prep_groups: at most that many workgroups per launch, each walking several windows / slots -- the
    // preparation runs on a side stream beside the first conv layers and has ~0.8 ms until the criterion needs it)
    long cap = g_index_prep_groups;
    if (cap < 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        cap = cus;
    }
    const auto capped = [cap](long wgs) { return (unsigned)(cap > 0 ? std::min<long>(wgs, cap) : wgs); };
    hipLaunchKernelGGL(nce_rows_kernel, dim3(capped(cdiv(n.BW, 4))), dim3(256), 0, st, batchIdx, seqIdx, ext, B, S, n.W, n.Nv, N);
    hipLaunchKernelGGL(nce_index_kernel, dim3(capped(cdiv(total, 256))), dim3(256), 0, st, ext, dest, count, B, S, n.W, K, N, n.koff);
    hipLaunchKernelGGL(nce_scan_kernel, dim3(1), dim3(1024), 0, st, count, row_ptr, cursor, rows);
    hipLaunchKernelGGL(nce_fill_kernel, dim3(capped(cdiv(total, 256))), dim3(256), 0, st, dest, cursor, perm, total);
    CPC_LAUNCH_CHECK();
    return 0;
}

// c (B,S,256) context, z (B,S,256) encoder output, wall (K*256, 256) = the K head weights stacked,
// ext (B*W, N) int32 rows into z.view(B*S,256) [i.e. criterion.py:199's extIdx laid out (b,t,n)].
// losses, acc: K floats each (criterion.py:256-257).
static int nce_forward(const float* c, const float* z, const float* wall, const int* ext, float* saved, float* scratch,
                       float* losses, float* acc, int B, int S, int K, int N, hipStream_t st, bool bounds_ready,
                       hipStream_t fin = nullptr) {
    NceLayout n;
    CPC_RETURN_IF(!nce_layout(B, S, K, N, n), CPC_ERR_SHAPE);
    N = n.N;                                 // (padded; n.Nv = as drawn)
    CPC_RETURN_IF(!c || !z || !wall || !ext || !saved || !scratch || !losses || !acc, CPC_ERR_ARG);
    float* pred = saved + n.pred;
    // operand bounds for the fp16-split GEMMs of this call and of the backward (one small launch: 11 MB read)
    // (written in every mode: the backward may run in another one)
    if (!bounds_ready) {                         // (third job: the max|T| slots of the one-pass kernel, zeroed)
        const float* xs[3] = {c, wall, nullptr};
        const long ns[3] = {(long)B * S * kC, (long)K * kC * kC, 0};
        int rc = absmax_slots(xs, ns, 3, saved + n.bounds, st);
        if (rc) return rc;
    }
    GemmBounds gb;
    gb.a = saved + n.bounds; gb.a_slots = kAmaxSlots;
    gb.b = saved + n.bounds + kAmaxSlots; gb.b_slots = kAmaxSlots;
    int rc = 0;
    if (nce_heads_dma(K)) {
        // the same product on the DMA-fed tile: an H2 copy of c (8 MB at B = 64), the weights in K-tile-major H2 rows (ahead of time
        // where the bounds were: cpc_nce_bounds), 256 x 256 tiles of the conv layers' main loop
        // (the layout of cpc_nce_bounds counts only for a caller that says its bounds -- in THIS workspace -- are ready)
        if (!(bounds_ready && g_wqh_ready))
            rc = gemm_weight_h2(wall, kC, 1, K * kC, kC, saved + n.wqh, saved + n.bounds + kAmaxSlots, nullptr, 1, 0, 0, 0, 0, st);
        g_wqh_ready = false;
        if (rc) return rc;
        rc = nce_rows_to_h2(c, saved + n.chf, saved + n.bounds, (long)B * S, st);
        if (rc) return rc;
        rc = gemm_nt_dma_rows(window_rows(saved + n.chf, B, S, n.W), saved + n.wqh, pred, (long)K * kC, K * kC, kC, saved + n.bounds,
                              saved + n.bounds + kAmaxSlots, st);
    } else {
        g_wqh_ready = false;
        rc = nt_gemm(window_rows(c, B, S, n.W), wall, kC, nullptr, pred, (long)K * kC, K * kC, kC, st, 0, 0, gb);
    }
    if (rc) return rc;
    if (nce_fused(N) >= 2 && !g_zh_ready) {                 // (cpc_nce_prepare_z: ahead of time, on another stream)
        rc = nce_prepare_zh(n, z, saved, B, S, st);
        if (rc) return rc;
    }
    g_zh_ready = false;
    return nce_scores_forward(n, pred, z, ext, saved, scratch, losses, acc, S, K, N, st, fin, nce_fused(N));
}

extern "C" int cpc_nce_forward(const float* c, const float* z, const float* wall, const int* ext, float* saved,
                               float* scratch, float* losses, float* acc, int B, int S, int K, int N,
                               void* stream) {
    return nce_forward(c, z, wall, ext, saved, scratch, losses, acc, B, S, K, N, (hipStream_t)stream, false);
}

// The operand bounds of the criterion's GEMMs, ahead of time: max|wall| is fixed once the optimiser has stepped, and a
// recurrent context is bounded a priori (|h| < 1 for a GRU started from 0 or from one of its own states), so neither has to
// wait for c -- this call writes them into `saved` (on any stream, e.g. beside the encoder) and cpc_nce_forward_prepared then
// skips the 24 us reduction that otherwise sits between the autoregressive network and the prediction GEMM.
// c_bound > 0: the a-priori bound of |c|; otherwise c must be given and is reduced here.
extern "C" int cpc_nce_bounds(const float* c, float c_bound, const float* wall, float* saved, int B, int S, int K, int N,
                              void* stream) {
    NceLayout n;
    CPC_RETURN_IF(!nce_layout(B, S, K, N, n), CPC_ERR_SHAPE);
    N = n.N;                                 // (padded; n.Nv = as drawn)
    CPC_RETURN_IF(!wall || !saved || (!(c_bound > 0.f) && !c), CPC_ERR_ARG);
    const float* xs[3] = {c_bound > 0.f ? nullptr : c, wall, nullptr};
    const long ns[3] = {(long)B * S * kC, (long)K * kC * kC, 0};
    const float cv[3] = {c_bound, 0.f, 0.f};
    int rc = absmax_slots(xs, ns, 3, saved + n.bounds, (hipStream_t)stream, cv);
    if (rc) return rc;
    if (nce_heads_dma(K)) {          // ... and the head weights in the rows the DMA-fed prediction product reads (needs max|wall|: just written)
        rc = gemm_weight_h2(wall, kC, 1, K * kC, kC, saved + n.wqh, saved + n.bounds + kAmaxSlots, nullptr, 1, 0, 0, 0, 0, (hipStream_t)stream);
        g_wqh_ready = rc == 0;
    }
    return rc;
}

extern "C" int cpc_nce_forward_prepared(const float* c, const float* z, const float* wall, const int* ext, float* saved,
                                        float* scratch, float* losses, float* acc, int B, int S, int K, int N,
                                        void* stream) {
    return nce_forward(c, z, wall, ext, saved, scratch, losses, acc, B, S, K, N, (hipStream_t)stream, true);
}

// cpc_nce_forward / cpc_nce_forward_prepared (bounds_ready != 0) with the loss / accuracy reduction on `finalize_stream`
// (ordered behind the scoring kernel by an event; the caller joins that stream before anybody reads losses / acc).
extern "C" int cpc_nce_forward_streams(const float* c, const float* z, const float* wall, const int* ext, float* saved,
                                       float* scratch, float* losses, float* acc, int B, int S, int K, int N, int bounds_ready,
                                       void* stream, void* finalize_stream) {
    return nce_forward(c, z, wall, ext, saved, scratch, losses, acc, B, S, K, N, (hipStream_t)stream, bounds_ready != 0,
                       (hipStream_t)finalize_stream);
}

// The weight- and bound-only share of cpc_nce_backward_streams, ahead of time on any stream that has seen cpc_nce_bounds:
// the per-head gradient scales from gloss, the GEMM operand bounds, the cleared max|dPred| / max|G| slots, the zeroed tail of
// dc, and wall^T.  cpc_nce_backward_prepared then starts with the score-gradient kernel.
extern "C" int cpc_nce_backward_prepare(const float* wall, const float* saved, const float* gloss, float* scratch, float* dc,
                                        int B, int S, int K, int N, void* stream) {
    NceLayout n;
    CPC_RETURN_IF(!nce_layout(B, S, K, N, n), CPC_ERR_SHAPE);
    N = n.N;                                 // (padded; n.Nv = as drawn)
    CPC_RETURN_IF(!wall || !saved || !gloss || !scratch || !dc, CPC_ERR_ARG);
    hipStream_t st = (hipStream_t)stream;
    const bool h2 = g_mfma_mode >= 2;
    hipLaunchKernelGGL(nce_gscale_kernel, dim3(1 + 128), dim3(64), 0, st, gloss, scratch + n.gscale, K,
                       1.0f / ((float)n.BW * (float)kC), h2 ? saved + n.bounds : (const float*)nullptr, dc, B, S, n.W);
    CPC_LAUNCH_CHECK();
    return nce_wallT(wall, scratch, n, K, N, st);
}

// Same criterion on predictions computed by the caller (any prediction network, e.g. --rnnMode transformer):
// pred (B*W, K*256), row (b,t), head k at columns k*256...  `saved` keeps logits and lse only.
extern "C" int cpc_nce_scores_forward(const float* pred, const float* z, const int* ext, float* saved, float* scratch,
                                      float* losses, float* acc, int B, int S, int K, int N, void* stream) {
    NceLayout n;
    CPC_RETURN_IF(!nce_layout(B, S, K, N, n), CPC_ERR_SHAPE);
    N = n.N;                                 // (padded; n.Nv = as drawn)
    CPC_RETURN_IF(!pred || !z || !ext || !saved || !scratch || !losses || !acc, CPC_ERR_ARG);
    hipStream_t st = (hipStream_t)stream;
    if (nce_fused(N) >= 2) {
        // the one-pass kernel on fp16 pieces for a foreign network's predictions too (round 6): scores AND the unit-gradient dPred
        // from ONE gather pass over the H2 copy of z; the backward then only scales T by the heads' upstream gradients
        // (cpc_nce_scores_backward) instead of gathering the candidate rows a second time (nce_bwd_dpred_kernel: 150 us at B = 64)
        const float* none[1] = {nullptr};
        const long zero_n[1] = {0};
        const float cv[1] = {0.f};
        int rc = absmax_slots(none, zero_n, 1, saved + n.bounds + 2 * kAmaxSlots, st, cv);      // max|T| slots, cleared
        if (rc) return rc;
        if (!g_zh_ready && (rc = nce_prepare_zh(n, z, saved, B, S, st))) return rc;
        g_zh_ready = false;
        return nce_scores_forward(n, pred, z, ext, saved, scratch, losses, acc, S, K, N, st, nullptr, 2, false);
    }
    return nce_scores_forward(n, pred, z, ext, saved, scratch, losses, acc, S, K, N, st);
}

// dpred (B*W, K*256) and dz (B,S,256) are overwritten.
extern "C" int cpc_nce_scores_backward(const float* pred, const float* z, const int* ext, const int* perm,
                                       const int* row_ptr, const float* saved, const float* gloss, float* scratch,
                                       float* dpred, float* dz, int B, int S, int K, int N, void* stream) {
    NceLayout n;
    CPC_RETURN_IF(!nce_layout(B, S, K, N, n), CPC_ERR_SHAPE);
    N = n.N;                                 // (padded; n.Nv = as drawn)
    CPC_RETURN_IF(!pred || !z || !ext || !perm || !row_ptr || !saved || !gloss || !scratch || !dpred || !dz, CPC_ERR_ARG);
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    if (nce_fused(N) >= 2) {           // (the forward ran the one-pass kernel: T is in `saved`; a forward and its backward share the setting)
        hipLaunchKernelGGL(nce_gscale_kernel, dim3(1), dim3(64), 0, st, gloss, scratch + n.gscale, K, 1.0f / ((float)n.BW * (float)kC),
                           (const float*)nullptr, (float*)nullptr, 0, 0, 0);
        const long n4 = (long)n.BW * K * (kC / 4);
        hipLaunchKernelGGL(nce_scale_tpred_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, saved + n.tpred, scratch + n.gscale, dpred, n4, K);
        CPC_LAUNCH_CHECK();
    } else {
        rc = nce_scores_backward(n, z, ext, saved, gloss, scratch, dpred, B, S, K, N, st);
        if (rc) return rc;
    }
    return nce_dz_rows_path(n, pred, saved, gloss, perm, row_ptr, scratch, scratch + n.gscale, dz, B, S, K, N, st, false);
}

// gloss: K upstream gradients dL/dloss_k (device).  Outputs (overwritten): dc, dz (B,S,256), dwall (K*256,256).
// perm / row_ptr: the candidate slots (slot = (b*W+t)*(N+K) + j; j < N negative j, j >= N positive
// of head j-N, whose destination row is b*S + t + (j-N) + 1) sorted by destination row of z:
// row_ptr has B*S+1 entries, perm[row_ptr[r] .. row_ptr[r+1]) are the slots landing on row r.
extern "C" int cpc_nce_backward(const float* c, const float* z, const float* wall, const int* ext,
                                const int* perm, const int* row_ptr, const float* saved, const float* gloss,
                                float* scratch, float* dc, float* dz, float* dwall, int B, int S, int K, int N,
                                void* stream) {
    CPC_RETURN_IF(!dz, CPC_ERR_ARG);
    return cpc_nce_backward_streams(c, z, wall, ext, perm, row_ptr, saved, gloss, scratch, dc, dz, dwall, B, S, K, N,
                                    stream, stream);
}

// The dz path alone (linear heads), from the score gradients the cpc_nce_backward_streams(dz = NULL) call left in
// `scratch`: `stream` must wait for that call.
extern "C" int cpc_nce_backward_dz(const float* c, const float* wall, const int* perm, const int* row_ptr, const float* saved,
                                   float* scratch, float* dz, int B, int S, int K, int N, void* stream) {
    NceLayout n;
    CPC_RETURN_IF(!nce_layout(B, S, K, N, n), CPC_ERR_SHAPE);
    N = n.N;                                 // (padded; n.Nv = as drawn)
    CPC_RETURN_IF(!c || !wall || !perm || !row_ptr || !saved || !scratch || !dz, CPC_ERR_ARG);
    return nce_dz_linear_path(n, c, wall, perm, row_ptr, scratch, dz, B, S, K, N, (hipStream_t)stream,
                              nce_fused(N) != 0 ? saved : nullptr);
}

// As cpc_nce_backward, with the dz path launched on `dz_stream` behind an event recorded on `stream` once the score
// gradients are written (the rest of dz_stream's ordering -- every consumer of dz waits for it -- is the caller's
// business); dz == NULL leaves the dz path out altogether (cpc_nce_backward_dz runs it later).
static int nce_backward_impl(const float* c, const float* z, const float* wall, const int* ext, const int* perm,
                             const int* row_ptr, const float* saved, const float* gloss, float* scratch, float* dc, float* dz,
                             float* dwall, int B, int S, int K, int N, void* stream, void* dz_stream, bool prepared);
extern "C" int cpc_nce_backward_streams(const float* c, const float* z, const float* wall, const int* ext,
                                        const int* perm, const int* row_ptr, const float* saved, const float* gloss,
                                        float* scratch, float* dc, float* dz, float* dwall, int B, int S, int K,
                                        int N, void* stream, void* dz_stream) {
    return nce_backward_impl(c, z, wall, ext, perm, row_ptr, saved, gloss, scratch, dc, dz, dwall, B, S, K, N, stream, dz_stream, false);
}
// cpc_nce_backward_streams(dz = NULL, dwall = NULL) behind a cpc_nce_backward_prepare of the same step: score gradients, dPred
// and dc on `stream`; the dz path and the heads' gradient follow through cpc_nce_backward_dz / _dwall.
extern "C" int cpc_nce_backward_prepared(const float* c, const float* z, const float* wall, const int* ext, const int* perm,
                                         const int* row_ptr, const float* saved, const float* gloss, float* scratch, float* dc,
                                         int B, int S, int K, int N, void* stream) {
    return nce_backward_impl(c, z, wall, ext, perm, row_ptr, saved, gloss, scratch, dc, nullptr, nullptr, B, S, K, N, stream, stream, true);
}
static int nce_backward_impl(const float* c, const float* z, const float* wall, const int* ext, const int* perm,
                             const int* row_ptr, const float* saved, const float* gloss, float* scratch, float* dc, float* dz,
                             float* dwall, int B, int S, int K, int N, void* stream, void* dz_stream, bool prepared) {
    NceLayout n;
    CPC_RETURN_IF(!nce_layout(B, S, K, N, n), CPC_ERR_SHAPE);
    N = n.N;                                 // (padded; n.Nv = as drawn)
    CPC_RETURN_IF(!c || !z || !wall || !ext || !perm || !row_ptr || !saved || !gloss || !scratch || !dc, CPC_ERR_ARG);
    hipStream_t st = (hipStream_t)stream, st_dz = (hipStream_t)dz_stream;
    const bool fused = nce_fused(N) != 0;    // the forward left T (unit-gradient dPred) and max|T| in `saved`: no score-gradient pass
    float* dpred = fused ? const_cast<float*>(saved) + n.tpred : scratch + n.dpred, *wallT = scratch + n.wallT;
    const bool h2 = g_mfma_mode >= 2;        // the forward left the operand bounds in `saved`
    int rc = 0;
    if (fused) {
        if (!prepared)
            hipLaunchKernelGGL(nce_gscale_kernel, dim3(1 + 128), dim3(64), 0, st, gloss, scratch + n.gscale, K,
                               1.0f / ((float)n.BW * (float)kC), h2 ? saved + n.bounds : (const float*)nullptr, dc, B, S, n.W);
        CPC_LAUNCH_CHECK();
    } else {
        rc = nce_scores_backward(n, z, ext, saved, gloss, scratch, dpred, B, S, K, N, st, h2 ? saved + n.bounds : nullptr, dc,
                                 scratch + n.dS, prepared);
        if (rc) return rc;
    }
    if (dz != nullptr) {
        if (st_dz != st) {
            hipEvent_t* ev = stream_events(st);
            CPC_RETURN_IF(!ev, CPC_ERR_ARG);
            CPC_RETURN_IF(hipEventRecord(ev[9], st) != hipSuccess || hipStreamWaitEvent(st_dz, ev[9], 0) != hipSuccess, CPC_ERR_ARG);
        }
        rc = nce_dz_linear_path(n, c, wall, perm, row_ptr, scratch, dz, B, S, K, N, st_dz, fused ? saved : nullptr);
        if (rc) return rc;
    }
    const float* bnd = scratch + n.gscale;   // [17] max|c|, [18] max|wall|, [19] max|wall| max|g|, [64..127] max|dPred| slots
    GemmBounds gdc, gdw;
    GemmGroup gdw_grp;
    if (h2) {
        gdc.a = gdw.a = fused ? saved + n.bounds + 2 * kAmaxSlots : bnd + 64; gdc.a_slots = gdw.a_slots = kAmaxSlots;
        gdc.b = fused ? bnd + 19 : bnd + 18; gdw.b = bnd + 17;
    }
    if (fused) { gdw_grp.out_scale = bnd; gdw_grp.out_scale_rows = kC; }      // dW_k = g_k T_k^T . c
    // dc[:, :W] = dPred . Wall  (NT against Wall^T [256][K*256]; fused: T against g_k-scaled rows)
    if (!prepared) rc = nce_wallT(wall, scratch, n, K, N, st);
    if (rc) return rc;
    SplitK sk;                           // N = 256: 58 row tiles at B = 64; the head gradient's partial buffer is free until it
    sk.part = scratch + n.part;          // starts (behind this GEMM, on either stream)
    sk.floats = tn_gemm_part_floats(n.BW, K * kC, kC);
    rc = nt_gemm(plain_rows(dpred, n.BW, K * kC), wallT, K * kC, nullptr, dc, kC, kC, K * kC, st, n.W,
                 (long)S * kC, gdc, GemmGroup(), sk);
    if (rc || !dwall) return rc;
    // dW_k = dPred_k^T . c[:, :W]
    return tn_gemm(plain_rows(dpred, n.BW, K * kC), K * kC, window_rows(c, B, S, n.W), kC, scratch + n.part,
                   dwall, 0, st, gdw, gdw_grp);
}

// The head-weight gradient alone, from the dPred that cpc_nce_backward_streams(dwall = NULL) left in `scratch`:
// nothing on the way to the encoder depends on it, so the caller may run it on any stream that waits for that call.
extern "C" int cpc_nce_backward_dwall(const float* c, const float* saved, float* scratch, float* dwall, int B, int S, int K,
                                      int N, void* stream) {
    NceLayout n;
    CPC_RETURN_IF(!nce_layout(B, S, K, N, n), CPC_ERR_SHAPE);
    N = n.N;                                 // (padded; n.Nv = as drawn)
    CPC_RETURN_IF(!c || !saved || !scratch || !dwall, CPC_ERR_ARG);
    const bool fused = nce_fused(N) != 0;
    GemmBounds gdw;                                               // left by cpc_nce_backward_streams (nce_gscale_kernel)
    GemmGroup grp;
    if (g_mfma_mode >= 2) {
        gdw.a = fused ? saved + n.bounds + 2 * kAmaxSlots : scratch + n.gscale + 64; gdw.a_slots = kAmaxSlots;
        gdw.b = scratch + n.gscale + 17;
    }
    if (fused) { grp.out_scale = scratch + n.gscale; grp.out_scale_rows = kC; }
    return tn_gemm(plain_rows(fused ? saved + n.tpred : scratch + n.dpred, n.BW, K * kC), K * kC, window_rows(c, B, S, n.W), kC,
                   scratch + n.part, dwall, 0, (hipStream_t)stream, gdw, grp);
}

// 1 (default): the linear heads' criterion in one gather pass -- the forward leaves T, the unit-gradient dPred, in its saved
// workspace and the backward starts with the dc GEMM; 0: the two-pass kernels (nce_fwd_kernel, nce_bwd_dpred_kernel).  A forward
// and its backward must run under the same setting.
// Head group of the calling thread's following cpc_nce_* calls (see g_head_off): heads k0 .. of a criterion with k_total prediction
// steps; (0, 0) = none (the default: a call's K heads are the whole criterion).  The caller brackets every group's calls --
// layout, prepare, forward, backward: all of them -- with the group and (0, 0); the composite step (cpc_train_step) does not use it.
extern "C" int cpc_nce_head_group(int k0, int k_total) {
    CPC_RETURN_IF(k0 < 0 || k_total < 0 || (k_total == 0 && k0 != 0) || (k_total > 0 && k0 >= k_total), CPC_ERR_ARG);
    g_head_off = k0;
    g_head_total = k_total;
    return 0;
}

// Tuning switch: workgroups per launch of the index preparation (cpc_nce_prepare); -1 = one per CU (default), 0 = uncapped.
extern "C" int cpc_set_index_prep_groups(int n) {
    CPC_RETURN_IF(n < -1, CPC_ERR_ARG);
    g_index_prep_groups = n;
    return 0;
}

extern "C" int cpc_set_nce_fused(int on) {
    CPC_RETURN_IF(on < 0 || on > 3, CPC_ERR_ARG);
    g_nce_fused = on;
    return 0;
}
extern "C" int cpc_get_nce_fused(void) { return g_nce_fused; }
extern "C" int cpc_set_nce_heads_dma(int on) { g_nce_heads_dma = on ? 1 : 0; return 0; }
extern "C" int cpc_set_nce_rows_apart(int on) { g_nce_rows_apart = on ? 1 : 0; return 0; }
/* measurement only (tools/time_nce.py): leave parts of nce_fwd_h2_kernel out -- results are wrong while mask != 0 */
extern "C" int cpc_set_nce_debug(int mask) { g_nce_dbg = mask; return 0; }

// Workgroups of the fp16-piece scoring kernel: 0 = one per four windows (dispatch order), -1 = two per CU, n > 0 = at most n;
// a capped grid walks its windows with a stride of the grid (cpc_set_nce_fused(2) only).
extern "C" int cpc_set_nce_grid(int wgs) {
    CPC_RETURN_IF(wgs < -1, CPC_ERR_ARG);
    g_nce_grid = wgs;
    return 0;
}

// The H2 copy of z the fp16-piece scoring kernel gathers from (cpc_set_nce_fused(2)), ahead of time: z is final when the encoder
// has run, long before the criterion -- a caller with a second stream queues this there (behind the encoder) and the calling
// thread's next cpc_nce_forward* skips it.  A no-op in the other modes.
extern "C" int cpc_nce_prepare_z(const float* z, float* saved, int B, int S, int K, int N, void* stream) {
    NceLayout n;
    CPC_RETURN_IF(!nce_layout(B, S, K, N, n), CPC_ERR_SHAPE);
    CPC_RETURN_IF(!z || !saved, CPC_ERR_ARG);
    if (nce_fused(n.N) < 2) return 0;
    g_zh_ready = true;
    return nce_prepare_zh(n, z, saved, B, S, (hipStream_t)stream);
}
