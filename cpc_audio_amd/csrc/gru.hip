// CPCAR: multi-layer GRU autoregressor (torch.nn.GRU semantics, batch_first, gate order r,z,n).
//
// Reference: cpc/model.py:175-176 (nn.GRU(dimEncoded, dimOutput, num_layers, batch_first=True)),
// :185-204 (forward, optional carried hidden state).
//
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)        z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
//   n = tanh   (W_in x + b_in + r * (W_hn h + b_hn))   h' = (1 - z) * n + z * h
//
// Structure on MI355X:
//   * the input projection of all S steps is ONE f32-MFMA GEMM (gemm.hip);
//   * the recurrence is S dependent steps.  Sequences are independent, so a step is
//     tiled over (16 batch rows) x (16 hidden units) x 3 gates = 64 workgroups at B = 64;
//     inside a workgroup the 4 waves split the K = 256 contraction (split-K), each wave
//     pulls its h / W_hh slices straight into MFMA fragments (float4 per lane, 4 k-steps
//     per load), the partial 16x16 tiles meet in LDS and the gate non-linearities are
//     applied by one thread per (b, j);
//   * the two-layer recurrence of the north-star model runs as ONE persistent launch
//     (gru2_persist_*_kernel below): every workgroup keeps its weight slice in registers for all S
//     steps and the step-to-step hand-over goes through the output arrays themselves, pre-filled
//     with a bit pattern no result can have and polled with cache-bypassing loads.  Measured on
//     MI355X (tools/probe_sync2.hip): 1.3-1.9 us per hand-over, against 6.4 us for a dependent
//     kernel launch (hipGraph replay == eager) and 13 us for fence + counter barriers;
//     the per-step kernels remain for other depths and as the path when the grid cannot be co-resident;
//   * backward (BPTT) mirrors it with K = 768:  dh_t = dY_t + dh_{t+1} * z_{t+1}
//     + dGh_{t+1} . W_hh,  then the gate derivatives; all weight gradients are batched
//     TN GEMMs over the B*S rows afterwards.
#include "cpc_common.h"
#include "cpc_internal.h"
#include "gemm_tile.h"

namespace cpc {

constexpr int kH = kC;          // hidden size (north-star config: 256)
constexpr int kG = 3 * kH;

// ------------------------------------------------------------------ forward step
// grid = (H/16, ceil(B/16)), 256 threads.
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(
    const float* __restrict__ hprev, long hp_bstride, const float* __restrict__ whh,
    const float* __restrict__ bhh, const float* __restrict__ gi, float* __restrict__ y,
    float* __restrict__ Rg, float* __restrict__ Zg, float* __restrict__ Ng, float* __restrict__ GHN,
    float* __restrict__ hN, int B, int S, int t) {
    __shared__ float part[4][3][256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int koff = 64 * w + 4 * kq;

    float4 a[4];
    {
        const bool ok = hprev != nullptr && (b0 + i) < B;
        const float* ap = hprev + (ok ? (long)(b0 + i) * hp_bstride + koff : 0);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
            a[ii] = ok ? *reinterpret_cast<const float4*>(ap + 16 * ii) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* wp = whh + (long)(g * kH + j0 + i) * kH + koff;
        float4 bw[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) bw[ii] = *reinterpret_cast<const float4*>(wp + 16 * ii);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(a[ii], jj), f4c(bw[ii], jj), acc[g], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[w][g][(kq * 4 + r) * 16 + i] = acc[g][r];
    __syncthreads();

    const int row = tid >> 4, col = tid & 15;
    const int b = b0 + row, j = j0 + col;
    if (b >= B) return;
    float gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
        gh[g] = ((part[0][g][tid] + part[1][g][tid]) + (part[2][g][tid] + part[3][g][tid])) + bhh[g * kH + j];
    const long bt = (long)b * S + t;
    const float* gip = gi + bt * kG;
    const float hp = hprev ? hprev[(long)b * hp_bstride + j] : 0.f;
    const float r = sigmoidf_(gip[j] + gh[0]);
    const float z = sigmoidf_(gip[kH + j] + gh[1]);
    const float n = tanhf(gip[2 * kH + j] + r * gh[2]);
    const float h = (1.0f - z) * n + z * hp;
    y[bt * kH + j] = h;
    Rg[bt * kH + j] = r;
    Zg[bt * kH + j] = z;
    Ng[bt * kH + j] = n;
    GHN[bt * kH + j] = gh[2];
    if (hN) hN[(long)b * kH + j] = h;
}

// ------------------------------------------------------------------ backward step
// dh_t = dY_t + dh_{t+1} * z_{t+1} + dGh_{t+1} . W_hh      (terms with t+1 absent at t = S-1)
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(
    const float* __restrict__ whhT, const float* __restrict__ dY, const float* __restrict__ y,
    const float* __restrict__ h0, const float* __restrict__ Rg, const float* __restrict__ Zg,
    const float* __restrict__ Ng, const float* __restrict__ GHN, float* __restrict__ dGi,
    float* __restrict__ dGh, float* __restrict__ DH, int B, int S, int t) {
    __shared__ float part[4][256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const bool has_next = (t + 1) < S;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_next) {                                   // block-uniform
        const int koff = 192 * w + 4 * kq;
        const bool ok = (b0 + i) < B;
        const float* ap = dGh + (ok ? ((long)(b0 + i) * S + t + 1) * kG + koff : 0);
        const float* wp = whhT + (long)(j0 + i) * kG + koff;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 a[6], bw[6];
#pragma unroll
            for (int ii = 0; ii < 6; ++ii) {
                a[ii] = ok ? *reinterpret_cast<const float4*>(ap + 16 * (half * 6 + ii)) : make_float4(0.f, 0.f, 0.f, 0.f);
                bw[ii] = *reinterpret_cast<const float4*>(wp + 16 * (half * 6 + ii));
            }
#pragma unroll
            for (int ii = 0; ii < 6; ++ii)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(a[ii], jj), f4c(bw[ii], jj), acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[w][(kq * 4 + r) * 16 + i] = acc[r];
    __syncthreads();

    const int row = tid >> 4, col = tid & 15;
    const int b = b0 + row, j = j0 + col;
    if (b >= B) return;
    const long bt = (long)b * S + t;
    float dh = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + dY[bt * kH + j];
    if (has_next) dh = fmaf(DH[(bt + 1) * kH + j], Zg[(bt + 1) * kH + j], dh);
    const float r = Rg[bt * kH + j], z = Zg[bt * kH + j], n = Ng[bt * kH + j], ghn = GHN[bt * kH + j];
    const float hp = t > 0 ? y[(bt - 1) * kH + j] : (h0 ? h0[(long)b * kH + j] : 0.f);
    const float dn = dh * (1.0f - z);
    const float dzg = dh * (hp - n);
    const float dan = dn * (1.0f - n * n);
    const float daz = dzg * z * (1.0f - z);
    const float dar = dan * ghn * r * (1.0f - r);
    float* gi = dGi + bt * kG;
    float* gh = dGh + bt * kG;
    gi[j] = dar;            gh[j] = dar;
    gi[kH + j] = daz;       gh[kH + j] = daz;
    gi[2 * kH + j] = dan;   gh[2 * kH + j] = dan * r;
    DH[bt * kH + j] = dh;
}

// Fragment-ordered (B,S,256) arrays: [step][batch tile][unit/4][row 0..15][4 floats].  One wave fetch of
// MFMA operand fragments = 1 KB contiguous, one step of one tile = 16 KB contiguous.  Used for the
// hand-over buffers of the persistent launch (polling the (B,S,*) output arrays directly would put the 16
// rows of a tile 128 KB -- a power of two, one memory channel -- apart and costs 3x the hand-over latency)
// and for the backward coefficient arrays.
constexpr int kXStride = 4 * 64;                     // floats between a lane's consecutive fragments (k += 16)
__device__ __forceinline__ long xtile(int t, int tile, int ntiles, int kwidth) {
    return ((long)t * ntiles + tile) * 16 * kwidth;
}
__device__ __forceinline__ int xpos(int row, int k) { return (k >> 2) * 64 + row * 4 + (k & 3); }

// ------------------------------------------------------------------ two-layer wavefront
// The north-star autoregressor has exactly two layers.  Launch s (s = 0..S) runs layer 0's step
// t = s and layer 1's step t = s-1 side by side (blockIdx.z = layer), so the 2*S dependent steps
// become S+1 dependent launches.  Layer 1 forms its input projection W_ih1 h0_t on the fly
// (waves 4-7) next to the recurrent product W_hh1 h1_{t-1} (waves 0-3); layer 0 reads the batched
// projection and splits its recurrent product over all 8 waves.  Gate inputs are fetched before
// the MFMA chain so their latency overlaps it.
struct Gru2Fwd {
    const float* x_gi0;        // layer 0: precomputed W_ih0 x + b_ih0, (B,S,3H)
    const float* h0[2];        // initial states (B,H) or NULL
    const float* whh[2];
    const float* bhh[2];
    const float* wih1;         // layer 1 input weights (3H,H)
    const float* bih1;
    float* y[2];               // layer outputs (B,S,H)
    float* R[2]; float* Z[2]; float* N[2]; float* GHN[2];
    float* hN;                 // (2,B,H)
    float* xh[2];              // persistent launch only: hand-over copies of y[l] (see xtile)
    float* coef[2];            // persistent launch only, optional: layer l's four backward coefficient arrays (cr, cz, cnh, cni,
                               // frag_stride floats apart, fragment order) -- written by the gate threads, which hold every input
    long frag_stride;          // of gru_bwd_coef_kernel in registers: that kernel's 75 MB read-back is not needed then
    int B, S;
    int spin_limit;            // persistent launch only: polling budget of a wave (cpc_set_gru_spin_limit)
    int ntiles, xcd_pack;      // persistent launch only: see PersistIds
    int tile0, total_tiles;    // persistent launch only: this launch covers batch tiles [tile0, tile0 + ntiles) of total_tiles
    int first_sleep;           // persistent launch only: see PollPace (< 0: self-steering)
    int poll_plain;            // persistent launch only: which waves take their FIRST look with plain (L2-cached) loads (cpc_set_gru_poll_plain)
    unsigned* xsync;           // persistent launch only: NULL, or two words per batch tile (0xFFFFFFFF) for the placement check of the
                               // XCD-local hand-over (tile_on_one_xcd; cpc_set_gru_xcd_local)
};

// Component-wise f32x4 arithmetic on MFMA accumulators written out scalar by scalar: f32x4 operators compile to v_pk_{add,mul}_f32,
// the instruction class behind the co-residency corruption of rounds 1-2 -- with these three kernels
// scalar as well the library contains NO packed fp32 arithmetic and build.py's allow-list is empty.
__device__ __forceinline__ f32x4 add4(const f32x4& a, const f32x4& b) {
    f32x4 r;
    r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = a[3] + b[3];
    return r;
}
__device__ __forceinline__ f32x4 scale4(const f32x4& a, float s) {
    f32x4 r;
    r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; r[3] = a[3] * s;
    return r;
}
__device__ __forceinline__ void mfma_slice(f32x4 (&acc)[3], const float* __restrict__ arow, bool ok,
                                           const float* __restrict__ w, int j0i, int koff, int nii) {
    // acc[g] += A[16 x (16*nii)] . W[g*H + j0 + i][koff ...]^T for the three gates
    float4 a[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
        a[ii] = (ok && ii < nii) ? *reinterpret_cast<const float4*>(arow + koff + 16 * ii) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float* wp = w + (long)(g * kH + j0i) * kH + koff;
        float4 bw[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
            bw[ii] = ii < nii ? *reinterpret_cast<const float4*>(wp + 16 * ii) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
            if (ii < nii) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(a[ii], jj), f4c(bw[ii], jj), acc[g], 0, 0, 0);
            }
    }
}

// grid = (H/16, ceil(B/16), 2), 512 threads
__global__ __launch_bounds__(512) void gru2_fwd_kernel(Gru2Fwd p, int s) {
    __shared__ float part[8][3][256];
    const int layer = blockIdx.z;
    const int t = layer == 0 ? s : s - 1;
    if (t < 0 || t >= p.S) return;                         // block-uniform
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int B = p.B, S = p.S;
    const float* yl = p.y[layer];
    const float* hprev = t == 0 ? p.h0[layer] : yl + (long)(t - 1) * kH;
    const long hstride = t == 0 ? kH : (long)S * kH;

    // ---- gate-math operands, fetched early (threads 0..255 own one (b, j) each)
    const int row = (tid & 255) >> 4, col = tid & 15;
    const int b = b0 + row, j = j0 + col;
    const bool gate_thread = tid < 256 && b < B;
    const long bt = (long)b * S + t;
    float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f, hp = 0.f, bh[3] = {0.f, 0.f, 0.f};
    if (gate_thread) {
        if (layer == 0) {
            const float* gip = p.x_gi0 + bt * kG;
            gi_r = gip[j]; gi_z = gip[kH + j]; gi_n = gip[2 * kH + j];
        } else {
            gi_r = p.bih1[j]; gi_z = p.bih1[kH + j]; gi_n = p.bih1[2 * kH + j];
        }
        hp = hprev ? hprev[(long)b * hstride + j] : 0.f;
#pragma unroll
        for (int g = 0; g < 3; ++g) bh[g] = p.bhh[layer][g * kH + j];
    }

    // ---- split-K MFMA
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool bok = (b0 + i) < B;
    if (layer == 0) {                    // recurrent product over 8 waves, 32 k each
        const bool ok = bok && hprev != nullptr;
        const float* arow = hprev + (ok ? (long)(b0 + i) * hstride : 0);
        mfma_slice(acc, arow, ok, p.whh[0], j0 + i, 32 * w + 4 * kq, 2);
    } else if (w < 4) {                  // recurrent product W_hh1 h1_{t-1}
        const bool ok = bok && hprev != nullptr;
        const float* arow = hprev + (ok ? (long)(b0 + i) * hstride : 0);
        mfma_slice(acc, arow, ok, p.whh[1], j0 + i, 64 * w + 4 * kq, 4);
    } else {                             // input projection W_ih1 h0_t
        const float* arow = p.y[0] + (bok ? ((long)(b0 + i) * S + t) * kH : 0);
        mfma_slice(acc, arow, bok, p.wih1, j0 + i, 64 * (w - 4) + 4 * kq, 4);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[w][g][(kq * 4 + r) * 16 + i] = acc[g][r];
    __syncthreads();
    if (!gate_thread) return;

    float gh[3];
    const int e = tid;                   // == row*16 + col
    if (layer == 0) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
            gh[g] = (((part[0][g][e] + part[1][g][e]) + (part[2][g][e] + part[3][g][e])) +
                     ((part[4][g][e] + part[5][g][e]) + (part[6][g][e] + part[7][g][e]))) + bh[g];
    } else {
#pragma unroll
        for (int g = 0; g < 3; ++g)
            gh[g] = ((part[0][g][e] + part[1][g][e]) + (part[2][g][e] + part[3][g][e])) + bh[g];
        gi_r += (part[4][0][e] + part[5][0][e]) + (part[6][0][e] + part[7][0][e]);
        gi_z += (part[4][1][e] + part[5][1][e]) + (part[6][1][e] + part[7][1][e]);
        gi_n += (part[4][2][e] + part[5][2][e]) + (part[6][2][e] + part[7][2][e]);
    }
    const float r = sigmoidf_(gi_r + gh[0]);
    const float z = sigmoidf_(gi_z + gh[1]);
    const float n = tanhf(gi_n + r * gh[2]);
    const float h = (1.0f - z) * n + z * hp;
    p.y[layer][bt * kH + j] = h;
    p.R[layer][bt * kH + j] = r;
    p.Z[layer][bt * kH + j] = z;
    p.N[layer][bt * kH + j] = n;
    p.GHN[layer][bt * kH + j] = gh[2];
    if (t == S - 1) p.hN[((long)layer * B + b) * kH + j] = h;
}

struct Gru2Bwd {
    const float* dy;           // gradient w.r.t. the top layer's output (B,S,H)
    const float* whhT[2];      // (H,3H) transposed recurrent weights, index = layer
    const float* wih1T;        // (H,3H): W_ih of layer 1 transposed
    const float* Z[2];
    const float* cr[2]; const float* cz[2]; const float* cnh[2]; const float* cni[2];   // see gru_bwd_coef_kernel
    float* dGi[2]; float* dGh[2]; float* DH[2];
    float* xdh[2];             // persistent launch only: hand-over copies of DH[l] (fragment order)
    int B, S;
    int spin_limit;            // persistent launch only: polling budget of a wave (cpc_set_gru_spin_limit)
    int ntiles, xcd_pack;      // persistent launch only: see PersistIds
    int tile0, total_tiles;    // persistent launch only: this launch covers batch tiles [tile0, tile0 + ntiles) of total_tiles
    int first_sleep;           // persistent launch only: see PollPace (< 0: self-steering)
    int poll_plain;            // persistent launch only: which waves take their FIRST look with plain (L2-cached) loads (cpc_set_gru_poll_plain)
    unsigned* xsync;           // persistent launch only: NULL, or two words per batch tile (0xFFFFFFFF) for the placement check of the
                               // XCD-local hand-over (tile_on_one_xcd; cpc_set_gru_xcd_local)
};

// Everything in the gate derivatives that does not depend on dh, for all (b, t, j) at once and in fragment
// order, so that the recurrence itself is   dGi = dh * (cr, cz, cni),  dGh = dh * (cr, cz, cnh):
//   a = (1-z)(1-n^2)   cni = a   cnh = a r   cr = a ghn r (1-r)   cz = (h_{t-1} - n) z (1-z)
// grid = ceil(B/16)*16*S*64/256 blocks of 256 threads (one float4 of units per thread)
__global__ __launch_bounds__(256) void gru_bwd_coef_kernel(
    const float* __restrict__ R, const float* __restrict__ Z, const float* __restrict__ N,
    const float* __restrict__ GHN, const float* __restrict__ y, const float* __restrict__ h0,
    float* __restrict__ cr, float* __restrict__ cz, float* __restrict__ cnh, float* __restrict__ cni, int B, int S) {
    const int ntiles = (B + 15) >> 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)ntiles * 16 * S * 64) return;
    const int u4 = (int)(idx & 63);
    const long bt = idx >> 6;
    const int b = (int)(bt / S), t = (int)(bt - (long)b * S);
    const long dst = xtile(t, b >> 4, ntiles, kH) + xpos(b & 15, 4 * u4);
    if (b >= B) {                                    // padding rows of the last tile: zero, so that the
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);      // recurrence can load them unconditionally
        *reinterpret_cast<float4*>(cr + dst) = zero;
        *reinterpret_cast<float4*>(cz + dst) = zero;
        *reinterpret_cast<float4*>(cnh + dst) = zero;
        *reinterpret_cast<float4*>(cni + dst) = zero;
        return;
    }
    const long src = bt * kH + 4 * u4;
    const float4 r = *reinterpret_cast<const float4*>(R + src), z = *reinterpret_cast<const float4*>(Z + src);
    const float4 n = *reinterpret_cast<const float4*>(N + src), g = *reinterpret_cast<const float4*>(GHN + src);
    float4 hp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t > 0) hp = *reinterpret_cast<const float4*>(y + src - kH);
    else if (h0) hp = *reinterpret_cast<const float4*>(h0 + (long)b * kH + 4 * u4);
    float4 ocr, ocz, onh, oni;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float re = f4c(r, e), ze = f4c(z, e), ne = f4c(n, e);
        const float a = (1.0f - ze) * (1.0f - ne * ne);
        (&oni.x)[e] = a;
        (&onh.x)[e] = a * re;
        (&ocr.x)[e] = a * f4c(g, e) * re * (1.0f - re);
        (&ocz.x)[e] = (f4c(hp, e) - ne) * ze * (1.0f - ze);
    }
    *reinterpret_cast<float4*>(cr + dst) = ocr;
    *reinterpret_cast<float4*>(cz + dst) = ocz;
    *reinterpret_cast<float4*>(cnh + dst) = onh;
    *reinterpret_cast<float4*>(cni + dst) = oni;
}

// acc += G[16 rows x (NU*16 units x 3 gates)] . W^T for this lane's unit fragments: k = g*H + unit0 + 16*ii.
// One accumulator per gate, each in (ii, jj) order, combined as (r + z) + n -- shared with the persistent
// kernel, which rebuilds G from dh and the coefficients.
template <int NU>
__device__ __forceinline__ f32x4 mfma_gate_rows(f32x4 acc, const float* __restrict__ grow, bool ok,
                                                const float* __restrict__ wrow, int unit0) {
    float4 a[NU][3], bw[NU][3];
#pragma unroll
    for (int ii = 0; ii < NU; ++ii)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int k = g * kH + unit0 + 16 * ii;
            a[ii][g] = *reinterpret_cast<const float4*>(grow + k);      // grow is valid for every lane (row 0 if !ok)
            bw[ii][g] = *reinterpret_cast<const float4*>(wrow + k);
        }
    if (!ok) {
#pragma unroll
        for (int ii = 0; ii < NU; ++ii)
#pragma unroll
            for (int g = 0; g < 3; ++g) a[ii][g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4 ag[3];                                     // one chain per gate: three independent MFMA streams
#pragma unroll
    for (int g = 0; g < 3; ++g) ag[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ii = 0; ii < NU; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                ag[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(a[ii][g], jj), f4c(bw[ii][g], jj), ag[g], 0, 0, 0);
    return add4(acc, add4(add4(ag[0], ag[1]), ag[2]));
}

// Launch s (s = 0..S): blockIdx.z = 0 -> top layer (1) step t = S-1-s; blockIdx.z = 1 -> bottom
// layer (0) step t = S-s, whose incoming gradient dGi1_t . W_ih1 is formed on the fly (waves 4-7).
__global__ __launch_bounds__(512) void gru2_bwd_kernel(Gru2Bwd p, int s) {
    __shared__ float part[8][256];
    const int layer = blockIdx.z == 0 ? 1 : 0;
    const int S = p.S, B = p.B;
    const int t = layer == 1 ? S - 1 - s : S - s;
    if (t < 0 || t >= S) return;                           // block-uniform
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const bool has_next = (t + 1) < S;

    const int row = (tid & 255) >> 4, col = tid & 15;
    const int b = b0 + row, j = j0 + col;
    const bool gate_thread = tid < 256 && b < B;
    const long bt = (long)b * S + t;
    float dh0 = 0.f, cr = 0.f, cz = 0.f, cnh = 0.f, cni = 0.f;
    if (gate_thread) {
        if (layer == 1) dh0 = p.dy[bt * kH + j];
        if (has_next) dh0 = fmaf(p.DH[layer][(bt + 1) * kH + j], p.Z[layer][(bt + 1) * kH + j], dh0);
        const long c = xtile(t, blockIdx.y, gridDim.y, kH) + xpos(row, j);
        cr = p.cr[layer][c]; cz = p.cz[layer][c]; cnh = p.cnh[layer][c]; cni = p.cni[layer][c];
    }

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool bok = (b0 + i) < B;
    if (layer == 1) {                     // dGh1_{t+1} . W_hh1 over 8 waves, 32 units x 3 gates each
        if (has_next) {
            const float* grow = p.dGh[1] + (bok ? ((long)(b0 + i) * S + t + 1) * kG : 0);
            acc = mfma_gate_rows<2>(acc, grow, bok, p.whhT[1] + (long)(j0 + i) * kG, 32 * w + 4 * kq);
        }
    } else if (w < 4) {                   // dGh0_{t+1} . W_hh0, 64 units x 3 gates per wave
        if (has_next) {
            const float* grow = p.dGh[0] + (bok ? ((long)(b0 + i) * S + t + 1) * kG : 0);
            acc = mfma_gate_rows<4>(acc, grow, bok, p.whhT[0] + (long)(j0 + i) * kG, 64 * w + 4 * kq);
        }
    } else {                              // incoming gradient dGi1_t . W_ih1
        const float* grow = p.dGi[1] + (bok ? ((long)(b0 + i) * S + t) * kG : 0);
        acc = mfma_gate_rows<4>(acc, grow, bok, p.wih1T + (long)(j0 + i) * kG, 64 * (w - 4) + 4 * kq);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) part[w][(kq * 4 + rr) * 16 + i] = acc[rr];
    __syncthreads();
    if (!gate_thread) return;
    const int e = tid;
    const float dh = (((part[0][e] + part[1][e]) + (part[2][e] + part[3][e])) +
                      ((part[4][e] + part[5][e]) + (part[6][e] + part[7][e]))) + dh0;
    float* gi = p.dGi[layer] + bt * kG;
    float* gh = p.dGh[layer] + bt * kG;
    const float dar = dh * cr, daz = dh * cz;
    gi[j] = dar;            gh[j] = dar;
    gi[kH + j] = daz;       gh[kH + j] = daz;
    gi[2 * kH + j] = dh * cni;   gh[2 * kH + j] = dh * cnh;
    p.DH[layer][bt * kH + j] = dh;
}

// ------------------------------------------------------------------ persistent two-layer recurrence
// One launch for all S steps.  Workgroup (layer, 16 hidden units, 16 sequences) exactly as in the
// wavefront kernels above and with the same MFMA / summation order (bit-identical results), but
//   * its W slice is loaded into registers once,
//   * h_{t-1} (forward) / the gate gradients of step t+1 (backward) are taken from the output arrays
//     while they are being produced: the host fills them with 0xFFFFFFFF (a NaN payload no arithmetic
//     result carries), producers store with agent-scope atomics (write-through, global_store sc1) and
//     consumers poll the fragments they need with agent-scope atomic loads until no lane sees the fill
//     pattern.  Each 4-byte value validates itself, so there is no flag, counter or fence on the chain.
// All workgroups must be resident at once (host side checks the occupancy).  Polling is bounded: if
// a producer never shows up (it cannot, short of a broken device) the wave gives up, lets the fill
// pattern -- a NaN -- propagate into the outputs and the loss, and the launch still terminates.
constexpr unsigned kNotReady = 0xFFFFFFFFu;
constexpr int kSpinLimit = 1 << 20;
// Set (bit 0) by a wave that gave up polling: the NaN it lets through reaches the loss, and this says why.  Read and
// cleared by cpc_device_error_flags() (capi.hip) -- a synchronising call, made by the train loops at their logging
// points and by the tests, never on the step path.
static __device__ unsigned g_gru_poll_timeout = 0;

__device__ __forceinline__ float4 load4_coherent(const float* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                       __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}
// the same 16 bytes by PLAIN loads (wavefront-scope atomics: global_load without sc bits -- served by this XCD's L2, which the
// workgroups of a tile on the XCD then share; what it returns may be stale, so only a FIRST look may use it: poll_row)
__device__ __forceinline__ float4 load4_plain(const float* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                       __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}
__device__ __forceinline__ bool ready4(float4 v) {
    return __float_as_uint(v.x) != kNotReady && __float_as_uint(v.y) != kNotReady &&
           __float_as_uint(v.z) != kNotReady && __float_as_uint(v.w) != kNotReady;
}
__device__ __forceinline__ void store_coherent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// XCD-local hand-over (cpc_set_gru_xcd_local).  A device-scope store is written through to memory and a device-scope load of a line
// its L2 holds clean goes back there: 1.5-2 k clocks each way.  When the 32 workgroups of a batch tile all sit on ONE XCD (the
// packed numbering, PersistIds) the producers can store PLAIN instead -- the per-CU cache is write-through, the value stops in that
// XCD's L2 -- and the consumers' device-scope loads find the dirty line there (measured coherent inside an XCD, tools/probe_sync4.hip;
// the round 2.0 -> 1.45 us alone).  Workgroup placement is the dispatcher's (id % 8 on an idle MI355X) and not a contract, so
// every tile checks it once, at the start of the launch: each workgroup clears the bit of its XCC_ID in a word of ones and counts
// itself in (device scope), waits for the other 31, and the tile takes the local path only if exactly one bit went.  A tile that
// straddles XCDs -- or a wait that runs out of budget -- keeps the device-scope stores: correct either way.
__device__ __forceinline__ void store_handover(float* p, float v, bool local) {
    if (local) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool tile_on_one_xcd(unsigned* sync, int spin_limit) {
    __shared__ unsigned verdict;
    if (threadIdx.x == 0) {
        const unsigned xid = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;       // hwreg(HW_REG_XCC_ID), bits 3:0
        __hip_atomic_fetch_and(sync, ~(1u << xid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);     // 0xFFFFFFFF + 32 arrivals = 31
        int budget = spin_limit < 4096 ? spin_limit : 4096;
        while (__hip_atomic_load(sync + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 31u && budget-- > 0) __builtin_amdgcn_s_sleep(8);
        const unsigned gone = ~__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        verdict = (budget > 0 && gone != 0u && (gone & (gone - 1u)) == 0u) ? 1u : 0u;
    }
    __syncthreads();
    return verdict != 0u;
}

// Fetch NII float4 fragments (k += 16 apart) of this lane's row, re-reading until every lane of the wave
// has complete data.  Lanes whose row is outside the batch contribute zeros.
// Pacing.  Every look is a device-scope load that travels to L2 and back whatever it finds, and sixteen workgroups per XCD
// looking flat out slow one another's hand-over down (measured at B = 64: forward 0.41 -> 0.28 ms, backward 0.77 -> 0.67 ms
// once the looks that cannot succeed are left out).  The data cannot be there before the producers' gate math is done, so a
// wave first sleeps `delay` x 64 clocks.  The right delay depends on the kernel, the batch and on what else runs on the
// chip, so each wave steers its own (PollPace): a look that had to be repeated came too early (delay += 2; 4 until round 6), four first-time
// hits in a row may have come late (delay -= 1); the steady state is about one repeated look in twenty steps.
// fixed >= 0 pins the delay instead (cpc_set_gru_poll_pacing).
struct PollPace {
    int delay, streak, fixed, up, clean;
    // fixed_ >= 0: pinned delay; -1: self-steering with the default steps (up 2, one down per 4 clean steps -- round 6: (2, 4)
    // against the (4, 4) of rounds 2-5 is worth 4 us forward and 9 us backward at B = 64, tools/ab_gru_pace.py); <= -2: self-steering
    // with up = (-fixed_) >> 4, clean = (-fixed_) & 15 (cpc_set_gru_poll_pacing: A/B of the steering constants)
    __device__ explicit PollPace(int fixed_) : delay(fixed_ > 0 ? fixed_ : 0), streak(0), fixed(fixed_), up(2), clean(4) {
        if (fixed_ <= -2) { up = (-fixed_) >> 4; clean = (-fixed_) & 15; if (clean < 1) clean = 1; }
    }
    __device__ __forceinline__ void update(int repeats) {     // wave-uniform
        if (fixed >= 0) return;
        if (repeats == 0) {
            if (++streak >= clean) { streak = 0; delay = delay > 0 ? delay - 1 : 0; }
        } else {
            streak = 0;
            delay = delay + up < 96 ? delay + up : 96;
        }
    }
};

// plain_first: the first look with plain loads (load4_plain) -- a hand-over address is read once per launch by a wave, the
// buffers were filled by a previous launch and written since by write-through stores only, so a line this XCD's L2 does not hold
// yet comes from memory as it is now, and the other workgroups of the tile on this XCD hit it there instead of crossing the
// fabric each; a line fetched too early stays stale in that L2, which the repeated looks (always device scope) get around.
template <int NII>
__device__ __forceinline__ void poll_row(const float* __restrict__ row, bool ok, float4 (&a)[NII], int& budget,
                                         PollPace& pace, bool plain_first = false) {
    for (int q = 0; q < pace.delay; ++q) __builtin_amdgcn_s_sleep(1);
    // Unconditional loads (a predicated load costs a branch and a full vmcnt(0) each): rows past the batch
    // are inside the buffer, never written, and masked out below.
    if (plain_first) {                                            // wave-uniform
#pragma unroll
        for (int ii = 0; ii < NII; ++ii) a[ii] = load4_plain(row + kXStride * ii);
    } else {
#pragma unroll
        for (int ii = 0; ii < NII; ++ii) a[ii] = load4_coherent(row + kXStride * ii);
    }
    int repeats = 0;
    for (;;) {
        bool rdy = true;
#pragma unroll
        for (int ii = 0; ii < NII; ++ii) rdy = rdy && ready4(a[ii]);
        if (__all(rdy || !ok) || budget <= 0) break;
        --budget;
        ++repeats;
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int ii = 0; ii < NII; ++ii)                          // re-read only what was incomplete
            if (ok && !ready4(a[ii])) a[ii] = load4_coherent(row + kXStride * ii);
    }
    pace.update(repeats);
    if (budget <= 0) atomicOr(&g_gru_poll_timeout, 1u);
    if (!ok) {
#pragma unroll
        for (int ii = 0; ii < NII; ++ii) a[ii] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int NII>
__device__ __forceinline__ void load_row_plain(const float* __restrict__ row, bool ok, float4 (&a)[NII]) {
#pragma unroll
    for (int ii = 0; ii < NII; ++ii)
        a[ii] = ok ? *reinterpret_cast<const float4*>(row + 16 * ii) : make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int NII>
__device__ __forceinline__ void load_gate_weights(float4 (&bw)[3][NII], const float* __restrict__ w, int j0i, int koff) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ii = 0; ii < NII; ++ii)
            bw[g][ii] = *reinterpret_cast<const float4*>(w + (long)(g * kH + j0i) * kH + koff + 16 * ii);
}

template <int NII>
__device__ __forceinline__ void mfma_gates(f32x4 (&acc)[3], const float4 (&a)[NII], const float4 (&bw)[3][NII]) {
#pragma unroll
    for (int ii = 0; ii < NII; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int g = 0; g < 3; ++g)              // three independent chains back to back
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(a[ii], jj), f4c(bw[g][ii], jj), acc[g], 0, 0, 0);
}

// The same products on the 16-bit matrix pipe: both operands as two fp16 pieces of x * s (s a power of two, see
// gemm_tile.h), three v_mfma_f32_16x16x32_f16 per product (hh + hl + lh; the dropped ll term is <= 2^-22 of the
// product) instead of eight exact-f32 16x16x4 issues per 8 k.  |h| < 1 for every recurrent operand (a convex
// combination of tanh outputs when h0 is absent), so sa is a constant; sb comes from the wave's own weight slice.
// Slot e of the 8-k operand is element (e & 3) of fragment (e >> 2), for A and B alike.
struct H2Frag { f16x8 h, l; };
__device__ __forceinline__ H2Frag split2h_8(const float4& u, const float4& v, float s) {
    const float x[8] = {u.x * s, u.y * s, u.z * s, u.w * s, v.x * s, v.y * s, v.z * s, v.w * s};
    H2Frag f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hi = (_Float16)x[e];
        f.h[e] = hi;
        f.l[e] = (_Float16)(x[e] - (float)hi);
    }
    return f;
}
template <int NII>
__device__ __forceinline__ float split_gate_weights(const float4 (&bw)[3][NII], H2Frag (&bh)[3][NII / 2]) {
    float m = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ii = 0; ii < NII; ++ii)
            m = fmaxf(fmaxf(m, fmaxf(fabsf(bw[g][ii].x), fabsf(bw[g][ii].y))), fmaxf(fabsf(bw[g][ii].z), fabsf(bw[g][ii].w)));
    const float sb = scale_for_amax(wave_max(m));
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int pr = 0; pr < NII / 2; ++pr) bh[g][pr] = split2h_8(bw[g][2 * pr], bw[g][2 * pr + 1], sb);
    return sb;
}
template <int NII>
__device__ __forceinline__ void mfma_gates_h2(f32x4 (&acc)[3], const float4 (&a)[NII], const H2Frag (&bh)[3][NII / 2],
                                              float sa) {
#pragma unroll
    for (int pr = 0; pr < NII / 2; ++pr) {
        const H2Frag af = split2h_8(a[2 * pr], a[2 * pr + 1], sa);
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af.h, bh[g][pr].h, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af.h, bh[g][pr].l, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af.l, bh[g][pr].h, acc[g], 0, 0, 0);
    }
}

// Phase trace of the persistent kernels (tools/time_gru_phases.py; library built with -DCPC_GRU_TIMING by tools/build_timing_lib.sh;
// in the product build every member is empty): shader-clock laps of one lane per wave, summed over the steps 16 .. S - 16, per
// (direction, workgroup, wave).  MFMA waves: 0 poll, 1 operand math + MFMAs until their results are in LDS-store flight, 2 LDS
// stores + barrier; gate waves: 3 barrier wait, 4 partial sums + gate math until the coherent store is issued, 5 the other
// stores and the next step's prefetch.
#ifdef CPC_GRU_TIMING
__device__ unsigned long long g_gru_phase[2][512][12][8];
struct PhaseClock {
    unsigned long long last, sum[8];
    __device__ PhaseClock() : last(0) { for (int i = 0; i < 8; ++i) sum[i] = 0; }
    __device__ __forceinline__ void start() {
        __builtin_amdgcn_sched_barrier(0);
        last = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void lap(int i, bool on) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long now = __builtin_readcyclecounter();
        if (on) sum[i] += now - last;
        last = now;
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void flush(int dir) {
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 512)
            for (int i = 0; i < 8; ++i) g_gru_phase[dir][blockIdx.x][threadIdx.x >> 6][i] = sum[i];
    }
};
#else
struct PhaseClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void lap(int, bool) {}
    __device__ __forceinline__ void flush(int) {}
};
#endif

struct PersistIds {
    int layer, j0, b0, tile, ntiles, G;
    bool valid;
    // G batch tiles of 32 workgroups each (2 layers x 16 unit tiles).  pack: workgroup b is dispatched to XCD b % 8, so the
    // grid is 256 * ceil(G / 8) and tile (slot / 32) * 8 + xcd takes the 32 slots of its XCD -- every hand-over of a tile
    // then stays inside one L2 instead of crossing the fabric (surplus workgroups exit at once; at B = 64 four XCDs
    // run the recurrence and four are left to the side-stream kernels).  Otherwise (default, faster): grid = 32 G, ids of one
    // tile G apart.
    // A launch may cover only the batch tiles [tile0, tile0 + G) of `total` (a batch too large for one resident grid runs as
    // several launches one after the other: sequences are independent); everything is addressed by the global tile.
    __device__ PersistIds(int G, int pack, int tile0, int total) {
        int rest;
        if (pack) {
            const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
            tile = (slot >> 5) * 8 + xcd;
            rest = slot & 31;
        } else {
            tile = blockIdx.x % G;
            rest = blockIdx.x / G;
        }
        valid = tile < G && tile0 + tile < total;
        tile += tile0;
        ntiles = total;
        this->G = G;
        b0 = tile * 16;
        layer = rest >> 4;
        j0 = (rest & 15) * 16;
    }
};

// Workgroup = 8 MFMA waves (poll -> MFMA -> partial tile to LDS) + 4 gate waves (one thread per (b, j):
// partial sums, non-linearities, stores, and the prefetch of the next step's saved operands).  The split
// keeps the gate threads' global loads/stores -- and their acknowledgements -- off the vmcnt of the polling
// waves, whose poll -> MFMA -> barrier chain is the critical path of the recurrence.
constexpr int kPersistThreads = 768;
constexpr int kMfmaWaves = 8;

// NT (1 or 2): batch tiles per workgroup.  A batch whose 32 workgroups per tile cannot all be resident at once (B = 256 on 256
// CUs: 16 tiles of 32) used to run as serial launches over chunks of tiles -- two full recurrences of step latency.  With NT = 2 a
// workgroup owns tiles `tile` and `tile + G` and alternates between them inside every time step: the W slice in its registers
// serves both, and while the hand-over of one tile's step is in flight (store -> visible -> load, the 1.3 us that bound a step)
// it computes the other's.  The two recurrences are independent and each keeps the arithmetic and the summation order of the
// single-tile kernel: bit-identical results (tests/test_emu_gru.py, tests/test_gpu_fused_step.py).
template <int LAYER, bool H2, int NT>
__device__ __forceinline__ void persist_fwd(const Gru2Fwd& p, float (&part)[2][8][3][256], const PersistIds& id, bool local) {
    const int j0 = id.j0;
    constexpr int NII = LAYER == 0 ? 2 : 4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int B = p.B, S = p.S;
    float* __restrict__ yl = p.y[LAYER];
    int tl[NT];                                      // this workgroup's batch tiles (tl[k] >= ntiles: none)
#pragma unroll
    for (int k = 0; k < NT; ++k) tl[k] = id.tile + k * id.G;

    if (w < kMfmaWaves) {
        const int i = lane & 15, kq = lane >> 4;
        const bool recurrent = LAYER == 0 || w < 4;  // this wave's product: W_hh h_{t-1}  (else W_ih1 h0_t)
        const int koff = LAYER == 0 ? 32 * w + 4 * kq : 64 * (w & 3) + 4 * kq;
        float4 bw[3][NII];
        load_gate_weights<NII>(bw, recurrent ? p.whh[LAYER] : p.wih1, j0 + i, koff);
        H2Frag bh[3][NII / 2];
        float sa = 1.f, inv = 1.f;
        if constexpr (H2) {
            sa = scale_for_amax(1.0f);
            inv = 1.0f / (sa * split_gate_weights<NII>(bw, bh));      // powers of two: exact
        }
        const float* __restrict__ xsrc = (recurrent ? p.xh[LAYER] : p.xh[0]) + xpos(i, koff);
        bool tok[NT], bok[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            tok[k] = tl[k] < id.ntiles;
            bok[k] = tok[k] && (tl[k] * 16 + i) < B;
        }
        int budget = p.spin_limit;
        PollPace pace(p.first_sleep);
        PhaseClock pc;
        pc.start();
        for (int t = 0; t < S; ++t) {
            const bool traced = t >= 16 && t < S - 16;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                float4 a[NII];
                if (!tok[k]) {                       // (workgroup-uniform: no second tile -- it still keeps the barrier count)
#pragma unroll
                    for (int ii = 0; ii < NII; ++ii) a[ii] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else if (recurrent) {
                    if (t == 0) load_row_plain<NII>(p.h0[LAYER] ? p.h0[LAYER] + (long)(tl[k] * 16 + i) * kH + koff : nullptr,
                                                    bok[k] && p.h0[LAYER] != nullptr, a);
                    else poll_row<NII>(xsrc + xtile(t - 1, tl[k], id.ntiles, kH), bok[k], a, budget, pace, (p.poll_plain & 4) != 0);
                } else {
                    poll_row<NII>(xsrc + xtile(t, tl[k], id.ntiles, kH), bok[k], a, budget, pace, (p.poll_plain & 1) != 0);
                }
                pc.lap(0, traced);
                f32x4 acc[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (H2) {
                    mfma_gates_h2<NII>(acc, a, bh, sa);
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g] = scale4(acc[g], inv);
                } else {
                    mfma_gates<NII>(acc, a, bw);
                }
                float (&pt)[8][3][256] = part[(t * NT + k) & 1];
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pt[w][g][(kq * 4 + r) * 16 + i] = acc[g][r];
                pc.lap(1, traced);
                __syncthreads();
                pc.lap(2, traced);
            }
        }
        pc.flush(0);
        return;
    }

    // ---- gate waves
    const int e = tid - kMfmaWaves * 64;             // == row * 16 + col
    const int j = j0 + (e & 15);
    int b[NT];
    bool tok[NT], live[NT];
    float bh[3] = {0.f, 0.f, 0.f}, gi[NT][3], hp[NT];
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = p.bhh[LAYER][g * kH + j];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        b[k] = tl[k] * 16 + (e >> 4);
        tok[k] = tl[k] < id.ntiles;
        live[k] = tok[k] && b[k] < B;
        gi[k][0] = gi[k][1] = gi[k][2] = 0.f;
        hp[k] = 0.f;
        if (live[k]) {
            if (LAYER == 1) {
#pragma unroll
                for (int g = 0; g < 3; ++g) gi[k][g] = p.bih1[g * kH + j];
            } else {
                const float* gip = p.x_gi0 + (long)b[k] * S * kG;
                gi[k][0] = gip[j]; gi[k][1] = gip[kH + j]; gi[k][2] = gip[2 * kH + j];
            }
            if (p.h0[LAYER]) hp[k] = p.h0[LAYER][(long)b[k] * kH + j];
        }
    }
    float* __restrict__ cf = p.coef[LAYER];
    const int cpos = xpos(e >> 4, j);
    PhaseClock pc;
    pc.start();
    for (int t = 0; t < S; ++t) {
        const bool traced = t >= 16 && t < S - 16;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            __syncthreads();
            pc.lap(3, traced);
            if (!live[k]) {
                if (cf && tok[k]) {                                   // padding rows of the last tile: zero coefficients, so that
                    const long c = xtile(t, tl[k], id.ntiles, kH) + cpos;     // the backward recurrence can load them unconditionally
                    cf[c] = 0.f; cf[c + p.frag_stride] = 0.f; cf[c + 2 * p.frag_stride] = 0.f; cf[c + 3 * p.frag_stride] = 0.f;
                }
                continue;
            }
            const long bt = (long)b[k] * S + t;
            float (&pt)[8][3][256] = part[(t * NT + k) & 1];
            float gh[3], gi_r = gi[k][0], gi_z = gi[k][1], gi_n = gi[k][2];
            if (LAYER == 0) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    gh[g] = (((pt[0][g][e] + pt[1][g][e]) + (pt[2][g][e] + pt[3][g][e])) +
                             ((pt[4][g][e] + pt[5][g][e]) + (pt[6][g][e] + pt[7][g][e]))) + bh[g];
            } else {
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    gh[g] = ((pt[0][g][e] + pt[1][g][e]) + (pt[2][g][e] + pt[3][g][e])) + bh[g];
                gi_r += (pt[4][0][e] + pt[5][0][e]) + (pt[6][0][e] + pt[7][0][e]);
                gi_z += (pt[4][1][e] + pt[5][1][e]) + (pt[6][1][e] + pt[7][1][e]);
                gi_n += (pt[4][2][e] + pt[5][2][e]) + (pt[6][2][e] + pt[7][2][e]);
            }
            const float r = sigmoidf_(gi_r + gh[0]);
            const float z = sigmoidf_(gi_z + gh[1]);
            const float n = tanhf(gi_n + r * gh[2]);
            const float h = (1.0f - z) * n + z * hp[k];
            store_handover(p.xh[LAYER] + xtile(t, tl[k], id.ntiles, kH) + xpos(e >> 4, j), h, local);   // first: others wait for it
            pc.lap(4, traced);
            yl[bt * kH + j] = h;
            p.R[LAYER][bt * kH + j] = r;
            p.Z[LAYER][bt * kH + j] = z;
            p.N[LAYER][bt * kH + j] = n;
            p.GHN[LAYER][bt * kH + j] = gh[2];
            if (t == S - 1) p.hN[((long)LAYER * B + b[k]) * kH + j] = h;
            if (cf) {                                                 // gru_bwd_coef_kernel's arithmetic, on the values it would re-read
                const float a = (1.0f - z) * (1.0f - n * n);
                const long c = xtile(t, tl[k], id.ntiles, kH) + cpos;
                cf[c] = a * gh[2] * r * (1.0f - r);                   // cr
                cf[c + p.frag_stride] = (hp[k] - n) * z * (1.0f - z); // cz   (hp: h_{t-1})
                cf[c + 2 * p.frag_stride] = a * r;                    // cnh
                cf[c + 3 * p.frag_stride] = a;                        // cni
            }
            hp[k] = h;
            if (LAYER == 0 && t + 1 < S) {                            // next step's input projection, a step ahead
                const float* gip = p.x_gi0 + (bt + 1) * kG;
                gi[k][0] = gip[j]; gi[k][1] = gip[kH + j]; gi[k][2] = gip[2 * kH + j];
            }
            pc.lap(5, traced);
        }
    }
    pc.flush(0);
}

// grid = 32 * ceil(B/16) workgroups (1-D), 768 threads; xh[0] and xh[1] pre-filled with 0xFF bytes
template <int NT>
__global__ __launch_bounds__(kPersistThreads) void gru2_persist_fwd_kernel(Gru2Fwd p) {
    __shared__ float part[2][8][3][256];
    const PersistIds id(p.ntiles, p.xcd_pack, p.tile0, p.total_tiles);
    if (!id.valid) return;
    const bool local = p.xsync != nullptr && tile_on_one_xcd(p.xsync + 2 * id.tile, p.spin_limit);
    if (id.layer == 0) persist_fwd<0, false, NT>(p, part, id, local);
    else persist_fwd<1, false, NT>(p, part, id, local);
}
// the same with the recurrent products on the fp16 pipe (two-piece split operands); h0 must be absent (|h| < 1)
template <int NT>
__global__ __launch_bounds__(kPersistThreads) void gru2_persist_fwd_h2_kernel(Gru2Fwd p) {
    __shared__ float part[2][8][3][256];
    const PersistIds id(p.ntiles, p.xcd_pack, p.tile0, p.total_tiles);
    if (!id.valid) return;
    const bool local = p.xsync != nullptr && tile_on_one_xcd(p.xsync + 2 * id.tile, p.spin_limit);
    if (id.layer == 0) persist_fwd<0, true, NT>(p, part, id, local);
    else persist_fwd<1, true, NT>(p, part, id, local);
}

template <int NU>
__device__ __forceinline__ void load_coef(float4 (&cf)[NU][3], const float* __restrict__ c0, const float* __restrict__ c1,
                                          const float* __restrict__ c2, long off) {
    // unconditional and unmasked (the coefficient kernel zero-fills the padding rows): nothing here may
    // depend on the loaded values, or their latency lands in front of the barrier
#pragma unroll
    for (int ii = 0; ii < NU; ++ii) {
        cf[ii][0] = *reinterpret_cast<const float4*>(c0 + off + kXStride * ii);
        cf[ii][1] = *reinterpret_cast<const float4*>(c1 + off + kXStride * ii);
        cf[ii][2] = *reinterpret_cast<const float4*>(c2 + off + kXStride * ii);
    }
}

// Backward hand-over carries dh (256 values per row), not the 768 gate gradients: the MFMA waves rebuild
// their operand fragments as dh * coefficient (the very products the gate threads store to dGi / dGh),
// which cuts the polled volume -- the cost that sets the step time -- to a third.
// (The fp16-split products of the forward, mfma_gates_h2, were tried here too -- with a per-wave, per-step power-of-two
// scale taken from the operand's own max, since gate gradients have no a-priori bound: parity was fine, but the extra
// VALU work (max, scale, 48 conversions per lane and step) and 23 spilled registers made the step 0.23 ms slower.)
// NT: batch tiles per workgroup, as in persist_fwd.  With NT = 2 the coefficients of a step are requested together with its
// first poll instead of one step ahead (two prefetched sets would not fit the register budget of three waves per SIMD); the
// other tile's work hides them.
template <int LAYER, int NT>
__device__ __forceinline__ void persist_bwd(const Gru2Bwd& p, float (&part)[2][8][256], const PersistIds& id, bool local) {
    constexpr int NU = LAYER == 1 ? 2 : 4;           // unit fragments per lane (x 3 gates = MFMA fragments)
    const int j0 = id.j0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int B = p.B, S = p.S;
    int tl[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) tl[k] = id.tile + k * id.G;

    if (w < kMfmaWaves) {
        const int i = lane & 15, kq = lane >> 4;
        const bool recurrent = LAYER == 1 || w < 4;  // dGh_{t+1} . W_hh   (else dGi1_t . W_ih1)
        const int unit0 = LAYER == 1 ? 32 * w + 4 * kq : 64 * (w & 3) + 4 * kq;
        float4 bw[NU][3];
        {
            const float* wrow = (recurrent ? p.whhT[LAYER] : p.wih1T) + (long)(j0 + i) * kG;
#pragma unroll
            for (int ii = 0; ii < NU; ++ii)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    bw[ii][g] = *reinterpret_cast<const float4*>(wrow + g * kH + unit0 + 16 * ii);
        }
        const int sl = recurrent ? LAYER : 1;                     // layer whose dh / coefficients this wave reads
        const float* __restrict__ c0 = p.cr[sl];
        const float* __restrict__ c1 = p.cz[sl];
        const float* __restrict__ c2 = recurrent ? p.cnh[sl] : p.cni[sl];
        const long lane_off = xpos(i, unit0);
        const float* __restrict__ xsrc = p.xdh[sl] + lane_off;
        bool tok[NT], bok[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            tok[k] = tl[k] < id.ntiles;
            bok[k] = tok[k] && (tl[k] * 16 + i) < B;
        }
        float4 cf[NU][3];
        if (NT == 1 && !recurrent) load_coef<NU>(cf, c0, c1, c2, xtile(S - 1, tl[0], id.ntiles, kH) + lane_off);
        int budget = p.spin_limit;
        PollPace pace(p.first_sleep);
        PhaseClock pc;
        pc.start();
        for (int t = S - 1; t >= 0; --t) {
            const int ts = recurrent ? t + 1 : t;                 // step whose gate gradients are this wave's operand
            const bool traced = t >= 16 && t < S - 16;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ts < S && tok[k]) {
                    if (NT > 1) load_coef<NU>(cf, c0, c1, c2, xtile(ts, tl[k], id.ntiles, kH) + lane_off);
                    float4 dh[NU];
                    poll_row<NU>(xsrc + xtile(ts, tl[k], id.ntiles, kH), bok[k], dh, budget, pace,
                                 (p.poll_plain & (recurrent ? 8 : 2)) != 0);
                    pc.lap(0, traced);
                    f32x4 ag[3];
#pragma unroll
                    for (int g = 0; g < 3; ++g) ag[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                    // every operand dh * coefficient FIRST, then the MFMAs back to back: a VALU multiply between two MFMAs costs
                    // the matrix pipe ~30 clocks each (measured with tools/time_gru_phases.py: 60-66 clocks per MFMA with the
                    // products computed on the way, against the instruction's 32)
                    // (in two halves: 48 live products on top of the weights and the coefficients would spill at three waves per SIMD)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        float4 op[NU / 2][3];
#pragma unroll
                        for (int ii = 0; ii < NU / 2; ++ii)
#pragma unroll
                            for (int g = 0; g < 3; ++g) {
                                const float4 d = dh[hf * (NU / 2) + ii], c = cf[hf * (NU / 2) + ii][g];
                                op[ii][g] = make_float4(d.x * c.x, d.y * c.y, d.z * c.z, d.w * c.w);
                            }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ii = 0; ii < NU / 2; ++ii)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                                for (int g = 0; g < 3; ++g)
                                    ag[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(op[ii][g], jj), f4c(bw[hf * (NU / 2) + ii][g], jj), ag[g], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc = add4(acc, add4(add4(ag[0], ag[1]), ag[2]));
                }
                float (&pt)[8][256] = part[(t * NT + k) & 1];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) pt[w][(kq * 4 + rr) * 16 + i] = acc[rr];
                pc.lap(1, traced);
                if (NT == 1 && t > 0)                             // next iteration's coefficients: step ts - 1 (behind the partials:
                    load_coef<NU>(cf, c0, c1, c2, xtile(ts - 1, tl[0], id.ntiles, kH) + lane_off);    // the gate waves wait for those)
                __syncthreads();
                pc.lap(2, traced);
            }
        }
        pc.flush(1);
        return;
    }

    // ---- gate waves
    const int e = tid - kMfmaWaves * 64;
    const int row = e >> 4;
    const int j = j0 + (e & 15);
    const int xp = xpos(row, j);
    int b[NT];
    bool live[NT];
    float dyv[NT], z[NT], cr[NT], cz[NT], cnh[NT], cni[NT];       // operands of the step about to run
    float dh_next[NT], z_next[NT];
    auto fetch = [&](int k, int t) __attribute__((always_inline)) {
        const long bt = (long)b[k] * S + t;
        if (LAYER == 1) dyv[k] = p.dy[bt * kH + j];
        z[k] = p.Z[LAYER][bt * kH + j];
        const long c = xtile(t, tl[k], id.ntiles, kH) + xp;
        cr[k] = p.cr[LAYER][c]; cz[k] = p.cz[LAYER][c]; cnh[k] = p.cnh[LAYER][c]; cni[k] = p.cni[LAYER][c];
    };
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        b[k] = tl[k] * 16 + row;
        live[k] = tl[k] < id.ntiles && b[k] < B;
        dyv[k] = z[k] = cr[k] = cz[k] = cnh[k] = cni[k] = 0.f;
        dh_next[k] = z_next[k] = 0.f;
        if (live[k]) fetch(k, S - 1);
    }
    PhaseClock pc;
    pc.start();
    for (int t = S - 1; t >= 0; --t) {
        const bool traced = t >= 16 && t < S - 16;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            __syncthreads();
            pc.lap(3, traced);
            if (!live[k]) continue;
            const long bt = (long)b[k] * S + t;
            float (&pt)[8][256] = part[(t * NT + k) & 1];
            float dh0 = LAYER == 1 ? dyv[k] : 0.f;
            if ((t + 1) < S) dh0 = fmaf(dh_next[k], z_next[k], dh0);
            const float dh = (((pt[0][e] + pt[1][e]) + (pt[2][e] + pt[3][e])) +
                              ((pt[4][e] + pt[5][e]) + (pt[6][e] + pt[7][e]))) + dh0;
            store_handover(p.xdh[LAYER] + xtile(t, tl[k], id.ntiles, kH) + xp, dh, local);   // first: others wait for it
            pc.lap(4, traced);
            float* gi = p.dGi[LAYER] + bt * kG;
            float* gh = p.dGh[LAYER] + bt * kG;
            const float dar = dh * cr[k], daz = dh * cz[k];
            // 200 MB of gate gradients for the weight-gradient GEMMs that follow: streamed past the caches the polling
            // traffic lives in (non-temporal: 0.834 vs 0.853 ms for the whole backward)
            __builtin_nontemporal_store(dar, gi + j); __builtin_nontemporal_store(dar, gh + j);
            __builtin_nontemporal_store(daz, gi + kH + j); __builtin_nontemporal_store(daz, gh + kH + j);
            __builtin_nontemporal_store(dh * cni[k], gi + 2 * kH + j); __builtin_nontemporal_store(dh * cnh[k], gh + 2 * kH + j);
            __builtin_nontemporal_store(dh, p.DH[LAYER] + bt * kH + j);
            dh_next[k] = dh;
            z_next[k] = z[k];
            if (t > 0) fetch(k, t - 1);                               // a step ahead: memory latency off the chain
            pc.lap(5, traced);
        }
    }
    pc.flush(1);
}

// grid / block as the forward; xdh[0] and xdh[1] pre-filled with 0xFF bytes
template <int NT>
__global__ __launch_bounds__(kPersistThreads) void gru2_persist_bwd_kernel(Gru2Bwd p) {
    __shared__ float part[2][8][256];
    const PersistIds id(p.ntiles, p.xcd_pack, p.tile0, p.total_tiles);
    if (!id.valid) return;
    const bool local = p.xsync != nullptr && tile_on_one_xcd(p.xsync + 2 * id.tile, p.spin_limit);
    if (id.layer == 0) persist_bwd<1, NT>(p, part, id, local);    // the top layer leads
    else persist_bwd<0, NT>(p, part, id, local);
}

// ------------------------------------------------------------------ host side
struct GruLayout {
    long R[8], Z[8], N[8], GHN[8], Y[8];     // saved (per layer); Y only for l < nl-1
    long saved_total;
    long gi, xh, xh_floats, fwd_total;       // forward scratch (xh: hand-over buffers of the persistent launch)
    long whhT, wihT, dGi, dGh, DH, mid[2], part, tmp;
    long whhT2, wihT2, dGi2, dGh2, DH2;      // second set for the two-layer wavefront
    long coef, xdh, frag_floats;             // two-layer path: 8 coefficient arrays, 2 hand-over buffers (fragment order)
    long sync_floats;                        // ... and, behind either pair of hand-over buffers, two words per batch tile (tile_on_one_xcd)
    long bwd_total;
};

static bool gru_layout(int B, int S, int nl, GruLayout& g) {
    if (B <= 0 || S <= 0 || nl <= 0 || nl > 8) return false;
    const long bsh = align64l((long)B * S * kH);
    long o = 0;
    for (int l = 0; l < nl; ++l) {
        g.R[l] = o; o += bsh;
        g.Z[l] = o; o += bsh;
        g.N[l] = o; o += bsh;
        g.GHN[l] = o; o += bsh;
        g.Y[l] = -1;
        if (l < nl - 1) { g.Y[l] = o; o += bsh; }
    }
    g.saved_total = o;
    const long tiles16 = (long)cdiv(B, 16) * 16;
    g.gi = 0;
    g.xh = align64l((long)B * S * kG);
    g.xh_floats = nl == 2 ? align64l((long)S * tiles16 * kH) : 0;
    g.sync_floats = nl == 2 ? align64l(2L * cdiv(B, 16)) : 0;
    g.fwd_total = g.xh + 2 * g.xh_floats + g.sync_floats;
    o = 0;
    g.whhT = o; o += (long)kH * kG;
    g.wihT = o; o += (long)kH * kG;
    g.dGi = o; o += align64l((long)B * S * kG);
    g.dGh = o; o += align64l((long)B * S * kG);
    g.DH = o; o += bsh;
    g.mid[0] = o; o += bsh;
    g.mid[1] = o; o += bsh;
    g.part = o; o += align64l(std::max(tn_gemm_part_floats(B * S, kG, kH), tn_gemm_batch_part_floats(4, B * S, kG, kH)));
    g.tmp = o; o += 4 * align64l((long)kRowsSumGroups * kG);      // one per bias reduction of the batched sum
    g.whhT2 = o; o += (long)kH * kG;
    g.wihT2 = o; o += (long)kH * kG;
    g.dGi2 = o; o += align64l((long)B * S * kG);
    g.dGh2 = o; o += align64l((long)B * S * kG);
    g.DH2 = o; o += bsh;
    g.frag_floats = nl == 2 ? align64l((long)S * tiles16 * kH) : 0;
    g.coef = o; o += 8 * g.frag_floats;
    g.xdh = o; o += 2 * g.frag_floats + g.sync_floats;
    g.bwd_total = o;
    return true;
}

}  // namespace cpc

using namespace cpc;

namespace {
int g_gru_spin_limit = kSpinLimit;
int g_gru_first_sleep[2] = {-1, -1};   // forward, backward (cpc_set_gru_poll_pacing); < 0: self-steering
int g_gru_poll_plain = 15; // cpc_set_gru_poll_plain
int g_gru_xcd_local = 1;   // cpc_set_gru_xcd_local: bit 0 forward, bit 1 backward (launches with one batch tile per workgroup: B <= 128
                           // on MI355X), bits 2 / 3 the same for any launch
// (defaults from profiles/r6_ab_gru_handover.txt, MI355X: B = 64 forward alone 317 -> 272 us with the local hand-over and plain
//  first looks, B = 128 351 -> 290, step at B = 64 2.77 -> 2.71 ms; the backward -- which runs beside the criterion's dz path -- gains
//  from the plain first looks only and LOSES with the packed numbering in the step (2.84 ms); with two tiles per workgroup
//  (B = 256) the local forward loses (518 -> 583 us) and so do the backward's plain looks (1620 -> 1758 us): both stay off there)
int g_gru_xcd_pack = 0;    // persistent launches: 0 (default) = tiles interleaved over the XCDs, 1 = one batch tile per XCD where the
                           // device has 8 (PersistIds; measured slower: B = 64 forward +42 us, backward +70 us -- what a tile gains
                           // in hand-over distance it loses to 32 instead of 16 polling workgroups on its L2), 2 = packed numbering
                           // forced (tests on the emulator)
int g_gru_mode = 2;        // 0: per-step launches; 1: persistent two-layer recurrence when the grid fits the device, exact-f32
                           // MFMAs; 2 (default): the same with the forward's recurrent products on the fp16 pipe (two-piece
                           // split, 3 MFMAs per product; exact-f32 when the caller supplies h0, whose size is unknown)

// Grid of a persistent launch over G batch tiles (PersistIds), or 0 if its 32 G working workgroups (kPersistThreads each)
// cannot all be resident at once.  *pack: one tile per XCD -- when asked for (g_gru_xcd_pack 1) and the device is 8 XCDs of
// cus / 8 CUs each with room for 32 * ceil(G / 8) workgroups per XCD; g_gru_xcd_pack == 2 forces the packed numbering on any
// device whose dispatcher hands out workgroups in id order as slots free up (the emulator; surplus ids exit at once).
// Batch tiles per persistent launch: all of them if their 32 G workgroups can be resident together, else the largest
// count that can (0: not even one tile).  g_gru_chunk_tiles > 0 caps it (tests: chunking on a device that would not need it).
int g_gru_chunk_tiles = 0;
static int persist_chunk(const void* kernel, int G) {
    int dev = 0, cus = 0, occ = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, kPersistThreads, 0) != hipSuccess) return 0;
    int fit = (int)(((long)cus * occ) / 32);
    if (g_gru_chunk_tiles > 0 && fit > g_gru_chunk_tiles) fit = g_gru_chunk_tiles;
    return fit >= G ? G : fit;
}

// How a persistent launch covers `total` batch tiles: *G tile slots (32 workgroups each) per launch, *NT tiles per slot (1, or 2
// when the batch does not fit the device in one launch and g_gru_tiles_per_wg allows), launches of G * NT tiles one after the
// other.  fit1 / fit2: tiles whose workgroups can be resident at once for the NT = 1 / NT = 2 kernels (persist_chunk).
int g_gru_tiles_per_wg = 2;      // cpc_set_gru_tiles_per_wg: 1 = serial chunks only (rounds 2-4), 2 = a workgroup may own two tiles
static void persist_plan(int total, int fit1, int fit2, int* G, int* NT) {
    if (fit1 >= total || g_gru_tiles_per_wg < 2 || fit2 <= 0) { *G = fit1 >= total ? total : fit1; *NT = 1; return; }
    const int slots = (total + 1) / 2;
    *G = slots < fit2 ? slots : fit2;
    *NT = 2;
}

// dirbit: 0 forward, 1 backward; NT: batch tiles per workgroup of the launch (the measured win is NT == 1: B <= 128 on MI355X)
static bool xcd_local_wanted(int dirbit, int NT) {
    if (g_gru_xcd_local & (4 << dirbit)) return true;
    return (g_gru_xcd_local & (1 << dirbit)) && NT == 1;
}
template <class K>
int persist_grid(K kernel, int G, int* pack, bool want_pack = false) {
    int dev = 0, cus = 0, occ = 0;
    *pack = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, kPersistThreads, 0) != hipSuccess) return 0;
    const long room = (long)cus * occ;
    if (32L * G > room) return 0;
    if (g_gru_xcd_pack == 2 || ((g_gru_xcd_pack == 1 || want_pack) && cus % 8 == 0 && 32L * cdiv(G, 8) <= (long)(cus / 8) * occ)) {
        *pack = 1;
        return 256 * cdiv(G, 8);
    }
    return 32 * G;
}
}  // namespace

namespace cpc {
// bit 0 of cpc_device_error_flags(): a persistent-recurrence wave ran out of its polling budget
int gru_error_flag_fetch(int clear, unsigned* out) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_gru_poll_timeout), sizeof(v)) != hipSuccess) return CPC_ERR_ARG;
    if (clear && v) {
        const unsigned zero = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_gru_poll_timeout), &zero, sizeof(zero)) != hipSuccess) return CPC_ERR_ARG;
    }
    *out = v;
    return 0;
}
}  // namespace cpc

// Polling budget of the persistent recurrence (re-reads per wave over the whole launch, ~1 us each) before it gives up,
// flags CPC_DEVERR_GRU_POLL_TIMEOUT and lets NaN through.  limit < 0 restores the default (2^20, about a second -- far
// beyond any co-scheduled side-stream kernel of the train step).  Tests use 0 to drive the error path.
extern "C" int cpc_set_gru_spin_limit(int limit) {
    g_gru_spin_limit = limit < 0 ? kSpinLimit : limit;
    return 0;
}

// Wait before a step's first look at the hand-over buffers in the persistent recurrence (forward / backward kernel), in units
// of 64 clocks; < 0 (default): every wave steers its own (PollPace).
extern "C" int cpc_set_gru_poll_pacing(int first_fwd, int first_bwd) {
    if (first_fwd > 200 || first_bwd > 200 || first_fwd < -255 || first_bwd < -255) return CPC_ERR_ARG;
    g_gru_first_sleep[0] = first_fwd;          // (< -1: steering constants, PollPace)
    g_gru_first_sleep[1] = first_bwd;
    return 0;
}

// Cap on the batch tiles (16 sequences each) of one persistent launch; 0 = whatever fits the device.  A larger batch runs as
// several launches one after the other.
extern "C" int cpc_set_gru_chunk_tiles(int tiles) {
    if (tiles < 0) return CPC_ERR_ARG;
    g_gru_chunk_tiles = tiles;
    return 0;
}

// Batch tiles (16 sequences) a workgroup of the persistent recurrence may own: 2 (default) lets a batch that does not fit the
// device in one launch -- B = 256 on 256 CUs -- run as ONE launch with the two tiles of a workgroup interleaved step by step;
// 1: serial launches over chunks of tiles.  Bit-identical results either way.
extern "C" int cpc_set_gru_tiles_per_wg(int n) {
    if (n != 1 && n != 2) return CPC_ERR_ARG;
    g_gru_tiles_per_wg = n;
    return 0;
}

// Which waves of the persistent recurrence take the FIRST look at a hand-over fragment with plain loads through their XCD's L2
// (poll_row) instead of device-scope ones: bit 0 forward layer 1's look at layer 0's output of the same step, bit 1 backward layer
// 0's look at layer 1's gradient of the same step (both produced steps earlier as a rule), bit 2 / bit 3 the recurrent looks of
// forward / backward (the backward's bits only count in launches with one batch tile per workgroup unless bit 4 is set as well).
// Results do not change (a stale line only costs a repeated look).  Default 15.
extern "C" int cpc_set_gru_poll_plain(int mask) {
    if (mask < 0 || mask > 31) return CPC_ERR_ARG;
    g_gru_poll_plain = mask;
    return 0;
}

// XCD-local hand-over of the persistent recurrence (store_handover / tile_on_one_xcd): bit 0 the forward launch, bit 1 the backward
// launch take the packed numbering (one batch tile per XCD, where the device has 8 with room for it) and, tile by tile, plain
// stores when the placement check finds the tile's 32 workgroups on one XCD -- in launches with one batch tile per workgroup
// (the measured win); bits 2 / 3: in any launch.  Same bits as the device-scope hand-over.  Default 1.
extern "C" int cpc_set_gru_xcd_local(int mask) {
    if (mask < 0 || mask > 15) return CPC_ERR_ARG;
    g_gru_xcd_local = mask;
    return 0;
}

extern "C" int cpc_set_gru_xcd_pack(int on) {
    if (on < 0 || on > 2) return CPC_ERR_ARG;
    g_gru_xcd_pack = on;
    return 0;
}

extern "C" int cpc_set_gru_mode(int mode) {
    if (mode != 0 && mode != 1 && mode != 2) return CPC_ERR_ARG;
    g_gru_mode = mode;
    return 0;
}

// sizes[0] = saved floats, [1] = forward scratch floats, [2] = backward scratch floats
extern "C" int cpc_gru_layout(int B, int S, int nl, long* sizes) {
    GruLayout g;
    CPC_RETURN_IF(!gru_layout(B, S, nl, g), CPC_ERR_SHAPE);
    sizes[0] = g.saved_total; sizes[1] = g.fwd_total; sizes[2] = g.bwd_total;
    return 0;
}

// x (B,S,256); h0 NULL or (nl,B,256); params: weight_ih, weight_hh, bias_ih, bias_hh per layer
// (torch.nn.GRU state-dict order); y (B,S,256) = last layer's output; hN (nl,B,256) final states.
static void launch_gru_coef(const GruLayout& g, const float* h0, const float* saved, const float* y, float* coef, int B, int S,
                            hipStream_t st);

extern "C" int cpc_gru_forward(const float* x, const float* h0, const float* const* params, float* saved,
                               float* scratch, float* y, float* hN, int B, int S, int nl, void* stream) {
    return cpc_gru_forward_coef(x, h0, params, saved, scratch, y, hN, nullptr, B, S, nl, stream);
}

// coef != NULL (nl == 2; cpc_gru_coef_floats floats): the eight coefficient arrays of the two-layer backward are filled on the way
// -- by the persistent forward's gate threads where that path runs, by gru_bwd_coef_kernel behind the forward otherwise --
// and cpc_gru_backward_coef(..., coef_done = 1) only has the hand-over buffers and the weight transposes left to prepare.
static int gru_forward_impl(const float* x, const float* h0, const float* const* params, float* saved, float* scratch, float* y,
                            float* hN, float* coef, int B, int S, int nl, void* stream, bool xh_ready);
extern "C" int cpc_gru_forward_coef(const float* x, const float* h0, const float* const* params, float* saved,
                                    float* scratch, float* y, float* hN, float* coef, int B, int S, int nl, void* stream) {
    return gru_forward_impl(x, h0, params, saved, scratch, y, hN, coef, B, S, nl, stream, false);
}
// The forward's only activation-independent launch -- the "not written yet" fill of the persistent recurrence's hand-over
// buffers in `scratch` -- ahead of time on any stream; cpc_gru_forward_coef_prepared (on a stream that has waited for it) then
// goes from the input projection straight into the recurrence.  nl == 2.
extern "C" int cpc_gru_forward_prepare(float* scratch, int B, int S, int nl, void* stream) {
    GruLayout g;
    CPC_RETURN_IF(nl != 2 || !gru_layout(B, S, nl, g), CPC_ERR_SHAPE);
    CPC_RETURN_IF(!scratch, CPC_ERR_ARG);
    if (hipMemsetAsync(scratch + g.xh, 0xFF, (2 * g.xh_floats + g.sync_floats) * sizeof(float), (hipStream_t)stream) != hipSuccess) return CPC_ERR_ARG;
    return 0;
}
extern "C" int cpc_gru_forward_coef_prepared(const float* x, const float* h0, const float* const* params, float* saved,
                                             float* scratch, float* y, float* hN, float* coef, int B, int S, int nl, void* stream) {
    return gru_forward_impl(x, h0, params, saved, scratch, y, hN, coef, B, S, nl, stream, true);
}
static int gru_forward_impl(const float* x, const float* h0, const float* const* params, float* saved, float* scratch, float* y,
                            float* hN, float* coef, int B, int S, int nl, void* stream, bool xh_ready) {
    GruLayout g;
    CPC_RETURN_IF(!gru_layout(B, S, nl, g), CPC_ERR_SHAPE);
    CPC_RETURN_IF(!x || !params || !saved || !scratch || !y || !hN, CPC_ERR_ARG);
    hipStream_t st = (hipStream_t)stream;
    const float* in = x;
    float* gi = scratch + g.gi;
    if (nl == 2) {                                   // two-layer wavefront (see gru2_fwd_kernel)
        int rc = nt_gemm(plain_rows(x, B * S, kH), params[0], kH, params[2], gi, kG, kG, kH, st);
        if (rc) return rc;
        Gru2Fwd p;
        p.x_gi0 = gi;
        for (int l = 0; l < 2; ++l) {
            p.h0[l] = h0 ? h0 + (long)l * B * kH : nullptr;
            p.whh[l] = params[4 * l + 1];
            p.bhh[l] = params[4 * l + 3];
            p.R[l] = saved + g.R[l]; p.Z[l] = saved + g.Z[l]; p.N[l] = saved + g.N[l]; p.GHN[l] = saved + g.GHN[l];
        }
        p.wih1 = params[4]; p.bih1 = params[6];
        p.y[0] = saved + g.Y[0]; p.y[1] = y;
        p.hN = hN; p.B = B; p.S = S; p.spin_limit = g_gru_spin_limit;
        p.first_sleep = g_gru_first_sleep[0];
        p.poll_plain = g_gru_poll_plain;
        p.xh[0] = p.xh[1] = nullptr;
        p.xsync = nullptr;
        p.coef[0] = p.coef[1] = nullptr;
        p.frag_stride = g.frag_floats;
        p.total_tiles = cdiv(B, 16);
        p.tile0 = 0;
        const bool h2 = g_gru_mode == 2 && !h0;
        // a batch whose 32 workgroups per tile cannot all be resident at once: two tiles per workgroup (B = 256 on 256 CUs: ONE
        // launch of 8 tile slots), and beyond that -- or with cpc_set_gru_tiles_per_wg(1) -- chunks of tiles, one launch after the other
        int NT = 1;
        {
            const int fit1 = persist_chunk(h2 ? (const void*)gru2_persist_fwd_h2_kernel<1> : (const void*)gru2_persist_fwd_kernel<1>, p.total_tiles);
            const int fit2 = persist_chunk(h2 ? (const void*)gru2_persist_fwd_h2_kernel<2> : (const void*)gru2_persist_fwd_kernel<2>, p.total_tiles);
            persist_plan(p.total_tiles, fit1, fit2, &p.ntiles, &NT);
        }
        const int nblocks = g_gru_mode < 1 || p.ntiles <= 0 ? 0
                            : h2 ? (NT == 2 ? persist_grid(gru2_persist_fwd_h2_kernel<2>, p.ntiles, &p.xcd_pack, xcd_local_wanted(0, NT))
                                            : persist_grid(gru2_persist_fwd_h2_kernel<1>, p.ntiles, &p.xcd_pack, xcd_local_wanted(0, NT)))
                                 : (NT == 2 ? persist_grid(gru2_persist_fwd_kernel<2>, p.ntiles, &p.xcd_pack, xcd_local_wanted(0, NT))
                                            : persist_grid(gru2_persist_fwd_kernel<1>, p.ntiles, &p.xcd_pack, xcd_local_wanted(0, NT)));
        if (nblocks > 0) {
            p.xh[0] = scratch + g.xh; p.xh[1] = scratch + g.xh + g.xh_floats;
            if (coef) { p.coef[0] = coef; p.coef[1] = coef + 4 * g.frag_floats; }
            if (!xh_ready && hipMemsetAsync(p.xh[0], 0xFF, (2 * g.xh_floats + g.sync_floats) * sizeof(float), st) != hipSuccess) return CPC_ERR_ARG;
            p.xsync = xcd_local_wanted(0, NT) && p.xcd_pack ? reinterpret_cast<unsigned*>(scratch + g.xh + 2 * g.xh_floats) : nullptr;
            step_timer_mark(3, st);
            for (p.tile0 = 0; p.tile0 < p.total_tiles; p.tile0 += p.ntiles * NT) {
                if (h2 && NT == 2) hipLaunchKernelGGL(gru2_persist_fwd_h2_kernel<2>, dim3(nblocks), dim3(kPersistThreads), 0, st, p);
                else if (h2) hipLaunchKernelGGL(gru2_persist_fwd_h2_kernel<1>, dim3(nblocks), dim3(kPersistThreads), 0, st, p);
                else if (NT == 2) hipLaunchKernelGGL(gru2_persist_fwd_kernel<2>, dim3(nblocks), dim3(kPersistThreads), 0, st, p);
                else hipLaunchKernelGGL(gru2_persist_fwd_kernel<1>, dim3(nblocks), dim3(kPersistThreads), 0, st, p);
            }
            step_timer_mark(4, st);
            CPC_LAUNCH_CHECK();
            return 0;
        }
        const dim3 grid(kH / 16, cdiv(B, 16), 2);
        for (int s = 0; s <= S; ++s) hipLaunchKernelGGL(gru2_fwd_kernel, grid, dim3(512), 0, st, p, s);
        if (coef) launch_gru_coef(g, h0, saved, y, coef, B, S, st);
        CPC_LAUNCH_CHECK();
        return 0;
    }
    for (int l = 0; l < nl; ++l) {
        const float* wih = params[4 * l], *whh = params[4 * l + 1], *bih = params[4 * l + 2], *bhh = params[4 * l + 3];
        float* out = (l == nl - 1) ? y : saved + g.Y[l];
        int rc = nt_gemm(plain_rows(in, B * S, kH), wih, kH, bih, gi, kG, kG, kH, st);
        if (rc) return rc;
        const float* h0l = h0 ? h0 + (long)l * B * kH : nullptr;
        const dim3 grid(kH / 16, cdiv(B, 16));
        for (int t = 0; t < S; ++t) {
            const float* hprev = t == 0 ? h0l : out + (long)(t - 1) * kH;
            const long hstride = t == 0 ? kH : (long)S * kH;
            hipLaunchKernelGGL(gru_step_fwd_kernel, grid, dim3(256), 0, st, hprev, hstride, whh, bhh, gi, out,
                               saved + g.R[l], saved + g.Z[l], saved + g.N[l], saved + g.GHN[l],
                               t == S - 1 ? hN + (long)l * B * kH : nullptr, B, S, t);
        }
        CPC_LAUNCH_CHECK();
        in = out;
    }
    return 0;
}

// dy (B,S,256) -> dx (B,S,256) and grads[4*nl] (same order as params; overwritten).
// h0 receives no gradient (the reference detaches the carried state, cpc/model.py:194-198).
// Floats of the buffer cpc_gru_backward_coef fills (0 when nl != 2): the coefficient arrays of the two-layer backward
// and its hand-over buffers.  A buffer serves ONE backward call (the hand-over buffers are consumed by it).
extern "C" long cpc_gru_coef_floats(int B, int S, int nl) {
    GruLayout g;
    if (nl != 2 || !gru_layout(B, S, nl, g)) return 0;
    // 8 coefficient arrays + the two hand-over buffers of the persistent backward + the four transposed weight matrices
    return 10 * g.frag_floats + g.sync_floats + 4L * kG * kH;
}

static void launch_gru_coef(const GruLayout& g, const float* h0, const float* saved, const float* y, float* coef,
                            int B, int S, hipStream_t st) {
    const float* yl[2] = {saved + g.Y[0], y};
    const float* h0l[2] = {h0, h0 ? h0 + (long)B * kH : nullptr};
    for (int l = 0; l < 2; ++l) {
        float* c = coef + 4 * l * g.frag_floats;
        hipLaunchKernelGGL(gru_bwd_coef_kernel, dim3(cdiv((long)cdiv(B, 16) * 16 * S * 64, 256)), dim3(256), 0, st,
                           saved + g.R[l], saved + g.Z[l], saved + g.N[l], saved + g.GHN[l], yl[l], h0l[l], c,
                           c + g.frag_floats, c + 2 * g.frag_floats, c + 3 * g.frag_floats, B, S);
    }
}

// Everything in the two-layer backward that depends on the forward pass only (gru_bwd_coef_kernel, the weight transposes), into `coef`
// (cpc_gru_coef_floats floats): a caller may run this any time after the forward, on any stream, and hand the result
// to cpc_gru_backward_with_coef -- it takes 47 us of HBM streaming off the path between the criterion and the
// recurrence.
extern "C" int cpc_gru_backward_coef(const float* h0, const float* const* params, const float* saved, const float* y,
                                     float* coef, int coef_done, int B, int S, int nl, void* stream) {
    GruLayout g;
    CPC_RETURN_IF(nl != 2 || !gru_layout(B, S, nl, g), CPC_ERR_SHAPE);
    CPC_RETURN_IF(!params || !saved || !y || !coef, CPC_ERR_ARG);
    if (!coef_done) launch_gru_coef(g, h0, saved, y, coef, B, S, (hipStream_t)stream);     // (else: cpc_gru_forward_coef wrote them)
    {   // (3H,H) -> (H,3H), the four weight matrices in one launch: W_hh0, W_ih0, W_hh1, W_ih1 behind the hand-over buffers
        float* wT = coef + 10 * g.frag_floats + g.sync_floats;
        const float* tin[4] = {params[1], params[0], params[5], params[4]};
        float* tout[4] = {wT, wT + (long)kG * kH, wT + 2L * kG * kH, wT + 3L * kG * kH};
        int rc = transpose_batch(tin, tout, 4, kG, kH, (hipStream_t)stream);
        if (rc) return rc;
    }
    // ... and the hand-over buffers of the persistent backward, pre-filled with the "not written yet" pattern
    if (hipMemsetAsync(coef + 8 * g.frag_floats, 0xFF, (2 * g.frag_floats + g.sync_floats) * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return CPC_ERR_ARG;
    CPC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cpc_gru_backward(const float* x, const float* h0, const float* const* params,
                                const float* saved, const float* y, const float* dy, float* scratch,
                                float* dx, float* const* grads, int B, int S, int nl, void* stream) {
    return cpc_gru_backward_with_coef(x, h0, params, saved, y, dy, nullptr, scratch, dx, grads, B, S, nl, stream);
}

// coef: NULL (computed here) or the output of cpc_gru_backward_coef for the same forward pass (nl == 2 only).
extern "C" int cpc_gru_backward_with_coef(const float* x, const float* h0, const float* const* params,
                                          const float* saved, const float* y, const float* dy, const float* coef,
                                          float* scratch, float* dx, float* const* grads, int B, int S, int nl,
                                          void* stream) {
    return cpc_gru_backward_streams(x, h0, params, saved, y, dy, coef, scratch, dx, grads, B, S, nl, stream, stream);
}

// As cpc_gru_backward_with_coef, with everything that only the optimiser reads -- the four weight gradients (one
// batched TN GEMM) and the four bias gradients -- on `wgrad_stream`, released by an event behind the recurrence; dx
// (what the encoder's backward waits for) stays on `stream`.  NO join: the caller orders every consumer of `grads`
// after `wgrad_stream`, and keeps `scratch` alive until then.  nl == 2 only; other depths run on `stream` alone.
extern "C" int cpc_gru_backward_streams(const float* x, const float* h0, const float* const* params,
                                        const float* saved, const float* y, const float* dy, const float* coef,
                                        float* scratch, float* dx, float* const* grads, int B, int S, int nl,
                                        void* stream, void* wgrad_stream) {
    GruLayout g;
    CPC_RETURN_IF(!gru_layout(B, S, nl, g), CPC_ERR_SHAPE);
    CPC_RETURN_IF(!x || !params || !saved || !scratch || !y || !dy || !dx || !grads, CPC_ERR_ARG);
    hipStream_t st = (hipStream_t)stream;
    hipStream_t wst = nl == 2 ? (hipStream_t)wgrad_stream : st;
    float* whhT = scratch + g.whhT, *wihT = scratch + g.wihT;
    float* dGi = scratch + g.dGi, *dGh = scratch + g.dGh, *DH = scratch + g.DH;
    const int M = B * S;
    if (nl == 2) {                                   // two-layer wavefront (see gru2_bwd_kernel)
        // (the transposed weights come with `coef` when the caller prepared it, cpc_gru_backward_coef)
        float* cwT = coef ? const_cast<float*>(coef) + 10 * g.frag_floats + g.sync_floats : nullptr;
        float* whhT_[2] = {coef ? cwT : scratch + g.whhT, coef ? cwT + 2L * kG * kH : scratch + g.whhT2};
        float* wihT_[2] = {coef ? cwT + (long)kG * kH : scratch + g.wihT, coef ? cwT + 3L * kG * kH : scratch + g.wihT2};
        float* dGi_[2] = {scratch + g.dGi, scratch + g.dGi2};
        float* dGh_[2] = {scratch + g.dGh, scratch + g.dGh2};
        float* DH_[2] = {scratch + g.DH, scratch + g.DH2};
        Gru2Bwd p;
        p.dy = dy; p.B = B; p.S = S; p.spin_limit = g_gru_spin_limit;
        p.first_sleep = g_gru_first_sleep[1];
        p.poll_plain = g_gru_poll_plain;                  // (narrowed below once the launch plan is known)
        const float* yl[2] = {saved + g.Y[0], y};
        const float* h0l[2] = {h0, h0 ? h0 + (long)B * kH : nullptr};
        int rc = 0;
        if (!coef) {   // (3H,H) -> (H,3H), the four weight matrices in one launch
            const float* tin[4] = {params[1], params[0], params[5], params[4]};
            float* tout[4] = {whhT_[0], wihT_[0], whhT_[1], wihT_[1]};
            rc = transpose_batch(tin, tout, 4, kG, kH, st);
            if (rc) return rc;
        }
        for (int l = 0; l < 2; ++l) {
            p.whhT[l] = whhT_[l];
            p.Z[l] = saved + g.Z[l];
            p.dGi[l] = dGi_[l]; p.dGh[l] = dGh_[l]; p.DH[l] = DH_[l];
            const float* c = (coef ? coef : scratch + g.coef) + 4 * l * g.frag_floats;
            p.cr[l] = c; p.cz[l] = c + g.frag_floats; p.cnh[l] = c + 2 * g.frag_floats; p.cni[l] = c + 3 * g.frag_floats;
            p.xdh[l] = (coef ? const_cast<float*>(coef) + 8 * g.frag_floats : scratch + g.xdh) + l * g.frag_floats;
        }
        if (!coef) launch_gru_coef(g, h0, saved, y, scratch + g.coef, B, S, st);
        p.wih1T = wihT_[1];
        p.xsync = nullptr;
        p.total_tiles = cdiv(B, 16);
        p.tile0 = 0;
        int NT = 1;
        persist_plan(p.total_tiles, persist_chunk((const void*)gru2_persist_bwd_kernel<1>, p.total_tiles),
                     persist_chunk((const void*)gru2_persist_bwd_kernel<2>, p.total_tiles), &p.ntiles, &NT);
        if (NT == 2 && !(g_gru_poll_plain & 16)) p.poll_plain = 0;     // two tiles per workgroup: the backward's plain looks lose
        const int nblocks = g_gru_mode < 1 || p.ntiles <= 0 ? 0
                            : NT == 2 ? persist_grid(gru2_persist_bwd_kernel<2>, p.ntiles, &p.xcd_pack, xcd_local_wanted(1, NT))
                                      : persist_grid(gru2_persist_bwd_kernel<1>, p.ntiles, &p.xcd_pack, xcd_local_wanted(1, NT));
        if (nblocks > 0) {
            if (!coef && hipMemsetAsync(p.xdh[0], 0xFF, (2 * g.frag_floats + g.sync_floats) * sizeof(float), st) != hipSuccess) return CPC_ERR_ARG;
            p.xsync = xcd_local_wanted(1, NT) && p.xcd_pack ? reinterpret_cast<unsigned*>(p.xdh[0] + 2 * g.frag_floats) : nullptr;
            // (in-step timing: the marker in FRONT of this launch is recorded by cpc_train_step before it releases the side
            // stream's gather kernels -- a marker packet between that release and this launch lets their workgroups take the
            // CUs first, and the persistent launch then waits 250 us for residency: measured)
            for (p.tile0 = 0; p.tile0 < p.total_tiles; p.tile0 += p.ntiles * NT) {
                if (NT == 2) hipLaunchKernelGGL(gru2_persist_bwd_kernel<2>, dim3(nblocks), dim3(kPersistThreads), 0, st, p);
                else hipLaunchKernelGGL(gru2_persist_bwd_kernel<1>, dim3(nblocks), dim3(kPersistThreads), 0, st, p);
            }
            step_timer_mark(6, st);
        } else {
            const dim3 grid(kH / 16, cdiv(B, 16), 2);
            for (int s = 0; s <= S; ++s) hipLaunchKernelGGL(gru2_bwd_kernel, grid, dim3(512), 0, st, p, s);
        }
        CPC_LAUNCH_CHECK();
        // dx = dGi0 . W_ih0 first: it is what the rest of the backward pass waits for
        SplitK sk;                       // N = 256: too few tiles for the chip; the partial buffer of the weight gradients is free
        sk.part = scratch + g.part;      // until they start (behind this GEMM, on either stream)
        sk.floats = std::max(tn_gemm_part_floats(B * S, kG, kH), tn_gemm_batch_part_floats(4, B * S, kG, kH));
        rc = nt_gemm(plain_rows(dGi_[0], M, kG), wihT_[0], kG, nullptr, dx, kH, kH, kG, st, 0, 0, GemmBounds(), GemmGroup(), sk);
        if (rc) return rc;
        if (wst != st) {
            hipEvent_t* ev = stream_events(st);
            CPC_RETURN_IF(!ev, CPC_ERR_ARG);
            if (hipEventRecord(ev[8], st) != hipSuccess || hipStreamWaitEvent(wst, ev[8], 0) != hipSuccess) return CPC_ERR_ARG;
        }
        const long tmp1 = align64l((long)kRowsSumGroups * kG);
        const RowsSumJob jobs[4] = {{dGi_[0], M, kG, scratch + g.tmp, grads[2]},
                                    {dGh_[0], M, kG, scratch + g.tmp + tmp1, grads[3]},
                                    {dGi_[1], M, kG, scratch + g.tmp + 2 * tmp1, grads[6]},
                                    {dGh_[1], M, kG, scratch + g.tmp + 3 * tmp1, grads[7]}};
        {   // the four weight gradients dW_ih^l = dGi_l^T . in_l, dW_hh^l = dGh_l^T . h^l_{t-1} as one batched GEMM
            RowMap am[4], bm[4];
            float* Cq[4];
            for (int l = 0; l < 2; ++l) {
                const float* in = l == 0 ? x : saved + g.Y[0];
                am[2 * l] = plain_rows(dGi_[l], M, kG);
                bm[2 * l] = plain_rows(in, M, kH);
                Cq[2 * l] = grads[4 * l];
                RowMap hm;                             // h_{t-1} rows: y[b, t-1] (zero row at t = 0; h0 term below)
                hm.base = yl[l]; hm.R = S; hm.bstride = (long)S * kH; hm.rstride = kH; hm.off = -kH;
                hm.tmul = 1; hm.tadd = -1; hm.Lin = S; hm.M = M;
                am[2 * l + 1] = plain_rows(dGh_[l], M, kG);
                bm[2 * l + 1] = hm;
                Cq[2 * l + 1] = grads[4 * l + 1];
            }
            rc = tn_gemm_batch(4, am, kG, bm, kH, scratch + g.part, Cq, 0, wst);
            if (rc) return rc;
        }
        for (int l = 0; l < 2; ++l) {
            if (h0l[l]) {                              // + dGh[:,0,:]^T . h0
                RowMap g0;
                g0.base = dGh_[l]; g0.R = 1; g0.bstride = (long)S * kG; g0.rstride = 0; g0.off = 0;
                g0.tmul = 0; g0.tadd = 0; g0.Lin = 0x7fffffff; g0.M = B;
                rc = tn_gemm(g0, kG, plain_rows(h0l[l], B, kH), kH, scratch + g.part, grads[4 * l + 1], 1, wst);
                if (rc) return rc;
            }
        }
        rc = rows_sum_multi(jobs, 4, wst);               // the four bias gradients in two launches
        return rc;
    }
    const float* dYl = dy;
    for (int l = nl - 1; l >= 0; --l) {
        const float* wih = params[4 * l], *whh = params[4 * l + 1];
        const float* in = l == 0 ? x : saved + g.Y[l - 1];
        const float* out = (l == nl - 1) ? y : saved + g.Y[l];
        const float* h0l = h0 ? h0 + (long)l * B * kH : nullptr;
        float* dXl = l == 0 ? dx : scratch + g.mid[l & 1];
        int rc = transpose(whh, whhT, kG, kH, st);       // (3H,H) -> (H,3H)
        if (rc) return rc;
        rc = transpose(wih, wihT, kG, kH, st);
        if (rc) return rc;
        const dim3 grid(kH / 16, cdiv(B, 16));
        for (int t = S - 1; t >= 0; --t)
            hipLaunchKernelGGL(gru_step_bwd_kernel, grid, dim3(256), 0, st, whhT, dYl, out, h0l,
                               saved + g.R[l], saved + g.Z[l], saved + g.N[l], saved + g.GHN[l], dGi, dGh, DH,
                               B, S, t);
        CPC_LAUNCH_CHECK();
        // weight / bias gradients over all B*S rows
        const RowMap gim = plain_rows(dGi, M, kG), ghm = plain_rows(dGh, M, kG);
        rc = tn_gemm(gim, kG, plain_rows(in, M, kH), kH, scratch + g.part, grads[4 * l], 0, st);
        if (rc) return rc;
        // h_{t-1} rows: y[b, t-1] (zero row at t = 0; the h0 term is added below)
        RowMap hm;
        hm.base = out; hm.R = S; hm.bstride = (long)S * kH; hm.rstride = kH; hm.off = -kH;
        hm.tmul = 1; hm.tadd = -1; hm.Lin = S; hm.M = M;
        rc = tn_gemm(ghm, kG, hm, kH, scratch + g.part, grads[4 * l + 1], 0, st);
        if (rc) return rc;
        if (h0l) {   // + dGh[:,0,:]^T . h0
            RowMap g0;
            g0.base = dGh; g0.R = 1; g0.bstride = (long)S * kG; g0.rstride = 0; g0.off = 0;
            g0.tmul = 0; g0.tadd = 0; g0.Lin = 0x7fffffff; g0.M = B;
            rc = tn_gemm(g0, kG, plain_rows(h0l, B, kH), kH, scratch + g.part, grads[4 * l + 1], 1, st);
            if (rc) return rc;
        }
        rc = rows_sum(dGi, M, kG, scratch + g.tmp, grads[4 * l + 2], st);
        if (rc) return rc;
        rc = rows_sum(dGh, M, kG, scratch + g.tmp, grads[4 * l + 3], st);
        if (rc) return rc;
        // dX = dGi . W_ih   (as NT against W_ih^T)
        rc = nt_gemm(gim, wihT, kG, nullptr, dXl, kH, kH, kG, st);
        if (rc) return rc;
        dYl = dXl;
    }
    return 0;
}

#ifdef CPC_GRU_TIMING
// host[2][512][12][8]: the phase sums of the last persistent forward (0) / backward (1) launch (PhaseClock)
extern "C" int cpc_debug_gru_phases(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(cpc::g_gru_phase), sizeof(unsigned long long) * 2 * 512 * 12 * 8) == hipSuccess ? 0 : CPC_ERR_ARG;
}
#endif
