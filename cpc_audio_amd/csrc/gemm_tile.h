// f32-MFMA tile main loops shared by the conv (implicit GEMM), GRU-projection and
// prediction-head kernels.
//
// gfx950 has exact-f32 MFMA (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD = the f32
// vector peak, bit-identical to an fmaf chain), so the contraction-heavy layers
// run on the matrix pipe without giving up the 1e-4 fp32 parity bar
// (/opt/skills/guides/cdna_hip_programming.md section 3).
//
// Operand conventions (lane l of a wave64, 32x32x2 form):
//   A fragment: A[i = l&31][k = l>>5]      B fragment: B[k = l>>5][j = l&31]
//   C/D:        col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5), reg in [0,16)
// Inside one BK=16 chunk the contraction index is visited as
//   kq in {0,1}, j in 0..3:   half-wave h = l>>5 supplies k = 8*kq + 4*h + j
// so a K-contiguous ("K-major") LDS row is read with ONE ds_read_b128 per 4 MFMA
// steps, and both operands use the same pairing (the sum over k is order-free).
#pragma once
#include <type_traits>

#include "cpc_common.h"

namespace cpc {

// Maps GEMM row m of an operand to memory.  Row m belongs to batch item b = m / R
// at position t = m % R and starts at  base + b*bstride + t*rstride + off.
// Element k of that row is a real element iff 0 <= t*tmul + tadd + (k >> 8) < Lin,
// otherwise it reads as zero: that is how conv zero-padding and the ragged ends of
// the transposed-conv windows are expressed without padded buffers.
struct RowMap {
    const float* base;
    int R;
    long bstride;
    int rstride;
    int off;
    int tmul;
    int tadd;
    int Lin;
    int M;
};

static inline RowMap plain_rows(const float* base, int M, int ld) {
    RowMap r;
    r.base = base; r.R = M > 0 ? M : 1; r.bstride = 0; r.rstride = ld; r.off = 0;
    r.tmul = 0; r.tadd = 0; r.Lin = 0x7fffffff; r.M = M;
    return r;
}

// Rows of a data-gradient phase GEMM (k = 2s): the 2-row windows [q-1, q] over dx (B, Lout, C).  Phase r of input step
// tau = q*s + r - p takes q over [0, Lout]; when Lin == s * Lout (every layer of the encoder at window sizes that are multiples
// of 160) exactly Lout of those Lout + 1 rows have tau inside [0, Lin): q >= 1 for the phases r < p, q <= Lout - 1 for the
// others.  Then R = Lout (the "exact" rows: the kernels see R == Lin of the map and start phase r at q0 = r < p), B * Lout rows
// per phase and no ragged last tile -- B * (Lout + 1) rows cost a whole extra round of 256-row tiles for 4 workgroups' worth
// of live rows at B = 64.
static inline RowMap dgrad_rows(const float* dx, int B, int Lin, int Lout, int s, int p) {
    const bool exact = Lin == s * Lout && p > 0 && p < s;
    RowMap r;
    r.base = dx; r.R = exact ? Lout : Lout + 1; r.bstride = (long)Lout * kC; r.rstride = kC; r.off = -kC;
    r.tmul = 1; r.tadd = -1; r.Lin = Lout; r.M = B * r.R;
    return r;
}

// im2col rows of an (B, Lin, C) channels-last activation for a conv (k, s, p):
// row (b,t) is the contiguous window x[b, t*s-p : t*s-p+k, :]  (k*C floats).
static inline RowMap conv_rows(const float* x, int B, int Lin, int Lout, int s, int p) {
    RowMap r;
    r.base = x; r.R = Lout; r.bstride = (long)Lin * kC; r.rstride = s * kC; r.off = -p * kC;
    r.tmul = s; r.tadd = -p; r.Lin = Lin; r.M = B * Lout;
    return r;
}

struct RowRef {
    const float* ptr;
    int tau0;
};

__device__ __forceinline__ RowRef resolve_row(const RowMap& rm, int m, int mlimit) {
    RowRef r;
    if (m < mlimit) {
        int b = m / rm.R;
        int t = m - b * rm.R;
        r.ptr = rm.base + (long)b * rm.bstride + (long)t * rm.rstride + rm.off;
        r.tau0 = t * rm.tmul + rm.tadd;
    } else {
        r.ptr = rm.base;
        r.tau0 = -(1 << 30);
    }
    return r;
}

// (batch, position) of a row, advanced without divisions (rows of a tile chunk are consecutive).
struct RowCursor {
    int b, t;
};
__device__ __forceinline__ RowCursor cursor_at(const RowMap& rm, int m) {
    RowCursor c;
    c.b = m / rm.R;
    c.t = m - c.b * rm.R;
    return c;
}
__device__ __forceinline__ RowCursor cursor_plus(const RowMap& rm, RowCursor c, int n) {
    c.t += n;
    while (c.t >= rm.R) { c.t -= rm.R; c.b += 1; }
    return c;
}
__device__ __forceinline__ RowRef cursor_ref(const RowMap& rm, const RowCursor& c, bool valid) {
    RowRef r;
    if (valid) {
        r.ptr = rm.base + (long)c.b * rm.bstride + (long)c.t * rm.rstride + rm.off;
        r.tau0 = c.t * rm.tmul + rm.tadd;
    } else {
        r.ptr = rm.base;
        r.tau0 = -(1 << 30);
    }
    return r;
}

// Branch-free: an out-of-range element reads a safe address and is zeroed afterwards, so the loads of a
// tile are straight-line code and the compiler can keep several chunks in flight behind counted
// s_waitcnt vmcnt(N) (a branch around a load makes it drain with vmcnt(0)).
__device__ __forceinline__ float4 load_row4(const RowRef& r, int k, int Lin, const float* safe) {
    const int tau = r.tau0 + (k >> kCLog2);
    const bool ok = (unsigned)tau < (unsigned)Lin;
    const float* p = ok ? r.ptr + k : safe;
    float4 v = *reinterpret_cast<const float4*>(p);
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}

// Same, but the zeroing is left to the consumer (`ok` travels with the data): a select right behind the load
// would make the compiler wait for the load one pipeline stage after issuing it instead of four.
__device__ __forceinline__ float4 load_row4_raw(const RowRef& r, int k, int Lin, const float* safe, bool& ok) {
    const int tau = r.tau0 + (k >> kCLog2);
    ok = (unsigned)tau < (unsigned)Lin;
    const float* p = ok ? r.ptr + k : safe;
    return *reinterpret_cast<const float4*>(p);
}

__device__ __forceinline__ float f4c(const float4& v, int j) {
    return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
}

// ---------------------------------------------------------------------------
// NT tile:  acc[BM x BN] += A[m0.., 0:K] * B[n0.., 0:K]^T, both operands K-major.
// ---------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N>
struct NtTile {
    static constexpr int BK = 16;
    static constexpr int LDK = BK + 4;   // 80-byte rows: 16B aligned, conflict-free b128 reads
    static constexpr int NTHREADS = 64 * WAVES_M * WAVES_N;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int A_SLOTS = BM * (BK / 4), B_SLOTS = BN * (BK / 4);
    static constexpr int A_PER = (A_SLOTS + NTHREADS - 1) / NTHREADS;
    static constexpr int B_PER = (B_SLOTS + NTHREADS - 1) / NTHREADS;
    static constexpr int STAGE = (BM + BN) * LDK;
    static constexpr int SMEM_FLOATS = 2 * STAGE;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");

    // row of the C tile (relative to m0) held in accumulator register `reg` of tile tm
    __device__ static __forceinline__ int c_row(int tm, int reg) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave / WAVES_N) * WM + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    }
    // column of the C tile (relative to n0) held by this lane for tile tn
    __device__ static __forceinline__ int c_col(int tn) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave % WAVES_N) * WN + tn * 32 + (lane & 31);
    }

    // bblk: floats between consecutive 16-k groups of a B row: 16 for a plain row-major B (ldb = K), N*16 for
    // the k-blocked layout [k/16][n][16] (ldb = 16), where one chunk of all rows is one contiguous block
    __device__ static void run(f32x16 (&acc)[TM][TN], const RowMap& am, int m0,
                               const float* __restrict__ Bmat, int ldb, int n0, int K,
                               float* smem, int bblk = 16) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;

        RowRef ar[A_PER];
        int a_k[A_PER], a_lds[A_PER];
        bool a_on[A_PER];
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int slot = tid + i * NTHREADS;
            a_on[i] = slot < A_SLOTS;
            int r = slot >> 2, kv = slot & 3;
            ar[i] = resolve_row(am, m0 + r, a_on[i] ? am.M : 0);
            a_k[i] = kv * 4;
            a_lds[i] = r * LDK + kv * 4;
        }
        const float* bp[B_PER];
        int b_lds[B_PER];
        bool b_on[B_PER];
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            int slot = tid + i * NTHREADS;
            b_on[i] = slot < B_SLOTS;
            int r = b_on[i] ? (slot >> 2) : 0, kv = slot & 3;
            bp[i] = Bmat + (long)(n0 + r) * ldb + kv * 4;
            b_lds[i] = BM * LDK + r * LDK + kv * 4;
        }

        float4 ra[A_PER], rb[B_PER];
        const int nk = K / BK;

auto gload = [&](int kc_) __attribute__((always_inline)) {
            const int k0 = kc_ * BK;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) ra[i] = load_row4(ar[i], k0 + a_k[i], am.Lin, am.base);
#pragma unroll
            for (int i = 0; i < B_PER; ++i) rb[i] = *reinterpret_cast<const float4*>(bp[i] + (long)kc_ * bblk);
        };
        auto sstore = [&](int st_) __attribute__((always_inline)) {
            float* s0 = smem + st_ * STAGE;
#pragma unroll
            for (int i = 0; i < A_PER; ++i)
                if (a_on[i]) *reinterpret_cast<float4*>(s0 + a_lds[i]) = ra[i];
#pragma unroll
            for (int i = 0; i < B_PER; ++i)
                if (b_on[i]) *reinterpret_cast<float4*>(s0 + b_lds[i]) = rb[i];
        };

        gload(0);
        sstore(0);
        __syncthreads();

        const int arow = wm * WM + (lane & 31);
        const int brow = wn * WN + (lane & 31);
        const int kofs = 4 * (lane >> 5);

        auto compute = [&](int cur) __attribute__((always_inline)) {
            const float* As = smem + cur * STAGE;
            const float* Bs = As + BM * LDK;
#pragma unroll
            for (int kq = 0; kq < BK / 8; ++kq) {
                float4 af[TM], bf[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    af[tm] = *reinterpret_cast<const float4*>(As + (arow + tm * 32) * LDK + kq * 8 + kofs);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    bf[tn] = *reinterpret_cast<const float4*>(Bs + (brow + tn * 32) * LDK + kq * 8 + kofs);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                f4c(af[tm], j), f4c(bf[tn], j), acc[tm][tn], 0, 0, 0);
            }
        };
        for (int kc = 0; kc + 1 < nk; ++kc) {
            gload(kc + 1);
            compute(kc & 1);
            sstore((kc & 1) ^ 1);
            __syncthreads();
        }
        compute((nk - 1) & 1);
        __syncthreads();
    }
};

// ---------------------------------------------------------------------------
// NtTileX3: the same NT product on the bf16 matrix pipe with fp32-level accuracy.
//
// Every fp32 operand x is split BY TRUNCATION into three bf16 pieces when it is staged into
// LDS:  h = top 16 bits of x,  m = top 16 bits of (x - h),  l = top 16 bits of (x - h - m)
// (both subtractions are exact in fp32), so x = h + m + l up to 2^-24 |x|.  A product
// a*b is then accumulated in fp32 as  ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh  (the three
// dropped terms are <= 2^-24 |ab|; bf16 x bf16 products are exact in the fp32 accumulator).
// Six v_mfma_f32_32x32x16_bf16 (32 cycles, 16 k) replace eight v_mfma_f32_32x32x2_f32
// (64 cycles, 2 k): 2.67x the f32-MFMA rate at the same parity bar.
//
// LDS: three bf16 planes per operand, K-major rows of BK halves padded by 8 (16-byte aligned,
// conflict-free ds_read_b128 over 16-lane groups).  Default: BK = 32, ONE LDS stage (92 KB at
// 128 x 256) with register prefetch: load(kc+1) | compute(kc) | barrier | split+store | barrier.
// The double-buffered BK = 16 form (STAGES = 2, 110 KB) measured slower on MI355X (conv1 152 vs 162
// TFLOP/s-equivalent, conv3 65 vs 107: more LDS per block, twice the barriers per k).
// ---------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(m);
    l = __float_as_uint(r2) & 0xFFFF0000u;
}

// four floats -> three 8-byte groups of four bf16 (element 0 in the low half of .x)
__device__ __forceinline__ void split3_pack4(const float4& v, uint2& ph, uint2& pm, uint2& pl) {
    unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
    split3(v.x, h0, m0, l0);
    split3(v.y, h1, m1, l1);
    split3(v.z, h2, m2, l2);
    split3(v.w, h3, m3, l3);
    ph = make_uint2((h0 >> 16) | h1, (h2 >> 16) | h3);
    pm = make_uint2((m0 >> 16) | m1, (m2 >> 16) | m3);
    pl = make_uint2((l0 >> 16) | l1, (l2 >> 16) | l3);
}

// ---- two fp16 pieces ("H2"): x*s = h + l with h = fp16(x*s) (round to nearest), l = fp16(x*s - h); the
// subtraction is exact and |x*s - h - l| <= 2^-22 |x*s|.  A product needs only hh + hl + lh (the dropped l*l
// term is <= 2^-22 |ab|): fp32-level accuracy at HALF the MFMAs of the three-piece bf16 split, provided the
// power-of-two scale s keeps |x*s| inside fp16's range -- callers derive it from a bound on max|x| (scale_for_amax).
__device__ __forceinline__ void split2h_pack4(const float4& v, float s, uint2& ph, uint2& pl) {
    const float x0 = v.x * s, x1 = v.y * s, x2 = v.z * s, x3 = v.w * s;
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1, h2 = (_Float16)x2, h3 = (_Float16)x3;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    const _Float16 l2 = (_Float16)(x2 - (float)h2), l3 = (_Float16)(x3 - (float)h3);
    ph = make_uint2(__builtin_bit_cast(unsigned, f16x2{h0, h1}), __builtin_bit_cast(unsigned, f16x2{h2, h3}));
    pl = make_uint2(__builtin_bit_cast(unsigned, f16x2{l0, l1}), __builtin_bit_cast(unsigned, f16x2{l2, l3}));
}
// power-of-two scale that maps a tensor with max|x| <= amax into [2^13, 2^14) (fp16 max is 65504)
__device__ __forceinline__ float scale_for_amax(float amax) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(amax, &e);                            // amax = m * 2^e, m in [0.5, 1)
    e = 14 - e;
    e = e > 60 ? 60 : (e < -60 ? -60 : e);             // |log2 s| <= 60: 1/(sa*sb) stays far inside fp32's range; an
    return ldexpf(1.0f, e);                            // operand with amax < 2^-46 is scaled as far as that allows
}

// A bound kept as `slots` partial maxima (<= 64; written by different workgroups so that no single address takes
// thousands of atomics, or by a reduction kernel with one workgroup per slot): every wave folds them itself.
constexpr int kAmaxSlots = 64;
__device__ __forceinline__ float fold_amax(const float* __restrict__ p, int slots) {
    if (slots <= 1) return *p;
    const int lane = threadIdx.x & 63;
    return wave_max(lane < slots ? p[lane] : 0.f);
}

// (O,I,W) conv weight -> K-tile-major H2 rows for the DMA kernels (conv_dma.hip): element (co, kg = kk*C + ci) of
// w * scale_for_amax(amax) goes to row (kg / 32) * 256 + co, a 128-byte row of four 8-element groups [h x 8 | l x 8].
__device__ __forceinline__ void permute_w_h2_elem(const float* __restrict__ w, unsigned char* __restrict__ wq, int k,
                                                  float amax, long idx) {
    const int co = (int)(idx / (k * kC));
    const int rem = (int)(idx - (long)co * k * kC);
    const int kk = rem >> kCLog2, ci = rem & (kC - 1);
    const float v = w[((long)co * kC + ci) * k + kk];
    _Float16 h, l;
    h2_split(v, scale_for_amax(amax), h, l);
    unsigned char* row = wq + ((long)(rem >> 5) * kC + co) * 128;
    _Float16* p = reinterpret_cast<_Float16*>(row + h2_byte_of(rem & 31));
    p[0] = h;
    p[8] = l;
}

// ... data gradient (enc_conv.hip, conv_dgrad_kernel): phase r < s, rows ci, contraction kg = j*C + co (j in {0,1}):
// element W[co][ci][r + (1-j)*s] * scale goes to phase block r (256 * 512 * 4 bytes), row (kg / 32) * 256 + ci
__device__ __forceinline__ void permute_w_dgrad_h2_elem(const float* __restrict__ w, unsigned char* __restrict__ wd, int s,
                                                        float amax, long idx) {
    const int k = 2 * s;
    const int r = (int)(idx / (kC * 2 * kC));
    const int rem = (int)(idx - (long)r * kC * 2 * kC);
    const int ci = rem / (2 * kC);
    const int jc = rem - ci * 2 * kC;
    const int j = jc >> kCLog2, co = jc & (kC - 1);
    _Float16 h, l;
    h2_split(w[((long)co * kC + ci) * k + r + (1 - j) * s], scale_for_amax(amax), h, l);
    unsigned char* row = wd + (long)r * (kC * 2 * kC * 4) + ((long)(jc >> 5) * kC + ci) * 128;
    _Float16* o = reinterpret_cast<_Float16*>(row + h2_byte_of(jc & 31));
    o[0] = h;
    o[8] = l;
}

// bf16 storage (mode 4): round to nearest even
__device__ __forceinline__ unsigned short bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_val(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
// (O,I,W) conv weight -> bf16 K-tile-major rows for the DMA kernels: forward [kg / 64][co][64], kg = kk*C + ci
__device__ __forceinline__ void permute_w_fwd_bf16_elem(const float* __restrict__ w, unsigned short* __restrict__ wq, int k,
                                                        long idx) {
    const int co = (int)(idx / (k * kC));
    const int rem = (int)(idx - (long)co * k * kC);
    const int kk = rem >> kCLog2, ci = rem & (kC - 1);
    wq[((long)(rem >> 6) * kC + co) * 64 + (rem & 63)] = bf16_rne(w[((long)co * kC + ci) * k + kk]);
}
// ... data gradient: phase r < s, rows ci, contraction kg = j*C + co (j in {0,1}): Wd[r][kg / 64][ci][64] = W[co][ci][r + (1-j)*s]
__device__ __forceinline__ void permute_w_dgrad_bf16_elem(const float* __restrict__ w, unsigned short* __restrict__ wd, int s,
                                                          long idx) {
    const int k = 2 * s;
    const int r = (int)(idx / (kC * 2 * kC));
    const int rem = (int)(idx - (long)r * kC * 2 * kC);
    const int ci = rem / (2 * kC);
    const int jc = rem - ci * 2 * kC;
    const int j = jc >> kCLog2, co = jc & (kC - 1);
    wd[(long)r * kC * 2 * kC + ((long)(jc >> 6) * kC + ci) * 64 + (jc & 63)] = bf16_rne(w[((long)co * kC + ci) * k + r + (1 - j) * s]);
}

template <int NP> struct SplitPlanes;
template <> struct SplitPlanes<3> {                  // three bf16 pieces, six products, no scaling
    static constexpr int NPROD = 6;
    __device__ static __forceinline__ void split(const float4& v, float, uint2 (&p)[3]) { split3_pack4(v, p[0], p[1], p[2]); }
    __device__ static __forceinline__ int pa(int q) { constexpr int t[6] = {2, 0, 1, 1, 0, 0}; return t[q]; }   // l*h, h*l, m*m,
    __device__ static __forceinline__ int pb(int q) { constexpr int t[6] = {0, 2, 1, 0, 1, 0}; return t[q]; }   // m*h, h*m, h*h
    __device__ static __forceinline__ f32x16 mfma(const s16x8& a, const s16x8& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct SplitPlanes<1> {                  // one bf16 piece, one product: the bf16-storage variant (operands arrive as bf16)
    static constexpr int NPROD = 1;
    __device__ static __forceinline__ int pa(int) { return 0; }
    __device__ static __forceinline__ int pb(int) { return 0; }
    __device__ static __forceinline__ f32x16 mfma(const s16x8& a, const s16x8& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct SplitPlanes<2> {                  // two fp16 pieces, three products, power-of-two operand scales
    static constexpr int NPROD = 3;
    __device__ static __forceinline__ void split(const float4& v, float s, uint2 (&p)[2]) { split2h_pack4(v, s, p[0], p[1]); }
    __device__ static __forceinline__ int pa(int q) { constexpr int t[3] = {1, 0, 0}; return t[q]; }            // l*h, h*l, h*h
    __device__ static __forceinline__ int pb(int q) { constexpr int t[3] = {0, 1, 0}; return t[q]; }
    __device__ static __forceinline__ f32x16 mfma(const s16x8& a, const s16x8& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// One BK-deep chunk of the split-bf16 product from K-major bf16 planes in LDS.
// planes: [h | m | l], each `plane` halves; rows of LDH halves.  The six partial products of a
// k-step are issued product-major so that consecutive MFMAs hit different accumulators.
// SWZ: rows were written with their 16-byte k-pairs XOR-swizzled by ((row >> 3) & (BK/8 - 1)) (the
// transposing TN loader does that to keep its 8-byte column stores bank-conflict free).
struct BSplitReg { uint2 h, m, l; };      // a 4-k group of a pre-split operand: three planes x four bf16

__device__ __forceinline__ void load_b(float4& dst, const float* p32, const unsigned short*, long k0, long) {
    dst = *reinterpret_cast<const float4*>(p32 + k0);
}
__device__ __forceinline__ void load_b(BSplitReg& dst, const float*, const unsigned short* p16, long k0, long plane) {
    dst.h = *reinterpret_cast<const uint2*>(p16 + k0);
    dst.m = *reinterpret_cast<const uint2*>(p16 + plane + k0);
    dst.l = *reinterpret_cast<const uint2*>(p16 + 2 * plane + k0);
}
// pre-split fp16 operand: the 16 bytes of a 4-k slot hold [h0 h1 h2 h3 | l0 l1 l2 l3] (same bytes and the same
// addresses as the fp32 slot, so it is fetched by the same single 16-byte load)
struct BSplitH2 { uint4 v; };
__device__ __forceinline__ void load_b(BSplitH2& dst, const float* p32, const unsigned short*, long k0, long) {
    dst.v = *reinterpret_cast<const uint4*>(p32 + k0);
}
__device__ __forceinline__ void planes_of(const float4& v, uint2& ph, uint2& pm, uint2& pl) { split3_pack4(v, ph, pm, pl); }
__device__ __forceinline__ void planes_of(const BSplitReg& v, uint2& ph, uint2& pm, uint2& pl) { ph = v.h; pm = v.m; pl = v.l; }

template <int TM, int TN, int BK, int LDH, bool SWZ = false, int NP = 3>
__device__ __forceinline__ void x3_compute(f32x16 (&acc)[TM][TN], const unsigned short* As, int planeA,
                                           const unsigned short* Bs, int planeB, int arow, int brow, int kofs) {
    constexpr int SWM = BK / 8 - 1;
    using SP = SplitPlanes<NP>;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
        s16x8 af[TM][NP], bf[TN][NP];
        const int pair = (ks * 16 + kofs) >> 3;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int row = arow + tm * 32;
            const int ko = SWZ ? ((pair ^ ((row >> 3) & SWM)) << 3) : (ks * 16 + kofs);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                af[tm][pl] = *reinterpret_cast<const s16x8*>(As + pl * planeA + row * LDH + ko);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int row = brow + tn * 32;
            const int ko = SWZ ? ((pair ^ ((row >> 3) & SWM)) << 3) : (ks * 16 + kofs);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                bf[tn][pl] = *reinterpret_cast<const s16x8*>(Bs + pl * planeB + row * LDH + ko);
        }
        // small terms first; consecutive MFMAs hit different accumulators
#pragma unroll
        for (int q = 0; q < SP::NPROD; ++q)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = SP::mfma(af[tm][SP::pa(q)], bf[tn][SP::pb(q)], acc[tm][tn]);
    }
}

// BSPLIT: the B operand (weights) is already stored as three bf16 planes [3][N][ldb] (written once per
// step by the weight re-layout kernels), so only the A operand is split while staging.
// NP: 3 = three bf16 pieces / six products (no operand scaling), 2 = two fp16 pieces / three products (operands
// pre-multiplied by the power-of-two scales sa, sb given to run(); the caller multiplies the result by 1/(sa*sb)).
// AH2 (NP == 2 only): the A operand lies in H2 storage (cpc_common.h: per 8 channels [h x 8 | l x 8], the 4 bytes per element
// and the 1 KB rows of fp32) scaled by sa -- the pieces are what the split would produce, so a 16-byte slot is loaded as it
// lies (float offset 4 kv of the chunk: k-group kv >> 1, piece kv & 1) and stored into its plane with one 16-byte LDS write:
// no conversion VALU for A at all.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK_ = 32, int STAGES = 1, bool SKEW = false, bool BSPLIT = false,
          int NP = 3, bool AH2 = false>
struct NtTileX3 {
    static_assert(!AH2 || NP == 2, "H2 operands are two fp16 pieces");
    using SP = SplitPlanes<NP>;
    // BSPLIT: the B operand is stored pre-split (NP == 3: three bf16 planes; NP == 2: interleaved fp16 slots, BSplitH2)
    static constexpr int BK = BK_;       // 32 with one LDS stage (default), or 16 double-buffered
    static constexpr int LDH = BK + 8;   // halves per LDS row (80 B / 48 B: 16-byte aligned, conflict-free b128)
    static constexpr int SPR = BK / 4;   // float4 slots per row
    static constexpr int NTHREADS = 64 * WAVES_M * WAVES_N;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int A_SLOTS = BM * (BK / 4), B_SLOTS = BN * (BK / 4);
    static constexpr int A_PER = (A_SLOTS + NTHREADS - 1) / NTHREADS;
    static constexpr int B_PER = (B_SLOTS + NTHREADS - 1) / NTHREADS;
    static constexpr bool A_EXACT = A_SLOTS % NTHREADS == 0, B_EXACT = B_SLOTS % NTHREADS == 0;   // no partial slot
    static constexpr int PLANE_A = BM * LDH, PLANE_B = BN * LDH;          // halves
    static constexpr int STAGE_H = NP * (PLANE_A + PLANE_B);              // halves per stage
    static constexpr int SMEM_FLOATS = STAGES * STAGE_H / 2;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
    static_assert(STAGES == 1 || STAGES == 2, "one or two LDS stages");

    __device__ static __forceinline__ int c_row(int tm, int reg) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave / WAVES_N) * WM + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int c_col(int tn) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave % WAVES_N) * WN + tn * 32 + (lane & 31);
    }

    __device__ static void run(f32x16 (&acc)[TM][TN], const RowMap& am, int m0,
                               const float* __restrict__ Bmat, int ldb, int n0, int K,
                               float* smem_f, long bplane = 0, int bblk = 16, int rot = 0,      // bblk: see NtTile::run
                               float sa = 1.0f, float sb = 1.0f, int tshift = 0) {
        // tshift: K consists of 2^tshift equal segments (conv taps) whose rows overlap in memory: tap j of output row t
        // and tap j - s of row t + 1 are the same input row.  Walking the chunks tap-fastest (all taps of one channel
        // block, then the next block) brings the two reads of every input segment a few chunks apart instead of half
        // the K walk, i.e. inside L2's reach: the plain order fetched layer 1's activation 2.4 times from HBM (PMC).
        // rot (pipelined schedule only): this workgroup walks the K chunks starting at chunk `rot` (mod K/BK).
        // All workgroups of a conv layer otherwise read the same 64-byte channel slice of rows that are a
        // multiple of 1 KB apart at the same moment, i.e. one L2 channel out of 16.
        unsigned short* smem0 = reinterpret_cast<unsigned short*>(smem_f);
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;

        RowRef ar[A_PER];
        int a_k[A_PER], a_lds[A_PER];
        bool a_on[A_PER];
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int slot = tid + i * NTHREADS;
            a_on[i] = slot < A_SLOTS;
            const int r = slot / SPR, kv = slot % SPR;
            ar[i] = resolve_row(am, m0 + r, a_on[i] ? am.M : 0);
            a_k[i] = kv * 4;
            a_lds[i] = AH2 ? (kv & 1) * PLANE_A + r * LDH + (kv >> 1) * 8 : r * LDH + kv * 4;
        }
        const float* bp[B_PER];
        const unsigned short* bp16[B_PER];
        int b_lds[B_PER];
        bool b_on[B_PER];
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int slot = tid + i * NTHREADS;
            b_on[i] = slot < B_SLOTS;
            const int r = b_on[i] ? (slot / SPR) : 0, kv = slot % SPR;
            const long bo = (long)(n0 + r) * ldb + (long)(kv >> 2) * bblk + (kv & 3) * 4;
            bp[i] = Bmat + bo;
            bp16[i] = reinterpret_cast<const unsigned short*>(Bmat) + bo;
            b_lds[i] = NP * PLANE_A + r * LDH + kv * 4;
        }
        using BReg = typename std::conditional<BSPLIT, typename std::conditional<NP == 3, BSplitReg, BSplitH2>::type, float4>::type;
        float4 ra[A_PER], ra1[A_PER];
        BReg rb[B_PER], rb1[B_PER];                // second register set for the deep-prefetch schedule
        const int nk = K / BK;

        auto gload_to = [&](int kc0_, float4 (&ra_)[A_PER], BReg (&rb_)[B_PER]) __attribute__((always_inline)) {
            int kc_ = kc0_ + rot;
            kc_ = kc_ >= nk ? kc_ - nk : kc_;
            kc_ = (kc_ & ((1 << tshift) - 1)) * (nk >> tshift) + (kc_ >> tshift);
            const int k0 = kc_ * BK;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) ra_[i] = load_row4(ar[i], k0 + a_k[i], am.Lin, am.base);
#pragma unroll
            for (int i = 0; i < B_PER; ++i) load_b(rb_[i], bp[i], bp16[i], (long)kc_ * (BK / 16) * bblk, bplane);
        };
        auto sstore_from = [&](int st_, const float4 (&ra)[A_PER], const BReg (&rb)[B_PER]) __attribute__((always_inline)) {
            unsigned short* smem = smem0 + st_ * STAGE_H;
#pragma unroll
            for (int i = 0; i < A_PER; ++i)
                if (A_EXACT || a_on[i]) {            // exact tilings stay branch-free: one basic block, so the
                    if constexpr (AH2) {             //  split can be scheduled into the MFMA shadow
                        uint4 raw;
                        raw.x = __float_as_uint(ra[i].x); raw.y = __float_as_uint(ra[i].y);
                        raw.z = __float_as_uint(ra[i].z); raw.w = __float_as_uint(ra[i].w);
                        *reinterpret_cast<uint4*>(smem + a_lds[i]) = raw;
                    } else {
                        uint2 pp[NP];
                        SP::split(ra[i], sa, pp);
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(smem + pl * PLANE_A + a_lds[i]) = pp[pl];
                    }
                }
#pragma unroll
            for (int i = 0; i < B_PER; ++i)
                if (B_EXACT || b_on[i]) {
                    uint2 pp[NP];
                    if constexpr (BSPLIT && NP == 3) planes_of(rb[i], pp[0], pp[1], pp[2]);
                    else if constexpr (BSPLIT) { pp[0] = make_uint2(rb[i].v.x, rb[i].v.y); pp[1] = make_uint2(rb[i].v.z, rb[i].v.w); }
                    else SP::split(rb[i], sb, pp);
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(smem + pl * PLANE_B + b_lds[i]) = pp[pl];
                }
        };
        const int arow = wm * WM + (lane & 31);
        const int brow = wn * WN + (lane & 31);
        const int kofs = 8 * (lane >> 5);
        auto compute = [&](int st_) __attribute__((always_inline)) {
            const unsigned short* smem = smem0 + st_ * STAGE_H;
            x3_compute<TM, TN, BK, LDH, false, NP>(acc, smem, PLANE_A, smem + NP * PLANE_A, PLANE_B, arow, brow, kofs);
        };

        auto gload = [&](int kc_) __attribute__((always_inline)) { gload_to(kc_, ra, rb); };
        auto sstore = [&](int st_) __attribute__((always_inline)) { sstore_from(st_, ra, rb); };
        if constexpr (STAGES == 2 && SKEW) {
            // Software pipeline over 16-k chunks, one barrier per chunk.  In iteration kc a wave
            //   * issues the LDS reads of chunk kc+1's fragments (other stage, second fragment register set),
            //   * runs the 6 x TM x TN MFMAs of chunk kc from fragments fetched one iteration earlier,
            //   * splits chunk kc+2 (VALU) and stores it over chunk kc's stage (free: everybody holds kc in regs),
            //   * issues the global loads of chunk kc+6;
            // so LDS reads, VALU and stores all sit in the MFMA shadow (sched_group_barrier below) and neither
            // LDS latency nor its bandwidth (98 KB of fragment reads per chunk per CU) is exposed after the
            // barrier.  Global prefetch depth: four register sets = chunks kc+2 .. kc+5 in flight (the per-CU
            // load rate is bytes-in-flight / latency, ~2 us under load).
            static_assert(BK == 16, "pipelined schedule is written for one k-step per chunk");
            float4 ra2[A_PER], ra3[A_PER];
            BReg rb2[B_PER], rb3[B_PER];
            s16x8 fa[2][TM][NP], fb[2][TN][NP];
            auto lfrag = [&](int st_, s16x8 (&af)[TM][NP], s16x8 (&bf)[TN][NP]) __attribute__((always_inline)) {
                const unsigned short* As = smem0 + st_ * STAGE_H;
                const unsigned short* Bs = As + NP * PLANE_A;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        af[tm][pl] = *reinterpret_cast<const s16x8*>(As + pl * PLANE_A + (arow + tm * 32) * LDH + kofs);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        bf[tn][pl] = *reinterpret_cast<const s16x8*>(Bs + pl * PLANE_B + (brow + tn * 32) * LDH + kofs);
            };
            auto mfma6 = [&](const s16x8 (&af)[TM][NP], const s16x8 (&bf)[TN][NP]) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < SP::NPROD; ++q)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = SP::mfma(af[tm][SP::pa(q)], bf[tn][SP::pb(q)], acc[tm][tn]);
            };
            auto interleave = [&]() __attribute__((always_inline)) {
                constexpr int NMFMA = TM * TN * SP::NPROD;
                constexpr int NRD = NP * (TM + TN);                        // ds_read_b128 per chunk
                constexpr int NWR = NP * (A_PER + B_PER);                  // 8-byte LDS stores per chunk
                constexpr int VPM = NP == 3 ? 3 : 4;                       // VALU per MFMA slot (fewer MFMAs to hide behind)
                constexpr int WEVERY = NMFMA / (A_PER + B_PER);            // MFMAs between the stores of two slots
#pragma unroll
                for (int q = 0; q < NMFMA; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
                    if (q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);   // VALU
                    if (q % WEVERY == WEVERY - 1) __builtin_amdgcn_sched_group_barrier(0x200, NP, 0);        // DS write
                    if (q % WEVERY == WEVERY / 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // VMEM read
                }
            };
            bool va[A_PER], va1[A_PER], va2[A_PER], va3[A_PER];
            auto gload_raw = [&](int kc0_, float4 (&ra_)[A_PER], BReg (&rb_)[B_PER], bool (&ok_)[A_PER]) __attribute__((always_inline)) {
                int kc_ = kc0_ + rot;
                kc_ = kc_ >= nk ? kc_ - nk : kc_;
                kc_ = (kc_ & ((1 << tshift) - 1)) * (nk >> tshift) + (kc_ >> tshift);
                const int k0 = kc_ * BK;
#pragma unroll
                for (int i = 0; i < A_PER; ++i) ra_[i] = load_row4_raw(ar[i], k0 + a_k[i], am.Lin, am.base, ok_[i]);
#pragma unroll
                for (int i = 0; i < B_PER; ++i) load_b(rb_[i], bp[i], bp16[i], (long)kc_ * (BK / 16) * bblk, bplane);
            };
            auto sstore_m = [&](int st_, const float4 (&ra_)[A_PER], const BReg (&rb_)[B_PER], const bool (&ok_)[A_PER]) __attribute__((always_inline)) {
                float4 rz[A_PER];
#pragma unroll
                for (int i = 0; i < A_PER; ++i) {
                    rz[i].x = ok_[i] ? ra_[i].x : 0.f; rz[i].y = ok_[i] ? ra_[i].y : 0.f;
                    rz[i].z = ok_[i] ? ra_[i].z : 0.f; rz[i].w = ok_[i] ? ra_[i].w : 0.f;
                }
                sstore_from(st_, rz, rb_);
            };
            gload_raw(0, ra, rb, va);
            gload_raw(min(1, nk - 1), ra1, rb1, va1);
            sstore_m(0, ra, rb, va);
            sstore_m(1, ra1, rb1, va1);
            gload_raw(min(2, nk - 1), ra, rb, va);
            gload_raw(min(3, nk - 1), ra1, rb1, va1);
            gload_raw(min(4, nk - 1), ra2, rb2, va2);
            gload_raw(min(5, nk - 1), ra3, rb3, va3);
            __syncthreads();
            lfrag(0, fa[0], fb[0]);
            __syncthreads();        // phase 0 overwrites stage 0: every wave must hold chunk 0's fragments first
            // nk is a multiple of 4 for every caller (K = 512, 1024, 2048): the four phases form ONE basic block,
            // so the prefetch loads cannot be sunk into a later phase (which is what the compiler does with
            // early exits between the phases: one vmcnt(0) drain and a burst of 9 loads every 4 chunks).
            for (int kc = 0; kc < nk; kc += 4) {
                lfrag(1, fa[1], fb[1]);
                mfma6(fa[0], fb[0]);
                sstore_m(0, ra, rb, va);
                gload_raw(min(kc + 6, nk - 1), ra, rb, va);
                interleave();
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);          // nothing moves across a phase boundary
                lfrag(0, fa[0], fb[0]);
                mfma6(fa[1], fb[1]);
                sstore_m(1, ra1, rb1, va1);
                gload_raw(min(kc + 7, nk - 1), ra1, rb1, va1);
                interleave();
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                lfrag(1, fa[1], fb[1]);
                mfma6(fa[0], fb[0]);
                sstore_m(0, ra2, rb2, va2);
                gload_raw(min(kc + 8, nk - 1), ra2, rb2, va2);
                interleave();
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                lfrag(0, fa[0], fb[0]);
                mfma6(fa[1], fb[1]);
                sstore_m(1, ra3, rb3, va3);
                gload_raw(min(kc + 9, nk - 1), ra3, rb3, va3);
                interleave();
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        gload(0);
        sstore(0);
        __syncthreads();
        if (STAGES == 2) {
            for (int kc = 0; kc + 1 < nk; ++kc) {
                gload(kc + 1);
                compute(kc & 1);
                sstore((kc & 1) ^ 1);
                __syncthreads();
            }
            compute((nk - 1) & 1);
        } else {
            for (int kc = 0; kc + 1 < nk; ++kc) {
                gload(kc + 1);
                compute(0);
                __syncthreads();
                sstore(0);
                __syncthreads();
            }
            compute(0);
        }
        __syncthreads();
    }
};

// ---------------------------------------------------------------------------
// TN tile:  acc[BM x BN] += sum_{m in [mbeg,mend)} A[m, c0..c0+BM)^T (x) B[m, n0..n0+BN)
// The contraction index is the ROW index of both operands ("M-major" LDS tiles,
// conflict-free ds_read_b32 with consecutive lanes on consecutive columns).
// Used for every weight gradient: dW = dOut^T . In.
// ---------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N>
struct TnTile {
    static constexpr int BK = 16;
    static constexpr int NTHREADS = 64 * WAVES_M * WAVES_N;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int A_SLOTS = BK * (BM / 4), B_SLOTS = BK * (BN / 4);
    static constexpr int A_PER = (A_SLOTS + NTHREADS - 1) / NTHREADS;
    static constexpr int B_PER = (B_SLOTS + NTHREADS - 1) / NTHREADS;
    static constexpr int STAGE = BK * (BM + BN);
    static constexpr int SMEM_FLOATS = 2 * STAGE;

    __device__ static __forceinline__ int c_row(int tm, int reg) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave / WAVES_N) * WM + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int c_col(int tn) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave % WAVES_N) * WN + tn * 32 + (lane & 31);
    }

    __device__ static void run(f32x16 (&acc)[TM][TN], const RowMap& am, int c0,
                               const RowMap& bm, int n0, int mbeg, int mend, float* smem) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;

        int a_row[A_PER], a_col[A_PER];
        bool a_on[A_PER];
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int slot = tid + i * NTHREADS;
            a_on[i] = slot < A_SLOTS;
            a_row[i] = slot / (BM / 4);
            a_col[i] = (slot % (BM / 4)) * 4;
        }
        int b_row[B_PER], b_col[B_PER];
        bool b_on[B_PER];
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            int slot = tid + i * NTHREADS;
            b_on[i] = slot < B_SLOTS;
            b_row[i] = slot / (BN / 4);
            b_col[i] = (slot % (BN / 4)) * 4;
        }
        float4 ra[A_PER], rb[B_PER];
        const int nk = (mend - mbeg + BK - 1) / BK;

RowCursor ca[A_PER], cb[B_PER];
#pragma unroll
        for (int i = 0; i < A_PER; ++i) ca[i] = cursor_at(am, mbeg + a_row[i]);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) cb[i] = cursor_at(bm, mbeg + b_row[i]);
        auto gload = [&](int kc_) __attribute__((always_inline)) {
            const int mm = mbeg + kc_ * BK;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const RowRef r = cursor_ref(am, ca[i], a_on[i] && (mm + a_row[i]) < mend);
                ra[i] = load_row4(r, c0 + a_col[i], am.Lin, am.base);
                ca[i] = cursor_plus(am, ca[i], BK);
            }
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const RowRef r = cursor_ref(bm, cb[i], b_on[i] && (mm + b_row[i]) < mend);
                rb[i] = load_row4(r, n0 + b_col[i], bm.Lin, bm.base);
                cb[i] = cursor_plus(bm, cb[i], BK);
            }
        };
        auto sstore = [&](int st_) __attribute__((always_inline)) {
            float* s0 = smem + st_ * STAGE;
#pragma unroll
            for (int i = 0; i < A_PER; ++i)
                if (a_on[i]) *reinterpret_cast<float4*>(s0 + a_row[i] * BM + a_col[i]) = ra[i];
#pragma unroll
            for (int i = 0; i < B_PER; ++i)
                if (b_on[i]) *reinterpret_cast<float4*>(s0 + BK * BM + b_row[i] * BN + b_col[i]) = rb[i];
        };

        if (nk <= 0) return;
        gload(0);
        sstore(0);
        __syncthreads();

        const int acol = wm * WM + (lane & 31);
        const int bcol = wn * WN + (lane & 31);
        const int kh = 4 * (lane >> 5);

        auto compute = [&](int cur) __attribute__((always_inline)) {
            const float* As = smem + cur * STAGE;
            const float* Bs = As + BK * BM;
#pragma unroll
            for (int kq = 0; kq < BK / 8; ++kq) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = kq * 8 + kh + j;
                    float a[TM], b[TN];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) a[tm] = As[kk * BM + acol + tm * 32];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) b[tn] = Bs[kk * BN + bcol + tn * 32];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
                }
            }
        };
        for (int kc = 0; kc + 1 < nk; ++kc) {
            gload(kc + 1);
            compute(kc & 1);
            sstore((kc & 1) ^ 1);
            __syncthreads();
        }
        compute((nk - 1) & 1);
        __syncthreads();
    }
};

// ---------------------------------------------------------------------------
// TnTileX3: the TN product (contraction over the row index m) on the bf16 pipe with 3-piece split
// operands.  bf16 MFMA fragments want 8 consecutive contraction indices per lane, so the loader
// transposes while staging: each thread fetches a 4(m) x 4(column) block (four float4 loads from
// four consecutive rows), splits it, and writes, per column and plane, 4 consecutive-m halves with
// one 8-byte LDS store.  LDS then holds the same K-major planes as NtTileX3 ([column][m]) and the
// compute step is shared.
// ---------------------------------------------------------------------------
// BH2 (NP == 2 only): the B operand is stored in H2 form (cpc_common.h: two fp16 pieces per element, already scaled by sb):
// the loader fetches the pieces and transposes them, no split VALU.
// BF16IN (NP == 1 only): both operands are bf16 tensors (the bf16-storage variant): fetched as stored, transposed, one product.
// AH2: the same for the A operand.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK_ = 32, int STAGES = 1, int NP = 3, bool BH2 = false, bool BF16IN = false,
          bool AH2 = false>
struct TnTileX3 {   // NP: see NtTileX3
    static constexpr int BK = BK_;
    static constexpr int LDH = BK + 8;
    static constexpr int NTHREADS = 64 * WAVES_M * WAVES_N;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int A_BLK = (BK / 4) * (BM / 4), B_BLK = (BK / 4) * (BN / 4);    // 4x4 blocks
    static constexpr int A_PER = (A_BLK + NTHREADS - 1) / NTHREADS;
    static constexpr int B_PER = (B_BLK + NTHREADS - 1) / NTHREADS;
    static constexpr int PLANE_A = BM * LDH, PLANE_B = BN * LDH;
    static constexpr int STAGE_H = NP * (PLANE_A + PLANE_B);
    static constexpr int SMEM_FLOATS = STAGES * STAGE_H / 2;

    __device__ static __forceinline__ int c_row(int tm, int reg) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave / WAVES_N) * WM + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int c_col(int tn) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave % WAVES_N) * WN + tn * 32 + (lane & 31);
    }

    // 4 rows x 4 columns of fp32 -> for each column (x,y,z,w of the float4s) the NP planes of its
    // 4 consecutive-m halves, stored at column-major LDS rows.  `scale`: operand scale of the fp16 split.
    __device__ static __forceinline__ void store_block(unsigned short* base, int plane, int col0, int mofs,
                                                       const float4 (&v)[4], float scale) {
        // 4-row block `mofs/4` of column group col0/4 goes to k-slot (mofs/4) ^ 2*((col0/8) & SWM): with the
        // (8 column groups x 2 row blocks) lane order below, a 16-lane group then covers 16 distinct banks.
        constexpr int SWM = BK / 8 - 1;
        const int mphys = (((mofs >> 2) ^ (2 * ((col0 >> 3) & SWM))) << 2);
        if constexpr (NP == 3) {
            unsigned h[4][4], m[4][4], l[4][4];      // [row][col]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                split3(v[r].x, h[r][0], m[r][0], l[r][0]);
                split3(v[r].y, h[r][1], m[r][1], l[r][1]);
                split3(v[r].z, h[r][2], m[r][2], l[r][2]);
                split3(v[r].w, h[r][3], m[r][3], l[r][3]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned short* dst = base + (col0 + c) * LDH + mphys;
                *reinterpret_cast<uint2*>(dst) = make_uint2((h[0][c] >> 16) | h[1][c], (h[2][c] >> 16) | h[3][c]);
                *reinterpret_cast<uint2*>(dst + plane) = make_uint2((m[0][c] >> 16) | m[1][c], (m[2][c] >> 16) | m[3][c]);
                *reinterpret_cast<uint2*>(dst + 2 * plane) = make_uint2((l[0][c] >> 16) | l[1][c], (l[2][c] >> 16) | l[3][c]);
            }
        } else {
            _Float16 h[4][4], l[4][4];               // [row][col]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x[4] = {v[r].x * scale, v[r].y * scale, v[r].z * scale, v[r].w * scale};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    h[r][c] = (_Float16)x[c];
                    l[r][c] = (_Float16)(x[c] - (float)h[r][c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned short* dst = base + (col0 + c) * LDH + mphys;
                *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, f16x2{h[0][c], h[1][c]}),
                                                            __builtin_bit_cast(unsigned, f16x2{h[2][c], h[3][c]}));
                *reinterpret_cast<uint2*>(dst + plane) = make_uint2(__builtin_bit_cast(unsigned, f16x2{l[0][c], l[1][c]}),
                                                                    __builtin_bit_cast(unsigned, f16x2{l[2][c], l[3][c]}));
            }
        }
    }

    // the same for a block that arrives as H2 pieces: v[r] = {h0h1, h2h3, l0l1, l2l3} (bit patterns) of row r
    __device__ static __forceinline__ void store_block_h2(unsigned short* base, int plane, int col0, int mofs,
                                                          const float4 (&v)[4]) {
        constexpr int SWM = BK / 8 - 1;
        const int mphys = (((mofs >> 2) ^ (2 * ((col0 >> 3) & SWM))) << 2);
        unsigned h[4][4], l[4][4];               // [row][col], 16-bit values
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned hx = __float_as_uint(v[r].x), hy = __float_as_uint(v[r].y);
            const unsigned lx = __float_as_uint(v[r].z), ly = __float_as_uint(v[r].w);
            h[r][0] = hx & 0xFFFFu; h[r][1] = hx >> 16; h[r][2] = hy & 0xFFFFu; h[r][3] = hy >> 16;
            l[r][0] = lx & 0xFFFFu; l[r][1] = lx >> 16; l[r][2] = ly & 0xFFFFu; l[r][3] = ly >> 16;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned short* dst = base + (col0 + c) * LDH + mphys;
            *reinterpret_cast<uint2*>(dst) = make_uint2(h[0][c] | (h[1][c] << 16), h[2][c] | (h[3][c] << 16));
            *reinterpret_cast<uint2*>(dst + plane) = make_uint2(l[0][c] | (l[1][c] << 16), l[2][c] | (l[3][c] << 16));
        }
    }
    // a block of bf16 values: v[r].x, v[r].y = the four 16-bit elements of row r
    __device__ static __forceinline__ void store_block_bf16(unsigned short* base, int col0, int mofs, const float4 (&v)[4]) {
        constexpr int SWM = BK / 8 - 1;
        const int mphys = (((mofs >> 2) ^ (2 * ((col0 >> 3) & SWM))) << 2);
        unsigned e[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned x = __float_as_uint(v[r].x), y = __float_as_uint(v[r].y);
            e[r][0] = x & 0xFFFFu; e[r][1] = x >> 16; e[r][2] = y & 0xFFFFu; e[r][3] = y >> 16;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<uint2*>(base + (col0 + c) * LDH + mphys) = make_uint2(e[0][c] | (e[1][c] << 16), e[2][c] | (e[3][c] << 16));
    }
    // elements k .. k+3 of a row of a bf16 tensor (row strides of the RowMap are in elements)
    __device__ static __forceinline__ float4 load_row4_bf16(const RowMap& rm, const RowCursor& cur, bool valid, int k) {
        const int tau = cur.t * rm.tmul + rm.tadd + (k >> kCLog2);
        const bool ok = valid && (unsigned)tau < (unsigned)rm.Lin;
        const unsigned short* base = reinterpret_cast<const unsigned short*>(rm.base);
        const unsigned short* p = ok ? base + (long)cur.b * rm.bstride + (long)cur.t * rm.rstride + rm.off + k : base;
        const uint2 d = *reinterpret_cast<const uint2*>(p);
        float4 v;
        v.x = ok ? __uint_as_float(d.x) : 0.f; v.y = ok ? __uint_as_float(d.y) : 0.f; v.z = 0.f; v.w = 0.f;
        return v;
    }
    // H2 pieces of elements k .. k+3 (k % 4 == 0) of an im2col row: tap k >> 8, channel k & 255 of a 1 KB H2 row
    __device__ static __forceinline__ float4 load_row4_h2(const RowRef& r, int k, int Lin, const float* safe) {
        const int tau = r.tau0 + (k >> kCLog2);
        const bool ok = (unsigned)tau < (unsigned)Lin;
        const unsigned char* p = ok ? reinterpret_cast<const unsigned char*>(r.ptr + (k & ~(kC - 1))) + h2_byte_of(k & (kC - 1))
                                    : reinterpret_cast<const unsigned char*>(safe);
        const uint2 hp = *reinterpret_cast<const uint2*>(p), lp = *reinterpret_cast<const uint2*>(p + 16);
        float4 v;
        v.x = ok ? __uint_as_float(hp.x) : 0.f; v.y = ok ? __uint_as_float(hp.y) : 0.f;
        v.z = ok ? __uint_as_float(lp.x) : 0.f; v.w = ok ? __uint_as_float(lp.y) : 0.f;
        return v;
    }

    __device__ static void run(f32x16 (&acc)[TM][TN], const RowMap& am, int c0,
                               const RowMap& bm, int n0, int mbeg, int mend, float* smem_f,
                               float sa = 1.0f, float sb = 1.0f) {
        static_assert((!BH2 && !AH2) || NP == 2, "H2 operands are two fp16 pieces");
        static_assert(BF16IN == (NP == 1), "one piece <=> bf16 tensors");
        unsigned short* smem0 = reinterpret_cast<unsigned short*>(smem_f);
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;
        int a_m[A_PER], a_c[A_PER], b_m[B_PER], b_c[B_PER];
        bool a_on[A_PER], b_on[B_PER];
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int blk = tid + i * NTHREADS;
            a_on[i] = blk < A_BLK;
            // 16 consecutive lanes = 8 column groups x 2 row blocks (128-byte global segments, and the
            // swizzled LDS stores of a 16-lane group land on 16 distinct banks)
            const int g = (blk & 7) + 8 * ((blk >> 4) % (BM / 32));
            const int mb = ((blk >> 3) & 1) + 2 * ((blk >> 4) / (BM / 32));
            a_m[i] = mb * 4;
            a_c[i] = g * 4;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int blk = tid + i * NTHREADS;
            b_on[i] = blk < B_BLK;
            const int g = (blk & 7) + 8 * ((blk >> 4) % (BN / 32));
            const int mb = ((blk >> 3) & 1) + 2 * ((blk >> 4) / (BN / 32));
            b_m[i] = mb * 4;
            b_c[i] = g * 4;
        }
        float4 ra[A_PER][4], rb[B_PER][4];
        const int nk = (mend - mbeg + BK - 1) / BK;
        if (nk <= 0) return;

        RowCursor ca[A_PER], cb[B_PER];          // first row of this thread's 4x4 block in the current chunk
#pragma unroll
        for (int i = 0; i < A_PER; ++i) ca[i] = cursor_at(am, mbeg + a_m[i]);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) cb[i] = cursor_at(bm, mbeg + b_m[i]);
        auto gload = [&](int kc_) __attribute__((always_inline)) {
            const int mm = mbeg + kc_ * BK;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (BF16IN) {
                        ra[i][r] = load_row4_bf16(am, cursor_plus(am, ca[i], r), a_on[i] && (mm + a_m[i] + r) < mend, c0 + a_c[i]);
                        continue;
                    }
                    const RowRef rr = cursor_ref(am, cursor_plus(am, ca[i], r), a_on[i] && (mm + a_m[i] + r) < mend);
                    if constexpr (AH2) ra[i][r] = load_row4_h2(rr, c0 + a_c[i], am.Lin, am.base);
                    else ra[i][r] = load_row4(rr, c0 + a_c[i], am.Lin, am.base);
                }
                ca[i] = cursor_plus(am, ca[i], BK);
            }
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (BF16IN) {
                        rb[i][r] = load_row4_bf16(bm, cursor_plus(bm, cb[i], r), b_on[i] && (mm + b_m[i] + r) < mend, n0 + b_c[i]);
                        continue;
                    }
                    const RowRef rr = cursor_ref(bm, cursor_plus(bm, cb[i], r), b_on[i] && (mm + b_m[i] + r) < mend);
                    if constexpr (BH2) rb[i][r] = load_row4_h2(rr, n0 + b_c[i], bm.Lin, bm.base);
                    else rb[i][r] = load_row4(rr, n0 + b_c[i], bm.Lin, bm.base);
                }
                cb[i] = cursor_plus(bm, cb[i], BK);
            }
        };
        auto sstore = [&](int st_) __attribute__((always_inline)) {
            unsigned short* smem = smem0 + st_ * STAGE_H;
#pragma unroll
            for (int i = 0; i < A_PER; ++i)
                if (a_on[i]) {
                    if constexpr (BF16IN) store_block_bf16(smem, a_c[i], a_m[i], ra[i]);
                    else if constexpr (AH2) store_block_h2(smem, PLANE_A, a_c[i], a_m[i], ra[i]);
                    else store_block(smem, PLANE_A, a_c[i], a_m[i], ra[i], sa);
                }
#pragma unroll
            for (int i = 0; i < B_PER; ++i)
                if (b_on[i]) {
                    if constexpr (BF16IN) store_block_bf16(smem + NP * PLANE_A, b_c[i], b_m[i], rb[i]);
                    else if constexpr (BH2) store_block_h2(smem + NP * PLANE_A, PLANE_B, b_c[i], b_m[i], rb[i]);
                    else store_block(smem + NP * PLANE_A, PLANE_B, b_c[i], b_m[i], rb[i], sb);
                }
        };
        const int arow = wm * WM + (lane & 31);
        const int brow = wn * WN + (lane & 31);
        const int kofs = 8 * (lane >> 5);
        auto compute = [&](int st_) __attribute__((always_inline)) {
            const unsigned short* smem = smem0 + st_ * STAGE_H;
            x3_compute<TM, TN, BK, LDH, true, NP>(acc, smem, PLANE_A, smem + NP * PLANE_A, PLANE_B, arow, brow, kofs);
        };
        gload(0);
        sstore(0);
        __syncthreads();
        if (STAGES == 2) {
            for (int kc = 0; kc + 1 < nk; ++kc) {
                gload(kc + 1);
                compute(kc & 1);
                sstore((kc & 1) ^ 1);
                __syncthreads();
            }
            compute((nk - 1) & 1);
        } else {
            for (int kc = 0; kc + 1 < nk; ++kc) {
                gload(kc + 1);
                compute(0);
                __syncthreads();
                sstore(0);
                __syncthreads();
            }
            compute(0);
        }
        __syncthreads();
    }
};

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
}

}  // namespace cpc
