// The DMA-fed tile shared by the conv kernels (conv_dma.hip) and the plain GEMMs of the transformer layer (gemm_dma.hip):
// DmaCfg / dma_gemm -- acc += A[rows, 0:K] . B[0:256, 0:K]^T with BOTH operands copied global -> LDS by global_load_lds --
// and the transposing LDS reads of the TN kernels.  (Moved out of conv_dma.hip in round 6, text unchanged.)
#pragma once
#include "cpc_common.h"
#include "cpc_internal.h"
#include "gemm_tile.h"

namespace cpc {

// BKE: contraction elements per LDS stage, NST: LDS stages, NP: storage of the operands -- 2 = H2 (two fp16 pieces per
// element, 4 bytes, three MFMAs per product: the fp32-accurate path), 1 = bf16 (2 bytes, one MFMA per product: the
// bf16-storage variant, cpc_set_mfma_mode(4)).  Rows of a stage are ROWB = 64 or 128 bytes in either storage.
// WR: rows of a wave's accumulator tile -- 64 (default: eight waves of 64 x 128 per 256-row tile, two per SIMD) or 128 (four waves of
// 128 x 128, ONE per SIMD, accumulators in the upper half of the 512-register file: a third less LDS traffic per MFMA, and a wave's
// LDS reads / DMA instructions sit between its own MFMAs instead of in its SIMD partner's matrix time -- dma_gemm's W128 loop)
template <int BM, int BKE_, int NST_, int NP_, int WR_ = 64>
struct DmaCfg {
    static constexpr int BN = kC;
    static constexpr int WR = WR_;
    static constexpr int BKE = BKE_, NST = NST_, NP = NP_;
    static constexpr int ESZ = 2 * NP;                 // bytes per element
    static constexpr int ROWB = BKE * ESZ;             // bytes per row and stage
    static constexpr int KPR = 128 / ESZ;              // contraction elements per 128-byte global weight row
    static constexpr int PPR = ROWB / 16;              // 16-byte pieces per row (4 or 8)
    static constexpr int RPP = 1024 / ROWB;            // rows per 1 KB DMA piece (16 or 8)
    static constexpr int SWSH = PPR == 8 ? 1 : 2;      // swizzle: piece ^= (row >> SWSH) & (PPR - 1)
    static constexpr int KS = BKE / 16;                // MFMA k-steps per stage
    static constexpr int WAVES_N = 2, WAVES_M = BM / WR, NW = WAVES_M * WAVES_N;
    static constexpr int NTHREADS = 64 * NW;
    static constexpr int TM = WR / 32, TN = 4;         // 32 x 32 accumulator tiles per wave (64 x 128 or 128 x 128)
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    static constexpr int A_PER = (BM / RPP) / NW, B_PER = (BN / RPP) / NW;    // 1 KB DMA pieces per wave and stage
    static constexpr int NPS = A_PER + B_PER;          // DMA instructions per wave and stage (vmcnt bookkeeping)
    static constexpr int SMEM_BYTES = NST * STAGE;
    static_assert(WR == 64 || WR == 128, "wave tiles of 64 or 128 rows");
    static_assert(NP == 1 || NP == 2, "bf16 or two fp16 pieces");
    static_assert(ROWB == 64 || ROWB == 128, "rows of 4 or 8 pieces");
    static_assert(NST >= 2 && NST <= 4, "2..4 stages");
    static_assert((BM / RPP) % NW == 0 && (BN / RPP) % NW == 0, "whole pieces per wave");
    static_assert(SMEM_BYTES >= BM * 2 * 4 * 2, "the row-statistics exchange reuses the stage buffers");
    static_assert(NPS * (NST - 2) < 64, "vmcnt is a 6-bit counter");
};

// row of the C tile held in accumulator register `reg` of tile tm / column held by this lane for tile tn
template <int WR = 64>
__device__ __forceinline__ int dma_c_row(int tm, int reg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave >> 1) * WR + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int dma_c_col(int tn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave & 1) * 128 + tn * 32 + (lane & 31);
}

// acc[64 x 128 per wave] += A[m0.., 0:K] . B[0:256, 0:K]^T with both operands DMA'd global -> LDS.
// am: rows of the A operand in ELEMENTS of C::ESZ bytes (im2col windows: element k of a row is real iff the position
// tau0 + (k >> 8) is inside [0, Lin), otherwise it reads `zeros`); wq: weight rows of 128 bytes, [K / KPR][256][128 B].
#ifdef CPC_DMA_TIMING
// tools/time_dma_slots.py: s_memtime stamps of two ping-pong iterations of one workgroup, [wave][iteration][stamp]
static __device__ unsigned long long g_dma_stamps[8 * 2 * 6];
#define CPC_STAMP(i) if (kt == 40 || kt == 41) stamp[(kt - 40) * 6 + (i)] = __builtin_readcyclecounter()
#else
#define CPC_STAMP(i)
#endif
// PP (needs BKE = 16, four stages): the two waves of a SIMD (w and w + 4) alternate roles, one multiplies k-step n from
// registers while the other reads its fragments of that k-step out of LDS and issues its DMA pieces of stage n + 3, with a
// barrier between the slots.  Without it all eight waves leave the stage barrier together, issue their DMA pieces and LDS reads
// together (the texture path takes 16 clocks per 1 KB piece, the eight waves' reads share the LDS) and only then start to
// multiply: the matrix pipe idles for more than half of every stage.
// SKEW: the slots are cut differently -- slot A = the l*h and h*l products (16 MFMAs) with the DMA pieces between them,
// slot B = the low planes of the next k-step's fragments -> the registers that just became free, the h*h products (8 MFMAs),
// then the high planes; while one wave of a SIMD is in A its partner is in B, so every slot carries 24 MFMAs per SIMD and
// both waves' non-matrix work runs in the other's matrix time.
template <class C, int NDC = -1, bool SKEW = false>
__device__ __forceinline__ void dma_gemm(f32x16 (&acc)[C::TM][C::TN], const RowMap& am, int m0,
                                         const unsigned char* __restrict__ wq, int K, const unsigned char* __restrict__ zeros,
                                         int rot_step, unsigned char* smem) {
    constexpr int TM = C::TM, TN = C::TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
    const int nkt = K / C::BKE;
    const int taps = K >> kCLog2;                                  // power of two for every caller (8, 4, 2)
    const int tshift = 31 - __builtin_clz(taps);
    // (the 128-row wave tile walks K in the 32-element chunks of the two-stage loop, two 16-k stages per chunk: the same order of
    // summation, the same bits)
    constexpr int WALK = C::WR == 128 ? 32 : C::BKE, SUB = WALK / C::BKE;
    const int nwalk = K / WALK;
    const int rot = (int)((blockIdx.x * (unsigned)rot_step) % (unsigned)nwalk);

    // ---- per-lane DMA sources.  A piece = RPP rows x ROWB bytes (1 KB); lane l of the piece covers row (l / PPR), LDS
    // slot (l % PPR), which holds global piece (l % PPR) ^ ((row >> SWSH) & (PPR - 1)) of that row.
    const unsigned char* a_src[C::A_PER];
    int a_tau0[C::A_PER];
    const unsigned char* b_src[C::B_PER];
#pragma unroll
    for (int i = 0; i < C::A_PER; ++i) {
        const int row = (wave * C::A_PER + i) * C::RPP + lane / C::PPR;
        const int piece = (lane % C::PPR) ^ ((row >> C::SWSH) & (C::PPR - 1));
        const int m = m0 + row;
        if (m < am.M) {
            const int b = m / am.R, t = m - b * am.R;
            a_src[i] = reinterpret_cast<const unsigned char*>(am.base) +
                       ((long)b * am.bstride + (long)t * am.rstride + am.off) * C::ESZ + piece * 16;
            a_tau0[i] = t * am.tmul + am.tadd;
        } else {
            a_src[i] = zeros;
            a_tau0[i] = -(1 << 30);
        }
    }
#pragma unroll
    for (int i = 0; i < C::B_PER; ++i) {
        const int row = (wave * C::B_PER + i) * C::RPP + lane / C::PPR;
        const int piece = (lane % C::PPR) ^ ((row >> C::SWSH) & (C::PPR - 1));
        b_src[i] = wq + (long)row * 128 + piece * 16;       // global weight rows are 128 B whatever the stage depth
    }
    const unsigned char* zsrc = zeros + (lane % C::PPR) * 16;
    // pieces [lo, hi) of a wave's NPS DMA pieces of stage kt (A pieces first)
    auto issue_part = [&](int kt, int stage, int lo, int hi) __attribute__((always_inline)) {
        int q = kt / SUB + rot;
        q = q >= nwalk ? q - nwalk : q;
        // tap-fastest walk (gemm_tile.h, tshift), the taps in the order 0, s, 1, s+1, ...: tap j of output row t and tap
        // j + s of row t - 1 are the same input row, so the two reads of every input row are ONE stage apart and the second
        // one hits L2 (in plain tap order they are s stages = 4 x 32 KB per CU apart: 671 MB fetched for a 268 MB
        // activation on layer 1, PMC)
        const int ti = q & (taps - 1);
        q = ((ti >> 1) + (taps >> 1) * (ti & 1)) * (nwalk >> tshift) + (q >> tshift);
        const int k0 = q * WALK + (kt % SUB) * C::BKE;
        const int tap = k0 >> kCLog2;
        const long koff = (long)k0 * C::ESZ;                               // bytes into an A row
        const long boff = (long)(k0 / C::KPR) * (C::BN * 128) + (k0 % C::KPR) * C::ESZ;
        unsigned char* as = smem + stage * C::STAGE + (wave * C::A_PER) * 1024;
        unsigned char* bs = smem + stage * C::STAGE + C::A_BYTES + (wave * C::B_PER) * 1024;
#pragma unroll
        for (int i = 0; i < C::A_PER; ++i)
            if (i >= lo && i < hi) {
                const bool ok = (unsigned)(a_tau0[i] + tap) < (unsigned)am.Lin;
                dma16_to_lds(ok ? a_src[i] + koff : zsrc, as + i * 1024);
            }
#pragma unroll
        for (int i = 0; i < C::B_PER; ++i)
            if (C::A_PER + i >= lo && C::A_PER + i < hi) dma16_to_lds(b_src[i] + boff, bs + i * 1024);
    };
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) { issue_part(kt, stage, 0, C::NPS); };

#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int sw = ((lane & 31) >> C::SWSH) & (C::PPR - 1), kg = lane >> 5;
    const int a_row0 = (wm * C::WR + (lane & 31)) * C::ROWB, b_row0 = (wn * 128 + (lane & 31)) * C::ROWB;

    if constexpr (C::WR == 128) {
        // ONE wave per SIMD, 128 x 128 per wave.  A stage is one 16-k step; the fragments of step kt are in registers (set kt & 1)
        // when iteration kt starts, so all FOUR stage buffers hold younger steps: kt + 1 (read into the other register set during
        // this iteration), kt + 2, kt + 3 (landed or in flight) and kt + 4, DMA'd into the buffer step kt was read from an iteration
        // ago -- a request has three iterations (~4600 clocks of MFMAs) to land.  Per iteration a wave issues 48 MFMAs and, one
        // behind every second MFMA, its 16 fragment reads and 8 DMA pieces: nothing of it waits for a partner wave's slot.  One
        // barrier per iteration: it publishes step kt + 2 and retires everybody's reads of step kt + 1.
        // Accumulation order per element as in the two-stage loop (per k-step l*h, h*l, h*h): the same bits.
        static_assert(C::KS == 1 && C::NST == 4 && C::NP == 2 && C::NPS == 8 && TM == 4 && TN == 4,
                      "one k-step per stage, four stages, H2 operands, eight DMA pieces per wave and stage");
        using SP = SplitPlanes<2>;
        s16x8 fa[2][TM][2], fb[2][TN][2];
        // fragment x (0..15) of the stage at `As` into register set `set`: planes 0 (x < 8) and 1, A tiles then B tiles
        auto read_frag = [&](const unsigned char* As, int set, int x) __attribute__((always_inline)) {
            const int pl = x >> 3, t = x & 7;
            const int off = ((2 * kg + pl) ^ sw) * 16;
            if (t < TM) fa[set][t][pl] = *reinterpret_cast<const s16x8*>(As + a_row0 + t * 32 * C::ROWB + off);
            else fb[set][t - TM][pl] = *reinterpret_cast<const s16x8*>(As + C::A_BYTES + b_row0 + (t - TM) * 32 * C::ROWB + off);
        };
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nkt) issue(j, j);
        CPC_WAIT_VMCNT(3 * C::NPS);
        __builtin_amdgcn_s_barrier();                           // step 0 has landed for everybody
#pragma unroll
        for (int x = 0; x < 16; ++x) read_frag(smem, 0, x);
        CPC_WAIT_LGKMCNT0();
        CPC_WAIT_VMCNT(2 * C::NPS);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                           // step 1 has landed, step 0 is in everybody's registers
        __builtin_amdgcn_sched_barrier(0);
        // DMA: issue the pieces of step kt + 4 (the main part of the loop); LEFT: DMA requests of mine that may stay in flight
        // across the closing barrier (step kt + 2 must have landed: everything older than the youngest LEFT)
        auto step = [&](int kt, auto set_tag, auto dma_tag, auto left_tag) __attribute__((always_inline)) {
            constexpr int set = decltype(set_tag)::value, LEFT = decltype(left_tag)::value;
            constexpr bool DMA = decltype(dma_tag)::value;
            const unsigned char* As = smem + ((kt + 1) & 3) * C::STAGE;
#pragma unroll
            for (int i = 0; i < 3 * TM * TN; ++i) {
                const int q = i / (TM * TN), tm = (i / TN) % TM, tn = i % TN;
                acc[tm][tn] = SP::mfma(fa[set][tm][SP::pa(q)], fb[set][tn][SP::pb(q)], acc[tm][tn]);
                // behind MFMAs 0, 2, .., 30 a fragment read of the next step (the last iteration reads a stale buffer into registers
                // nobody uses); behind MFMAs 3, 9, .., 45 a DMA piece -- 192 clocks of MFMAs apart: a piece occupies the wave's issue
                // for 60+ clocks (MI355X_MICROARCH.md), packed back to back they stall the matrix pipe
                if (i % 2 == 0 && i < 32) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_frag(As, set ^ 1, i / 2);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (DMA && i % 6 == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_part(kt + 4, kt & 3, i / 6, i / 6 + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            CPC_WAIT_LGKMCNT0();
            CPC_WAIT_VMCNT(LEFT);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using Yes = std::integral_constant<bool, true>;
        using No = std::integral_constant<bool, false>;
        int kt = 0;
        for (; kt + 4 < nkt; kt += 2) {                         // (nkt is a multiple of 16: K = 256 * 2^j)
            step(kt, I0{}, Yes{}, std::integral_constant<int, 2 * C::NPS>{});
            step(kt + 1, I1{}, Yes{}, std::integral_constant<int, 2 * C::NPS>{});
        }
        step(kt, I0{}, No{}, std::integral_constant<int, C::NPS>{});        // kt = nkt - 4: steps up to nkt - 1 are issued
        step(kt + 1, I1{}, No{}, I0{});
        step(kt + 2, I0{}, No{}, I0{});
        step(kt + 3, I1{}, No{}, I0{});
        __syncthreads();
        return;
    }
    if constexpr (SKEW) {
        static_assert(C::KS == 1 && C::NST == 4 && C::NP == 2 && C::NPS == 4, "one k-step per stage, four stages, H2 operands");
        using SP = SplitPlanes<2>;
        const int grp = wave >> 2;                              // waves w and w + 4 share a SIMD
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nkt) issue(j, j);
        CPC_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();                           // stages 0..2 have landed for everybody
        s16x8 af[TM][2], bf[TN][2];
        auto read_plane = [&](int kt, int pl) __attribute__((always_inline)) {
            const unsigned char* As = smem + (kt & 3) * C::STAGE;
            const unsigned char* Bs = As + C::A_BYTES;
            const int off = ((2 * kg + pl) ^ sw) * 16;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                af[tm][pl] = *reinterpret_cast<const s16x8*>(As + a_row0 + tm * 32 * C::ROWB + off);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                bf[tn][pl] = *reinterpret_cast<const s16x8*>(Bs + b_row0 + tn * 32 * C::ROWB + off);
        };
        read_plane(0, 0);
        read_plane(0, 1);
        CPC_WAIT_LGKMCNT0();
        if (grp == 1) __builtin_amdgcn_s_barrier();             // the second group runs one slot behind
#ifdef CPC_DMA_TIMING
        unsigned long long stamp[12] = {};
#endif
        for (int kt = 0; kt < nkt; ++kt) {
            CPC_STAMP(0);
            // ---- slot A: l*h and h*l of k-step kt; my DMA pieces of stage kt + 3 (into the buffer of stage kt - 1, read two
            // slots ago) between the MFMAs
            const bool more = kt + 3 < nkt;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2 * TM * TN; ++i) {
                const int q = i / (TM * TN), tm = (i / TN) % TM, tn = i % TN;
                acc[tm][tn] = SP::mfma(af[tm][SP::pa(q)], bf[tn][SP::pb(q)], acc[tm][tn]);
                if (i % 4 == 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) issue_part(kt + 3, (kt + 3) & 3, i / 4, i / 4 + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            CPC_STAMP(1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            CPC_STAMP(2);
            // ---- slot B: the low planes of k-step kt + 1 (their registers are free), h*h of k-step kt, the high planes
            const bool next = kt + 1 < nkt;
            if (next) read_plane(kt + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = SP::mfma(af[tm][0], bf[tn][0], acc[tm][tn]);
            __builtin_amdgcn_sched_barrier(0);
            CPC_STAMP(3);
            if (next) read_plane(kt + 1, 0);
            CPC_WAIT_LGKMCNT0();
            if (more) { CPC_WAIT_VMCNT(C::NPS); }               // my pieces of stage kt + 2 have landed (kt + 3 in flight)
            else { CPC_WAIT_VMCNT(0); }
            __builtin_amdgcn_sched_barrier(0);
            CPC_STAMP(4);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            CPC_STAMP(5);
        }
#ifdef CPC_DMA_TIMING
        if (blockIdx.x == 37 && lane == 0)
            for (int i = 0; i < 12; ++i) g_dma_stamps[wave * 12 + i] = stamp[i];
#endif
        if (grp == 0) __builtin_amdgcn_s_barrier();
        __syncthreads();
        return;
    }
    if constexpr (NDC >= 0) {
        static_assert(C::KS == 1 && C::NST == 4 && C::NP == 2 && NDC <= C::NPS, "ping-pong: one k-step per stage, four stages, H2 operands");
        using SP = SplitPlanes<2>;
        constexpr int NDL = C::NPS - NDC;                       // pieces issued in the load slot
        const int grp = wave >> 2;                              // waves w and w + 4 share a SIMD
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nkt) issue(j, j);
        if (nkt > 2) { CPC_WAIT_VMCNT(2 * C::NPS); }
        else { CPC_WAIT_VMCNT(0); }
        __builtin_amdgcn_s_barrier();                           // stage 0 has landed for everybody
        if (grp == 1) __builtin_amdgcn_s_barrier();             // the second group runs one slot behind
#ifdef CPC_DMA_TIMING
        unsigned long long stamp[12] = {};
#endif
        for (int kt = 0; kt < nkt; ++kt) {
            CPC_STAMP(0);
            // ---- load slot: fragments of k-step kt -> registers; the first NDL DMA pieces of stage kt + 3 into the buffer of
            // stage kt - 1 (whose last readers finished their load slot before the barrier that opened this one)
            const unsigned char* As = smem + (kt & 3) * C::STAGE;
            const unsigned char* Bs = As + C::A_BYTES;
            s16x8 af[TM][2], bf[TN][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const int off = ((2 * kg + pl) ^ sw) * 16;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    af[tm][pl] = *reinterpret_cast<const s16x8*>(As + a_row0 + tm * 32 * C::ROWB + off);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    bf[tn][pl] = *reinterpret_cast<const s16x8*>(Bs + b_row0 + tn * 32 * C::ROWB + off);
            }
            const bool more = kt + 3 < nkt;
            if (NDL > 0 && more) issue_part(kt + 3, (kt + 3) & 3, 0, NDL);
            CPC_STAMP(1);
            CPC_WAIT_LGKMCNT0();                                // my reads are done before anybody may overwrite the buffer
            // my pieces of stage kt + 1 have landed; stage kt + 2 and what this slot issued of kt + 3 stay in flight
            if (more) { CPC_WAIT_VMCNT(C::NPS + NDL); }
            else if (kt + 2 < nkt) { CPC_WAIT_VMCNT(C::NPS); }
            else { CPC_WAIT_VMCNT(0); }
            CPC_STAMP(2);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            CPC_STAMP(3);
            // ---- multiply slot, the other NDC pieces spread between the MFMAs (a piece issued among MFMAs costs the wave
            // far less than one issued in a burst of loads, MI355X_MICROARCH.md)
            constexpr int NM = SP::NPROD * TM * TN;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int q = i / (TM * TN), tm = (i / TN) % TM, tn = i % TN;
                acc[tm][tn] = SP::mfma(af[tm][SP::pa(q)], bf[tn][SP::pb(q)], acc[tm][tn]);
                if constexpr (NDC > 0) {
                    constexpr int every = NM / NDC;
                    if (i % every == every - 3) {               // after MFMAs 3, 9, 15, 21 (NDC = 4)
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) issue_part(kt + 3, (kt + 3) & 3, NDL + i / every, NDL + i / every + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            CPC_STAMP(4);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            CPC_STAMP(5);
        }
#ifdef CPC_DMA_TIMING
        if (blockIdx.x == 37 && lane == 0)
            for (int i = 0; i < 12; ++i) g_dma_stamps[wave * 12 + i] = stamp[i];
#endif
        if (grp == 0) __builtin_amdgcn_s_barrier();
        __syncthreads();
        return;
    }
    // NST - 1 stages ahead: at the top of iteration kt the stages kt .. kt + NST - 2 have been issued; stage kt must have
    // landed (vmcnt leaves the NST - 2 younger ones in flight), the barrier publishes it to the other waves and retires
    // everybody's reads of stage kt - 1, whose buffer the DMA of stage kt + NST - 1 then overwrites.
#pragma unroll
    for (int j = 0; j < C::NST - 1; ++j)
        if (j < nkt) issue(j, j);
    int stage = 0;
#ifdef CPC_DMA_TIMING
    unsigned long long stamp[12] = {};
#endif
    for (int kt = 0; kt < nkt; ++kt) {
        CPC_STAMP(0);
        const int younger = min(C::NST - 2, nkt - 1 - kt);      // issued after stage kt and allowed to be in flight (uniform)
        if (younger >= 2) { CPC_WAIT_VMCNT(2 * C::NPS); }
        else if (younger == 1) { CPC_WAIT_VMCNT(C::NPS); }
        else { CPC_WAIT_VMCNT(0); }
        CPC_STAMP(1);
        __builtin_amdgcn_s_barrier();
        CPC_STAMP(2);
        if (kt + C::NST - 1 < nkt) {
            int st = stage + C::NST - 1;
            st = st >= C::NST ? st - C::NST : st;
            issue(kt + C::NST - 1, st);
        }
        CPC_STAMP(3);
        const unsigned char* As = smem + stage * C::STAGE;
        const unsigned char* Bs = As + C::A_BYTES;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            if constexpr (C::NP == 2) {
                using SP = SplitPlanes<2>;
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
            } else {
                s16x8 af[TM], bf[TN];
                const int off = ((2 * ks + kg) ^ sw) * 16;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    af[tm] = *reinterpret_cast<const s16x8*>(As + a_row0 + tm * 32 * C::ROWB + off);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    bf[tn] = *reinterpret_cast<const s16x8*>(Bs + b_row0 + tn * 32 * C::ROWB + off);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[tm]),
                                                                              __builtin_bit_cast(bf16x8, bf[tn]), acc[tm][tn], 0, 0, 0);
            }
        }
        CPC_STAMP(4);
        CPC_STAMP(5);
        stage = stage + 1 == C::NST ? 0 : stage + 1;
    }
#ifdef CPC_DMA_TIMING
    if (blockIdx.x == 37 && lane == 0)
        for (int i = 0; i < 12; ++i) g_dma_stamps[wave * 12 + i] = stamp[i];
#endif
    __syncthreads();                            // the stage buffers are free (the epilogues reuse them)
}

// ds_read_b64_tr_b16 at `base + OFF` (OFF: immediate).  By inline asm, because hipcc drains the whole DMA queue (s_waitcnt vmcnt(0))
// in front of every __builtin_amdgcn_ds_read_tr16_b64 that follows a global_load_lds -- it cannot tell that the read touches another
// stage than the requests in flight (plain ds_read_b128 of the forward kernel are spared that) -- which left round 2's
// weight-gradient kernel with NO overlap of DMA and MFMAs: request a stage, wait for it, multiply (rocprofv3: 4-4.6 us per 32-row
// stage against 1.3 us of MFMAs, whatever the grid size).  The asm reads are invisible to the waitcnt pass: the kernel waits for
// them itself (lds_wait_tr16 below) before the first MFMA that consumes them.
template <int OFF>
__device__ __forceinline__ s16x4 lds_read_tr16(const unsigned char* base) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIPEMU)
    s16x4 v;
    const unsigned a = (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)base;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
    return v;
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + OFF));
#endif
}
// s_waitcnt lgkmcnt(0) that the twelve fragments (24 half reads) of a k-step pass THROUGH: nothing that consumes them can be
// scheduled in front of it, and no copy of them is made before it
__device__ __forceinline__ void lds_wait_tr16(s16x4 (&a)[2][2][2], s16x4 (&b)[4][2][2]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIPEMU)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0][0][0]), "+v"(a[0][0][1]), "+v"(a[0][1][0]), "+v"(a[0][1][1]), "+v"(a[1][0][0]), "+v"(a[1][0][1]),
                   "+v"(a[1][1][0]), "+v"(a[1][1][1]), "+v"(b[0][0][0]), "+v"(b[0][0][1]), "+v"(b[0][1][0]), "+v"(b[0][1][1]),
                   "+v"(b[1][0][0]), "+v"(b[1][0][1]), "+v"(b[1][1][0]), "+v"(b[1][1][1]), "+v"(b[2][0][0]), "+v"(b[2][0][1]),
                   "+v"(b[2][1][0]), "+v"(b[2][1][1]), "+v"(b[3][0][0]), "+v"(b[3][0][1]), "+v"(b[3][1][0]), "+v"(b[3][1][1]));
#endif
}

__device__ __forceinline__ void lds_wait_tr16(s16x4 (&a)[2][2], s16x4 (&b)[4][2]) {      // one piece (bf16 storage): 12 half reads
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIPEMU)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]),
                   "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[3][0]), "+v"(b[3][1]));
#endif
}

}  // namespace cpc
