// Internal (non-ABI) helpers shared between translation units of libcpc_hip.
#pragma once
#include "cpc_common.h"
#include "gemm_tile.h"

namespace cpc {

// out[0:n] = sum over `nrows` rows of `part` (row length n), summed in a fixed order.
// tmp must hold kRowsSumGroups*n floats.
constexpr int kRowsSumGroups = 128;
// (G independent reductions: part, tmp, out of problem g at + g * part_gs / tmp_gs / out_gs floats)
int rows_sum(const float* part, int nrows, int n, float* tmp, float* out, hipStream_t stream, int G = 1, long part_gs = 0,
             long tmp_gs = 0, long out_gs = 0);
// several independent reductions (each with its own tmp) in two launches; bit-identical with rows_sum per job
constexpr int kRowsSumMaxJobs = 8;
struct RowsSumJob { const float* part; int nrows; int n; float* tmp; float* out; };
int rows_sum_multi(const RowsSumJob* jobs, int njobs, hipStream_t stream);

// Device-side upper bounds of max|A| and max|B| (each `slots` <= 64 floats whose maximum is the bound).  Given both -- and in
// cpc_set_mfma_mode >= 2 -- a GEMM runs on the fp16 pipe with two-piece split operands (3 MFMAs per product) instead of
// three bf16 pieces (6); the bounds need not be tight (a factor 2^10 costs nothing measurable, gemm_tile.h).
struct GemmBounds {
    const float* a = nullptr;
    const float* b = nullptr;
    int a_slots = 1, b_slots = 1;
    long a_gs = 0, b_gs = 0;    // grouped launches (GemmGroup): problem g reads its bounds at a + g * a_gs / b + g * b_gs
};
// max|x| of up to 4 arrays in one launch, as kAmaxSlots partial maxima each (no memset, no atomics): out[j*64 .. j*64+63]
// (x[j] == NULL: the bound of job j is known a priori -- its slots all carry cval[j])
int absmax_slots(const float* const* x, const long* n, int njobs, float* out, hipStream_t st, const float* cval = nullptr);
// the same for G equally sized arrays per job (array g of job j at x[j] + g * x_gs[j]): out[j * kAmaxSlots + g * out_gs + slot]
int absmax_group(const float* const* x, const long* n, const long* x_gs, int njobs, int G, float* out, long out_gs, hipStream_t st);

// G equally shaped problems in one launch (blockIdx.z / .y picks the problem): problem g reads and writes at the given
// pointers + g * stride (floats).  A stride of 0 shares the operand between the problems.
struct GemmGroup {
    int G = 1;
    long a = 0, b = 0, bias = 0, c = 0;
    long part = 0;      // tn_gemm: stride of the split partials (0: problem g's S * N1 * N2 floats right behind problem g-1's)
    // tn_gemm: rows [k * out_scale_rows, (k + 1) * out_scale_rows) of C are multiplied by out_scale[k] (device floats) on their
    // way out of the split reduction (the K prediction heads' weight gradient from the unit-gradient dPred, nce.hip)
    const float* out_scale = nullptr;
    int out_scale_rows = 0;
};
// A scratch buffer a narrow NT product may use to split its K walk over several workgroups per output tile (N = 256 gives
// M / 128 tiles of the wide kernel -- 58 for the criterion's dc: a quarter of the chip): partial products land in `part`
// (floats >= splits * M * N, splits <= 8) and one more launch sums them into C in a fixed order.
struct SplitK {
    float* part = nullptr;
    long floats = 0;
};
// What the feed-forward GEMMs of the transformer layer do to C on its way out (transformer.hip), fused into the wide fp16-piece
// tile's epilogue -- nt_gemm_fuses() says whether a product runs on that tile; otherwise the caller launches the elementwise
// kernel behind the GEMM (same arithmetic, same dropout masks).
//   kind 1: C = dropout(relu(A.B^T + bias)): keep / (1 - p) by Philox site 1 (philox.h; stream seed + problem index); C's rows
//           are kFfnWidth wide
//   kind 2: C = (A.B^T) * scale where mask > 0, else 0 (mask: a tensor of C's shape and row pitch, problem g at + g * mask_gs)
//           (+ colsum: what the bias gradient of the layer in front needs, without another pass over C)
// amax (or NULL): max|C| into kAmaxSlots slots by atomicMax (zeroed by the caller), problem g at + g * amax_gs.
struct GemmEpilogue {
    int kind = 0;
    float drop_p = 0.f;
    unsigned long long seed = 0;
    const float* mask = nullptr;
    long mask_gs = 0;
    float scale = 1.f;
    float* amax = nullptr;
    long amax_gs = 0;
    float* colsum = nullptr;      // (or NULL) per-row-tile column sums of C: colsum[(m0 / 128) * N + col], problem g at + g * colsum_gs
    long colsum_gs = 0;           // -- the bias gradient's partials, summed over the row tiles by rows_sum afterwards
};
bool nt_gemm_fuses(int M, int N, int K, int ldc, const GemmBounds& gb, const GemmGroup& grp);
int nt_gemm_fused(const RowMap& am, const float* Bmat, int ldb, const float* bias, float* C, long ldc, int N, int K,
                  hipStream_t st, GemmBounds gb, GemmGroup grp, GemmEpilogue ep);
// C[M,N] = A . B[N,K]^T (+ bias);  N % 128 == 0, K % 16 == 0
int nt_gemm(const RowMap& am, const float* Bmat, int ldb, const float* bias, float* C, long ldc, int N,
            int K, hipStream_t st, int c_R = 0, long c_bstride = 0, GemmBounds gb = GemmBounds(), GemmGroup grp = GemmGroup(),
            SplitK sk = SplitK());
// C[N1,N2] (+)= sum_m A[m,:N1]^T (x) B[m,:N2];  part: tn_gemm_part_floats(M,N1,N2) floats
void tn_gemm_plan(int M, int N1, int N2, int* splits, int* rows);
long tn_gemm_part_floats(int M, int N1, int N2);
// (grouped: part holds grp.G * tn_gemm_part_floats floats; grp.c strides C)
int tn_gemm(const RowMap& am, int N1, const RowMap& bm, int N2, float* part, float* C, int accumulate,
            hipStream_t st, GemmBounds gb = GemmBounds(), GemmGroup grp = GemmGroup());
// nprob <= 4 TN problems with equal M, N1, N2 in one GEMM launch + one reduction; part: tn_gemm_batch_part_floats
long tn_gemm_batch_part_floats(int nprob, int M, int N1, int N2);
int tn_gemm_batch(int nprob, const RowMap* am, int N1, const RowMap* bm, int N2, float* part, float* const* C,
                  int accumulate, hipStream_t st);
// out[Cn][R] = in[R][Cn]^T  (G matrices in_gs / out_gs floats apart)
int transpose(const float* in, float* out, int R, int Cn, hipStream_t st, int G = 1, long in_gs = 0, long out_gs = 0);
// n <= 4 matrices of one shape in one launch
int transpose_batch(const float* const* in, float* const* out, int n, int R, int Cn, hipStream_t st);

// ---- gemm_dma.hip: plain GEMMs on the DMA-fed tile, operands in H2 storage (every stride in floats = H2 elements; bounds as
// kAmaxSlots partial maxima; G problems per launch, problem g at + g * the *_gs strides)
bool gemm_dma_wanted(int M, int G);
bool gemm_dma_relu_fused();                                          // lin1's ReLU + dropout + H2 storage in its epilogue?                                  // cpc_set_gemm_dma: do a call's launches fill the chip?
// B(n, k) = w[n * sn + k * sk] -> the K-tile-major H2 rows gemm_nt_dma reads (N * K floats), scaled for max|w| (`amax`);
// l1 (or NULL; zeroed by the caller): max_n sum_k |B(n, k)| -- |A . B^T| <= max|A| * that
int gemm_weight_h2(const float* w, long sn, long sk, int N, int K, float* wq, const float* amax, float* l1, int G, long w_gs,
                   long wq_gs, long amax_gs, long l1_gs, hipStream_t st);
int rows_to_h2(const float* x, float* xh, const float* bound, long rows, int G, long x_gs, long xh_gs, long bound_gs, hipStream_t st);
// C (fp32) = A . B^T + bias; amax_out (or NULL; zeroed by the caller): max|C|.  N % 256 == 0, K = 256 * 2^j
int gemm_nt_dma(const float* a_h2, int lda, const float* wq, const float* bias, float* C, long ldc, int M, int N, int K,
                const float* a_bound, const float* w_amax, float* amax_out, int G, long a_gs, long wq_gs, long bias_gs, long c_gs,
                long a_bound_gs, long w_amax_gs, long amax_gs, hipStream_t st);
// one product whose A rows are a RowMap over an H2 tensor (strides in H2 elements), C fp32 with rows ldc apart, no bias
int gemm_nt_dma_rows(const RowMap& am_h2, const float* wq, float* C, long ldc, int N, int K, const float* a_bound, const float* w_amax,
                     hipStream_t st);
// C (H2) = dropout(relu(A . B^T + bias)) (Philox site 1, stream seed + problem), scaled for (max|A| * w_l1 + max|bias|) / (1 - p),
// which goes to every slot of out_slots; bits: [C != 0], one bit per element (N / 8 bytes per row); flag[0] = 1.  N = 2048.
// (w_amax, w_l1, bias_amax: slot arrays w_gs apart per problem; out_slots / flag: out_gs apart)
int gemm_nt_dma_relu(const float* a_h2, int lda, const float* wq, const float* bias, float* c_h2, long ldc, float* bits, float drop_p,
                     unsigned long long seed, int M, int N, int K, const float* a_bound, const float* w_amax, const float* w_l1,
                     const float* bias_amax, float* out_slots, float* flag, int G, long a_gs, long wq_gs, long bias_gs, long c_gs,
                     long bits_gs, long a_bound_gs, long w_gs, long out_gs, hipStream_t st);
// C (H2, scaled for max|A| * l1 * scale) = (A . B^T) * scale where the bit of mask_h2 (one bit per element of C, N / 8 bytes per
// row, mask_gs in floats) is set, else 0;
// colsum (or NULL): [ceil(M / 256)][N] column sums per row tile; out_slots (or NULL): kAmaxSlots floats, every one = that bound
int gemm_nt_dma_masked(const float* a_h2, int lda, const float* wq, float* c_h2, long ldc, const float* mask_h2, float scale, int M,
                       int N, int K, const float* a_bound, const float* w_amax, const float* w_l1, float* colsum, float* out_slots, int G,
                       long a_gs, long wq_gs, long c_gs, long mask_gs, long a_bound_gs, long w_amax_gs, long w_l1_gs, long colsum_gs,
                       hipStream_t st);
// C[N1, N2] = A^T . B over M rows (both H2); part: gemm_tn_dma_part_floats(M, N1, N2, G) floats per problem
long gemm_tn_dma_part_floats(int M, int N1, int N2, int G);
int gemm_tn_dma(const float* a_h2, int lda, int N1, const float* b_h2, int ldb, int N2, int M, float* part, float* C,
                const float* a_bound, const float* b_bound, int G, long a_gs, long b_gs, long part_gs, long c_gs, long a_bound_gs,
                long b_bound_gs, hipStream_t st);

// 1 (default): NT GEMMs run on the bf16 matrix pipe with 3-piece split operands (NtTileX3, fp32-level
// accuracy); 0: exact-f32 MFMA (NtTile).  Set through cpc_set_mfma_mode().
extern int g_mfma_mode;

// per-(device, caller stream) pool of events for the two-stream entry points: [0..4] encoder backward, [5] everything of the
// weight-gradient stream but layer 1's weight gradient is done (recorded there by the encoder's backward: what a mid gradient
// bucket / an open-tailed step waits for), [7] layer 1's weight gradient reduced, [8] GRU backward, [9] criterion backward
// (score gradients -> dz stream), [10] criterion forward (loss reduction on its own stream), [11] the recurrence's weight gradients
// when the composite step runs them on the preparation stream, [12..20] the composite train step (train_step.hip), [21] conv1's updated weight and its
// layouts ready for the next step (cpc_train_step_tail), [22] the same for every other parameter but conv0's, [6] / [23] open
// tail: the last stand-alone norm backward is done (main) / the batched column sums are (sums stream)
// [24] the encoder's forward is done (composite step: the criterion's H2 copy of z is made beside the recurrence)
// [25] layer 1's weight-gradient GEMM is done (recorded in front of its split reduction: cpc_set_tail_schedule(3))
constexpr int kStreamEvents = 26;
constexpr int kEvWgradRest = 5, kEvNorm1 = 6, kEvWgrad1 = 7, kEvGruWgrad = 11, kEvNextConv1 = 21, kEvNextRest = 22, kEvSums = 23,
              kEvEncoderDone = 24, kEvWgrad1Gemm = 25;
hipEvent_t* stream_events(hipStream_t caller_stream);

// ---- hooks of the composite step into the per-stage entry points (train_step.hip sets them around its calls; per host
// thread; all default to "off") -------------------------------------------------------------------------------------------
struct StepHooks {
    int parity = 0;               // which of the two y0 buffers / input-bound sets of the encoder workspace this step uses
    bool weights_ready = false;   // the conv weight layouts, max|w| and input bounds of this parity were prepared by the previous
                                  // step's tail (cpc_train_step_tail): the forward launches no preparation
    hipEvent_t conv1_wait[2] = {nullptr, nullptr};   // the forward's stream waits for these in front of layer 1 (updated weights)
    bool open_tail = false;       // encoder backward: the caller's stream is joined with everything of the weight-gradient stream
                                  // BUT layer 1's weight gradient (event kEvWgradRest); kEvWgrad1 is recorded behind that one
    hipStream_t sums_stream = nullptr;   // open tail: the batched column sums of the norm backwards (bias / norm gradients of layers
                                  // 1..4) run HERE, released behind layer 1's norm backward, instead of at the end of the chain
    hipEvent_t* timers = nullptr; // in-step timing (cpc_set_step_timing): 10 timing-enabled events, or nullptr
};
StepHooks& step_hooks();
// record timing event `idx` on `st` if in-step timing is on: [0] before conv0, [1] behind conv0, [7] in front of conv1 (behind
// its stream waits), [2] behind conv1, [3] / [4] around the forward recurrence's persistent launch(es), [5] / [6] around the
// backward recurrence's, [8] / [9] around the criterion's scoring kernel
inline void step_timer_mark(int idx, hipStream_t st) {
    hipEvent_t* t = step_hooks().timers;
    if (t) (void)hipEventRecord(t[idx], st);
}

// conv_dma.hip: forward conv layer with both operands DMA'd into LDS (H2 storage, cpc_common.h)
int conv_fwd_dma(const float* x_h2, const float* wq, const float* bias, const float* nw, const float* nb, float* y,
                 int y_h2, float* xhat, float* rstd, const float* x_amax, const float* y_amax, const float* zeros, int B,
                 int Lin, int k, int s, int p, int bm, hipStream_t st);
int permute_w_h2(const float* w, float* wq, int k, const float* amax, hipStream_t st);
// bf16-storage variant (mode 4): forward / data-gradient conv layers on bf16 tensors, bf16 -> fp32 copy
int conv_fwd_dma_bf16(const void* x, const void* wq, const float* bias, const float* nw, const float* nb, void* y, int y_f32,
                      void* xhat, float* rstd, const float* zeros, int B, int Lin, int k, int s, int p, hipStream_t st);
int conv_dgrad_dma_bf16(const void* dx, const void* wd, void* dprev, const float* zeros, int B, int Lin, int k, int s, int p,
                        hipStream_t st);
int conv_dgrad_dma_h2(const void* dx_h2, const float* wd, float* dprev, const float* zeros, const float* dx_bound,
                      float* amax_out, int B, int Lin, int k, int s, int p, hipStream_t st);
// weight gradient with dx and x both in H2 storage: DMA + transposing LDS reads (conv_dma.hip)
void conv_wgrad_dma_plan(int M, int k, int* splits, int* rows, int wgs = 0);     // wgs: 0 = the current setting, 512 = the largest
int conv_wgrad_dma_bf16(const void* dx, const void* x, float* part, const float* zeros, int B, int Lin, int k, int s, int p,
                        int* splits_out, hipStream_t st);
int conv_wgrad_dma(const void* dx_h2, const void* x_h2, float* part, const float* dx_bound, const float* x_bound,
                   const float* zeros, int B, int Lin, int k, int s, int p, int* splits_out, hipStream_t st);
int bf16_decode(const void* src, float* dst, long n, hipStream_t st);
int conv0_forward_bf16(const float* wave, const float* w, const float* bias, const float* nw, const float* nb, void* y,
                       float* mean, float* rstd, int B, int L, hipStream_t st);
int conv0_backward(const float* wave, const float* w, const float* bias, const float* nw, const float* nb,
                   const float* mean, const float* rstd, const void* dy, int dy_bf16, float* scratch, float* dW0, float* dB0,
                   float* dNW0, float* dNB0, int B, int L, hipStream_t stream);

// device-side error words of the translation units that own them (cpc_device_error_flags)
int gru_error_flag_fetch(int clear, unsigned* out);
int nce_error_flag_fetch(int clear, unsigned* out);
int enc_error_flag_fetch(int clear, unsigned* out);

static inline long align64l(long v) { return (v + 63) & ~63L; }

}  // namespace cpc
