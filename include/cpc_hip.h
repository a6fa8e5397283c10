/* cpc_hip.h -- C ABI of libcpc_hip.so, the MI355X (gfx950) implementation of the
 * CPC-audio train-step hot path.
 *
 * The reference (facebookresearch/CPC_audio) has no FFI: its hot path is a set of
 * torch.nn.Module classes.  This ABI is what those modules' forward/backward bind to
 * in the drop-in package (cpc_audio_amd/model.py, criterion.py via ctypes); each entry
 * point cites the reference code it replaces.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (or int32 where stated), owned by the
 *     caller (torch's caching allocator); nothing is allocated or retained here;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no syncs;
 *   - activations are channels-last: (B, L, C) with C = 256 contiguous;
 *   - return value 0 = ok, CPC_ERR_* = argument error, 1000+e = hipError_t e.
 */
#ifndef CPC_HIP_H
#define CPC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CPC_OK 0
#define CPC_ERR_SHAPE 1        /* unsupported / inconsistent sizes */
#define CPC_ERR_ARG 2          /* null pointer or bad flag */

/* ABI version, bumped on any signature change. */
int cpc_abi_version(void);
/* The two-stream entry points (*_streams) keep a small pool of hipEvents per (device, caller stream), created on first use and
 * reused for the life of the process -- right for long-lived streams.  A caller that creates and destroys streams hands each
 * one back before destroying it (no *_streams call of this library may be in flight on it). */
int cpc_release_stream(void* stream);
/* *overlap = 1 if a kernel on stream_b completes while stream_a is busy, 0 if the two execute in submission order: the runtime
 * maps hipStreams onto a few hardware queues (4 per priority level) in creation order, and two streams on one queue serialise
 * whatever their events allow.  Launches a 3 ms one-wavefront wait kernel on stream_a and blocks the calling thread until both
 * streams drain: a start-up probe, used by the host side to choose the side streams of cpc_train_step. */
int cpc_streams_overlap(void* stream_a, void* stream_b, int* overlap);

/* Arithmetic of the NT GEMMs (conv forward/dgrad, projections, heads):
 *   1 bf16 matrix pipe, fp32 operands split by truncation into three bf16 pieces, six
 *     bf16 MFMAs per product, fp32 accumulate: error <= 2^-23 per product (fp32 level), 2.67x the rate;
 *   0 exact-f32 MFMA (v_mfma_f32_32x32x2_f32);
 *   2 fp16 matrix pipe for the 128-row conv tiles, operands scaled by a power of two (from a bound on their max|.|) and split into two
 *     fp16 pieces, three fp16 MFMAs per product (hh + hl + lh), fp32 accumulate: error <= 2^-21 per product,
 *     half the MFMAs of mode 1.  Applies to the conv layers (forward, dgrad, wgrad), whose operand bounds
 *     come for free (ChannelNorm affine; max|gradient| accumulated by the producing kernel); the small generic
 *     GEMMs (projections, heads, weight gradients of the AR / criterion) run as in mode 1 unless their operand bounds are
 *     known (cpc_set_gemm_split);
 *   3 (default) as 2, and the encoder (cpc_encoder_forward / _backward) keeps the output of layer 0 -- at B >= ~100 also
 *     of layer 1 -- in H2 storage (below): conv1 (conv2), 70 % (87 %) of the conv stack's FLOPs, read both GEMM operands
 *     global -> LDS by DMA (csrc/conv_dma.hip, 256- / 128-row tiles); their weight gradients read the pieces as stored.  The
 *     per-layer entry points (cpc_conv_layer_*, cpc_conv_gemm_forward, cpc_norm_backward) behave as in mode 2. */
int cpc_set_mfma_mode(int mode);
int cpc_get_mfma_mode(void);

/* Device-side error flags of the current device, accumulated since they were last cleared (clear != 0 clears them):
 *   CPC_DEVERR_GRU_POLL_TIMEOUT  a workgroup of the persistent recurrence (cpc_gru_forward / _backward) gave up
 *                                waiting for another one; its outputs carry NaN from that step on
 *   CPC_DEVERR_NEGATIVE_INDEX    cpc_nce_prepare was handed a draw outside batchIdx in [0,B) / seqIdx in [0,S)
 *                                (criterion.py:181-189 draws [0,B) and [1,S)); the index was clamped
 *   CPC_DEVERR_CONV_EXCHANGE     a workgroup of the N-split conv forward (cpc_set_fwd_nsplit) gave up waiting for its partner's
 *                                ChannelNorm statistics; its rows carry NaN
 * The reference raises Python exceptions for such things; kernels cannot, so the wrapper (ops.check_device_errors) turns
 * the mask into a RuntimeError.  The call synchronises with the device: logging points and tests, not the step path.
 * Returns the mask (>= 0) or a negative number if the flags cannot be read. */
#define CPC_DEVERR_GRU_POLL_TIMEOUT 1
#define CPC_DEVERR_NEGATIVE_INDEX 2
#define CPC_DEVERR_CONV_EXCHANGE 4
int cpc_device_error_flags(int clear);

/* ---------------------------------------------------------------- encoder ----
 * CPCEncoder.forward, cpc/model.py:99-105:  5 x relu(ChannelNorm(conv_i(x))).
 * ChannelNorm: cpc/model.py:50-58 (mean / UNBIASED variance over channels, eps 1e-5).
 */

/* Layer 0 (conv0 1->256 k10 s5 p3, model.py:83,100) fused with norm + ReLU.
 * wave (B,L); w (256,1,10); bias,nw,nb (256); y (B,L0,256); mean,rstd (B*L0). */
int cpc_conv0_forward(const float* wave, const float* w, const float* bias, const float* nw,
                      const float* nb, float* y, float* mean, float* rstd, int B, int L,
                      void* stream);
/* "H2" activation storage (csrc/cpc_common.h): an fp32 tensor of 256-channel rows kept as two fp16 pieces per element,
 * x * s = h + l with s = the power of two that maps a bound *amax >= max|x| into fp16's range -- exactly the two operand
 * pieces of the 3-product fp16 GEMM tiles -- laid out per 8 channels as [h x 8 | l x 8] (32 bytes; 1 KB per row, as fp32).
 * In mode 3 (cpc_set_mfma_mode) the encoder keeps the outputs of layers 0 and 1 in this form: conv1 / conv2 then read both
 * GEMM operands global -> LDS by DMA.  The entry points below expose the pieces (tests, benchmarks, other callers). */
/* conv0 writing y in H2 storage scaled by scale(*y_amax) (y_amax == NULL: plain fp32, = cpc_conv0_forward) */
int cpc_conv0_forward_h2(const float* wave, const float* w, const float* bias, const float* nw, const float* nb,
                         void* y, float* mean, float* rstd, const float* y_amax, int B, int L, void* stream);
/* fp32 (n_rows,256) <-> H2 with the scale of *amax */
int cpc_h2_encode(const float* src, void* dst, long n_rows, const float* amax, void* stream);
int cpc_h2_decode(const void* src, float* dst, long n_rows, const float* amax, void* stream);
/* conv weight (256,256,k) -> the DMA kernel's K-tile-major H2 rows; wq: 256*k*256 + 64 floats (max|w| behind the tiles) */
int cpc_conv_weight_relayout_h2(const float* w, float* wq, int k, void* stream);
/* One conv layer (k = 2s) + bias + ChannelNorm + ReLU on the DMA kernel (conv_dma.hip, one launch): x in H2 storage scaled
 * by scale(*x_amax); y in H2 storage scaled by scale(*y_amax) (which must bound |y|), or fp32 when y_amax == NULL; xhat,
 * rstd fp32.  zeros: 32 floats of zeros (padding rows).  bm: rows per workgroup, 128 / 256 (0: by problem size). */
int cpc_conv_gemm_forward_h2(const void* x_h2, const float* wq, const float* bias, const float* nw, const float* nb,
                             void* y, float* xhat, float* rstd, const float* x_amax, const float* y_amax,
                             const float* zeros, int B, int Lin, int k, int s, int p, int bm, void* stream);
/* tuning knobs of the DMA kernel inside the composite encoder: rows per workgroup (0 / 128 / 256) and the K-walk
 * rotation step between neighbouring workgroups (0: lockstep) */
int cpc_set_conv0_tuning(int groups, int nontemporal);   /* layer-0 forward kernel: 16-step groups per wave (1..8, default 4); 1 / 0 (default):
                                                          * activation rows as non-temporal / plain stores */
int cpc_set_dma_tile(int bm);
int cpc_set_h2_layers(int n);            /* mode 3, which layers read their input (and write their gradient) in H2 storage: 1 = conv1, 2 = conv1 and
                                            conv2, both on the DMA kernels; 4 = all four -- conv1 (conv2 from B ~ 100 on) on the DMA kernels, the
                                            short layers on the register-staged tiles, which stage H2 rows as they lie, and every weight
                                            gradient on the DMA + transposing-read kernel with one batched reduction; 0 = by problem size */
int cpc_set_dma_layer2(int on);          /* 1: layer 2 (forward and data gradient) on the DMA-fed kernels whatever the batch size; 0 (default): from B ~ 100 on */
int cpc_set_conv_small_tile(int bm);     /* rows per workgroup of the register-staged conv tiles below 32000 rows: 32 (default) or 64 */
int cpc_set_dgrad_nsplit(int min_wgs);    /* > 0: the H2-fed data gradient of layers below the 128-row regime runs on 128 x 128 tiles (two workgroups
                                            per row tile, 128 input channels each) where that still gives min_wgs workgroups; 0: 32-row tiles */
int cpc_set_conv_small_pipe(int on);     /* 1 (default): the 32- / 64-row tiles of the H2-fed register-staged conv kernels (cpc_set_h2_layers(4)) run the software-
                                            pipelined 16-k schedule of the 128-row tiles (four chunks of global loads in flight); 0: one 32-k stage */
int cpc_set_wgrad_dma_groups(int wgs);  /* workgroups the DMA weight gradient aims at (row splits = wgs / taps); 64..512 */
int cpc_set_wgrad_dma_stages(int n);     /* its LDS pipeline: 2 stages of 32 contraction rows (one in flight) or 4 (default) of 16 (three in flight) */
int cpc_set_wgrad_dma_min_rows(int rows); /* ... and the fewest rows one of its splits walks (default 512; 128..8192, a multiple of 64) */
int cpc_set_wgrad1_early(int on);        /* two-stream encoder backward: 0 (default) layer 1's weight gradient behind its data gradient, 1 beside it */
int cpc_set_h2_dx(int on);               /* mode 3: 1 (default) keeps the gradient dx of every layer whose input is in H2 storage in H2 storage
                                            too (layer 1; layer 2 where conv2 reads H2 input): data gradient on the DMA kernel, weight
                                            gradient on DMA + transposing LDS reads; 2: the same with the weight gradient on the
                                            register-staged tile (A/B); 0: fp32 dx */
int cpc_set_gemm_split(int on);          /* 1 (default): plain GEMMs with known operand bounds (the criterion's, see cpc_nce_forward) run on two
                                            fp16 pieces in mode >= 2, wide products on the 128 x 256 pipelined tile; 0: three bf16 pieces always;
                                            2: two pieces but never the wide tile; 3: the wide tile whatever the grid size (tests) */
int cpc_set_gemm_fuse(int on);           /* 1 (default): the transformer layer's feed-forward ReLU (forward, dropout 0) and ReLU derivative
                                            (backward) as epilogues of their GEMMs on the wide fp16-piece tile; 0: elementwise kernels
                                            behind the GEMMs (same bits) */
int cpc_set_dma_rotation(int step);
int cpc_set_dma_pipeline(int variant);   /* main-loop schedule of the forward DMA kernel on 256-row tiles (tuning / measurement switch; results
                                          * agree to rounding order).  2 (default): tap-pair walk -- an input row reaches LDS once for both
                                          * taps that read it -- where k == 2*stride and Lout % 256 == 0, variant 1 elsewhere; 1: two 32-k
                                          * LDS stages; 0: four 16-k stages, three in flight; 3, 4, 5: ping-pong slots (one wave of a SIMD
                                          * multiplies while its partner loads) with 0 / 2 / 4 DMA pieces issued among the MFMAs; 6: skewed
                                          * slots.  What each measured: DESIGN.md section 4.10 */
long cpc_conv0_backward_scratch_floats(int B, int L);
/* Backward of layer 0 (no dgrad: the waveform needs no gradient, train.py:81-87).
 * dy = gradient w.r.t. y.  Outputs dW0 (256,1,10), dB0, dNW0, dNB0 (256) are overwritten. */
int cpc_conv0_backward(const float* wave, const float* w, const float* bias, const float* nw,
                       const float* nb, const float* mean, const float* rstd, const float* dy,
                       float* scratch, float* dW0, float* dB0, float* dNW0, float* dNB0, int B,
                       int L, void* stream);

/* Layers 1..4 (256->256, k = 2s; model.py:85-92,101-104): x (B,Lin,256) ->
 * y = relu(norm(conv)) and xhat (pre-affine normalised), both (B,Lout,256); rstd (B*Lout).
 * w is PyTorch (O,I,W); wp is 256*k*256*3/2 floats of scratch (re-laid-out weight, see below). */
int cpc_conv_layer_forward(const float* x, const float* w, const float* bias, const float* nw,
                           const float* nb, float* wp, float* y, float* xhat, float* rstd, int B,
                           int Lin, int k, int s, int p, void* stream);
/* Weight re-layout (O,I,W) -> K-major rows wp[co][kk*256+ci]; three bf16 planes in the default
 * split-bf16 mode.  wp: 256*k*256*3/2 floats. */
int cpc_conv_weight_relayout(const float* w, float* wp, int k, void* stream);
/* The forward GEMM kernel alone on a weight prepared by cpc_conv_weight_relayout (one launch). */
/* x_amax: device float holding an upper bound of max|x| (cpc_absmax, or the bound of the producing ChannelNorm);
 * read in mode 2 only, may be NULL otherwise. */
int cpc_conv_gemm_forward(const float* x, const float* wp, const float* bias, const float* nw,
                          const float* nb, float* y, float* xhat, float* rstd, const float* x_amax, int B,
                          int Lin, int k, int s, int p, void* stream);
/* *out = max(*out, max_i |x_i|), exact and order-independent (*out must be initialised, e.g. 0). */
int cpc_absmax(const float* x, long n, float* out, void* stream);
/* ReLU' + ChannelNorm backward over M rows; small3 = [d norm.w | d norm.b | d conv.bias]. */
/* dx_amax (may be NULL): device float that receives max(*dx_amax, max|dx|) (initialise to 0); it is the operand
 * bound the mode-2 dgrad / wgrad of this layer read. */
int cpc_norm_backward(const float* dy, const float* xhat, const float* y, const float* rstd,
                      const float* nw, float* dx, float* colpart, float* tmp, float* small3,
                      float* dx_amax, int M, void* stream);
/* wd: 256*k*256*3/2 floats of scratch (per-phase re-laid-out weight). */
/* dx_amax: device float bounding max|dx| (mode 2; NULL = computed with an extra pass over dx);
 * dprev_amax (fuse only, may be NULL): receives max(*dprev_amax, max|dprev|). */
int cpc_conv_layer_dgrad(const float* dx, const float* w, float* wd, int fuse,
                         const float* xhat_prev, const float* y_prev, const float* rstd_prev,
                         const float* nw_prev, float* dprev, float* colpart, float* tmp,
                         float* small3, const float* dx_amax, float* dprev_amax, int B, int Lin, int k,
                         int s, int p, void* stream);
/* dx_amax, x_amax: device floats bounding max|dx| and max|x| (read in mode 2 only). */
int cpc_conv_layer_wgrad(const float* dx, const float* x, float* part, float* dW, const float* dx_amax,
                         const float* x_amax, int B, int Lin, int k, int s, int p, int splits,
                         int rows_per_split, void* stream);

/* 0: one launch per time step; 1: the two-layer recurrence runs as one persistent launch whenever all of its
 * workgroups can be resident at once (gru.hip) -- bit-identical with 0; 2 (default): as 1, with the forward's
 * recurrent products on the fp16 matrix pipe (operands split into two fp16 pieces, three MFMAs per product, fp32
 * accumulate: differences at the 1e-7 level; exact-f32 products whenever the caller supplies h0). */
int cpc_set_gru_mode(int mode);
/* Persistent recurrence: 1 places the 32 workgroups of each 16-sequence batch tile on one XCD (grid of 256 * ceil(tiles / 8)
 * workgroups, workgroup b on XCD b % 8), so that the step-to-step hand-over stays inside one L2; 0 (default, measured
 * faster) interleaves the tiles over the XCDs (grid of 32 * tiles); 2 forces the packed numbering whatever the device
 * reports (tests). */
int cpc_set_gru_xcd_pack(int on);
int cpc_set_gru_xcd_local(int mask);    /* bit 0 forward, bit 1 backward: one batch tile per XCD, hand-over through that XCD's L2 (gru.hip) */
int cpc_set_gru_poll_plain(int mask);   /* first looks of the persistent recurrence through the XCD's L2 (gru.hip); default 0 */
/* Persistent recurrence: cap on the 16-sequence batch tiles of one launch (0 = whatever fits the device, the default); a
 * larger batch runs as several launches one after the other (B = 256 on 256 CUs: two launches of 8 tiles). */
int cpc_set_gru_chunk_tiles(int tiles);
/* Batch tiles a workgroup of the persistent recurrence may own: 2 (default) = a batch that does not fit the device in one launch
 * (B = 256 per GPU, BASELINE configs[2], on 256 CUs) runs as ONE launch, each workgroup interleaving its two tiles step by step
 * (cpc/model.py:193's nn.GRU over the whole batch); 1 = serial launches over chunks of tiles.  Same bits either way. */
int cpc_set_gru_tiles_per_wg(int n);
/* Persistent recurrence: the wait before a step's first look at the hand-over buffers (forward / backward kernel), in
 * units of 64 clocks; < 0 (default): every wave steers its own so that looks that cannot succeed yet are not issued
 * (they load the L2s the hand-over itself goes through). */
int cpc_set_gru_poll_pacing(int first_fwd, int first_bwd);
/* Polling budget of one wave of the persistent recurrence (re-reads over the whole launch) before it gives up and
 * flags CPC_DEVERR_GRU_POLL_TIMEOUT; limit < 0 restores the default (2^20).  Tests use 0 to drive the error path. */
int cpc_set_gru_spin_limit(int limit);

/* Tuning / test knob: rows per block of the conv GEMM tiles (0 = auto, 32, 64, 128). */
int cpc_set_conv_tile(int bm);

/* Whole encoder.  params / grads: 20 pointers in the reference's state-dict order
 * (conv{i}.weight, conv{i}.bias, batchNorm{i}.weight, batchNorm{i}.bias, i = 0..4).
 * cpc_encoder_layout fills sizes[0..21]: [0] saved floats, [1] fwd scratch floats,
 * [2] bwd scratch floats, [3..7] L0..L4, then offsets into `saved` (see enc_conv.hip). */
int cpc_encoder_layout(int B, int L, long* sizes);
/* fp32 copy of the saved output of layer `layer` (0..3) of a cpc_encoder_forward, whatever its storage: dst (B,L_layer,256).
 * Tests / debugging; call it in the mode the forward ran in. */
int cpc_encoder_saved_activation(const float* saved, int layer, float* dst, int B, int L, void* stream);
int cpc_encoder_forward(const float* wave, const float* const* params, float* saved,
                        float* scratch, float* z, int B, int L, void* stream);
/* The weight-only preparation cpc_encoder_forward starts with (GEMM layouts and max|w| of conv1..4, bounds of the layers'
 * inputs from the ChannelNorm affines; what cpc/model.py:83-92's nn.Conv1d does inside MIOpen, hoisted), alone, for a caller
 * that runs it ahead of the forward: mask bit i (1..4) = layer i, bit 0 = the four bounds.  Used by cpc_train_step_tail. */
int cpc_encoder_prepare_weights(const float* const* params, float* saved, float* scratch, int B, int L, int mask, void* stream);
int cpc_encoder_backward(const float* wave, const float* const* params, const float* saved,
                         const float* z, const float* dz, float* scratch, float* const* grads,
                         int B, int L, void* stream);

/* The same with the four conv weight-gradient GEMMs on `wgrad_stream` (they are not on the dx chain); ordering
 * between the two streams is internal (hip events, no host synchronisation) and `stream` waits for `wgrad_stream`
 * before the call returns control of the outputs.  wgrad_stream == stream: identical to cpc_encoder_backward. */
int cpc_encoder_backward_streams(const float* wave, const float* const* params, const float* saved,
                                 const float* z, const float* dz, float* scratch, float* const* grads, int B,
                                 int L, void* stream, void* wgrad_stream);

/* ------------------------------------------------------------ plain GEMMs ----
 * C[M,N] = A[M,K] . B[N,K]^T + bias[N]   (N % 128 == 0, K % 16 == 0; bias may be NULL).
 * This is torch.nn.Linear's arithmetic (criterion.py:90-91,108; the GRU projections). */
int cpc_gemm_nt(const float* A, int lda, const float* B, int ldb, const float* bias, float* C,
                int ldc, int M, int N, int K, void* stream);
/* C[N1,N2] (+)= A[M,N1]^T . B[M,N2]   (N1, N2 % 128 == 0): every weight gradient. */
long cpc_gemm_tn_scratch_floats(int M, int N1, int N2);
int cpc_gemm_tn(const float* A, int lda, const float* B, int ldb, float* part, float* C, int M,
                int N1, int N2, int accumulate, void* stream);

/* ------------------------------------------------------------ autoregressor ----
 * CPCAR.forward, cpc/model.py:185-204: nn.GRU(256, 256, num_layers=nl, batch_first=True)
 * (model.py:175-176) with optional initial state h0 (the carried `self.hidden`, :193-198).
 * params / grads: weight_ih_l, weight_hh_l, bias_ih_l, bias_hh_l for l = 0..nl-1
 * (torch state-dict order, gate rows r,z,n).  x, y, dy, dx: (B,S,256); h0, hN: (nl,B,256).
 * cpc_gru_layout fills sizes[0..2] = saved / forward-scratch / backward-scratch floats. */
int cpc_gru_layout(int B, int S, int nl, long* sizes);
int cpc_gru_forward(const float* x, const float* h0, const float* const* params, float* saved,
                    float* scratch, float* y, float* hN, int B, int S, int nl, void* stream);
int cpc_gru_backward(const float* x, const float* h0, const float* const* params,
                     const float* saved, const float* y, const float* dy, float* scratch, float* dx,
                     float* const* grads, int B, int S, int nl, void* stream);

/* The part of the two-layer backward that depends on the forward pass only (per-step gate-derivative coefficients
 * and the pre-filled hand-over buffers of the persistent launch: cpc_gru_coef_floats(B,S,nl) floats; 0 unless
 * nl == 2; one buffer serves one backward call).  cpc_gru_backward_coef may run any time after the forward
 * on any stream; cpc_gru_backward_with_coef(coef != NULL) then skips that work (coef == NULL: same as
 * cpc_gru_backward). */
long cpc_gru_coef_floats(int B, int S, int nl);
int cpc_gru_backward_coef(const float* h0, const float* const* params, const float* saved, const float* y, float* coef,
                          int coef_done, int B, int S, int nl, void* stream);
/* cpc_gru_forward that also fills the coefficient arrays of `coef` (cpc_gru_coef_floats floats) on the way: the persistent
 * forward's gate threads hold every input of those coefficients in registers.  Follow with cpc_gru_backward_coef(coef_done = 1),
 * which then only prepares the hand-over buffers and the transposed weights. */
int cpc_gru_forward_coef(const float* x, const float* h0, const float* const* params, float* saved, float* scratch, float* y,
                         float* hN, float* coef, int B, int S, int nl, void* stream);
int cpc_gru_backward_with_coef(const float* x, const float* h0, const float* const* params, const float* saved,
                               const float* y, const float* dy, const float* coef, float* scratch, float* dx,
                               float* const* grads, int B, int S, int nl, void* stream);

/* cpc_gru_forward_coef split for callers that keep launches off their critical stream: cpc_gru_forward_prepare fills the
 * persistent recurrence's hand-over buffers in `scratch` (the forward's only activation-independent launch) on any stream;
 * cpc_gru_forward_coef_prepared, on a stream that has waited for it, skips that fill.  nl == 2. */
int cpc_gru_forward_prepare(float* scratch, int B, int S, int nl, void* stream);
int cpc_gru_forward_coef_prepared(const float* x, const float* h0, const float* const* params, float* saved, float* scratch,
                                  float* y, float* hN, float* coef, int B, int S, int nl, void* stream);

/* As cpc_gru_backward_with_coef, with the weight and bias gradients (what only the optimiser reads) on
 * `wgrad_stream`, released by an event behind the recurrence; dx stays on `stream`.  There is NO join: the caller
 * orders every consumer of `grads` after `wgrad_stream` and keeps `scratch` alive until then.  (nl == 2; other depths
 * run on `stream` alone.) */
int cpc_gru_backward_streams(const float* x, const float* h0, const float* const* params, const float* saved,
                             const float* y, const float* dy, const float* coef, float* scratch, float* dx,
                             float* const* grads, int B, int S, int nl, void* stream, void* wgrad_stream);

/* ---------------------------------------------------------------- transformer layer ----
 * One TransformerLayer of cpc/transformers.py:103-111 (buildTransformerAR, :130-139), d_model 256, 8 heads,
 * d_ff 2048, sequence S <= 128, dropout not applied.  Used as the auto-regressive network (--arMode
 * transformer, cpc/feature_loader.py:138-141, S = 128) and as a prediction network (--rnnMode transformer,
 * cpc/criterion/criterion.py:82-88, S = 128 - K).  BASELINE.json config 4.
 * params / grads: 13 pointers in the reference's state-dict order -- multihead.Wo, Wk, Wq, Wv .weight (256,256),
 * multihead.Att.Krelpos (32,S) (NULL = abspos layer without the relative term), ln_multihead.weight, .bias,
 * ffnetwork.lin1.weight (2048,256), .bias, ffnetwork.lin2.weight (256,2048), .bias, ln_ffnetwork.weight, .bias.
 * sizes[0] = saved floats, [1] = forward scratch floats, [2] = backward scratch floats, [3..7] = offsets inside
 * `saved` of qkv (B*S,768), A (B*8,S,S), o, y (B*S,256), hid (B*S,2048) = relu(lin1(y)).
 * 128 < S <= 512 (a layer built for a longer window -- the 400 frames of a 64000-sample feature-extraction chunk,
 * cpc/feature_loader.py:247-266): FORWARD ONLY, without dropout -- attention as a running-softmax walk over 128-key blocks
 * (attn_fwd_long_kernel); the attention probabilities A are not kept (size 0), the backward entry points refuse such S. */
int cpc_transformer_layout(int B, int S, long* sizes);
int cpc_transformer_layer_forward(const float* x, const float* const* params, float* saved, float* scratch,
                                  float* out, int B, int S, void* stream);
int cpc_transformer_layer_backward(const float* x, const float* const* params, const float* saved,
                                   const float* dy, float* scratch, float* dx, float* const* grads,
                                   int B, int S, void* stream);
/* Training mode with dropout probability p (0 <= p < 1) on the attention probabilities and on the feed-forward hidden
 * layer (cpc/transformers.py:18,50,93,100; the reference hard-codes 0.1).  Masks are a pure function of `seed` and the
 * element index (Philox4x32-10): give the backward call the seed of its forward call.  p = 0 is the plain layer. */
int cpc_transformer_layer_forward_dropout(const float* x, const float* const* params, float* saved, float* scratch,
                                          float* out, int B, int S, float p, unsigned long long seed, void* stream);
int cpc_transformer_layer_backward_dropout(const float* x, const float* const* params, const float* saved,
                                           const float* dy, float* scratch, float* dx, float* const* grads, int B,
                                           int S, float p, unsigned long long seed, void* stream);
/* the hidden layer (B*S, 2048) of the forward call that filled `saved`, as fp32 whatever its storage (the DMA-fed feed-forward
 * GEMMs keep it as two fp16 pieces per element); tests / inspection; synchronises `stream` */
int cpc_transformer_hidden(const float* saved, float* out, int B, int S, void* stream);
/* feed-forward GEMMs of the transformer layer (cpc/transformers.py:86-101) on the DMA-fed tiles: 0 off, 1 (default) where a
 * call's launches fill the chip (the K predictors as a group), 2 always; + 4: without the 128-row tail launch of the one-tile-wide
 * products, + 8: ReLU / dropout as a pass behind lin1 instead of in its epilogue (A/B); must not change between a forward and its backward */
int cpc_set_gemm_dma(int mode);
/* wave tile of the DMA-fed plain NT products' 256-row tiles: 64 = eight waves of 64 x 128 (two per SIMD), 128 = four waves of
 * 128 x 128 (one per SIMD, accumulators in AGPRs, a four-stage 16-k loop: csrc/dma_tile.h); same products in the same order: same bits */
int cpc_set_dma_wave_rows(int rows);
int cpc_set_gemm_tail_cus(int cus);     /* CU count the DMA-fed NT products' tail split plans for (0 = the device's; tests) */
int cpc_set_attn_fwd(int variant);       /* forward attention kernel (S <= 128): 1 = two workgroups per CU (default), 0 = one; same bits */
/* G transformer layers of one shape on ONE input x (B,S,256), every kernel launched once for all of them: the K predictors
 * of the criterion in --rnnMode transformer (cpc/criterion/criterion.py:82-88, :97-118).  params[i] / grads[i]: the G
 * tensors of kind i stacked, layer g at + g * numel; saved / scratch: G workspaces of cpc_transformer_layout's sizes back
 * to back; out / dy: (B*S, G*256) with layer g at columns g*256..; dx (B,S,256): the SUM of the layers' input gradients.
 * Dropout masks of layer g: those of a single-layer call with seed + g. */
int cpc_transformer_group_forward(const float* x, const float* const* params, float* saved, float* scratch, float* out,
                                  int B, int S, int G, float p, unsigned long long seed, void* stream);
int cpc_transformer_group_backward(const float* x, const float* const* params, const float* saved, const float* dy,
                                   float* scratch, float* dx, float* const* grads, int B, int S, int G, float p,
                                   unsigned long long seed, void* stream);
/* Test helper: out[i] = keep_i / (1 - p) of dropout site 0 (attention probabilities, flat ((b*8 + head)*S + i)*S + j; n a
 * multiple of S*S) or 1 (hidden layer, flat row*2048 + col; S ignored) under `seed`. */
int cpc_dropout_keep_mask(float* out, long n, int site, int S, float p, unsigned long long seed, void* stream);

/* ---------------------------------------------------------------- criterion ----
 * CPCUnsupersivedCriterion.forward (criterion.py:225-257) with linear prediction heads
 * (PredictionNetwork, criterion.py:90-91,97-118) and the negatives of sampleClean
 * (criterion.py:174-219) supplied as row indices.
 *   c, z : (B,S,256) context / encoded features;  W = S - K windows per sequence
 *   wall : (K*256, 256) the K head weights `wPrediction.predictors.k.weight` stacked
 *   ext  : (B*W, N) int32, ext[(b*W+t)*N + n] = row of z.view(B*S,256) used as negative n
 *          of window (b,t)  ==  criterion.py:199's extIdx[b,n,t]
 *   losses, acc : K floats (criterion.py:256-257)
 * K <= 16 per call (cpc_nce_head_group walks more), any N > 0.  sizes[0..2] = saved / fwd-scratch / bwd-scratch floats,
 * sizes[3..5] = offsets of pred, logits (B*W,K,1+N), lse (B*W,K) inside `saved`. */
int cpc_nce_layout(int B, int S, int K, int N, long* sizes);
/* Negatives per window as the kernels lay them out: N (criterion.py:176-189 draws any number) rounded up to the 16-wide MFMA
 * tile.  The padding candidates are valid rows whose logits the scoring kernels force to -3e38: weight 0 in the softmax, the
 * arg-max and every gradient.  ext is (B*W, padded) int32, logits (B*W, K, 1 + padded), perm / work count padded + K candidates
 * per window; the draws (batchIdx / seqIdx) stay B*N*W. */
int cpc_nce_padded_negatives(int N);
/* More than 16 prediction steps (criterion.py:225-257 takes any nPredicts): the score tiles hold 16 heads, so such a criterion
 * is walked in groups of at most 16.  After cpc_nce_head_group(k0, k_total) every cpc_nce_* call of the CALLING THREAD works on
 * heads k0 .. k0+K-1 of a criterion with k_total steps: W = S - k_total windows per sequence, head k's positive is
 * z[b, t + k0 + k + 1]; `wall` / `pred` / losses / acc are the group's K heads.  Per-head results are independent, dc / dz of the
 * groups add.  (0, 0) ends it; every call of a group -- layout, prepare, forward, backward -- is made under the same setting. */
int cpc_nce_head_group(int k0, int k_total);
/* Index preparation: the two int64 draws of sampleClean (criterion.py:181-189; B*N*W each, flat (b,n,t)
 * order) -> ext (the rows of criterion.py:191-199, laid out (b,t,n) with the N rows of a window in ASCENDING order: the
 * criterion is invariant under a permutation of a window's negatives, and sorted lists keep the gathers of the scoring
 * kernels inside L2) and the destination-sorted candidate slots (perm, row_ptr) the backward uses.
 * work: B*W*(N+K) + 2*B*S + 2 ints. */
int cpc_nce_prepare(const long* batchIdx, const long* seqIdx, int* ext, int* perm, int* row_ptr, int* work,
                    int B, int S, int K, int N, void* stream);
int cpc_nce_forward(const float* c, const float* z, const float* wall, const int* ext, float* saved,
                    float* scratch, float* losses, float* acc, int B, int S, int K, int N,
                    void* stream);
/* The operand bounds of the criterion's GEMMs (max|c|, max|wall|, kept in `saved`) ahead of time, on any stream: max|wall| is
 * fixed once the optimiser has stepped and a recurrent context is bounded a priori (c_bound > 0: |c| <= c_bound, e.g. 1 for a
 * GRU; otherwise c is reduced here), so the reduction need not sit between the autoregressive network and the prediction
 * GEMM.  cpc_nce_forward_prepared is cpc_nce_forward on a `saved` whose bounds were written by cpc_nce_bounds. */
int cpc_nce_bounds(const float* c, float c_bound, const float* wall, float* saved, int B, int S, int K, int N, void* stream);
int cpc_nce_forward_prepared(const float* c, const float* z, const float* wall, const int* ext, float* saved,
                             float* scratch, float* losses, float* acc, int B, int S, int K, int N, void* stream);
/* cpc_nce_forward (bounds_ready == 0) / cpc_nce_forward_prepared (!= 0) with the loss / accuracy reduction on finalize_stream,
 * ordered behind the scoring kernel by an event: the backward reads the saved logits, never the losses, so a caller that joins
 * finalize_stream before reading losses / acc keeps that launch off its critical stream. */
int cpc_nce_forward_streams(const float* c, const float* z, const float* wall, const int* ext, float* saved, float* scratch,
                            float* losses, float* acc, int B, int S, int K, int N, int bounds_ready, void* stream,
                            void* finalize_stream);
/* The share of the backward that depends on the weights, the operand bounds in `saved` (cpc_nce_bounds) and gloss only --
 * gradient scales, GEMM operand bounds, cleared maximum slots, the zeroed tail rows of dc, wall^T -- ahead of time on any stream
 * that has seen cpc_nce_bounds; cpc_nce_backward_prepared, on a stream that has waited for it, is then
 * cpc_nce_backward_streams(dz = NULL, dwall = NULL) without those launches (cpc_nce_backward_dz / _dwall follow as usual). */
int cpc_nce_backward_prepare(const float* wall, const float* saved, const float* gloss, float* scratch, float* dc, int B, int S,
                             int K, int N, void* stream);
int cpc_nce_backward_prepared(const float* c, const float* z, const float* wall, const int* ext, const int* perm,
                              const int* row_ptr, const float* saved, const float* gloss, float* scratch, float* dc, int B,
                              int S, int K, int N, void* stream);
/* gloss: K upstream gradients dL/dloss_k.  dc, dz (B,S,256) and dwall are overwritten.
 * perm (B*W*(N+K)) / row_ptr (B*S+1): candidate slots sorted by destination row of z (slot =
 * (b*W+t)*(N+K)+j; j<N: negative j -> row ext[..j]; j>=N: positive of head j-N -> row b*S+t+j-N+1);
 * they turn the gradient scatter into an atomics-free, reproducible per-row reduction.  With linear heads
 * (pred_k = W_k c) dz is re-associated through c:  dz[j] = sum_k W_k . G[j,k,:],  G[j,k,:] = sum over the slots landing
 * on row j of dS[slot,k] * c[window(slot)]  -- a per-row gather-GEMM over 1 KB rows of c, then one dense GEMM with the
 * stacked heads; no per-candidate gradient rows are materialised (criterion.py:108-116,199-217 backward). */
int cpc_nce_backward(const float* c, const float* z, const float* wall, const int* ext,
                     const int* perm, const int* row_ptr, const float* saved, const float* gloss,
                     float* scratch, float* dc, float* dz, float* dwall, int B, int S, int K, int N,
                     void* stream);

/* As cpc_nce_backward with the dz path (Wcat + gather-GEMM G + dz GEMM; it depends on the score gradients this call
 * writes first, not on dpred / dc / dwall) launched on dz_stream behind an event, so that it can run beside the
 * auto-regressive network's backward; dz == NULL leaves the dz path out (run it later with cpc_nce_backward_dz, same
 * scratch, on a stream that waits for this call).  Every consumer of dz must wait for dz_stream. */
int cpc_nce_backward_streams(const float* c, const float* z, const float* wall, const int* ext,
                             const int* perm, const int* row_ptr, const float* saved, const float* gloss,
                             float* scratch, float* dc, float* dz, float* dwall, int B, int S, int K, int N,
                             void* stream, void* dz_stream);

int cpc_nce_backward_dz(const float* c, const float* wall, const int* perm, const int* row_ptr, const float* saved,
                        float* scratch, float* dz, int B, int S, int K, int N, void* stream);

/* cpc_nce_backward_streams also accepts dwall == NULL: the head-weight gradient (criterion.py:44-50, the K
 * nn.Linear weights) is then left out and formed later by this call from the dPred kept in `scratch`; nothing on
 * the way to the encoder depends on it.  `stream` must wait for the cpc_nce_backward_streams call. */
int cpc_nce_backward_dwall(const float* c, const float* saved, float* scratch, float* dwall, int B, int S, int K, int N,
                           void* stream);
/* The linear heads' criterion in ONE gather pass (1; default since round 6: 2, below): the scoring kernel of cpc_nce_forward* carries the softmax-weighted
 * sum of the candidate rows along with the log-sum-exp (an online softmax over criterion.py:108-116's candidates) and leaves
 * T = d loss_k / d pred_k for a unit upstream gradient in `saved`; the backward then has no score-gradient pass over the 1 KB
 * candidate rows (criterion.py:200-201's gather, 1.06 GB at B = 64) -- the heads' upstream gradients are folded into the dc
 * GEMM's weight operand and the weight gradient's reduction, the dz path multiplies the softmax rows the forward leaves per
 * candidate slot.  0: the two-pass kernels.  A forward and its backward run under the same setting. */
int cpc_set_nce_fused(int on);
/* (on = 2, round 6: the same one-pass criterion with the scoring kernel on the 16-bit matrix pipe -- both products as hh + hl + lh
 * of two fp16 pieces, the arithmetic of the conv layers -- gathering from an H2 copy of z that goes global -> LDS by DMA, the
 * contraction over candidates fed by the transposing LDS read; on = 3: the dz path's gather-GEMM likewise, from an H2 copy of c.
 * Same interface, same saved tensors.) */
int cpc_get_nce_fused(void);
/* cpc_set_nce_fused(2 / 3): workgroups of the scoring kernel -- 0 = one per four windows, -1 (default) = two per CU, n > 0 = at most n (a capped
 * grid walks its windows with the grid's stride, so that all resident waves sweep their ascending candidate lists in step). */
int cpc_set_nce_grid(int wgs);
/* cpc_set_nce_fused(2 / 3): 1 (default) = the softmax rows the dz path reads are written by a launch of their own, on the stream
 * the loss reduction runs on (cpc_nce_forward_streams' finalize_stream: off the forward's chain), 0 = inside the scoring kernel. */
int cpc_set_nce_rows_apart(int on);
/* The prediction product pred = c . wall^T of the K linear heads (cpc/criterion/criterion.py:108-116): 1 (default) = on the DMA-fed
 * 256 x 256 tile of csrc/gemm_dma.hip (c as an H2 copy, the stacked weights re-laid beside the operand bounds), 0 = on the generic
 * register-staged tile.  Same fp16-piece arithmetic, another order of summation. */
int cpc_set_nce_heads_dma(int on);
/* Measurement switch of the cpc_set_nce_fused(2) scoring kernel (tools/time_nce.py: what each part of it costs): bit 0 leaves the
 * softmax-row pass out, 1 the T epilogue, 2 the weighted row sum, 3 the logits stores.  Results are WRONG while mask != 0. */
int cpc_set_nce_debug(int mask);
/* cpc_set_nce_fused(2 / 3): the H2 copy of z (criterion.py:200-201's gather source) ahead of time on `stream` -- z is final when
 * the encoder has run; the calling thread's next cpc_nce_forward* then skips the two small launches.  No-op in other modes. */
int cpc_nce_prepare_z(const float* z, float* saved, int B, int S, int K, int N, void* stream);
/* Tuning switch: at most n workgroups per launch of cpc_nce_prepare's kernels (each then walks several windows / slots);
 * -1 (default) = the device's CU count, 0 = one per 4 windows / 256 slots.  Same lists either way. */
int cpc_set_index_prep_groups(int n);
/* Tuning switch of the composite step (single-rank calls, phases 3): 1 (default) = the recurrence's weight / bias gradients run on
 * the preparation stream, which is idle during the backward, instead of in front of the conv layers' on the weight-gradient
 * stream (0).  Same kernels, same values. */
int cpc_set_gru_wgrad_stream(int on_prep);

/* The same criterion for predictions formed by the caller -- any prediction network of
 * cpc/criterion/criterion.py:44-118, e.g. K transformer layers (--rnnMode transformer, :82-88):
 * pred (B*W, K*256), row (b,t), head k at columns k*256..; implements :115-116 (mean over 256 of pred * candidate)
 * and :245-257.  Layout / scratch sizes as cpc_nce_layout (the `pred` part of `saved` stays unused).
 * backward overwrites dpred (B*W, K*256) and dz (B,S,256). */
int cpc_nce_scores_forward(const float* pred, const float* z, const int* ext, float* saved, float* scratch,
                           float* losses, float* acc, int B, int S, int K, int N, void* stream);
int cpc_nce_scores_backward(const float* pred, const float* z, const int* ext, const int* perm,
                            const int* row_ptr, const float* saved, const float* gloss, float* scratch,
                            float* dpred, float* dz, int B, int S, int K, int N, void* stream);

/* ---- the whole step ---------------------------------------------------------------------------------------------------
 * cpc/train.py:78-87 for the north-star configuration (CPCEncoder + 2-layer GRU CPCAR + K linear InfoNCE heads): model
 * forward, criterion forward, allLosses.sum().backward() -- every launch of the entry points above, issued from ONE call on
 * four caller-owned streams with the cross-stream order of the package's Python train loop (ops.py / train.Trainer), i.e. the
 * same kernels in the same order: results are bit-identical to that loop.  What it saves is host time (~60 launches, three
 * autograd nodes and a dozen allocator calls per step from Python), which is what an eager data-parallel rank is bound by.
 *   wave (B,1,L); batchIdx, seqIdx: the two int64 draws of sampleClean (criterion.py:181-189), B*N*W each, ready on
 *   side_stream (or earlier on main_stream); h0: NULL or the carried GRU state (2,B,256), hN (2,B,256) receives the final one
 *   (cpc/model.py:193-198); c_bound > 0: an a-priori bound of |c| (1 for a GRU started from zero or from one of its own final
 *   states), <= 0: max|c| is reduced in line; params / grads: 29 pointers -- the 20 encoder tensors (cpc_encoder_forward's
 *   order), the 8 GRU tensors (cpc_gru_forward's order), the K head weights stacked (K*256, 256); grads are OVERWRITTEN
 *   (= zero_grad + backward); gloss: K floats dL/dloss_k (ones for train.py:85's sum); workspace: cpc_train_step_layout
 *   sizes[0] floats, reused from step to step (z and c of the step live in it at sizes[1], sizes[2] until the next call);
 *   losses, acc: K floats each (criterion.py:256-257).
 * phases (bit mask): 1 = forward + backward down to the encoder's input gradient -- on return the heads' gradient is queued
 * on side_stream, the recurrence's on wgrad_stream, everything else of the non-encoder gradients is final on main_stream (a
 * data-parallel caller starts reducing that bucket here); 2 = the encoder's backward, after which main_stream has waited for
 * side_stream and wgrad_stream: every gradient is final on main_stream.  3 = both.  No host synchronisation anywhere.
 * Cross-step pipelining of a single-rank loop (cpc/train.py:78-91 is strictly serial: backward, optimizer.step, next forward):
 *   + 4 (with 2) open tail: main_stream does not wait for the step's LAST kernel, layer 1's weight gradient on wgrad_stream
 *     (0.13 ms past the end of main_stream's chain), and the bias / norm gradient sums of layers 1..4 run on prep_stream beside
 *     the chain's last kernels.  The caller then updates conv0's parameters on main_stream, conv1.weight on wgrad_stream and the
 *     rest on prep_stream (cpc_adam_step three times, cpc_train_step_wait for their gradients), calls cpc_train_step_tail, and gives
 *     the NEXT step + 8 (its weight layouts are ready; main_stream waits for conv1's in front of layer 1) and alternates + 16
 *     (second y0 buffer / bound set of the workspace: the next layer 0 runs while layer 1's weight gradient still reads y0).
 *   A caller that touches parameters or gradients outside these calls first joins with cpc_train_step_wait(main, 2, main). */
int cpc_train_step_layout(int B, int L, int K, int N, long* sizes);
int cpc_train_step(const float* wave, const long* batchIdx, const long* seqIdx, const float* h0, float c_bound,
                   const float* const* params, float* const* grads, const float* gloss, float* workspace, float* losses,
                   float* acc, float* hN, int B, int L, int K, int N, int phases, void* main_stream, void* side_stream,
                   void* prep_stream, void* wgrad_stream);
/* The index lists of the NEXT step's draws, one step ahead (optional): queue it on side_stream after the current step's
 * cpc_train_step, with the next step's draws ready there; the next cpc_train_step -- same sizes and workspace -- is then given
 * batchIdx = seqIdx = NULL.  The preparation then runs beside the tail of the current step (layer 1's weight gradient alone on
 * the matrix pipes) instead of beside the next step's first conv layers. */
int cpc_train_step_prefetch(const long* batchIdx, const long* seqIdx, float* workspace, int B, int L, int K, int N,
                            void* side_stream);
/* The tail of an open-tailed step (phases + 4), after the optimiser's three launches (conv0's parameters on main_stream,
 * conv1.weight on wgrad_stream, every other parameter on prep_stream behind cpc_train_step_wait(main, 0 / 3 / 4, prep)): the
 * next step's weight preparation, each share on the stream its parameters arrive on, for parity next_parity, and the events the
 * next step's layer 1 waits for.  params: the 20 encoder tensors (cpc_encoder_forward's order). */
int cpc_train_step_tail(const float* const* params, float* workspace, int B, int L, int K, int N, int next_parity,
                        void* main_stream, void* prep_stream, void* wgrad_stream);
/* waiting_stream waits for events the last cpc_train_step on main_stream recorded: 0 = everything of wgrad_stream but layer 1's
 * weight gradient (the recurrence's and conv2..4's weight gradients: a data-parallel caller's mid gradient bucket), 1 = what an
 * open-tailed step left open (layer 1's weight gradient + the bias / norm gradient sums on prep_stream), 2 =
 * cpc_train_step_tail's end (every updated weight and layout), 3 = the heads' weight gradient, 4 = the bias / norm sums alone. */
int cpc_train_step_wait(void* main_stream, int which, void* waiting_stream);
/* In-step timing (diagnostic): while on, cpc_train_step records timing events around layer 0, layer 1, the two persistent
 * recurrence launches and the criterion's scoring kernel on main_stream; cpc_get_step_timing waits for the last and writes the
 * 5 durations of the most recent step in microseconds (conv0, conv1, forward recurrence, backward recurrence, scoring kernel;
 * each includes one marker's cost). */
/* The forward of a short conv layer (conv2..4 with H2 input, below the 128-row-tile regime; cpc/model.py:87-92,101-104) on
 * 128 x 128 tiles, two workgroups per 128-row tile with the ChannelNorm statistics (cpc/model.py:50-58) exchanged between the pair
 * through global memory, wherever that gives at least min_wgs workgroups (default 256; 0 = the full-row tiles).  spin_limit: polls
 * before a workgroup gives up on its partner (< 0: the default 2^22; tests use 0 to drive CPC_DEVERR_CONV_EXCHANGE). */
int cpc_set_fwd_nsplit(int min_wgs, int spin_limit);
int cpc_set_step_timing(int on);
/* Where the next step's layer 0 takes up an open tail: 0 behind the whole tail (layer 1's weight gradient, its split reduction, the
 * update of conv1.weight and that weight's layouts), 1 under it -- only layer 1 waits (layer 0 gets a quarter of its wave slots
 * beside layer 1's weight gradient) --, 2 (default since the end of round 6: -13 us per step) behind the weight gradient and its
 * reduction, beside the update and the layouts, 3 already behind the weight gradient's GEMM (an event more on that stream: no gain). */
int cpc_set_tail_schedule(int conv0_early);
int cpc_get_step_timing(float* us);
/* Measurement switches of cpc_train_step's schedule.  prep_point: where the criterion's index preparation (190 MB of index
 * traffic at B = 64) is released on side_stream -- 0 at the step's start (beside conv0, the one HBM-bound layer: 50 -> 96 us), 1
 * (default) behind conv0 (beside conv1 / conv2), 2 behind the encoder (beside the recurrence), 3 behind conv1 (beside conv2..conv4).  dz_early: 1 = the dz path on main_stream BEFORE the recurrence's
 * backward (which then has the memory system to itself) instead of beside it on side_stream (0, default); + 2 = the small
 * weight-only launches of the criterion / recurrence stay on main_stream; + 4 = the weight layouts of conv layers 1..4 are
 * prepared beside layer 0 on prep_stream instead of in front of it on main_stream (measured 8 us slower). */
int cpc_set_step_schedule(int prep_point, int dz_early);

/* ---- optimiser -------------------------------------------------------------------------------------------------
 * One Adam step (cpc/train.py:335-337: torch.optim.Adam(params, lr, betas, eps); :88-89 optimizer.step()) on n dense
 * fp32 tensors in one launch: exp_avg <- lerp(exp_avg, grad, 1 - beta1), exp_avg_sq <- beta2 exp_avg_sq + (1 - beta2)
 * grad^2, param <- param - (lr / bias_correction1) exp_avg / (sqrt(exp_avg_sq) / bias_correction2_sqrt + eps), all in
 * place.  bias_correction1 = 1 - beta1^step, bias_correction2_sqrt = sqrt(1 - beta2^step), step counted from 1.
 * No weight decay / amsgrad / maximize (the reference uses none). */
int cpc_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  const long* numel, int n, double lr, double beta1, double beta2, double eps, double bias_correction1,
                  double bias_correction2_sqrt, void* stream);
/* The same with the step counter on the device: nothing changes from call to call on the host side, so the launch can be
 * part of a captured HIP graph.  step: one device double = updates done so far (incremented by the call); coef: 8 device
 * floats of scratch. */
int cpc_adam_step_capturable(float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const long* numel, int n, double lr, double beta1, double beta2,
                             double eps, double* step, float* coef, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CPC_HIP_H */
