// The whole north-star train step behind ONE C call: every launch of forward and backward (cpc/train.py:78-87: model,
// criterion, allLosses.sum().backward()) enqueued on the caller's four streams from here, in the order and with the
// cross-stream dependencies the Python train loop (ops.py, train.Trainer._eager_step) uses -- same entry points, same
// kernels, same arithmetic: bit-identical results.  What it removes is the host's share: ~60 launches, three autograd nodes,
// a dozen allocator calls and as many torch events issued from Python cost 1.4-1.8 ms per 3.1 ms step (5 ms on a busy box);
// from here the step costs the host the launches alone.  With that the eager step no longer depends on HIP-graph replay,
// which is not available where the gradient all-reduce runs (N > 1).
//
// Streams (all owned by the caller):
//   main    the dx chain: encoder, recurrence, criterion forward, dPred / dc, recurrence backward, encoder backward
//   side    the criterion's step-independent preparation (index lists of the negative draws, GEMM operand bounds), later the dz
//           path (per-destination gather-GEMM + dense GEMM) beside the recurrence's backward, then the heads' weight gradient
//   prep    forward-only preparation of the recurrence's backward (weight transposes, hand-over buffers)
//   wgrad   weight gradients of the recurrence and of the conv layers, beside the dx chain
// Phases (bit mask; a data-parallel caller runs them as separate calls with its gradient all-reduce in between):
//   1  forward + backward down to the encoder's input gradient dz_total: every gradient but the encoder's is final or queued
//      (heads: side stream; recurrence: wgrad stream) when it returns
//   2  encoder backward; main waits for side and wgrad before it returns: all gradients final on main
//   4  (with 2) OPEN TAIL: main does not wait for the step's last kernel -- layer 1's weight gradient, which runs 0.13 ms past
//      the end of the main stream's chain -- but only for everything else; the caller updates conv1.weight on the wgrad stream
//      behind it and everything else on main, calls cpc_train_step_tail, and the NEXT step (bits 8, 16) starts under the tail
//   8  (with 1) the conv weight layouts / bounds of this step were prepared by cpc_train_step_tail at the end of the previous
//      one; main waits for conv1's (event kEvNextConv1, recorded on the wgrad stream) in front of layer 1
//  16  this step uses the second y0 buffer / bound set of the workspace (the caller alternates while it leaves tails open)
#include "cpc_common.h"
#include "cpc_internal.h"

namespace cpc {
namespace {

constexpr int kEncParams = 20, kGruParams = 8;      // + 1: the K stacked head weights

struct StepLayout {
    int S, W;
    long enc[22], gru[3], nce[6], coef;
    long enc_saved, enc_fscr, z, gru_saved, gru_fscr, c, gcoef, nce_saved, nce_fscr, ext, perm, row_ptr, work;
    long nce_bscr, dc, dz, gru_bscr, dx, enc_bscr, total;
};

bool step_layout(int B, int L, int K, int N, StepLayout& s) {
    if (B <= 0 || cpc_encoder_layout(B, L, s.enc) != 0) return false;
    s.S = (int)s.enc[7];
    s.W = s.S - K;
    if (s.W <= 0 || cpc_gru_layout(B, s.S, 2, s.gru) != 0 || cpc_nce_layout(B, s.S, K, N, s.nce) != 0) return false;
    s.coef = cpc_gru_coef_floats(B, s.S, 2);
    if (s.coef <= 0) return false;
    const int Np = cpc_nce_padded_negatives(N);      // candidates per window as the criterion's kernels lay them out
    const long act = (long)B * s.S * kC, slots = (long)B * s.W * (Np + K), rows = (long)B * s.S;
    long o = 0;
    auto take = [&](long n) { const long at = o; o += align64l(n); return at; };
    s.enc_saved = take(s.enc[0]); s.enc_fscr = take(std::max(1L, s.enc[1])); s.z = take(act);
    s.gru_saved = take(s.gru[0]); s.gru_fscr = take(s.gru[1]); s.c = take(act); s.gcoef = take(s.coef);
    s.nce_saved = take(s.nce[0]); s.nce_fscr = take(s.nce[1]);
    s.ext = take((long)B * s.W * Np); s.perm = take(slots); s.row_ptr = take(rows + 1); s.work = take(slots + 2 * rows + 2);
    s.nce_bscr = take(s.nce[2]); s.dc = take(act); s.dz = take(act);
    s.gru_bscr = take(s.gru[2]); s.dx = take(act); s.enc_bscr = take(s.enc[2]);
    s.total = o;
    return true;
}

// a += b  (n % 4 == 0): the two gradients of the encoder output -- through the criterion and through the recurrence -- summed
// as autograd does between the two backward nodes (one fp32 addition per element; the order of the operands does not matter)
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 x = reinterpret_cast<float4*>(a)[i];
    const float4 y = reinterpret_cast<const float4*>(b)[i];
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    reinterpret_cast<float4*>(a)[i] = x;
}

// tuning / measurement switches (cpc_set_step_schedule)
int g_prep_point = 1;     // where the criterion's index preparation is released on the side stream: 0 step begin (beside conv0: the one
                          // HBM-bound layer goes from 50 to 96 us), 1 (default) behind conv0 (beside conv1 / conv2: +20 us there),
                          // 2 behind the encoder (beside the recurrence, whose hand-over it disturbs), 3 behind conv1 (beside conv2..4)
int g_no_early = 0;
int g_weight_prep_apart = 0;   // 1: the conv weight layouts on the preparation stream beside layer 0 instead of in front of it on main
                               // (measured: +8 us per step -- the 36 us they free on main, layer 0 and layer 1's wait give back)
int g_gru_wgrad_prep = 1; // 1 (default): the recurrence's weight / bias gradients on the PREPARATION stream, which is idle during the
                          // backward, instead of in front of the conv layers' on the weight-gradient stream (single-rank steps,
                          // phases 3; cpc_set_gru_wgrad_stream).  The conv layers' weight gradients then start 0.2 ms earlier and
                          // layer 2's no longer runs beside layer 1's data gradient: 2.764 against 2.779 ms per step sustained
                          // (open tail), 2.784 against 2.798 (closed), three alternations each (profiles/r5_ab_gru_wgrad_stream.txt)
int g_dz_early = 0;       // 1: the dz path on MAIN before the recurrence's backward (which then has the memory system to itself)
                          // instead of beside it on the side stream

inline bool rec(hipEvent_t e, hipStream_t s) { return hipEventRecord(e, s) == hipSuccess; }
inline bool wait(hipStream_t s, hipEvent_t e) { return hipStreamWaitEvent(s, e, 0) == hipSuccess; }

// in-step timing (cpc_set_step_timing): timing-enabled events recorded around layer 0, layer 1 and the two recurrence launches
// INSIDE the step (StepHooks::timers, step_timer_mark); a diagnostic of its own steps, never of the timed ones -- every marker
// costs the stream 6-8 us
int g_step_timing = 0;
hipEvent_t g_timers[10];
bool g_timers_made = false;

// the hooks of one cpc_train_step call, restored when it returns
struct HookScope {
    StepHooks saved;
    HookScope() : saved(step_hooks()) {}
    ~HookScope() { step_hooks() = saved; }
};

}  // namespace

// enc_conv.hip: an event cpc_encoder_forward records on its stream right behind layer 0's launch (nullptr: none)
void enc_set_after_conv0_event(hipEvent_t ev);
void enc_set_forward_event(int layer, hipEvent_t ev);
void enc_set_weight_prep_stream(hipStream_t st, hipEvent_t done);

}  // namespace cpc

using namespace cpc;

extern "C" int cpc_set_gru_wgrad_stream(int on_prep) {
    g_gru_wgrad_prep = on_prep ? 1 : 0;
    return 0;
}

extern "C" int cpc_set_step_schedule(int prep_point, int dz_early) {
    CPC_RETURN_IF(prep_point < 0 || prep_point > 3 || dz_early < 0 || dz_early > 7, CPC_ERR_ARG);
    g_prep_point = prep_point;
    g_dz_early = dz_early & 1;
    g_no_early = (dz_early >> 1) & 1;      // + 2: the small weight-only launches stay on the main stream where round 3 had them (A/B)
    g_weight_prep_apart = (dz_early >> 2) & 1;      // + 4: the conv weight layouts beside layer 0 on the preparation stream (A/B)
    return 0;
}

// sizes[0] = floats of the step's workspace; [1], [2] = offsets of z and c (B, S, 256) inside it (outputs of phase 1, valid
// until the next step); [3] = S; [4..7] = offsets of ext (B, W, N int32), dz_total, dc, dx (tests)
extern "C" int cpc_train_step_layout(int B, int L, int K, int N, long* sizes) {
    StepLayout s;
    CPC_RETURN_IF(!sizes || !step_layout(B, L, K, N, s), CPC_ERR_SHAPE);
    sizes[0] = s.total; sizes[1] = s.z; sizes[2] = s.c; sizes[3] = s.S;
    sizes[4] = s.ext; sizes[5] = s.dz; sizes[6] = s.dc; sizes[7] = s.dx;
    return 0;
}

extern "C" int cpc_train_step(const float* wave, const long* batchIdx, const long* seqIdx, const float* h0, float c_bound,
                              const float* const* params, float* const* grads, const float* gloss, float* workspace,
                              float* losses, float* acc, float* hN, int B, int L, int K, int N, int phases, void* main_stream,
                              void* side_stream, void* prep_stream, void* wgrad_stream) {
    StepLayout s;
    CPC_RETURN_IF(!step_layout(B, L, K, N, s), CPC_ERR_SHAPE);
    CPC_RETURN_IF(!wave || !params || !grads || !workspace || (phases & ~31) || !(phases & 3), CPC_ERR_ARG);
    CPC_RETURN_IF(((phases & 4) && !(phases & 2)) || ((phases & 8) && !(phases & 1)), CPC_ERR_ARG);
    hipStream_t M = (hipStream_t)main_stream, S0 = (hipStream_t)side_stream, S1 = (hipStream_t)prep_stream,
                S2 = (hipStream_t)wgrad_stream;
    // (equal handles are allowed: the launches then simply queue up in issue order, which respects every dependency below)
    hipEvent_t* pool = stream_events(M);
    CPC_RETURN_IF(!pool, CPC_ERR_ARG);
    HookScope scope;
    StepHooks& hk = step_hooks();
    hk.parity = (phases & 16) ? 1 : 0;
    hk.weights_ready = (phases & 8) != 0;
    hk.conv1_wait[0] = (phases & 8) ? pool[kEvNextConv1] : nullptr;
    hk.conv1_wait[1] = (phases & 8) ? pool[kEvNextRest] : nullptr;
    hk.open_tail = (phases & 4) != 0 && S2 != M;
    hk.sums_stream = hk.open_tail && S1 != M ? S1 : nullptr;
    hk.timers = (g_step_timing && g_timers_made) ? g_timers : nullptr;
    hipEvent_t* ev = pool + 12;     // [0] begin, [1] index lists + bounds ready, [2] recurrence-backward preparation ready,
                                    // [3] score gradients ready, [4] dz done, [5] head gradient done, [6] conv0 launched / encoder done,
                                    // [7] forward recurrence's hand-over buffers filled, [8] conv weight layouts ready
    float* ws = workspace;
    const float* const* enc_p = params;
    const float* const* gru_p = params + kEncParams;
    const float* wall = params[kEncParams + kGruParams];
    float* const* enc_g = grads;
    float* const* gru_g = grads + kEncParams;
    float* dwall = grads[kEncParams + kGruParams];
    float* z = ws + s.z, *c = ws + s.c, *coef = ws + s.gcoef, *dz = ws + s.dz, *dc = ws + s.dc, *dx = ws + s.dx;
    int* ext = reinterpret_cast<int*>(ws + s.ext), *perm = reinterpret_cast<int*>(ws + s.perm);
    int* row_ptr = reinterpret_cast<int*>(ws + s.row_ptr), *work = reinterpret_cast<int*>(ws + s.work);
    const int S = s.S;
    int rc = 0;
    if (phases & 1) {
        // batchIdx == seqIdx == NULL: the index lists of this step's draws are already in the workspace (cpc_train_step_prefetch,
        // queued on side_stream behind the previous step)
        CPC_RETURN_IF((batchIdx == nullptr) != (seqIdx == nullptr) || !gloss || !losses || !acc || !hN, CPC_ERR_ARG);
        // ---- forward ----
        // the step begins here on main: the other streams fork from this point (the workspace is the previous step's, whose
        // last users main has waited for)
        CPC_RETURN_IF(!rec(ev[0], M) || !wait(S1, ev[0]), CPC_ERR_ARG);
        // (after an open-tailed step the optimiser's update of everything but conv0 / conv1.weight ran on the preparation stream,
        // S1 itself: the side stream reads the heads' weights and must come behind it -- long done by now)
        if (phases & 8) CPC_RETURN_IF(!wait(S0, pool[kEvNextRest]), CPC_ERR_ARG);
        const bool bounds_early = c_bound > 0.f;
        const bool early = bounds_early && !g_no_early;      // the criterion's operand bounds do not depend on c: its backward's weight-only share too
        auto prepare = [&]() -> int {      // index lists of the draws + operand bounds of the prediction GEMMs: depend on no activation
            int r = batchIdx ? cpc_nce_prepare(batchIdx, seqIdx, ext, perm, row_ptr, work, B, S, K, N, S0) : 0;
            if (r) return r;
            if (bounds_early) r = cpc_nce_bounds(nullptr, c_bound, wall, ws + s.nce_saved, B, S, K, N, S0);
            if (r) return r;
            // ... and what the criterion's backward needs of the weights and of gloss alone (gradient scales, GEMM bounds, cleared
            // maximum slots, the zeroed tail of dc, wall^T): three small launches off the chain criterion -> recurrence
            if (early) r = cpc_nce_backward_prepare(wall, ws + s.nce_saved, gloss, ws + s.nce_bscr, dc, B, S, K, N, S0);
            if (r) return r;
            return rec(ev[1], S0) ? 0 : CPC_ERR_ARG;
        };
        if (g_prep_point == 0) {
            CPC_RETURN_IF(!wait(S0, ev[0]), CPC_ERR_ARG);
            if ((rc = prepare())) return rc;
        }
        if (g_prep_point == 3) enc_set_forward_event(1, ev[6]);
        else enc_set_after_conv0_event(g_prep_point == 1 ? ev[6] : nullptr);
        // (switch, off: the weight layouts of layers 1..4 -- 36 us of small kernels on the parameters alone -- first thing on the
        // preparation stream, beside layer 0, which needs only their four bounds)
        enc_set_weight_prep_stream(g_weight_prep_apart ? S1 : nullptr, ev[8]);
        rc = cpc_encoder_forward(wave, enc_p, ws + s.enc_saved, ws + s.enc_fscr, z, B, L, M);
        enc_set_after_conv0_event(nullptr);
        enc_set_weight_prep_stream(nullptr, nullptr);
        if (rc) return rc;
        // the hand-over buffers of the forward recurrence pre-filled beside the encoder instead of between the input projection
        // and the recurrence (queued behind the weight layouts: the recurrence is 0.4 ms away)
        rc = cpc_gru_forward_prepare(ws + s.gru_fscr, B, S, 2, S1);
        if (rc) return rc;
        CPC_RETURN_IF(!rec(ev[7], S1), CPC_ERR_ARG);
        if (g_prep_point == 1 || g_prep_point == 3) {
            CPC_RETURN_IF(!wait(S0, ev[6]), CPC_ERR_ARG);
            if ((rc = prepare())) return rc;
        } else if (g_prep_point == 2) {
            CPC_RETURN_IF(!rec(ev[6], M) || !wait(S0, ev[6]), CPC_ERR_ARG);
            if ((rc = prepare())) return rc;
        }
        // the fp16-piece scoring kernel (cpc_set_nce_fused(2 / 3)) gathers from an H2 copy of z: made on the side stream as soon as the
        // encoder is done (behind the index lists), beside the recurrence; the event main waits for in front of the criterion is
        // recorded again behind it
        if (cpc_get_nce_fused() >= 2 && S0 != M) {
            CPC_RETURN_IF(!rec(pool[kEvEncoderDone], M) || !wait(S0, pool[kEvEncoderDone]), CPC_ERR_ARG);
            rc = cpc_nce_prepare_z(z, ws + s.nce_saved, B, S, K, N, S0);
            if (rc) return rc;
            CPC_RETURN_IF(!rec(ev[1], S0), CPC_ERR_ARG);
        }
        CPC_RETURN_IF(!wait(M, ev[7]), CPC_ERR_ARG);
        rc = cpc_gru_forward_coef_prepared(z, h0, gru_p, ws + s.gru_saved, ws + s.gru_fscr, c, hN, coef, B, S, 2, M);
        if (rc) return rc;
        // what the recurrence's backward needs beyond the coefficients the forward writes on its way (weight transposes,
        // hand-over buffers: disjoint parts of `coef`) depends on the parameters only: beside the forward, on its own stream
        rc = cpc_gru_backward_coef(h0, gru_p, ws + s.gru_saved, c, coef, 1, B, S, 2, S1);
        if (rc) return rc;
        CPC_RETURN_IF(!rec(ev[2], S1) || !wait(M, ev[1]), CPC_ERR_ARG);
        // (the loss / accuracy reduction runs on the side stream: the backward reads the saved logits, not the losses; the side
        // stream is joined before the step ends)
        // (... and, with the fp16-piece scoring kernel, the softmax rows the dz path reads -- which therefore must run on S0 as well)
        rc = cpc_nce_forward_streams(c, z, wall, ext, ws + s.nce_saved, ws + s.nce_fscr, losses, acc, B, S, K, N, bounds_early ? 1 : 0, M,
                                     early && !g_dz_early ? S0 : M);
        if (rc) return rc;
        // ---- backward ----
        // criterion: score gradients, dPred and dc on main; the dz path and the heads' weight gradient are held back
        rc = early ? cpc_nce_backward_prepared(c, z, wall, ext, perm, row_ptr, ws + s.nce_saved, gloss, ws + s.nce_bscr, dc, B, S, K,
                                               N, M)
                   : cpc_nce_backward_streams(c, z, wall, ext, perm, row_ptr, ws + s.nce_saved, gloss, ws + s.nce_bscr, dc, nullptr,
                                              nullptr, B, S, K, N, M, M);
        if (rc) return rc;
        if (g_dz_early) {
            rc = cpc_nce_backward_dz(c, wall, perm, row_ptr, ws + s.nce_saved, ws + s.nce_bscr, dz, B, S, K, N, M);
            if (rc) return rc;
        }
        step_timer_mark(5, M);           // (in front of the event that releases the side stream: see cpc_gru_backward_streams)
        CPC_RETURN_IF(!rec(ev[3], M) || !wait(M, ev[2]), CPC_ERR_ARG);
        // recurrence: dx on main, its weight / bias gradients straight into `grads` on the wgrad stream (no join here)
        const bool gru_wgrad_prep = g_gru_wgrad_prep && (phases & 3) == 3 && S1 != M && S1 != S2;
        rc = cpc_gru_backward_streams(z, h0, gru_p, ws + s.gru_saved, c, dc, coef, ws + s.gru_bscr, dx, gru_g, B, S, 2, M,
                                      gru_wgrad_prep ? S1 : S2);
        if (rc) return rc;
        if (gru_wgrad_prep) CPC_RETURN_IF(!rec(pool[kEvGruWgrad], S1), CPC_ERR_ARG);
        // ... and only now, with the persistent recurrence in flight (its 768-thread workgroups could not become resident beside
        // a chip full of gather blocks), the dz path and behind it the heads' gradient on the side stream
        CPC_RETURN_IF(!wait(S0, ev[3]), CPC_ERR_ARG);
        if (!g_dz_early) {
            rc = cpc_nce_backward_dz(c, wall, perm, row_ptr, ws + s.nce_saved, ws + s.nce_bscr, dz, B, S, K, N, S0);
            if (rc) return rc;
            CPC_RETURN_IF(!rec(ev[4], S0), CPC_ERR_ARG);
        }
        // (the heads' weight gradient stays behind the dz path on the side stream: moved behind the recurrence's weight gradients on
        // the preparation stream -- it needs neither -- it lands beside the short layers' data gradients and costs 40 us, measured)
        rc = cpc_nce_backward_dwall(c, ws + s.nce_saved, ws + s.nce_bscr, dwall, B, S, K, N, S0);
        if (rc) return rc;
        CPC_RETURN_IF(!rec(ev[5], S0), CPC_ERR_ARG);
        if (!g_dz_early) CPC_RETURN_IF(!wait(M, ev[4]), CPC_ERR_ARG);
        const long n4 = (long)B * S * kC / 4;
        hipLaunchKernelGGL(add_inplace_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, M, dz, dx, n4);
        CPC_LAUNCH_CHECK();
    }
    if (phases & 2) {
        rc = cpc_encoder_backward_streams(wave, enc_p, ws + s.enc_saved, z, dz, ws + s.enc_bscr, enc_g, B, L, M, S2);
        if (rc) return rc;          // (joins the wgrad stream -- the recurrence's gradients were queued there before the conv layers')
        CPC_RETURN_IF(!wait(M, ev[5]), CPC_ERR_ARG);
        // (the recurrence's gradients on the preparation stream: a closed step joins it too; after an open-tailed one the optimiser's
        // share for them runs on that very stream, behind them)
        if (g_gru_wgrad_prep && (phases & 3) == 3 && !(phases & 4) && S1 != M && S1 != S2)
            CPC_RETURN_IF(!wait(M, pool[kEvGruWgrad]), CPC_ERR_ARG);
    }
    return 0;
}

// The tail of an open-tailed step (phases & 4), called after the optimiser's three launches -- conv0's parameters on main behind
// conv0's backward, conv1.weight on the wgrad stream behind its gradient, every other parameter on the preparation stream (where
// the step left the bias / norm gradients of layers 1..4; the caller made it wait for events 0 and 3 of cpc_train_step_wait) --:
// the weight-only preparation of the NEXT step, each share on the stream its parameters arrive on, written for parity
// `next_parity`, and the two events the next step waits for in front of layer 1 (phases & 8).  Main gets one 4-workgroup launch
// (layer 1's input bound, from layer 0's new affine) between conv0's update and the next step's conv0.  Same kernels on the same
// values as the preparation at the head of a step: bit-identical results.
extern "C" int cpc_train_step_tail(const float* const* params, float* workspace, int B, int L, int K, int N, int next_parity,
                                   void* main_stream, void* prep_stream, void* wgrad_stream) {
    StepLayout s;
    CPC_RETURN_IF(!step_layout(B, L, K, N, s), CPC_ERR_SHAPE);
    CPC_RETURN_IF(!params || !workspace, CPC_ERR_ARG);
    hipStream_t M = (hipStream_t)main_stream, S1 = (hipStream_t)prep_stream, S2 = (hipStream_t)wgrad_stream;
    hipEvent_t* pool = stream_events(M);
    CPC_RETURN_IF(!pool, CPC_ERR_ARG);
    HookScope scope;
    step_hooks().parity = next_parity & 1;
    float* ws = workspace;
    int rc = cpc_encoder_prepare_weights(params, ws + s.enc_saved, ws + s.enc_fscr, B, L, 32, M);
    if (rc) return rc;
    rc = cpc_encoder_prepare_weights(params, ws + s.enc_saved, ws + s.enc_fscr, B, L, 1 | 4 | 8 | 16, S1);
    if (rc) return rc;
    CPC_RETURN_IF(!rec(pool[kEvNextRest], S1), CPC_ERR_ARG);
    rc = cpc_encoder_prepare_weights(params, ws + s.enc_saved, ws + s.enc_fscr, B, L, 2, S2);
    if (rc) return rc;
    CPC_RETURN_IF(!rec(pool[kEvNextConv1], S2), CPC_ERR_ARG);
    return 0;
}

// Make `waiting_stream` wait for events the step recorded for the caller (pool of `main_stream`):
//   0  everything of the weight-gradient stream but layer 1's weight gradient (the recurrence's and the short layers' weight
//      gradients: a data-parallel caller's mid gradient bucket, dist.FlatGradAllReduce; an open tail's update of "the rest")
//   1  what an open-tailed step (phases & 4) left open: layer 1's weight gradient and the batched bias / norm gradient sums --
//      closes the tail for a caller that will not call cpc_train_step_tail
//   2  the tail of an open-tailed step (cpc_train_step_tail): every updated weight and layout -- a caller that touches the
//      parameters outside cpc_train_step joins with this first
//   3  the heads' weight gradient (side stream)
//   4  the batched bias / norm gradient sums of an open-tailed step alone
extern "C" int cpc_train_step_wait(void* main_stream, int which, void* waiting_stream) {
    CPC_RETURN_IF(which < 0 || which > 4, CPC_ERR_ARG);
    hipEvent_t* pool = stream_events((hipStream_t)main_stream);
    CPC_RETURN_IF(!pool, CPC_ERR_ARG);
    hipStream_t w = (hipStream_t)waiting_stream;
    bool ok = true;
    if (which == 0) ok = wait(w, pool[kEvWgradRest]);
    else if (which == 1) ok = wait(w, pool[kEvWgrad1]) && wait(w, pool[kEvSums]) && wait(w, pool[kEvGruWgrad]);
    else if (which == 2) ok = wait(w, pool[kEvNextConv1]) && wait(w, pool[kEvNextRest]);
    else if (which == 3) ok = wait(w, pool[12 + 5]);
    else ok = wait(w, pool[kEvSums]);
    CPC_RETURN_IF(!ok, CPC_ERR_ARG);
    return 0;
}

// In-step timing: while on, every cpc_train_step records timing events around layer 0, layer 1 and the two persistent recurrence
// launches and the criterion's scoring kernel on its main stream.  cpc_get_step_timing waits for the last of them and returns the
// five durations of the most recent step in microseconds: [0] conv0, [1] conv1, [2] forward recurrence, [3] backward recurrence,
// [4] the scoring kernel (the markers' own cost -- 6-8 us each, measured by the caller with back-to-back events -- is included).
extern "C" int cpc_set_step_timing(int on) {
    if (on && !g_timers_made) {
        for (int i = 0; i < 10; ++i)
            if (hipEventCreate(&g_timers[i]) != hipSuccess) return CPC_ERR_ARG;
        g_timers_made = true;
    }
    g_step_timing = on ? 1 : 0;
    return 0;
}
extern "C" int cpc_get_step_timing(float* us) {
    CPC_RETURN_IF(!us || !g_timers_made, CPC_ERR_ARG);
    CPC_RETURN_IF(hipEventSynchronize(g_timers[6]) != hipSuccess, CPC_ERR_ARG);
    const int a[5] = {0, 7, 3, 5, 8}, b[5] = {1, 2, 4, 6, 9};
    for (int i = 0; i < 5; ++i) {
        float ms = 0.f;
        CPC_RETURN_IF(hipEventElapsedTime(&ms, g_timers[a[i]], g_timers[b[i]]) != hipSuccess, CPC_ERR_ARG);
        us[i] = ms * 1000.f;
    }
    return 0;
}

// The index lists of the NEXT step's negative draws, prepared one step ahead: the preparation (190 MB of index traffic at B = 64)
// depends on nothing but the draws, and the end of a step -- layer 1's weight gradient running alone on the matrix pipes -- leaves
// the memory system idle, while at the start of a step it competes with the conv layers.  Call it after cpc_train_step (phase 2)
// of the current step, with the next step's draws ready on `side_stream`; everything of the current step that reads the lists
// precedes it on that stream.  The next cpc_train_step (same B, L, K, N, workspace) is then called with batchIdx = seqIdx = NULL.
extern "C" int cpc_train_step_prefetch(const long* batchIdx, const long* seqIdx, float* workspace, int B, int L, int K, int N,
                                       void* side_stream) {
    StepLayout s;
    CPC_RETURN_IF(!step_layout(B, L, K, N, s), CPC_ERR_SHAPE);
    CPC_RETURN_IF(!batchIdx || !seqIdx || !workspace, CPC_ERR_ARG);
    float* ws = workspace;
    return cpc_nce_prepare(batchIdx, seqIdx, reinterpret_cast<int*>(ws + s.ext), reinterpret_cast<int*>(ws + s.perm),
                           reinterpret_cast<int*>(ws + s.row_ptr), reinterpret_cast<int*>(ws + s.work), B, s.S, K, N,
                           (hipStream_t)side_stream);
}
