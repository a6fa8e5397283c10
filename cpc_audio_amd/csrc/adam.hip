// Adam update of every parameter tensor of the step in ONE launch (cpc/train.py:335-337 builds
// torch.optim.Adam(params, lr, betas, eps); :88-89 calls optimizer.step() once per batch).
// torch's fused multi-tensor kernel walks the 2.9 M parameters of this model in ~60 workgroups of 64 K elements
// (two launches, 45 + 43 us on MI355X); the update is a pure stream of 7 x 11.6 MB, so here it is cut into 4 K-element
// pieces (~750 workgroups) and takes the time the bytes take.
//   m <- m + (1 - b1)(g - m)          (torch's lerp form)
//   v <- b2 v + (1 - b2) g g
//   p <- p - (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps),      bc_i = 1 - b_i^step   (computed by the host in double)
// No weight decay, no amsgrad, no maximize: the reference uses none (train.py:335-337).
#include "cpc_common.h"
#include "cpc_internal.h"

namespace cpc {

constexpr int kAdamMaxTensors = 48;
constexpr int kAdamChunk = 4096;           // elements per workgroup: 256 threads x 4 float4

struct AdamBatch {
    float* p[kAdamMaxTensors];
    const float* g[kAdamMaxTensors];
    float* m[kAdamMaxTensors];
    float* v[kAdamMaxTensors];
    int n[kAdamMaxTensors];
    int blk0[kAdamMaxTensors + 1];         // first workgroup of tensor i
    int count;
};

struct AdamCoef { float b1c, b2, b2c, step_size, inv_bc2_sqrt, eps; };

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamCoef& c) {
    m = m + c.b1c * (g - m);
    v = c.b2 * v + c.b2c * g * g;
    const float denom = sqrtf(v) * c.inv_bc2_sqrt + c.eps;
    p = p - c.step_size * m / denom;
}

// Capturable form (the whole train step replayed as one HIP graph: kernel arguments are frozen at capture, so nothing
// that changes from step to step may be a host scalar): the step counter lives on the device, this one-thread kernel
// advances it and forms the step's coefficients -- in double, as the host path and torch do -- for adam_kernel_dev.
__global__ void adam_coef_kernel(double* __restrict__ step, AdamCoef* __restrict__ out, double lr, double beta1, double beta2,
                                 double eps) {
    const double t = *step + 1.0;
    *step = t;
    const double bc1 = 1.0 - pow(beta1, t), bc2s = sqrt(1.0 - pow(beta2, t));
    AdamCoef c;
    c.b1c = (float)(1. - beta1); c.b2 = (float)beta2; c.b2c = (float)(1. - beta2);
    c.step_size = (float)(lr / bc1); c.inv_bc2_sqrt = (float)(1. / bc2s); c.eps = (float)eps;
    *out = c;
}

__device__ __forceinline__ void adam_body(const AdamBatch& b, const AdamCoef& c);

__global__ __launch_bounds__(256) void adam_kernel(AdamBatch b, AdamCoef c) { adam_body(b, c); }
__global__ __launch_bounds__(256) void adam_kernel_dev(AdamBatch b, const AdamCoef* __restrict__ cp) { adam_body(b, *cp); }

__device__ __forceinline__ void adam_body(const AdamBatch& b, const AdamCoef& c) {
    int t = 0;
    while (t + 1 < b.count && (int)blockIdx.x >= b.blk0[t + 1]) ++t;       // block-uniform
    const int n = b.n[t];
    const int e0 = ((int)blockIdx.x - b.blk0[t]) * kAdamChunk;
    float* __restrict__ p = b.p[t];
    const float* __restrict__ g = b.g[t];
    float* __restrict__ m = b.m[t];
    float* __restrict__ v = b.v[t];
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    if (vec) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0 + (q * 256 + (int)threadIdx.x) * 4;
            if (e + 3 < n) {
                float4 pp = *reinterpret_cast<float4*>(p + e);
                const float4 gg = *reinterpret_cast<const float4*>(g + e);
                float4 mm = *reinterpret_cast<float4*>(m + e), vv = *reinterpret_cast<float4*>(v + e);
                adam_one(pp.x, gg.x, mm.x, vv.x, c);
                adam_one(pp.y, gg.y, mm.y, vv.y, c);
                adam_one(pp.z, gg.z, mm.z, vv.z, c);
                adam_one(pp.w, gg.w, mm.w, vv.w, c);
                *reinterpret_cast<float4*>(p + e) = pp;
                *reinterpret_cast<float4*>(m + e) = mm;
                *reinterpret_cast<float4*>(v + e) = vv;
            } else {
                for (int i = e; i < n && i < e + 4; ++i) adam_one(p[i], g[i], m[i], v[i], c);
            }
        }
    } else {
        for (int i = e0 + threadIdx.x; i < n && i < e0 + kAdamChunk; i += 256) adam_one(p[i], g[i], m[i], v[i], c);
    }
}

}  // namespace cpc

using namespace cpc;

// One Adam step on n tensors (fp32, dense).  params / exp_avg / exp_avg_sq are updated in place; bias_correction1 =
// 1 - beta1^step and bias_correction2_sqrt = sqrt(1 - beta2^step) come from the caller; the scalars are doubles so that
// 1 - beta and lr / bias_correction1 are rounded to fp32 once, as in torch.
static int adam_launch(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       const long* numel, int n, const AdamCoef* host_coef, const AdamCoef* dev_coef, void* stream);

extern "C" int cpc_adam_step(float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const long* numel, int n, double lr, double beta1, double beta2,
                             double eps, double bias_correction1, double bias_correction2_sqrt, void* stream) {
    CPC_RETURN_IF(n < 0 || (n > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel)), CPC_ERR_ARG);
    CPC_RETURN_IF(!(bias_correction1 > 0.) || !(bias_correction2_sqrt > 0.), CPC_ERR_ARG);
    AdamCoef c;      // formed in double like torch does (1 - 0.999f would already be off by 1.3e-5 relative)
    c.b1c = (float)(1. - beta1); c.b2 = (float)beta2; c.b2c = (float)(1. - beta2);
    c.step_size = (float)(lr / bias_correction1); c.inv_bc2_sqrt = (float)(1. / bias_correction2_sqrt); c.eps = (float)eps;
    return adam_launch(params, grads, exp_avg, exp_avg_sq, numel, n, &c, nullptr, stream);
}

// The same update with the step counter on the device (graph-capturable: no per-step host scalar).  step: one device
// double, the number of updates done so far (incremented here); coef: 8 device floats of scratch that stay valid until
// the launch has run.  lr / betas / eps are frozen into a captured graph (the reference's schedule changes lr per epoch:
// re-capture, or run eagerly, when it changes).
extern "C" int cpc_adam_step_capturable(float* const* params, const float* const* grads, float* const* exp_avg,
                                        float* const* exp_avg_sq, const long* numel, int n, double lr, double beta1,
                                        double beta2, double eps, double* step, float* coef, void* stream) {
    CPC_RETURN_IF(n < 0 || (n > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel)) || !step || !coef, CPC_ERR_ARG);
    AdamCoef* cd = reinterpret_cast<AdamCoef*>(coef);
    hipLaunchKernelGGL(adam_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, cd, lr, beta1, beta2, eps);
    CPC_LAUNCH_CHECK();
    return adam_launch(params, grads, exp_avg, exp_avg_sq, numel, n, nullptr, cd, stream);
}

static int adam_launch(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       const long* numel, int n, const AdamCoef* host_coef, const AdamCoef* dev_coef, void* stream) {
    int i = 0;
    while (i < n) {
        AdamBatch b;
        int cnt = 0, nblk = 0;
        for (; i < n && cnt < kAdamMaxTensors; ++i) {
            CPC_RETURN_IF(numel[i] < 0 || numel[i] > 0x7fffffffL - kAdamChunk, CPC_ERR_SHAPE);
            if (numel[i] == 0) continue;
            CPC_RETURN_IF(!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i], CPC_ERR_ARG);
            b.p[cnt] = params[i]; b.g[cnt] = grads[i]; b.m[cnt] = exp_avg[i]; b.v[cnt] = exp_avg_sq[i];
            b.n[cnt] = (int)numel[i];
            b.blk0[cnt] = nblk;
            nblk += cdiv((int)numel[i], kAdamChunk);
            ++cnt;
        }
        if (!cnt) continue;
        b.blk0[cnt] = nblk;
        for (int q = cnt; q < kAdamMaxTensors; ++q) {
            b.p[q] = nullptr; b.g[q] = nullptr; b.m[q] = nullptr; b.v[q] = nullptr; b.n[q] = 0; b.blk0[q + 1] = nblk;
        }
        b.count = cnt;
        if (dev_coef) hipLaunchKernelGGL(adam_kernel_dev, dim3(nblk), dim3(256), 0, (hipStream_t)stream, b, dev_coef);
        else hipLaunchKernelGGL(adam_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, b, *host_coef);
        CPC_LAUNCH_CHECK();
    }
    return 0;
}
