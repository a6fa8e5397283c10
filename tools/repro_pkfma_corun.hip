// Minimal reproducer attempt for the co-residency corruption of DESIGN.md section 4.6.
//
// Observation (round 1, tools/probe_corun.py): conv0_bwd_kernel, when the compiler had formed packed fp32 arithmetic in
// it (v_pk_fma_f32 whose broadcast operand is an op_sel-selected half of a ds_read2_b32 register pair), returned wrong
// partial sums in ~20 % of its workgroups whenever a 16-bit-MFMA GEMM kernel shared the chip with it -- and never
// alone, nor beside an f32-MFMA GEMM or copy kernels.
//
// This program isolates the two ingredients: a VICTIM that runs exactly that instruction pattern (inline asm, so the
// compiler cannot choose differently) in three variants
//     0  scalar v_fma_f32                       (what -fno-slp-vectorize produces)
//     1  v_pk_fma_f32, plain packed operands
//     2  v_pk_fma_f32 with op_sel / op_sel_hi broadcasting one half of a ds_read2_b32 pair   (the suspect)
// and an AGGRESSOR on a second stream, resident on the same CUs:
//     0  none     1  v_mfma_f32_32x32x2_f32     2  v_mfma_f32_32x32x16_f16     3  v_mfma_f32_32x32x8_f16
// Every (victim, aggressor) pair is run several times; the victim's output is compared bit for bit with its solo run
// and with the scalar variant (identical arithmetic: each lane performs the same fused multiply-adds in the same order).
//
//   hipcc -O3 --offload-arch=gfx950 tools/repro_pkfma_corun.hip -o tools/_bin/repro_pkfma_corun && tools/_bin/repro_pkfma_corun
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } \
    } while (0)

constexpr int kTaps = 10;          // conv0 has 10 taps: five ds_read2_b32 pairs per step
constexpr int kIters = 2048;

// Each thread: acc[j] (j < 10, two independent rows each) += x_t * w[j] over kIters steps; w from LDS.
template <int VARIANT>
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     float* __restrict__ out) {
    __shared__ float lw[kIters * kTaps];
    for (int i = threadIdx.x; i < kIters * kTaps; i += 256) lw[i] = w[i];
    __syncthreads();
    const int gid = blockIdx.x * 256 + threadIdx.x;
    f32x2 acc[kTaps];
#pragma unroll
    for (int j = 0; j < kTaps; ++j) acc[j] = f32x2{0.f, 0.f};
    const float* xp = x + (long)gid * 2;
    f32x2 xv = f32x2{xp[0], xp[1]};
    for (int t = 0; t < kIters; ++t) {
        const unsigned addr = (unsigned)(size_t)(&lw[t * kTaps]);   // LDS byte address (low 32 bits of the generic ptr)
        f32x2 wp[kTaps / 2];
#pragma unroll
        for (int q = 0; q < kTaps / 2; ++q) {
            if (VARIANT == 2) {
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(wp[q]) : "v"(addr), "n"(2 * q), "n"(2 * q + 1));
            } else {
                wp[q] = f32x2{lw[t * kTaps + 2 * q], lw[t * kTaps + 2 * q + 1]};
            }
        }
        if (VARIANT == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < kTaps / 2; ++q) {
            if (VARIANT == 0) {
                acc[2 * q].x = __builtin_fmaf(xv.x, wp[q].x, acc[2 * q].x);
                acc[2 * q].y = __builtin_fmaf(xv.y, wp[q].x, acc[2 * q].y);
                acc[2 * q + 1].x = __builtin_fmaf(xv.x, wp[q].y, acc[2 * q + 1].x);
                acc[2 * q + 1].y = __builtin_fmaf(xv.y, wp[q].y, acc[2 * q + 1].y);
            } else if (VARIANT == 1) {
                const f32x2 w0 = f32x2{wp[q].x, wp[q].x}, w1 = f32x2{wp[q].y, wp[q].y};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[2 * q]) : "v"(xv), "v"(w0));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[2 * q + 1]) : "v"(xv), "v"(w1));
            } else {
                // lo lane: x.lo * w.lo + acc.lo ; hi lane: x.hi * w.lo + acc.hi
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[2 * q]) : "v"(xv), "v"(wp[q]));
                // lo lane: x.lo * w.hi + acc.lo ; hi lane: x.hi * w.hi + acc.hi
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[2 * q + 1]) : "v"(xv), "v"(wp[q]));
            }
        }
        xv.x = xv.x * 0.999f + 1e-3f;      // the operand changes every step (same sequence in every variant)
        xv.y = xv.y * 1.001f - 1e-3f;
    }
#pragma unroll
    for (int j = 0; j < kTaps; ++j) {
        out[((long)gid * kTaps + j) * 2] = acc[j].x;
        out[((long)gid * kTaps + j) * 2 + 1] = acc[j].y;
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void aggressor_kernel(float* __restrict__ sink, int iters) {
    f32x16 acc0 = {0}, acc1 = {0};
    const float s = 1.0f + threadIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 1) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, 1.0f, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, s, acc1, 0, 0, 0);
        } else if (KIND == 2) {
            const f16x8 a = {(_Float16)s, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, (_Float16)s};
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
        } else {
            const f16x4 a = {(_Float16)s, 1, 1, 1}, b = {1, 1, 1, (_Float16)s};
            acc0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x8f16(b, a, acc1, 0, 0, 0);
        }
    }
    float r = 0.f;
    for (int q = 0; q < 16; ++q) r += acc0[q] + acc1[q];
    if (r == 123.456f) sink[0] = r;
}

typedef void (*VictimFn)(const float*, const float*, float*);

int main() {
    const int nblk = 2048, n = nblk * 256;
    std::vector<float> hx(2 * n), hw(kIters * kTaps);
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.1f;
    float *dx, *dw, *dout, *sink;
    const size_t out_bytes = sizeof(float) * (size_t)n * kTaps * 2;
    CK(hipMalloc(&dx, sizeof(float) * hx.size()));
    CK(hipMalloc(&dw, sizeof(float) * hw.size()));
    CK(hipMalloc(&dout, out_bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemcpy(dx, hx.data(), sizeof(float) * hx.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), sizeof(float) * hw.size(), hipMemcpyHostToDevice));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    std::vector<float> ref0(n * kTaps * 2), ref(n * kTaps * 2), cur(n * kTaps * 2);
    VictimFn victims[3] = {victim_kernel<0>, victim_kernel<1>, victim_kernel<2>};
    const char* vname[3] = {"scalar v_fma_f32", "v_pk_fma_f32 plain", "v_pk_fma_f32 op_sel on ds_read2_b32 pairs"};
    const char* aname[4] = {"none", "v_mfma_f32_32x32x2_f32", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x8_f16"};
    int total_bad = 0;
    for (int v = 0; v < 3; ++v) {
        CK(hipMemsetAsync(dout, 0xFF, out_bytes, s1));
        hipLaunchKernelGGL(victims[v], dim3(nblk), dim3(256), 0, s1, dx, dw, dout);
        CK(hipStreamSynchronize(s1));
        CK(hipMemcpy(ref.data(), dout, out_bytes, hipMemcpyDeviceToHost));
        if (v == 0) ref0 = ref;
        const bool same_as_scalar = std::memcmp(ref.data(), ref0.data(), out_bytes) == 0;
        printf("victim %d (%s): solo run %s the scalar variant\n", v, vname[v], same_as_scalar ? "==" : "!=");
        for (int a = 0; a < 4; ++a) {
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemsetAsync(dout, 0xFF, out_bytes, s1));
                CK(hipStreamSynchronize(s1));
                const int iters = 400000;           // ~ several ms per aggressor workgroup: outlives the victim
                if (a == 1) hipLaunchKernelGGL(aggressor_kernel<1>, dim3(512), dim3(256), 0, s2, sink, iters / 4);
                if (a == 2) hipLaunchKernelGGL(aggressor_kernel<2>, dim3(512), dim3(256), 0, s2, sink, iters);
                if (a == 3) hipLaunchKernelGGL(aggressor_kernel<3>, dim3(512), dim3(256), 0, s2, sink, iters / 2);
                hipLaunchKernelGGL(victims[v], dim3(nblk), dim3(256), 0, s1, dx, dw, dout);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(cur.data(), dout, out_bytes, hipMemcpyDeviceToHost));
                long bad = 0, bad_blocks = 0;
                for (int b = 0; b < nblk; ++b) {
                    long bb = 0;
                    const size_t off = (size_t)b * 256 * kTaps * 2;
                    for (size_t i = 0; i < (size_t)256 * kTaps * 2; ++i)
                        bb += std::memcmp(&cur[off + i], &ref[off + i], 4) != 0;
                    bad += bb;
                    bad_blocks += bb != 0;
                }
                total_bad += bad != 0;
                printf("  beside %-26s rep %d: %ld differing values in %ld of %d workgroups\n", aname[a], rep, bad, bad_blocks, nblk);
            }
        }
    }
    printf("RESULT: %s\n", total_bad ? "corruption reproduced" : "no differing bit in any pair");
    return 0;
}
