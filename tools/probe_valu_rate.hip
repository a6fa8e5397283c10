// Issue rate of the VALU instructions conv0's epilogue is made of, per SIMD, with 1..4 waves per SIMD resident:
// cycles per wave64 instruction = kernel time x clock / (instructions per wave x waves per SIMD).
// 8 independent dependency chains per wave (so a single wave is not bound by result latency).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_valu_rate.hip -o tools/_bin/probe_valu_rate; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(64) void rate_kernel(float* out, int iters, float seed) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * 0.001f + i;
    const float s = seed * 0.5f + 1.0f, s2 = seed + 3.0f;
    const unsigned long long m64 = 0x5555555555555555ull;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p2[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
    const f2 q2 = {s, s2};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[u]) : "v"(s));
            if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u]) : "v"(s));
            if (OP == 2) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[u]));
            if (OP == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[u]) : "v"(s));
            if (OP == 4) asm volatile("v_fma_mixlo_f16 %0, %1, %0, 0" : "+v"(v[u]) : "v"(s));
            if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[u]) : "v"(s));
            if (OP == 6) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[u]));
            if (OP == 7) asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[u]));
            if (OP == 8) asm volatile("v_rsq_f32 %0, %0" : "+v"(v[u]));
            if (OP == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[u]) : "v"(s));
            if (OP == 10) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[u]) : "s"(s));                    // SGPR operand
            if (OP == 11) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[u]) : "v"(s), "s"(m64));   // explicit SGPR pair
            if (OP == 12) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[u]) : "v"(s), "v"(s2));
            if (OP == 13) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[u]), "v"(s) : "vcc");
            if (OP == 14) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0x5" : "+v"(v[u]) : "v"(s));
            if (OP == 15) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[u]) : "v"(s));
            if (OP == 16) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[u]) : "v"(s) : "vcc");
            if (OP == 17) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[u]) : "v"(s), "v"(s2));
            if (OP == 18) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[u]) : "v"(s), "v"(s2));
            if (OP == 19) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2[u & 3]) : "v"(q2));
            if (OP == 20) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[u]));
            if (OP == 21) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(v[u]));
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += v[i] + p2[i & 3].x + p2[i & 3].y;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int OP>
static void run(const char* name, float* out, double ghz) {
    const int iters = 4096;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * 4 * wps;            // 64-thread blocks: wps waves per SIMD on 256 CUs (if spread evenly)
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double cyc = best * 1e-3 * ghz * 1e9 / ((double)iters * 8 * wps);
        printf("%-22s %d waves/SIMD: %8.1f us  %5.2f cycles per instruction and SIMD (at %.2f GHz)\n", name, wps, best * 1e3, cyc, ghz);
    }
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    run<0>("v_mul_f32", out, ghz);
    run<1>("v_fma_f32", out, ghz);
    run<9>("v_sub_f32", out, ghz);
    run<7>("v_max_f32", out, ghz);
    run<2>("v_cvt_f32_f16", out, ghz);
    run<3>("v_cvt_pk_f16_f32", out, ghz);
    run<4>("v_fma_mixlo_f16", out, ghz);
    run<5>("v_cndmask_b32", out, ghz);
    run<6>("s_nop 1 + v_mov_dpp", out, ghz);
    run<8>("v_rsq_f32", out, ghz);
    run<10>("v_mul_f32 (SGPR src)", out, ghz);
    run<11>("v_cndmask_e64 sgpr", out, ghz);
    run<12>("v_bfi_b32", out, ghz);
    run<13>("v_cmp_gt_f32 vcc", out, ghz);
    run<16>("v_cmp + v_cndmask", out, ghz);
    run<14>("v_mov_dpp bank_mask 5", out, ghz);
    run<20>("v_add_f32_dpp", out, ghz);
    run<15>("v_and_b32", out, ghz);
    run<21>("v_ashrrev_i32", out, ghz);
    run<17>("v_perm_b32", out, ghz);
    run<18>("v_med3_f32", out, ghz);
    run<19>("v_pk_mul_f32 (2 flop)", out, ghz);
    return 0;
}
