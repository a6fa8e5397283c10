// What does the SHAPE of a streamed store cost?  conv0's forward writes 268 MB of activation rows (1 KB each) per launch at
// B = 64; the scalar kernel of rounds 1-2 wrote one whole row per wave instruction, the MFMA kernel's accumulator layout gives a
// lane 4 consecutive channels of 4 rows, i.e. four 256-byte pieces of four rows (4 KB apart) per instruction.  This probe writes
// the same 268 MB with either shape, nontemporal or not, 16 rows per wave and `groups` such blocks per wave, and reports GB/s.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_store_pattern.hip -o tools/_bin/probe_store_pattern; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: instruction i of a 16-row block writes row i (64 lanes x 16 B = 1 KB contiguous)
// MODE 1: instruction (r, q) writes piece q (256 B) of rows r, 4 + r, 8 + r, 12 + r   (the MFMA accumulator layout)
// MODE 2: as 1, instructions ordered q-fastest inside a row quad (rows complete sooner)
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ y, long rows, int groups) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const long g0 = ((long)blockIdx.x * 4 + wv) * groups;
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    for (int gi = 0; gi < groups; ++gi) {
        const long r0 = (g0 + gi) * 16;
        if (r0 >= rows) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4* p;
            if (MODE == 0) p = reinterpret_cast<f32x4*>(y + (r0 + i) * 256) + lane;
            else {
                const int r = MODE == 1 ? i >> 2 : i & 3, q = MODE == 1 ? i & 3 : i >> 2;
                p = reinterpret_cast<f32x4*>(y + (r0 + 4 * kq + r) * 256 + 64 * q) + n;
            }
            v.x += 1.f;
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

template <int MODE, bool NT>
static void run(float* y, long rows, int groups, const char* name) {
    const long waves = (rows / 16 + groups - 1) / groups;
    const int blocks = (int)((waves + 3) / 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 8; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((store_kernel<MODE, NT>), dim3(blocks), dim3(256), 0, 0, y, rows, groups);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2 && ms < best) best = ms;
    }
    printf("%-44s groups %2d: %7.1f us  %6.0f GB/s\n", name, groups, best * 1e3, rows * 1024.0 / best / 1e6);
}

int main() {
    const long rows = 64L * 4096;
    float* y;
    hipMalloc(&y, rows * 1024);
    for (int groups : {1, 4, 8}) {
        run<0, true>(y, rows, groups, "row per instruction, nontemporal");
        run<0, false>(y, rows, groups, "row per instruction");
        run<1, true>(y, rows, groups, "4 x 256 B per instruction (r-major), nt");
        run<1, false>(y, rows, groups, "4 x 256 B per instruction (r-major)");
        run<2, true>(y, rows, groups, "4 x 256 B per instruction (q-major), nt");
    }
    return 0;
}
