// What ds_read_b64_tr_b16 returns (gfx950), for the weight-gradient tile that wants its contraction index transposed out of a
// row-major LDS image.  Build: hipcc --offload-arch=gfx950 -O2 tools/probe_tr16.hip -o tools/_bin/probe_tr16; run on the GPU box.
//   test 1: lane l reads the 8 bytes at element 4 l of a linear ramp (lds[e] = e)
//   test 2: lane p of every 16-lane group points at row (p >> 2), columns 4 (p & 3).. of a [rows][pitch] image with
//           lds = 1000 * row + col, group G looking at columns 16 G ..: the layout an MFMA fragment needs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int test) {
    __shared__ short lds[8192];
    const int l = threadIdx.x;
    if (test == 1) {
        for (int i = l; i < 8192; i += 64) lds[i] = (short)i;
    } else {
        for (int i = l; i < 8192; i += 64) lds[i] = (short)(1000 * (i / 128) + (i % 128));      // pitch 128 elements
    }
    __syncthreads();
    const short* src;
    if (test == 1) src = lds + 4 * l;
    else { const int G = l >> 4, p = l & 15; src = lds + (p >> 2) * 128 + 16 * G + 4 * (p & 3); }
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)src);
    for (int j = 0; j < 4; ++j) out[4 * l + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    for (int test = 1; test <= 2; ++test) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, test);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("test %d\n", test);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
