// What bounds the criterion's scoring kernel once its MFMA time is cut (round 6): the gather of 1 KB candidate rows.
//   (1) issue rate of the MFMA forms the fp16-piece version would use (cycles per instruction per SIMD, 1 / 2 waves per SIMD);
//   (2) the gather alone, shaped like the kernel -- 7424 windows x 9 tiles of 16 rows out of an 8.4 MB table, one wavefront per
//       window, ascending row lists per window -- with the rows brought in (a) by 16-byte global loads into registers (today's
//       kernel) or (b) by global_load_lds_dwordx4 into a 16 KB LDS tile per wave, followed by `nmfma` dependent-free MFMAs
//       standing for the tile's arithmetic; grid (A) one workgroup per four windows in dispatch order (today), or (P)
//       persistent: 2 workgroups per CU walking windows w, w + G, ... so that all waves sweep the table in step.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_nce_gather.hip -o tools/_bin/probe_nce_gather; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void mfma_rate_kernel(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a8, b8;
    f16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8[i] = (_Float16)(0.5f + i); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    const float af = threadIdx.x * 0.01f, bf = 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[u], 0, 0, 0);
            if (OP == 1) acc[u] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[u], 0, 0, 0);
            if (OP == 2) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[u], 0, 0, 0);
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int OP>
static void mfma_rate(const char* name, float* out, double ghz) {
    const int iters = 2048;
    for (int wps : {1, 2}) {
        const int blocks = 256 * wps;                   // 256-thread blocks: 4 waves = one per SIMD
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        const double cyc = best * 1e-3 * ghz * 1e9 / ((double)iters * 8 * wps);
        printf("%-28s waves/SIMD %d: %.3f ms -> %.1f cycles per MFMA per SIMD (at %.2f GHz)\n", name, wps, best, cyc, ghz);
    }
}

// ---- the gather
constexpr int kRow = 256;                               // floats per row (1 KB)
template <int MODE>                                      // 0: registers, 1: LDS DMA
__global__ __launch_bounds__(256, 2) void gather_kernel(const float* __restrict__ z, const int* __restrict__ ext, int BW, int N,
                                                        int nmfma, int persistent, int nt_store, float* __restrict__ out) {
    __shared__ float4 tiles[4][1024];                   // 16 KB per wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 15, r4 = lane >> 4;
    float4* tile = tiles[wv];
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(lane * 0.001f + i); b8[i] = (_Float16)(0.5f + i); }
    float keep = 0.f;
    const int stride = persistent ? gridDim.x * 4 : BW;
    for (int bt = blockIdx.x * 4 + wv; bt < BW; bt += stride) {
        for (int nt = 0; nt < N / 16; ++nt) {
            const int* e = ext + (long)bt * N + nt * 16;
            if (MODE == 0) {
                float4 v[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float* rp = z + (long)e[4 * q + r4] * kRow + 4 * c;
#pragma unroll
                    for (int g = 0; g < 4; ++g) v[q][g] = *reinterpret_cast<const float4*>(rp + 64 * g);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int g = 0; g < 4; ++g) keep += v[q][g].x + v[q][g].w;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {          // one 1 KB row per wave instruction, piece (lane ^ r) of row r lands in slot lane
                    const float* src = z + (long)e[r] * kRow + 4 * (lane ^ r);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(tile + r * 64), 16, 0, 0);
                }
                __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
                __builtin_amdgcn_wave_barrier();
                const float4 t0 = tile[c * 64 + ((4 * r4) ^ c)];
                keep += t0.x + t0.w;
            }
            for (int m = 0; m < nmfma; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[m & 3], 0, 0, 0);
        }
        // the kernel's output streams: T (12 KB per window) -- nothing else here
        float* op = out + (long)bt * 3072 + 4 * lane;
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const f32x4 o = f32x4{keep, acc[0][0], acc[1][1], acc[2][2] + acc[3][3]};
            if (nt_store) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(op + 256 * g));
            else *reinterpret_cast<f32x4*>(op + 256 * g) = o;
        }
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate * 1e-6;
    printf("%s, %d CUs, %.2f GHz\n", prop.name, prop.multiProcessorCount, ghz);
    float* out;
    CK(hipMalloc(&out, 256 * 4));
    mfma_rate<0>("v_mfma_f32_16x16x32_f16", out, ghz);
    mfma_rate<1>("v_mfma_f32_16x16x16_f16", out, ghz);
    mfma_rate<2>("v_mfma_f32_16x16x4_f32", out, ghz);

    const int B = 64, S = 128, W = 116, N = 144, BW = B * W, rows = B * S;     // 128 negatives + 16 (positives tile) per window
    std::vector<int> ext((size_t)BW * N);
    std::mt19937 rng(1);
    for (int bt = 0; bt < BW; ++bt) {
        int* e = &ext[(size_t)bt * N];
        for (int j = 0; j < N; ++j) e[j] = (int)(rng() % rows);
        std::sort(e, e + N);
    }
    float* z; int* dext; float* T;
    CK(hipMalloc(&z, (size_t)rows * kRow * 4));
    CK(hipMemset(z, 0, (size_t)rows * kRow * 4));
    CK(hipMalloc(&dext, ext.size() * 4));
    CK(hipMemcpy(dext, ext.data(), ext.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&T, (size_t)BW * 3072 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)BW * N * 1024;
    for (int mode = 0; mode < 2; ++mode)
        for (int persistent = 0; persistent < 2; ++persistent)
            for (int nmfma : {0, 24, 72, 256})
                for (int nts = 0; nts < 2; ++nts) {
                    if (nts && nmfma != 72) continue;
                    const int grid = persistent ? 512 : (BW + 3) / 4;
                    float best = 1e9f;
                    for (int rep = 0; rep < 4; ++rep) {
                        CK(hipEventRecord(e0));
                        if (mode == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(grid), dim3(256), 0, 0, z, dext, BW, N, nmfma, persistent, nts, T);
                        else hipLaunchKernelGGL(gather_kernel<1>, dim3(grid), dim3(256), 0, 0, z, dext, BW, N, nmfma, persistent, nts, T);
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        best = std::min(best, ms);
                    }
                    printf("gather %s grid %s nmfma %3d nt_store %d: %7.1f us  = %.2f TB/s of row gathers\n", mode ? "lds-dma  " : "registers",
                           persistent ? "persistent(512)" : "per-4-windows  ", nmfma, nts, best * 1e3, bytes / (best * 1e-3) / 1e12);
                }
    return 0;
}
