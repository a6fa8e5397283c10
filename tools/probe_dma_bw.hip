// What rate can a CU pull L2-resident (or HBM) data into LDS with global_load_lds_dwordx4, and what does it depend on?
// The conv DMA kernels (conv_dma.hip) run at ~10 B/clk per CU; this probe separates request shape (64 B / 128 B / 1 KB
// contiguous per row), row stride (activation rows 4 KB apart vs packed weight rows), bytes in flight, source footprint
// (every workgroup the same 2 MB = L2 hits, or private regions) and destination (LDS by DMA vs VGPRs).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_dma_bw.hip -o tools/_bin/probe_dma_bw; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt((((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14)))

struct Args {
    const unsigned char* src;
    long region;        // bytes per workgroup region (power of two)
    long wg_stride;     // bytes between the regions of neighbouring workgroups (0: all share one)
    int rowb;           // contiguous bytes per row (64, 128, 256, 1024)
    int rowstride;      // bytes between rows
    int iters;          // batches
    int sync;           // 1: wait for everything, then issue the next batch (<= DEPTH in flight); 0: one batch ahead
    unsigned* sink;
};

__device__ __forceinline__ long row_off(long row, const Args& a) {
    const long lin = row * a.rowstride;
    const long wraps = lin >> (63 - __builtin_clzl(a.region));          // region and rowstride are powers of two
    return (lin & (a.region - 1)) + ((wraps * a.rowb) & (a.rowstride - 1));
}

template <int DEPTH, int MODE>
__global__ __launch_bounds__(512) void probe(Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int ppr = a.rowb / 16, rpp = 1024 / a.rowb;
    const unsigned char* base = a.src + (long)blockIdx.x * a.wg_stride;
    unsigned char* lds = smem + wave * (2 * DEPTH) * 1024;
    unsigned acc = 0;
    long n = 0;
    auto issue = [&](int batch) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const long g = n * nw + wave;
            const long row = g * rpp + lane / ppr;
            const unsigned char* p = base + row_off(row, a) + (lane % ppr) * 16;
            if constexpr (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(lds + ((batch & 1) * DEPTH + i) * 1024), 16, 0, 0);
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(p);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
            ++n;
        }
    };
    if (a.sync) {
        for (int it = 0; it < a.iters; ++it) {
            issue(it);
            WAIT_VMCNT(0);
            __builtin_amdgcn_s_barrier();
        }
    } else {
        issue(0);
        for (int it = 1; it < a.iters; ++it) {
            issue(it);
            if constexpr (MODE == 0) {
                WAIT_VMCNT(DEPTH);
                __builtin_amdgcn_s_barrier();
            }
        }
        WAIT_VMCNT(0);
    }
    __syncthreads();
    if constexpr (MODE == 0) acc = *reinterpret_cast<const unsigned*>(smem + threadIdx.x * 16);
    if (acc == 0x12345678u) a.sink[0] = acc;
}

template <int DEPTH, int MODE>
static double run(Args a, int grid, int threads, int lds_bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<DEPTH, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((probe<DEPTH, MODE>), dim3(grid), dim3(threads), lds_bytes, 0, a);      // warm
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<DEPTH, MODE>), dim3(grid), dim3(threads), lds_bytes, 0, a);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double bytes = (double)grid * (threads / 64) * DEPTH * 1024.0 * a.iters;
    return bytes / (best * 1e-3) / 1e12;      // TB/s
}

int main() {
    const long total = 1L << 30;
    unsigned char* src;
    unsigned* sink;
    hipMalloc(&src, total);
    hipMalloc(&sink, 64);
    hipMemset(src, 1, total);
    struct Src { const char* name; long region, wg_stride; };
    const Src srcs[] = {{"shared 2 MB (L2 hits)", 2L << 20, 0}, {"private 64 KB (L2 hits)", 64L << 10, 64L << 10},
                        {"private 2 MB (HBM/MALL)", 2L << 20, 2L << 20}};
    struct Shape { const char* name; int rowb, rowstride; };
    const Shape shapes[] = {{"1 KB contiguous", 1024, 1024}, {"128 B rows packed", 128, 128}, {"128 B rows / 4 KB", 128, 4096},
                            {"64 B rows packed", 64, 64}, {"64 B rows / 4 KB", 64, 4096}, {"256 B rows / 4 KB", 256, 4096}};
    printf("TB/s chip-wide (x 1e12 / 256 CUs / 2.4e9 = B/clk/CU: 5 TB/s = 8.1)\n");
    for (const Src& s : srcs)
        for (const Shape& sh : shapes) {
            Args a{src, s.region, s.wg_stride, sh.rowb, sh.rowstride, 0, 0, sink};
            printf("%-24s %-20s", s.name, sh.name);
            // 256 workgroups x 8 waves, LDS DMA, one batch ahead, depth 4 / 8 / 16 per wave (32 / 64 / 128 KB per CU and batch)
            a.iters = 2048; a.sync = 0;
            printf(" | dma ahead d4 %5.2f", run<4, 0>(a, 256, 512, 8 * 8 * 1024));
            a.iters = 1024;
            printf(" d8 %5.2f", run<8, 0>(a, 256, 512, 8 * 16 * 1024));
            a.iters = 1024; a.sync = 1;
            printf(" | dma sync d8 %5.2f", run<8, 0>(a, 256, 512, 8 * 16 * 1024));
            a.iters = 2048; a.sync = 0;
            printf(" | 2 wg/cu d4 %5.2f", run<4, 0>(a, 512, 512, 8 * 8 * 1024));
            a.iters = 1024;
            printf(" | vgpr d8 %5.2f", run<8, 1>(a, 256, 512, 1024));
            printf(" vgpr d8 x2wg %5.2f", run<8, 1>(a, 512, 512, 1024));
            printf("\n");
        }
    return 0;
}
