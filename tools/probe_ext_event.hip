// Does hipExtLaunchKernelGGL's stopEvent save the marker packet a hipEventRecord costs a stream?
// Chain of N short kernels on stream s0, each followed by "an event another stream waits for": (a) hipEventRecord behind the
// launch, (b) the event as the launch's stopEvent, (c) no event at all.  Prints us per link; also checks that a waiter on s1 really
// runs behind the kernel whose stopEvent it waited for.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void work(float* p, int n, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
    p[i] = v;
}
__global__ void stamp(const float* src, float* dst) { dst[0] = src[0]; }
int main() {
    const int n = 1 << 20, N = 200;
    float *a, *b, *c;
    hipMalloc(&a, n * 4); hipMalloc(&b, 4 * N); hipMalloc(&c, 4);
    hipMemset(a, 0, n * 4);
    hipStream_t s0, s1;
    hipStreamCreate(&s0); hipStreamCreate(&s1);
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode == 1) hipExtLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, s0, nullptr, ev[i], 0, a, n, 20);
                else hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, s0, a, n, 20);
                if (mode == 0) hipEventRecord(ev[i], s0);
                if (mode != 2) { hipStreamWaitEvent(s1, ev[i], 0); hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s1, a, b + i); }
            }
            hipDeviceSynchronize();
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("mode %d (%s): %.2f us per link\n", mode, mode == 0 ? "hipEventRecord" : mode == 1 ? "stopEvent" : "no event", us / N);
        }
    }
    // ordering check: s1's stamp behind kernel i must see the value kernel i wrote (a[0] after i+1 passes differs per i)
    hipMemset(a, 0, n * 4);
    hipDeviceSynchronize();
    for (int i = 0; i < 8; ++i) {
        hipExtLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, s0, nullptr, ev[i], 0, a, n, 2000);
        hipStreamWaitEvent(s1, ev[i], 0);
        hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s1, a, b + i);
    }
    hipDeviceSynchronize();
    float hb[8];
    hipMemcpy(hb, b, sizeof(hb), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("stamp %d: %.4f\n", i, hb[i]);
    return 0;
}
