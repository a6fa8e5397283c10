// Micro-probe (developer tool): can workgroups of ONE XCD hand data over through that XCD's L2 -- without the trip over the
// fabric that device-scope (sc1) accesses take -- and how do the two kinds of hand-over behave while another kernel streams
// through HBM?  Same protocol as probe_sync3 (buffer pre-filled with 0xFFFFFFFF, producers write finite floats, consumers poll
// 16-byte fragments); group = blockIdx % 8 = one XCD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_sync4.hip -o tools/_bin/probe_sync4 && tools/_bin/probe_sync4
// V0: sc1 loads, sc1 stores (what gru.hip does)
// V4: "loads" = returning 64-bit atomic OR of zero (executed in the L2), plain stores (the vector L1 is write-through)
// V5: sc0 nt loads, plain stores
// V6: sc1 loads, plain stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %s:%d\n", (int)e, __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

template <int V>
__device__ __forceinline__ f32x4 ld16(const float* p) {
    f32x4 v;
    if (V == 0 || V == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (V == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else {
        u64 lo, hi; const u64 zero = 0;
        asm volatile("global_atomic_or_x2 %0, %2, %3, off sc0\n global_atomic_or_x2 %1, %2, %3, off offset:8 sc0\n s_waitcnt vmcnt(0)"
                     : "=&v"(lo), "=&v"(hi) : "v"(p), "v"(zero) : "memory");
        v[0] = __uint_as_float((unsigned)lo); v[1] = __uint_as_float((unsigned)(lo >> 32));
        v[2] = __uint_as_float((unsigned)hi); v[3] = __uint_as_float((unsigned)(hi >> 32));
    }
    return v;
}
template <int V>
__device__ __forceinline__ void st4(float* p, float v) {
    if (V == 0) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}

template <int V, int NLD>
__global__ __launch_bounds__(512) void xcd_kernel(float* buf, int rounds, int nb, int vpb, unsigned* errors,
                                                  long long* spins, unsigned* xcc, int active_groups) {
    extern __shared__ float own[];
    if (rounds < 0) own[threadIdx.x] = 0.f;
    const int id = blockIdx.x;
    const int group = id % 8, member = id / 8;
    const int tid = threadIdx.x;
    unsigned xid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
    if (tid == 0) xcc[id] = xid & 15;
    if (group >= active_groups) return;
    const long per_round = (long)nb * vpb;
    float* gb = buf + (long)group * rounds * per_round;
    long long nspin = 0;
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        if (r > 0) {
            const float* src = gb + (long)(r - 1) * per_round;
            f32x4 v[NLD];
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < NLD; ++q) v[q] = ld16<V>(src + 4 * (tid + 512 * q));
#pragma unroll
                for (int q = 0; q < NLD; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ok = ok && __float_as_uint(v[q][e]) != 0xFFFFFFFFu;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++nspin > 200000LL) break;
            }
#pragma unroll
            for (int q = 0; q < NLD; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) if (v[q][e] != (float)r) ++bad;
        }
        __syncthreads();
        for (int k = tid; k < vpb; k += 512) st4<V>(gb + (long)r * per_round + member * vpb + k, (float)(r + 1));
    }
    if (bad) atomicAdd(errors, bad);
    if (tid == 0) spins[id] = nspin;
}

// 64 KB of LDS per workgroup, so that it cannot share a CU with a probe workgroup holding 128 KB (loaded == 2)
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ a, float4* __restrict__ b, long n, int passes) {
    extern __shared__ float pad[];
    if (n < 0) pad[threadIdx.x] = 0.f;
    for (int p = 0; p < passes; ++p)
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) b[i] = a[i];
}

template <int V, int NLD>
static void run(const char* what, int active_groups, int nb, int rounds, int loaded) {
    const int vpb = NLD * 512 * 4 / nb;
    const int nblocks = 8 * nb;
    const size_t bytes = (size_t)8 * rounds * nb * vpb * 4;
    float* buf; unsigned* err; long long* spins; unsigned* xcc;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&err, 8)); CHECK(hipMalloc(&spins, nblocks * 8)); CHECK(hipMalloc(&xcc, nblocks * 4));
    static float4 *sa = nullptr, *sb = nullptr; static hipStream_t side;
    const long sn = (256L << 20) / 16;
    if (!sa) { CHECK(hipMalloc(&sa, sn * 16)); CHECK(hipMalloc(&sb, sn * 16)); CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking)); }
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f; unsigned herr = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(err, 0, 8)); CHECK(hipMemset(buf, 0xFF, bytes)); CHECK(hipMemset(spins, 0, nblocks * 8));
        CHECK(hipDeviceSynchronize());
        if (loaded) hipLaunchKernelGGL(stream_kernel, dim3(1024), dim3(256), loaded == 2 ? 65536 : 0, side, sa, sb, sn, 40);   // ~ 5 ms of copying
        CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((xcd_kernel<V, NLD>), dim3(nblocks), dim3(512), loaded == 2 ? 131072 : 0, st, buf, rounds, nb, vpb, err, spins, xcc, active_groups);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned h; CHECK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost)); herr += h;
    }
    std::vector<long long> hs(nblocks); std::vector<unsigned> hx(nblocks);
    CHECK(hipMemcpy(hs.data(), spins, nblocks * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hx.data(), xcc, nblocks * 4, hipMemcpyDeviceToHost));
    long long mx = 0; for (auto v : hs) mx = v > mx ? v : mx;
    int mism = 0; for (int i = 0; i < nblocks; ++i) if (hx[i] != hx[i % 8]) ++mism;
    printf("V%d %-24s %s %d groups x %2d blocks, %2d x16B/lane: %6.2f us/round  bad %u  max spins %lld  same-xcd violations %d\n", V, what,
           loaded == 2 ? "HBM copy on other CUs" : loaded ? "beside an HBM copy   " : "alone                ", active_groups, nb, NLD, 1000.f * best / rounds, herr, mx, mism);
    (void)hipFree(buf); (void)hipFree(err); (void)hipFree(spins); (void)hipFree(xcc); (void)hipStreamDestroy(st);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    CHECK(hipFuncSetAttribute((const void*)stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
#define BIG_LDS(V, N) CHECK(hipFuncSetAttribute((const void*)xcd_kernel<V, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072))
    BIG_LDS(0, 2); BIG_LDS(0, 4); BIG_LDS(6, 4); BIG_LDS(4, 2); BIG_LDS(4, 4); BIG_LDS(5, 4);
    const int R = 512;
    for (int loaded = 0; loaded < 3; ++loaded) {
        run<0, 2>("sc1 / sc1", 4, 16, R, loaded);
        run<0, 4>("sc1 / sc1", 4, 32, R, loaded);
        run<6, 4>("sc1 / plain store", 4, 32, R, loaded);
        run<4, 2>("L2 atomic / plain store", 4, 16, R, loaded);
        run<4, 4>("L2 atomic / plain store", 4, 32, R, loaded);
        run<4, 4>("L2 atomic / plain store", 8, 32, R, loaded);
        run<5, 4>("sc0 nt / plain store", 4, 32, R, loaded);
    }
    return 0;
}
