// Micro-probe (developer tool, not part of the library): cost of a flag-based exchange step between
// co-resident workgroups, the building block of a persistent (single-launch) GRU recurrence.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_sync.hip -o /tmp/probe_sync && /tmp/probe_sync
// Each group of NB workgroups runs `rounds` rounds of: wait until every member finished the previous
// round -> read the 16 KB the group wrote in that round -> write the own 1 KB slice -> signal.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %s:%d\n", (int)e, __FILE__, __LINE__); exit(1); } } while (0)

// MODE 0: plain stores/loads + agent-scope release/acquire fences
// MODE 1: data moved with agent-scope relaxed atomics (cache-bypassing), ordering by s_waitcnt + flag
template <int MODE>
__global__ __launch_bounds__(512) void exchange_kernel(float* buf, unsigned* cnt, int rounds, int ngroups, int nb,
                                                       unsigned* errors, long long* spins) {
    const int id = blockIdx.x;
    const int group = id % ngroups, member = id / ngroups;
    const int tid = threadIdx.x;
    float* gb = buf + (long)group * 2 * nb * 256;
    unsigned* c = cnt + group * 64;
    long long nspin = 0;
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        if (tid == 0) {
            const unsigned target = (unsigned)nb * r;
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++nspin > 200000000LL) break;           // never hang the box
            }
        }
        __syncthreads();
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (r > 0) {
            const float* src = gb + (long)((r - 1) & 1) * nb * 256;
            float s = 0.f;
            for (int k = tid; k < nb * 256; k += 512) {
                float v;
                if (MODE == 0) v = src[k];
                else v = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s += v;
                if (v != (float)r) ++bad;
            }
            if (s < 0.f) errors[1] = 1;
        }
        if (tid < 256) {
            float* dst = gb + (long)(r & 1) * nb * 256 + member * 256 + tid;
            if (MODE == 0) *dst = (float)(r + 1);
            else __hip_atomic_store(dst, (float)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (MODE == 1) __builtin_amdgcn_s_waitcnt(0);      // own stores acknowledged
        __syncthreads();
        if (tid == 0) {
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (bad) atomicAdd(errors, bad);
    if (tid == 0) spins[id] = nspin;
}

// MODE 2: no counter at all -- every value travels as an 8-byte {value, round tag} word; consumers poll
// the words they need until the tag matches (one store + one load on the dependency chain).
__global__ __launch_bounds__(512) void tagged_kernel(unsigned long long* buf, int rounds, int ngroups, int nb,
                                                     unsigned* errors, long long* spins) {
    const int id = blockIdx.x;
    const int group = id % ngroups, member = id / ngroups;
    const int tid = threadIdx.x;
    unsigned long long* gb = buf + (long)group * 2 * nb * 256;
    long long nspin = 0;
    unsigned bad = 0;
    const int per = nb * 256 / 512;
    for (int r = 0; r < rounds; ++r) {
        if (r > 0) {
            const unsigned long long* src = gb + (long)((r - 1) & 1) * nb * 256;
            float s = 0.f;
            for (;;) {
                bool ok = true;
                s = 0.f;
                for (int q = 0; q < per; ++q) {
                    const unsigned long long v = __hip_atomic_load(src + tid + 512 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(v >> 32) == (unsigned)r;
                    s += __uint_as_float((unsigned)v);
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++nspin > 100000000LL) break;
            }
            if (s != (float)(per * r)) ++bad;
        }
        __syncthreads();      // stands in for the LDS split-K reduction of the real kernel
        if (tid < 256) {
            const unsigned long long w = ((unsigned long long)(unsigned)(r + 1) << 32) | __float_as_uint((float)(r + 1));
            __hip_atomic_store(gb + (long)(r & 1) * nb * 256 + member * 256 + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (bad) atomicAdd(errors, bad);
    if (tid == 0) spins[id] = nspin;
}

// MODE 3: as MODE 2, but the words are exchanged through the XCD's own L2 (sc0 = bypass the per-CU
// vector cache only).  Valid only if all members of a group sit on one XCD (block id % 8), which the
// kernel verifies from HW_REG_XCC_ID.
__device__ __forceinline__ void ld8_sc0(const unsigned long long* p, unsigned long long (&v)[8]) {
    asm volatile(
        "global_load_dwordx2 %0, %8, off offset:-4096 sc0\n"
        "global_load_dwordx2 %1, %8, off sc0\n"
        "global_load_dwordx2 %2, %9, off offset:-4096 sc0\n"
        "global_load_dwordx2 %3, %9, off sc0\n"
        "global_load_dwordx2 %4, %10, off offset:-4096 sc0\n"
        "global_load_dwordx2 %5, %10, off sc0\n"
        "global_load_dwordx2 %6, %11, off offset:-4096 sc0\n"
        "global_load_dwordx2 %7, %11, off sc0\n"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p + 512), "v"(p + 1536), "v"(p + 2560), "v"(p + 3584)
        : "memory");
}
__global__ __launch_bounds__(512) void tagged_l2_kernel(unsigned long long* buf, int rounds, int ngroups, int nb,
                                                        unsigned* errors, long long* spins, unsigned* xcc) {
    const int id = blockIdx.x;
    const int group = id % ngroups, member = id / ngroups;
    const int tid = threadIdx.x;
    unsigned xid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
    if (tid == 0) xcc[id] = xid & 15;
    unsigned long long* gb = buf + (long)group * 2 * nb * 256;
    long long nspin = 0;
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        if (r > 0) {
            const unsigned long long* src = gb + (long)((r - 1) & 1) * nb * 256;
            float s = 0.f;
            for (;;) {
                bool ok = true;
                s = 0.f;
                unsigned long long v[8];
                ld8_sc0(src + tid, v);                       // nb == 16: 4096 words, 8 per thread, stride 512
                for (int q = 0; q < 8; ++q) {
                    ok = ok && (unsigned)(v[q] >> 32) == (unsigned)r;
                    s += __uint_as_float((unsigned)v[q]);
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++nspin > 20000000LL) break;
            }
            if (s != (float)(8 * r)) ++bad;
        }
        __syncthreads();
        if (tid < 256) {
            const unsigned long long w = ((unsigned long long)(unsigned)(r + 1) << 32) | __float_as_uint((float)(r + 1));
            unsigned long long* dst = gb + (long)(r & 1) * nb * 256 + member * 256 + tid;
            asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(dst), "v"(w) : "memory");
        }
    }
    if (bad) atomicAdd(errors, bad);
    if (tid == 0) spins[id] = nspin;
}

static void run_tagged_l2(int ngroups, int rounds) {
    const int nb = 16;
    const int nblocks = ngroups * nb;
    unsigned long long* buf; unsigned* err; long long* spins; unsigned* xcc;
    CHECK(hipMalloc(&buf, (size_t)ngroups * 2 * nb * 256 * 8));
    CHECK(hipMalloc(&err, 8));
    CHECK(hipMalloc(&spins, nblocks * 8));
    CHECK(hipMalloc(&xcc, nblocks * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned herr = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemset(err, 0, 8));
        CHECK(hipMemset(buf, 0, (size_t)ngroups * 2 * nb * 256 * 8));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(tagged_l2_kernel, dim3(nblocks), dim3(512), 0, 0, buf, rounds, ngroups, nb, err, spins, xcc);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned h; CHECK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        herr += h;
    }
    std::vector<long long> hs(nblocks);
    std::vector<unsigned> hx(nblocks);
    CHECK(hipMemcpy(hs.data(), spins, nblocks * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hx.data(), xcc, nblocks * 4, hipMemcpyDeviceToHost));
    long long mx = 0; for (auto v : hs) mx = v > mx ? v : mx;
    int mism = 0;
    for (int i = 0; i < nblocks; ++i) if (hx[i] != hx[i % ngroups]) ++mism;
    printf("tagged-L2 groups %d x %d blocks, %d rounds: %.3f ms  = %.2f us/round   bad %u  max spins %lld  xcc of blocks 0..15:",
           ngroups, nb, rounds, best, 1000.f * best / rounds, herr, mx);
    for (int i = 0; i < 16 && i < nblocks; ++i) printf(" %u", hx[i]);
    printf("  group/xcc mismatches %d\n", mism);
}

static void run_tagged(int ngroups, int nb, int rounds) {
    const int nblocks = ngroups * nb;
    unsigned long long* buf; unsigned* err; long long* spins;
    CHECK(hipMalloc(&buf, (size_t)ngroups * 2 * nb * 256 * 8));
    CHECK(hipMalloc(&err, 8));
    CHECK(hipMalloc(&spins, nblocks * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned herr = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemset(err, 0, 8));
        CHECK(hipMemset(buf, 0, (size_t)ngroups * 2 * nb * 256 * 8));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(tagged_kernel, dim3(nblocks), dim3(512), 0, 0, buf, rounds, ngroups, nb, err, spins);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned h; CHECK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        herr += h;
    }
    std::vector<long long> hs(nblocks);
    CHECK(hipMemcpy(hs.data(), spins, nblocks * 8, hipMemcpyDeviceToHost));
    long long mx = 0; for (auto v : hs) mx = v > mx ? v : mx;
    printf("tagged groups %d x %d blocks, %d rounds: %.3f ms  = %.2f us/round   bad %u  max spins %lld\n",
           ngroups, nb, rounds, best, 1000.f * best / rounds, herr, mx);
}

template <int MODE>
static void run(int ngroups, int nb, int rounds) {
    const int nblocks = ngroups * nb;
    float* buf; unsigned* cnt; unsigned* err; long long* spins;
    CHECK(hipMalloc(&buf, (size_t)ngroups * 2 * nb * 256 * 4));
    CHECK(hipMalloc(&cnt, 8 * 64 * 4));
    CHECK(hipMalloc(&err, 8));
    CHECK(hipMalloc(&spins, nblocks * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned herr[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemset(cnt, 0, 8 * 64 * 4));
        CHECK(hipMemset(err, 0, 8));
        CHECK(hipMemset(buf, 0, (size_t)ngroups * 2 * nb * 256 * 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(exchange_kernel<MODE>, dim3(nblocks), dim3(512), 0, 0, buf, cnt, rounds, ngroups, nb, err, spins);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned h[2]; CHECK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
        herr[0] += h[0]; herr[1] += h[1];
    }
    std::vector<long long> hs(nblocks);
    CHECK(hipMemcpy(hs.data(), spins, nblocks * 8, hipMemcpyDeviceToHost));
    long long mx = 0; for (auto v : hs) mx = v > mx ? v : mx;
    printf("mode %d groups %d x %d blocks, %d rounds: %.3f ms  = %.2f us/round   stale reads %u  max spins %lld\n",
           MODE, ngroups, nb, rounds, best, 1000.f * best / rounds, herr[0], mx);
    (void)hipFree(buf); (void)hipFree(cnt); (void)hipFree(err); (void)hipFree(spins);
}

int main() {
    const int rounds = 256;
    run_tagged_l2(8, rounds);
    run_tagged_l2(4, rounds);
    run_tagged_l2(2, rounds);
    run_tagged_l2(1, rounds);
    run_tagged(4, 16, rounds);
    run_tagged(8, 16, rounds);
    run_tagged(4, 32, rounds);
    run_tagged(8, 32, rounds);
    run_tagged(16, 16, rounds);
    run_tagged(3, 16, rounds);
    run_tagged(1, 16, rounds);
    run<0>(4, 32, rounds);      // 4 groups: each group's 32 blocks share an XCD (block id % 8 == group)
    run<1>(4, 32, rounds);
    run<0>(8, 32, rounds);
    run<1>(8, 32, rounds);
    run<0>(4, 16, rounds);
    run<1>(4, 16, rounds);
    return 0;
}
