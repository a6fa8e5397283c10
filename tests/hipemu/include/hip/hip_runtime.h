// hipemu -- a tiny host-side SIMT interpreter for HIP kernels.
//
// TEST TOOL ONLY.  It exists so that the index / layout logic of the gfx950
// kernels in cpc_audio_amd/csrc can be unit-tested in a container with no GPU:
// the SAME .hip sources are compiled for x86 with this directory first on the
// include path, so `#include <hip/hip_runtime.h>` resolves here.  The product
// package never loads the emulated library (cpc_audio_amd/_lib.py only ever opens
// libcpc_hip.so built by hipcc for gfx950); only tests/test_emu_*.py do.
//
// Model: every HIP thread of a block is a user-level fiber; the fibers of one
// block run on one OS thread in lane order, each until it reaches a barrier or a
// wave collective (shuffle / MFMA).  Different blocks run on different OS threads.
// `__shared__` becomes `static thread_local` (one copy per OS thread == per
// resident block).  Wave size is 64.  MFMA fragment layouts follow
// /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIPEMU 1

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(8) uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

typedef struct ihipStream_t* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
// events: every launch of the emulator completes before it returns, so ordering between streams is trivially kept
typedef struct ihipEvent_t* hipEvent_t;
#define hipEventDisableTiming 2
#define hipEventDisableSystemFence 0x20000000
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }

namespace hipemu {

enum { RUNNABLE = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };
constexpr int WAVE = 64;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int state = DONE;
    int lin = 0;          // linear thread id in block
    dim3 tid;
};

struct WaveBuf {          // double-buffered exchange area for collectives
    uint32_t a[2][WAVE];
    uint32_t b[2][WAVE];
    uint16_t va[2][WAVE][8];   // 8 x 16-bit operand fragments (bf16 MFMA)
    uint16_t vb[2][WAVE][8];
    int phase = 0;        // parity of the collective currently being deposited
    int waiting = 0;
    int live = 0;
};

struct BlockCtx {
    dim3 bid, bdim, gdim;
    int nthreads = 0;
    int block_waiting = 0;
    int live = 0;
    std::vector<Fiber> fibers;
    std::vector<WaveBuf> waves;
    void* main_sp = nullptr;
    const std::function<void()>* body = nullptr;
};

extern thread_local BlockCtx* blk;
extern thread_local Fiber* cur;

void yield_to_main();           // implemented in hipemu.cpp
void launch_impl(dim3 grid, dim3 block, const std::function<void()>& body);

inline int lane_id() { return cur->lin & (WAVE - 1); }
inline WaveBuf& my_wave() { return blk->waves[cur->lin / WAVE]; }

// all live lanes of the calling wave rendezvous here
inline void wave_sync() {
    WaveBuf& w = my_wave();
    cur->state = WAIT_WAVE;
    yield_to_main();
}

inline void block_sync() {
    cur->state = WAIT_BLOCK;
    yield_to_main();
}

template <class T> inline uint32_t bits(T v) { static_assert(sizeof(T) == 4, "32-bit only"); uint32_t u; memcpy(&u, &v, 4); return u; }
template <class T> inline T unbits(uint32_t u) { T v; memcpy(&v, &u, 4); return v; }

// deposit one 32-bit value per lane, rendezvous, return the buffer to read from
inline const uint32_t* exchange(uint32_t v) {
    WaveBuf& w = my_wave();
    int ph = w.phase;                 // all lanes see the same phase for this collective
    w.a[ph][lane_id()] = v;
    wave_sync();
    return w.a[ph];
}

template <class T> inline T shfl_idx(T v, int src) {
    const uint32_t* buf = exchange(bits(v));
    return unbits<T>(buf[src & (WAVE - 1)]);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

inline f32x16 mfma_f32_32x32x2f32(float a, float b, f32x16 c, int, int, int) {
    WaveBuf& w = my_wave();
    int ph = w.phase, l = lane_id();
    w.a[ph][l] = bits(a);
    w.b[ph][l] = bits(b);
    wave_sync();
    const uint32_t* A = w.a[ph];      // A[i][k] = A[i + 32k]
    const uint32_t* B = w.b[ph];      // B[k][j] = B[j + 32k]
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k)
            acc = fmaf(unbits<float>(A[row + 32 * k]), unbits<float>(B[col + 32 * k]), acc);
        c[r] = acc;
    }
    return c;
}

inline f32x4 mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
    WaveBuf& w = my_wave();
    int ph = w.phase, l = lane_id();
    w.a[ph][l] = bits(a);
    w.b[ph][l] = bits(b);
    wave_sync();
    const uint32_t* A = w.a[ph];      // A[i][k] = A[i + 16k]
    const uint32_t* B = w.b[ph];      // B[k][j] = B[j + 16k]
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k)
            acc = fmaf(unbits<float>(A[row + 16 * k]), unbits<float>(B[col + 16 * k]), acc);
        c[r] = acc;
    }
    return c;
}

// DPP data movement (the subset the kernels use) and v_readlane, as wave collectives
inline int dpp_src_lane(int ctrl, int l) {
    const int row = l & ~15, i = l & 15;
    if (ctrl >= 0 && ctrl <= 0xFF) return (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);       // quad_perm
    if (ctrl == 0x140) return row | (15 - i);                                              // row_mirror
    if (ctrl == 0x141) return row | (i < 8 ? 7 - i : 23 - i);                              // row_half_mirror
    fprintf(stderr, "hipemu: unsupported dpp_ctrl 0x%x\n", ctrl);
    abort();
}
// row_mask bit r enables the lanes of row r (16 lanes), bank_mask bit k bank k of every row (its lanes 4k .. 4k+3); a disabled
// lane keeps `old`
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool) {
    const uint32_t* buf = exchange((uint32_t)src);
    const int l = lane_id();
    const bool on = ((row_mask >> (l >> 4)) & 1) && ((bank_mask >> ((l >> 2) & 3)) & 1);
    return on ? (int)buf[dpp_src_lane(ctrl, l)] : old;
}
inline int readlane(int v, int lane) {
    const uint32_t* buf = exchange((uint32_t)v);
    return (int)buf[lane & (WAVE - 1)];
}

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

inline float bf16_bits_to_float(uint16_t h) { return unbits<float>((uint32_t)h << 16); }

// v_mfma_f32_32x32x16_bf16: lane l supplies A[i = l&31][k = 8*(l>>5) + e], B[k = 8*(l>>5) + e][j = l&31]
inline f32x16 mfma_f32_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c, int, int, int) {
    WaveBuf& w = my_wave();
    int ph = w.phase, l = lane_id();
    memcpy(w.va[ph][l], &a, 16);
    memcpy(w.vb[ph][l], &b, 16);
    wave_sync();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k)
            acc = fmaf(bf16_bits_to_float(w.va[ph][row + 32 * (k >> 3)][k & 7]),
                       bf16_bits_to_float(w.vb[ph][col + 32 * (k >> 3)][k & 7]), acc);
        c[r] = acc;
    }
    return c;
}

// v_mfma_f32_16x16x32_f16: lane l supplies A[i = l&15][k = 8*(l>>4) + e], B[k = 8*(l>>4) + e][j = l&15];
// D as for 16x16x4: c[r] = D[4*(l>>4) + r][l&15]
typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
inline f32x4 mfma_f32_16x16x32_f16(f16x8_ a, f16x8_ b, f32x4 c, int, int, int) {
    WaveBuf& w = my_wave();
    int ph = w.phase, l = lane_id();
    memcpy(w.va[ph][l], &a, 16);
    memcpy(w.vb[ph][l], &b, 16);
    wave_sync();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            _Float16 x, y;
            memcpy(&x, &w.va[ph][row + 16 * (k >> 3)][k & 7], 2);
            memcpy(&y, &w.vb[ph][col + 16 * (k >> 3)][k & 7], 2);
            acc = fmaf((float)x, (float)y, acc);
        }
        c[r] = acc;
    }
    return c;
}

// v_mfma_f32_16x16x16_f16: lane l supplies A[i = l&15][k = 4*(l>>4) + e], B[k = 4*(l>>4) + e][j = l&15] (four halves each);
// D as for 16x16x4
typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
inline f32x4 mfma_f32_16x16x16f16(f16x4_ a, f16x4_ b, f32x4 c, int, int, int) {
    WaveBuf& w = my_wave();
    int ph = w.phase, l = lane_id();
    memcpy(w.va[ph][l], &a, 8);
    memcpy(w.vb[ph][l], &b, 8);
    wave_sync();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            _Float16 x, y;
            memcpy(&x, &w.va[ph][row + 16 * (k >> 2)][k & 3], 2);
            memcpy(&y, &w.vb[ph][col + 16 * (k >> 2)][k & 3], 2);
            acc = fmaf((float)x, (float)y, acc);
        }
        c[r] = acc;
    }
    return c;
}

// v_mfma_f32_32x32x16_f16: same operand layout with IEEE half elements
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
inline f32x16 mfma_f32_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c, int, int, int) {
    WaveBuf& w = my_wave();
    int ph = w.phase, l = lane_id();
    memcpy(w.va[ph][l], &a, 16);
    memcpy(w.vb[ph][l], &b, 16);
    wave_sync();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            _Float16 x, y;
            memcpy(&x, &w.va[ph][row + 32 * (k >> 3)][k & 7], 2);
            memcpy(&y, &w.vb[ph][col + 32 * (k >> 3)][k & 7], 2);
            acc = fmaf((float)x, (float)y, acc);
        }
        c[r] = acc;
    }
    return c;
}

}  // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::blk->bid)
#define blockDim (hipemu::blk->bdim)
#define gridDim (hipemu::blk->gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::block_sync(); }
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int l = hipemu::lane_id();
    int base = l & ~(width - 1);
    return hipemu::shfl_idx(v, base + (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    return hipemu::shfl_idx(v, hipemu::lane_id() ^ mask);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int src = l + (int)d;
    if ((src & ~(width - 1)) != (l & ~(width - 1))) src = l;
    return hipemu::shfl_idx(v, src);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int src = l - (int)d;
    if (src < 0 || (src & ~(width - 1)) != (l & ~(width - 1))) src = l;
    return hipemu::shfl_idx(v, src);
}

static inline float atomicAdd(float* p, float v) {
    uint32_t* u = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
    for (;;) {
        float f;
        memcpy(&f, &old, 4);
        f += v;
        uint32_t nu;
        memcpy(&nu, &f, 4);
        if (__atomic_compare_exchange_n(u, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            float r;
            memcpy(&r, &old, 4);
            return r;
        }
    }
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline void unsafeAtomicAdd(float* p, float v) { (void)atomicAdd(p, v); }
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
#define HIP_SYMBOL(x) x
template <class T> static inline hipError_t hipMemcpyFromSymbol(void* dst, const T& sym, size_t n) { memcpy(dst, &sym, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyToSymbol(T& sym, const void* src, size_t n) { memcpy(&sym, src, n); return hipSuccess; }

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }

#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu::mfma_f32_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu::mfma_f32_16x16x4f32
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 hipemu::mfma_f32_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 hipemu::mfma_f32_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 hipemu::mfma_f32_16x16x32_f16
#define __builtin_amdgcn_mfma_f32_16x16x16f16 hipemu::mfma_f32_16x16x16f16
static inline unsigned __float_as_uint(float x) { return hipemu::bits(x); }
static inline float __uint_as_float(unsigned u) { return hipemu::unbits<float>(u); }
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_update_dpp hipemu::update_dpp
#define __builtin_amdgcn_readlane hipemu::readlane
// ---- inter-workgroup exchange support (persistent kernels): agent-scope atomics are host atomics,
// s_sleep hands the OS thread to the other (co-resident) blocks.
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
template <class T> static inline T hipemu_atomic_load(const T* p, int order) { T v; __atomic_load(p, &v, order); return v; }
template <class T, class V> static inline void hipemu_atomic_store(T* p, V v, int order) { T t = (T)v; __atomic_store(p, &t, order); }
#define __hip_atomic_load(p, order, scope) hipemu_atomic_load((p), (order))
#define __hip_atomic_fetch_and(p, v, order, scope) __atomic_fetch_and((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
static inline unsigned __builtin_amdgcn_s_getreg(int) { return 0u; }     // hardware registers (XCC_ID ...): one XCD, id 0
#define __hip_atomic_store(p, v, order, scope) hipemu_atomic_store((p), (v), (order))
static inline void __builtin_amdgcn_s_sleep(int) { sched_yield(); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline bool __all(bool pred) {
    const uint32_t* buf = hipemu::exchange(pred ? 1u : 0u);
    const int lo = 0, hi = hipemu::WAVE;
    bool r = true;
    const int base = (hipemu::cur->lin / hipemu::WAVE) * hipemu::WAVE;
    for (int l = lo; l < hi; ++l)
        if (base + l < hipemu::blk->nthreads) r = r && buf[l] != 0;
    return r;
}
static inline bool __any(bool pred) {
    const uint32_t* buf = hipemu::exchange(pred ? 1u : 0u);
    bool r = false;
    const int base = (hipemu::cur->lin / hipemu::WAVE) * hipemu::WAVE;
    for (int l = 0; l < hipemu::WAVE; ++l)
        if (base + l < hipemu::blk->nthreads) r = r || buf[l] != 0;
    return r;
}
static inline unsigned long long __ballot(bool pred) {
    const uint32_t* buf = hipemu::exchange(pred ? 1u : 0u);
    unsigned long long r = 0;
    const int base = (hipemu::cur->lin / hipemu::WAVE) * hipemu::WAVE;
    for (int l = 0; l < hipemu::WAVE; ++l)
        if (base + l < hipemu::blk->nthreads && buf[l] != 0) r |= 1ull << l;
    return r;
}
// device queries: the "device" has as many CUs as the emulator has worker threads
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
namespace hipemu { int worker_count(); }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = hipemu::worker_count(); return hipSuccess; }
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 1; return hipSuccess; }
#define __builtin_amdgcn_wave_barrier() hipemu::wave_sync()
#define __builtin_amdgcn_s_setprio(x) ((void)0)
// LDS DMA: lane l's `size` bytes land at (wave-uniform) lds base + l * size, synchronously in the emulator
namespace hipemu {
static inline void global_load_lds(const void* g, void* lds_base, unsigned size, int offset) {
    memcpy(reinterpret_cast<char*>(lds_base) + offset + (size_t)(threadIdx.x & 63) * size, g, size);
}
}  // namespace hipemu
// ds_read_b64_tr_b16 (gfx950): every lane reads 8 bytes at its own address; inside each group of 16 lanes the 16 x 4 halves
// are handed out transposed -- lane q of the group receives, for j = 0..3, element (q & 3) of source lane 4 j + (q >> 2)
// (measured on MI355X, tools/probe_tr16.hip: with source lane p pointing at row p >> 2, columns 4 (p & 3).. of a row-major
// image, lane q gets rows 0..3 of column q)
typedef short s16x4 __attribute__((ext_vector_type(4)));
namespace hipemu {
static inline s16x4 ds_read_tr16_b64(const void* lane_addr) {
    WaveBuf& w = my_wave();
    const int ph = w.phase, l = lane_id();
    memcpy(w.va[ph][l], lane_addr, 8);
    wave_sync();
    const int g0 = l & ~15, q = l & 15;
    s16x4 out;
    for (int j = 0; j < 4; ++j) {
        short e[4];
        memcpy(e, w.va[ph][g0 + 4 * j + (q >> 2)], 8);
        out[j] = e[q & 3];
    }
    return out;
}
}  // namespace hipemu
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipemu::ds_read_tr16_b64((const void*)(p))
#define __builtin_amdgcn_global_load_lds(g, l, size, offset, aux) \
    hipemu::global_load_lds((const void*)(g), (void*)(l), (size), (offset))
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_barrier() hipemu::block_sync()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...)                  \
    do {                                                                            \
        (void)(smem); (void)(stream);                                               \
        hipemu::launch_impl((grid), (block), [&]() { kernel(__VA_ARGS__); });       \
    } while (0)
