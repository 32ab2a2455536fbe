// TEST INFRASTRUCTURE ONLY -- never part of the shipped library.
//
// A tiny functional emulator of the HIP execution model (grid / block /
// 64-lane wave / LDS / __syncthreads / wave shuffles / f32 MFMA) so the *same*
// kernel sources under lungmask_amd/csrc can be compiled by g++ and exercised
// at small sizes in the GPU-less dev container (`pytest -m "not gpu"`).
// One fibre per GPU thread, cooperative scheduling, barriers yield.  Blocks of
// a launch run in parallel on OS threads (OpenMP).  It models semantics, not
// timing.  The product library (`liblungmask_hip.so`) is always built by hipcc
// for gfx950 and never contains this file; `lungmask_amd._native` refuses to
// load anything else.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define LM_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__ __restrict
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct lm_emu_idx {
    unsigned x, y, z;
};
inline thread_local lm_emu_idx threadIdx, blockIdx;
struct alignas(16) float4 {
    float x, y, z, w;
};
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct alignas(8) float2 {
    float x, y;
};
struct alignas(8) int2 {
    int x, y;
};
struct alignas(16) int4 {
    int x, y, z, w;
};
struct alignas(16) uint4 {
    unsigned x, y, z, w;
};
inline thread_local dim3 blockDim, gridDim;

// ------------------------------------------------------------------ runtime
typedef int hipError_t;
typedef void* hipStream_t;
struct lm_emu_event {
    std::chrono::steady_clock::time_point t;
};
typedef lm_emu_event* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
inline const char* hipGetErrorString(hipError_t e) { return e == 0 ? "success" : "emu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
constexpr unsigned hipHostRegisterDefault = 0;
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipErrorInvalidValue; }  // the emulator takes the page-touch branch
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) {
    *p = nullptr;
    if (posix_memalign(p, 256, n ? n : 256) != 0) return hipErrorOutOfMemory;
    memset(*p, 0xCD, n);  // poison: kernels must not rely on zeroed memory
    return hipSuccess;
}
template <class T>
inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t r = 0; r < height; ++r) memcpy(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
    return hipSuccess;
}
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipStreamDefault = 0, hipDeviceAttributeMultiprocessorCount = 1 };
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 4; return hipSuccess; }  // 4 "CUs": persistent kernels walk several items per block
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { static int dummy2; *s = &dummy2; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static int dummy; *s = &dummy; return hipSuccess; }  // (non-null: "created")
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new lm_emu_event(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

// ------------------------------------------------------------------ fibres
namespace lm_emu {

extern "C" void lm_emu_switch(void** save_sp, void* load_sp);

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
    lm_emu_idx tid{};
};

struct Barrier {
    unsigned count = 0, gen = 0;
};

struct WaveState {
    Barrier bar;
    float a[2][64], b[2][64];
    unsigned long long u64[2][64];
    int parity = 0;
};

struct BlockCtx {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    Barrier block_bar;
    unsigned nthreads = 0, alive = 0;
    void* sched_sp = nullptr;
    int cur = -1;
    const std::function<void()>* body = nullptr;
    std::vector<char> dyn_smem;
    static constexpr size_t kStack = 256 * 1024;
};

inline thread_local BlockCtx* g_ctx = nullptr;

inline void yield() {
    BlockCtx* c = g_ctx;
    Fiber& f = c->fibers[c->cur];
    lm_emu_switch(&f.sp, c->sched_sp);
    threadIdx = f.tid;  // restored on resume
}

inline void barrier_wait(Barrier& b, unsigned expected) {
    unsigned gen = b.gen;
    if (++b.count >= expected) {
        b.count = 0;
        b.gen++;
    } else {
        while (b.gen == gen) yield();
    }
}

inline unsigned linear_tid() { return threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); }
inline WaveState& my_wave() { return g_ctx->waves[linear_tid() >> 6]; }
inline unsigned wave_width() {
    unsigned w = linear_tid() >> 6;
    unsigned n = g_ctx->nthreads - w * 64;
    return n < 64 ? n : 64;
}
inline void wave_sync() { barrier_wait(my_wave().bar, wave_width()); }

extern "C" inline void lm_emu_fiber_entry() {
    BlockCtx* c = g_ctx;
    Fiber& f = c->fibers[c->cur];
    threadIdx = f.tid;
    (*c->body)();
    f.done = true;
    c->alive--;
    lm_emu_switch(&f.sp, c->sched_sp);
    abort();
}

inline void run_block(BlockCtx& c, const std::function<void()>& body, dim3 block) {
    g_ctx = &c;
    c.nthreads = block.x * block.y * block.z;
    if (c.fibers.size() < c.nthreads) c.fibers.resize(c.nthreads);
    c.waves.assign((c.nthreads + 63) / 64, WaveState());
    c.block_bar = Barrier();
    c.alive = c.nthreads;
    c.body = &body;
    for (unsigned t = 0; t < c.nthreads; ++t) {
        Fiber& f = c.fibers[t];
        if (!f.stack) {
            void* p = nullptr;
            if (posix_memalign(&p, 64, BlockCtx::kStack) != 0) abort();
            f.stack = (char*)p;
        }
        f.done = false;
        f.tid.x = t % block.x;
        f.tid.y = (t / block.x) % block.y;
        f.tid.z = t / (block.x * block.y);
        uintptr_t top = ((uintptr_t)f.stack + BlockCtx::kStack) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;                        // fake return address of the entry
        *--sp = (void*)&lm_emu_fiber_entry;     // `ret` target
        for (int i = 0; i < 6; ++i) *--sp = nullptr;  // rbp rbx r12..r15
        f.sp = sp;
    }
    while (c.alive) {
        for (unsigned t = 0; t < c.nthreads; ++t) {
            if (c.fibers[t].done) continue;
            c.cur = (int)t;
            lm_emu_switch(&c.sched_sp, c.fibers[t].sp);
        }
    }
}

inline char* dyn_smem() { return g_ctx->dyn_smem.data(); }

inline void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    const long nblocks = (long)grid.x * grid.y * grid.z;
#pragma omp parallel
    {
        static thread_local BlockCtx ctx;
        ctx.dyn_smem.resize(std::max<size_t>(smem, 16) + 64);
#pragma omp for schedule(dynamic, 1)
        for (long b = 0; b < nblocks; ++b) {
            gridDim = grid;
            blockDim = block;
            blockIdx.x = (unsigned)(b % grid.x);
            blockIdx.y = (unsigned)((b / grid.x) % grid.y);
            blockIdx.z = (unsigned)(b / ((long)grid.x * grid.y));
            run_block(ctx, body, block);
        }
    }
}

}  // namespace lm_emu

#define LM_LAUNCH(kernel, grid, block, smem, stream, ...) \
    lm_emu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
#define LM_DYN_SMEM(name) char* name = (char*)(((uintptr_t)lm_emu::dyn_smem() + 63) & ~(uintptr_t)63)

inline void __syncthreads() { lm_emu::barrier_wait(lm_emu::g_ctx->block_bar, lm_emu::g_ctx->alive); }

// ------------------------------------------------------------------ atomics (global + LDS)
template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float atomicAdd(float* p, float v) {
    float old = *p, nv;
    do { nv = old + v; } while (!__atomic_compare_exchange(p, &old, &nv, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
}
template <class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicCAS(T* p, T cmp, T v) {
    __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return cmp;
}
template <class T> inline T atomicMin(T* p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
template <class T> inline T atomicMax(T* p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ------------------------------------------------------------------ bit / math intrinsics
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline unsigned __brev(unsigned v) {
    unsigned r = 0;
    for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
using std::max;
using std::min;

// ------------------------------------------------------------------ wave-level (64 lanes)
inline unsigned long long lm_emu_exchange_u64(unsigned long long v, int src_lane) {
    lm_emu::WaveState& w = lm_emu::my_wave();
    const int lane = lm_emu::linear_tid() & 63;
    w.u64[0][lane] = v;
    lm_emu::wave_sync();
    unsigned long long r = w.u64[0][src_lane & 63];
    lm_emu::wave_sync();
    return r;
}
template <class T> inline T __shfl(T v, int src, int = 64) {
    unsigned long long u = 0;
    memcpy(&u, &v, sizeof(T));
    u = lm_emu_exchange_u64(u, src);
    T r;
    memcpy(&r, &u, sizeof(T));
    return r;
}
template <class T> inline T __shfl_xor(T v, int m, int = 64) { return __shfl(v, (int)((lm_emu::linear_tid() & 63) ^ m)); }
template <class T> inline T __shfl_down(T v, unsigned d, int = 64) {
    int lane = lm_emu::linear_tid() & 63;
    return __shfl(v, lane + (int)d < 64 ? lane + (int)d : lane);
}
template <class T> inline T __shfl_up(T v, unsigned d, int = 64) {
    int lane = lm_emu::linear_tid() & 63;
    return __shfl(v, lane >= (int)d ? lane - (int)d : lane);
}
inline unsigned long long __ballot(int pred) {
    lm_emu::WaveState& w = lm_emu::my_wave();
    const int lane = lm_emu::linear_tid() & 63;
    w.u64[0][lane] = pred ? 1ull : 0ull;
    lm_emu::wave_sync();
    unsigned long long r = 0;
    const unsigned n = lm_emu::wave_width();
    for (unsigned i = 0; i < n; ++i) r |= w.u64[0][i] << i;
    lm_emu::wave_sync();
    return r;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) {
    unsigned n = lm_emu::wave_width();
    unsigned long long full = n == 64 ? ~0ull : ((1ull << n) - 1);
    return __ballot(pred) == full;
}

// ------------------------------------------------------------------ MFMA (f32 in / f32 acc)
typedef float lm_f32x16 __attribute__((vector_size(64)));
typedef float lm_f32x4 __attribute__((vector_size(16)));

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D reg r of lane l: col j=l&31, row i=(r&3)+8*(r>>2)+4*(l>>5).  k-ordered fmaf chain.
inline lm_f32x16 lm_emu_mfma_f32_32x32x2f32(float a, float b, lm_f32x16 c) {
    lm_emu::WaveState& w = lm_emu::my_wave();
    const int lane = lm_emu::linear_tid() & 63;
    w.a[0][lane] = a;
    w.b[0][lane] = b;
    lm_emu::wave_sync();
    const int j = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        acc = fmaf(w.a[0][i], w.b[0][j], acc);            // k = 0
        acc = fmaf(w.a[0][32 + i], w.b[0][32 + j], acc);  // k = 1
        c[r] = acc;
    }
    lm_emu::wave_sync();
    return c;
}

// ------------------------------------------------------------------ fp16 (software) + f16 MFMA + LDS-DMA
typedef unsigned short lm_h16;  // IEEE binary16 bit pattern
inline float lm_h2f(lm_h16 h) {
    const unsigned sign = (unsigned)(h & 0x8000u) << 16;
    const unsigned exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    unsigned bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {  // subnormal
            float f = (float)man * 5.9604644775390625e-08f;  // 2^-24
            memcpy(&bits, &f, 4);
            bits |= sign;
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
inline lm_h16 lm_f2h(float f) {  // round to nearest even, subnormals kept
    unsigned x;
    memcpy(&x, &f, 4);
    const unsigned sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (lm_h16)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (lm_h16)(sign | 0x7c00u);  // overflow -> inf (>= 65520)
    if (x < 0x38800000u) {                                  // subnormal or zero in half
        const float a = fabsf(f) * 16777216.0f;             // * 2^24 -> units of the smallest subnormal
        const float r = nearbyintf(a);
        return (lm_h16)(sign | (unsigned)r);
    }
    const unsigned mant = x & 0x7fffffu, exp = (x >> 23) - 112u;
    unsigned h = (exp << 10) | (mant >> 13);
    const unsigned rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (lm_h16)(sign | h);
}
struct alignas(16) lm_h16x8 {
    lm_h16 v[8];
};

// v_mfma_f32_32x32x16_f16: A[i=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][col=l&31], fp32 accumulate.
inline lm_f32x16 lm_emu_mfma_f32_32x32x16_f16(lm_h16x8 a, lm_h16x8 b, lm_f32x16 c) {
    const int lane = lm_emu::linear_tid() & 63;
    const int wv = lm_emu::linear_tid() >> 6;
    static thread_local float As[16][64][8], Bs[16][64][8];  // per wave of the block
    for (int j = 0; j < 8; ++j) {
        As[wv][lane][j] = lm_h2f(a.v[j]);
        Bs[wv][lane][j] = lm_h2f(b.v[j]);
    }
    lm_emu::wave_sync();
    const int col = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int kb = 0; kb < 2; ++kb)
            for (int j = 0; j < 8; ++j) acc = fmaf(As[wv][32 * kb + i][j], Bs[wv][32 * kb + col][j], acc);
        c[r] = acc;
    }
    lm_emu::wave_sync();
    return c;
}

// v_mfma_f32_16x16x32_f16: A[i=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][col=l&15]; D reg r of lane l: row 4*(l>>4)+r, col l&15.
inline lm_f32x4 lm_emu_mfma_f32_16x16x32_f16(lm_h16x8 a, lm_h16x8 b, lm_f32x4 c) {
    const int lane = lm_emu::linear_tid() & 63;
    const int wv = lm_emu::linear_tid() >> 6;
    static thread_local float As[16][64][8], Bs[16][64][8];  // per wave of the block
    for (int j = 0; j < 8; ++j) {
        As[wv][lane][j] = lm_h2f(a.v[j]);
        Bs[wv][lane][j] = lm_h2f(b.v[j]);
    }
    lm_emu::wave_sync();
    const int col = lane & 15, q = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * q + r;
        float acc = c[r];
        for (int kb = 0; kb < 4; ++kb)
            for (int j = 0; j < 8; ++j) acc = fmaf(As[wv][16 * kb + i][j], Bs[wv][16 * kb + col][j], acc);
        c[r] = acc;
    }
    lm_emu::wave_sync();
    return c;
}

// global_load_lds_dwordx4: LDS destination = wave-uniform base + lane*16; per-lane global source.
inline void lm_emu_global_load_lds16(const void* gsrc, void* lds_wave_base) {
    const int lane = lm_emu::linear_tid() & 63;
    memcpy((char*)lds_wave_base + lane * 16, gsrc, 16);
}
struct alignas(8) uint2 {
    unsigned x, y;
};
