// Device/runtime front door for every kernel source in this directory.
// Product build: hipcc --offload-arch=gfx950 (the only shipped configuration).
// -DLM_EMU_BUILD: g++ build against tests/emu/hip_emu.h, used ONLY by the
// CPU-side test-suite to execute these same kernels functionally without a GPU.
#pragma once
#ifdef LM_EMU_BUILD
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define LM_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)
#define LM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
typedef float lm_f32x16 __attribute__((ext_vector_type(16)));
typedef float lm_f32x4 __attribute__((ext_vector_type(4)));
#endif
#include <cstdint>
// 1 for the hipcc/gfx950 product build, 0 for the g++ test emulation (lm_is_gpu_build(): the binding refuses the latter)
#ifdef LM_EMU_BUILD
#define LM_IS_GPU_BUILD 0
#else
#define LM_IS_GPU_BUILD 1
#endif

// ---- fp16 pieces of the split-f16 ("3-product") path -------------------------------------------
#ifndef LM_EMU_BUILD
typedef _Float16 lm_h16;
typedef _Float16 lm_h16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float lm_h2f(lm_h16 h) { return (float)h; }
__device__ __forceinline__ lm_h16 lm_f2h(float f) { return (lm_h16)f; }  // v_cvt_f16_f32: round-to-nearest-even
#endif

// v_mfma_f32_32x32x16_f16 (gfx950): A[i=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][col=l&31], fp32 accumulate; C/D as above.
__device__ __forceinline__ lm_f32x16 lm_mfma_f32_32x32x16_f16(lm_h16x8 a, lm_h16x8 b, lm_f32x16 c) {
#ifdef LM_EMU_BUILD
    return lm_emu_mfma_f32_32x32x16_f16(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// v_mfma_f32_16x16x32_f16 (gfx950): A[i=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][col=l&15]; D reg r of lane l is
// (row i = 4*(l>>4)+r, col = l&15).  Half the accumulator traffic per MAC of the 32x32x16 form: under the chip's power budget
// it sustains +15 % more FLOP/s on this network's operands (tools/ubench/mfma_power.hip).
__device__ __forceinline__ lm_f32x4 lm_mfma_f32_16x16x32_f16(lm_h16x8 a, lm_h16x8 b, lm_f32x4 c) {
#ifdef LM_EMU_BUILD
    return lm_emu_mfma_f32_16x16x32_f16(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

// Ordering point between LDS writes and reads of OTHER lanes of the same wave.  The hardware executes a
// wave's LDS instructions in order, so nothing is emitted on the GPU; the test emulator runs lanes as
// independent fibres and needs the rendezvous.
__device__ __forceinline__ void lm_wave_lds_fence() {
#ifdef LM_EMU_BUILD
    lm_emu::wave_sync();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// 16-byte store of streaming output (tensors far larger than the caches, written once): the non-temporal form does not
// allocate the line in L2 / the memory-side cache on its way out.
__device__ __forceinline__ void lm_store16_stream(void* dst, uint4 v) {
#if defined(LM_EMU_BUILD) || defined(LM_NO_NT_STORES)
    *reinterpret_cast<uint4*>(dst) = v;
#else
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u4*>(dst));
#endif
}

// The value of lane ^ 1 (the x + 1 partner of the 2x2 average pool): a DPP quad permutation on the GPU -- a modifier of the
// consuming VALU instruction or one v_mov_dpp -- instead of __shfl_xor's ds_bpermute_b32 through the LDS crossbar.
__device__ __forceinline__ float lm_lane_xor1(float v) {
#ifdef LM_EMU_BUILD
    return __shfl_xor(v, 1);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
#endif
}

// A value the program knows to be wave-uniform, made provably so for the compiler (SGPR instead of a
// per-lane VGPR + waterfall loop).
__device__ __forceinline__ int lm_uniform(int x) {
#ifdef LM_EMU_BUILD
    return x;
#else
    return __builtin_amdgcn_readfirstlane(x);
#endif
}

// 16-byte global -> LDS DMA (global_load_lds_dwordx4): destination is the WAVE-UNIFORM base + lane*16.
__device__ __forceinline__ void lm_global_load_lds16(const void* gsrc, void* lds_wave_base) {
#ifdef LM_EMU_BUILD
    lm_emu_global_load_lds16(gsrc, lds_wave_base);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

// v_mfma_f32_32x32x2_f32: exact-f32 matrix core op (k-ordered fmaf chain).
// lane l holds A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]; D reg r of lane l is
// (row i=(r&3)+8*(r>>2)+4*(l>>5), col j=l&31).
__device__ __forceinline__ lm_f32x16 lm_mfma_f32_32x32x2(float a, float b, lm_f32x16 c) {
#ifdef LM_EMU_BUILD
    return lm_emu_mfma_f32_32x32x2f32(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

// Every LDS-DMA this wave has issued has landed (and every store has left).  Stated explicitly in front of the
// barriers that publish DMA'd data: the compiler's own vmcnt(0) in __syncthreads() was found missing on one
// control-flow path of one instantiation (a rare wrong tile on the GPU, never in the emulator).
__device__ __forceinline__ void lm_dma_wait_all() {
#ifndef LM_EMU_BUILD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// 4-byte variant (global_load_lds_dword): 64 lanes fill 256 contiguous LDS bytes.
__device__ __forceinline__ void lm_global_load_lds4(const void* gsrc, void* lds_wave_base) {
#ifdef LM_EMU_BUILD
    memcpy((char*)lds_wave_base + (lm_emu::linear_tid() & 63) * 4, gsrc, 4);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
#endif
}

// ---- buffer-descriptor LDS-DMA (buffer_load_dwordx4 ... offen lds) ---------------------------------------------
// A raw buffer descriptor (base, num_records bytes): a lane whose voffset + soffset lies outside [0, num_records)
// writes ZEROS to its 16 LDS bytes, so out-of-image halo pixels need neither branches nor a zero page: their lanes
// carry voffset = LM_DMA_OOB.  The instruction is hand-issued (the compiler neither counts it in vmcnt nor drains it):
// lm_dma_wait_all() + a barrier publish the data.  M0 (LDS destination = wave-uniform base + lane * 16) is written in
// the same statement.
#define LM_DMA_OOB 0x80000000u
#ifdef LM_EMU_BUILD
struct lm_rsrc {
    const char* base;
    unsigned bytes;
};
__device__ __forceinline__ lm_rsrc lm_make_rsrc(const void* base, size_t bytes) { return lm_rsrc{(const char*)base, (unsigned)bytes}; }
__device__ __forceinline__ void lm_dma16(const lm_rsrc& r, unsigned voff, unsigned soff, void* lds_wave_base) {
    char* d = (char*)lds_wave_base + (lm_emu::linear_tid() & 63) * 16;
    const unsigned long long off = (unsigned long long)voff + soff;
    if (off + 16 <= r.bytes) memcpy(d, r.base + off, 16);
    else memset(d, 0, 16);
}
__device__ __forceinline__ void lm_dma4_global(const void* gsrc, void* lds_wave_base) {
    memcpy((char*)lds_wave_base + (lm_emu::linear_tid() & 63) * 4, gsrc, 4);
}
__device__ __forceinline__ void lm_dma4(const lm_rsrc& r, unsigned voff, unsigned soff, void* lds_wave_base) {
    char* d = (char*)lds_wave_base + (lm_emu::linear_tid() & 63) * 4;
    const unsigned long long off = (unsigned long long)voff + soff;
    if (off + 4 <= r.bytes) memcpy(d, r.base + off, 4);
    else memset(d, 0, 4);
}
__device__ __forceinline__ void lm_barrier_dma() { __syncthreads(); }
__device__ __forceinline__ void lm_barrier_lds() { __syncthreads(); }
#define LM_PIN(x) \
    do {          \
    } while (0)
#else
typedef int lm_rsrc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lm_rsrc lm_make_rsrc(const void* base, size_t bytes) {
    const unsigned long long a = (unsigned long long)base;
    lm_rsrc r;
    r[0] = (int)(unsigned)a;
    r[1] = (int)(unsigned)(a >> 32);  // stride 0: raw buffer
    r[2] = (int)(unsigned)bytes;
    r[3] = 0x00020000;                // gfx9 raw-buffer dword 3 (DATA_FORMAT = 32)
    return r;
}
__device__ __forceinline__ void lm_dma16(lm_rsrc r, unsigned voff, unsigned soff, void* lds_wave_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base), "v"(voff), "s"(r), "s"(soff)
                 : "memory");
}
// 4-byte form of the buffer-descriptor DMA (buffer_load_dword ... offen lds): 64 lanes fill 256 contiguous LDS bytes, lanes whose
// offset lies outside the buffer write zero; inactive lanes (EXEC = 0) write nothing
__device__ __forceinline__ void lm_dma4(lm_rsrc r, unsigned voff, unsigned soff, void* lds_wave_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :
                 : "s"((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base), "v"(voff), "s"(r), "s"(soff)
                 : "memory");
}
// 4-byte form on a flat pointer (global_load_lds_dword): 64 lanes fill 256 contiguous LDS bytes
__device__ __forceinline__ void lm_dma4_global(const void* gsrc, void* lds_wave_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
                 :
                 : "s"((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base), "v"(gsrc)
                 : "memory");
}
// Workgroup barrier that publishes this wave's LDS-DMAs (and retires its hand-issued LDS reads) ...
__device__ __forceinline__ void lm_barrier_dma() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and one that only orders LDS accesses (no DMA of this wave may be needed by anyone after it): global stores stay in flight
__device__ __forceinline__ void lm_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier ; LM_BARRIER_LDS_ONLY" ::: "memory"); }
// Opaque use of a value: orders an asm statement against the instructions that produce / consume x
#define LM_PIN(x) asm volatile("" : "+v"(x))
#endif

// ---- in-kernel cycle accounting of the persistent conv kernel (lab builds only: -DLM_H3_TRACE) ------------------
// Per wave, the shader-clock cycles between consecutive marks are summed per category and written to lm_h3_trace_ptr
// ([workgroup][wave][8] unsigned) at kernel end.  The product build compiles all of it to nothing.
#if defined(LM_H3_TRACE) && defined(LM_EMU_BUILD)
#undef LM_H3_TRACE  // the trace reads the shader clock: hardware only
#endif
#if defined(LM_H3_TIMELINE) && !defined(LM_EMU_BUILD)
// Timeline form (lab builds: -DLM_H3_TIMELINE): workgroups 0 and LM_TL_WG2 record (mark id, shader clock) per wave into
// lm_h3_trace_ptr ([2][8][256] unsigned: id << 28 | clock & 0xfffffff; the host presets it to 0xffffffff).
extern __device__ unsigned* lm_h3_trace_ptr;
#ifndef LM_TL_WG2
#define LM_TL_WG2 133
#endif
#define LM_TRACE_INIT()                                                                                    \
    unsigned tl_n_ = 0;                                                                                    \
    const bool tl_on_ = lm_h3_trace_ptr && (blockIdx.x == 0 || blockIdx.x == LM_TL_WG2);                   \
    unsigned* const tl_o_ = lm_h3_trace_ptr + ((size_t)(blockIdx.x == 0 ? 0 : 1) * 8 + (threadIdx.x >> 6)) * 256
// (events go straight to global memory from lane 0: the 3x3 form has no LDS left; the stores count in vmcnt, i.e. a wave's
// DMA wait also waits for its last few trace stores -- a few hundred cycles per chunk of perturbation at most)
#define LM_TRACE_MARK(K)                                                                                   \
    do {                                                                                                   \
        if (tl_on_ && tl_n_ < 256) {                                                                       \
            const unsigned n_ = (unsigned)__builtin_amdgcn_s_memtime();                                    \
            if ((threadIdx.x & 63) == 0) tl_o_[tl_n_] = ((unsigned)(K) << 28) | (n_ & 0xfffffffu);         \
            ++tl_n_;                                                                                       \
        }                                                                                                  \
    } while (0)
#define LM_TRACE_SUB(K) LM_TRACE_MARK(K)
#define LM_TRACE_FLUSH() \
    do {                 \
    } while (0)
#elif defined(LM_H3_TRACE)
extern __device__ unsigned* lm_h3_trace_ptr;
#define LM_TRACE_SUB(K) \
    do {                \
    } while (0)
#define LM_TRACE_INIT()                                     \
    unsigned tr_last_ = (unsigned)__builtin_amdgcn_s_memtime(); \
    unsigned tr_sum_[6] = {0u, 0u, 0u, 0u, 0u, 0u}
#define LM_TRACE_MARK(K)                                              \
    do {                                                              \
        const unsigned n_ = (unsigned)__builtin_amdgcn_s_memtime();   \
        tr_sum_[K] += n_ - tr_last_;                                  \
        tr_last_ = n_;                                                \
    } while (0)
#define LM_TRACE_FLUSH()                                                                                   \
    do {                                                                                                   \
        if (lm_h3_trace_ptr && (threadIdx.x & 63) == 0) {                                                  \
            unsigned* o_ = lm_h3_trace_ptr + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;            \
            for (int k_ = 0; k_ < 6; ++k_) o_[k_] = tr_sum_[k_];                                            \
        }                                                                                                  \
    } while (0)
#else
#define LM_TRACE_INIT() \
    do {                \
    } while (0)
#define LM_TRACE_MARK(K) \
    do {                 \
    } while (0)
#define LM_TRACE_FLUSH() \
    do {                 \
    } while (0)
#define LM_TRACE_SUB(K) \
    do {                \
    } while (0)
#endif

// The remainder halves keep only their LM_LO_BITS leading mantissa bits (10 = all of them).  v = hi + lo then holds to
// 2^-(12 + LM_LO_BITS) relative instead of 2^-22; the 1e-3 parity bar of the log-probabilities needs ~2^-16.  Why throw bits
// away: MI355X clocks to a power budget and the matrix pipes' power follows the operands' toggling bits -- two thirds of the
// conv kernel's matrix instructions have a remainder operand (tools/ubench/mfma_power.hip prices it).  Rounding is to nearest
// (ties away from zero) on the bit pattern: add half of the dropped range, clear the dropped bits; a remainder is at most
// 2^-11 of the f16 range, so the add never reaches the sign bit and two halves can be handled in one 32-bit register.
#ifndef LM_LO_BITS
#define LM_LO_BITS 10
#endif
__host__ __device__ __forceinline__ unsigned lm_round_lo_pair(unsigned two_halves) {
    if (LM_LO_BITS >= 10) return two_halves;
    constexpr unsigned half_ulp = (1u << (9 - (LM_LO_BITS < 10 ? LM_LO_BITS : 9))), keep = 0xffffu & ~((1u << (10 - (LM_LO_BITS < 10 ? LM_LO_BITS : 9))) - 1u);
    return (two_halves + (half_ulp | half_ulp << 16)) & (keep | keep << 16);
}

// fp32 x4 -> split-f16: hi = f16(v), lo = f16(v - hi) (UNSCALED: for |v| < 2^-3 the remainder is an f16 denormal, which
// conversions and the matrix instructions honour -- tools/ubench/mfma_denorm.hip -- so v = hi + lo to 2^-25 absolute or
// 2^-22 relative, whichever is larger; less when LM_LO_BITS < 10), each packed as 4 halves (8 bytes).
// The vector form makes hipcc emit v_cvt_pk_f16_f32 / v_pk_add_f32 (2 VALU per value).
__device__ __forceinline__ void lm_split4(float v0, float v1, float v2, float v3, uint2* hi, uint2* lo) {
#ifdef LM_EMU_BUILD
    lm_h16 h[4] = {lm_f2h(v0), lm_f2h(v1), lm_f2h(v2), lm_f2h(v3)};
    lm_h16 l[4] = {lm_f2h(v0 - lm_h2f(h[0])), lm_f2h(v1 - lm_h2f(h[1])), lm_f2h(v2 - lm_h2f(h[2])), lm_f2h(v3 - lm_h2f(h[3]))};
    memcpy(hi, h, 8);
    memcpy(lo, l, 8);
#else
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {v0, v1, v2, v3};
    const h4 h = __builtin_convertvector(v, h4);  // 2 x v_cvt_pk_f16_f32
    __builtin_memcpy(hi, &h, 8);
#ifndef LM_SPLIT_NO_MIX
    // lo = f16(v - f32(hi)) in ONE mixed-precision fma per value (f32 v * 1.0 - f16 hi, rounded once to f16: v - hi is exact in
    // f32, so this is bit for bit the convert / subtract / convert sequence, f16 denormals included -- tools/ubench/split_mix.hip
    // checks it on the hardware): 6 VALU per 4 values instead of 12.
    unsigned l01, l23;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l01) : "v"(v0), "v"(hi->x));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l01) : "v"(v1), "v"(hi->x));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l23) : "v"(v2), "v"(hi->y));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l23) : "v"(v3), "v"(hi->y));
    lo->x = l01;
    lo->y = l23;
#else
    const f4 r = v - __builtin_convertvector(h, f4);
    const h4 l = __builtin_convertvector(r, h4);
    __builtin_memcpy(lo, &l, 8);
#endif
#endif
    lo->x = lm_round_lo_pair(lo->x);
    lo->y = lm_round_lo_pair(lo->y);
}
// f16 range guard on the hi halves themselves: running maximum of |hi| as bit patterns, two halves per register (one AND + one
// packed max per 4 values in the epilogue instead of a float max per value).  |hi| >= 0x7800 (32768.0) also catches inf / NaN.
__device__ __forceinline__ unsigned lm_pk_absmax_u16(unsigned acc, unsigned two_halves) {
    const unsigned m = two_halves & 0x7fff7fffu;
#ifdef LM_EMU_BUILD
    const unsigned lo = (acc & 0xffffu) > (m & 0xffffu) ? (acc & 0xffffu) : (m & 0xffffu);
    const unsigned hi = (acc >> 16) > (m >> 16) ? (acc >> 16) : (m >> 16);
    return lo | (hi << 16);
#else
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 a, b;
    __builtin_memcpy(&a, &acc, 4);
    __builtin_memcpy(&b, &m, 4);
    a = __builtin_elementwise_max(a, b);  // v_pk_max_u16
    unsigned r;
    __builtin_memcpy(&r, &a, 4);
    return r;
#endif
}
__device__ __forceinline__ bool lm_pk_out_of_f16_guard(unsigned acc) { return (acc & 0xffffu) >= 0x7800u || (acc >> 16) >= 0x7800u; }

// one value (the first conv writes single channels)
__device__ __forceinline__ lm_h16 lm_round_lo1(lm_h16 l) {
    unsigned short u;
    memcpy(&u, &l, 2);
    u = (unsigned short)lm_round_lo_pair(u);
    memcpy(&l, &u, 2);
    return l;
}

// the values a consumer of the split pair sees: v' = f32(hi) + f32(lo)
__device__ __forceinline__ void lm_unsplit4(uint2 hi, uint2 lo, float* out) {
#ifdef LM_EMU_BUILD
    lm_h16 h[4], l[4];
    memcpy(h, &hi, 8);
    memcpy(l, &lo, 8);
    for (int k = 0; k < 4; ++k) out[k] = lm_h2f(l[k]) + lm_h2f(h[k]);
#else
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    h4 h, l;
    __builtin_memcpy(&h, &hi, 8);
    __builtin_memcpy(&l, &lo, 8);
    const f4 r = __builtin_convertvector(l, f4) + __builtin_convertvector(h, f4);
    out[0] = r[0];
    out[1] = r[1];
    out[2] = r[2];
    out[3] = r[3];
#endif
}

// Two fp32 values in one 64-bit register pair, and acc += x[SEL] * w on both halves: v_pk_fma_f32 with one dword of x broadcast
// (two independent IEEE fused multiply-adds -- the same bits as two fmaf calls).
#ifdef LM_EMU_BUILD
typedef float lm_f32x2 __attribute__((vector_size(8)));
#else
typedef float lm_f32x2 __attribute__((ext_vector_type(2)));
#endif
template <int SEL>
__device__ __forceinline__ void lm_pk_fma_bcast(lm_f32x2& acc, lm_f32x2 x, lm_f32x2 w) {
#ifdef LM_EMU_BUILD
    acc[0] = fmaf(x[SEL], w[0], acc[0]);
    acc[1] = fmaf(x[SEL], w[1], acc[1]);
#else
    if (SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(x), "v"(w));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(w));
#endif
}

// acc = x * w + acc as ONE v_fma_f32 that the compiler cannot pair up again (it SLP-packs adjacent scalar multiply-adds into
// v_pk_fma_f32 under -O3, and beside matrix instructions the packed form is the dearer one: MI355X_MICROARCH.md prices 1
// v_pk_fma_f32 at +22 cycles over 2 v_fma_f32).  The same bits as fmaf.
__device__ __forceinline__ void lm_fma_f32_single(float& acc, float x, float w) {
#ifdef LM_EMU_BUILD
    acc = fmaf(x, w, acc);
#else
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(w));
#endif
}

// DPP row broadcast: every lane reads `w` from lane K of its own row of 16 lanes (row_newbcast).  acc += w[row lane K] * x, and the
// plain move.  A wave-uniform table of up to 16 values therefore lives in ONE register (lane k of every row holds value k) and
// costs no LDS read, no scalar register and no extra instruction at the point of use.  All lanes of the row must be active.
template <int K>
__device__ __forceinline__ void lm_fmac_rowbcast(float& acc, float w, float x) {
#ifdef LM_EMU_BUILD
    acc = fmaf(__shfl(w, ((int)(lm_emu::linear_tid() & 63) & ~15) | K), x, acc);
#else
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x), "n"(K));
#endif
}
template <int K>
__device__ __forceinline__ float lm_mov_rowbcast(float w) {
#ifdef LM_EMU_BUILD
    return __shfl(w, ((int)(lm_emu::linear_tid() & 63) & ~15) | K);
#else
    float r = 0.f;
    asm("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(w), "n"(K));
    return r;
#endif
}

// v_permlane32_swap_b32: a[lanes 32..63] <-> b[lanes 0..31] (gfx950).  In the conv epilogue lanes l and l + 32 hold the two 4-channel
// halves of the same pixel's 8-channel group: after swapping (hi, lo) word by word, lane l owns all eight hi halves and lane l + 32
// all eight lo halves -- one 16-byte store each instead of two 8-byte pieces staged through LDS.
__device__ __forceinline__ void lm_permlane32_swap(unsigned& a, unsigned& b) {
#ifdef LM_EMU_BUILD
    const unsigned pa = __shfl_xor(a, 32), pb = __shfl_xor(b, 32);
    if ((lm_emu::linear_tid() & 63) < 32) b = pa;
    else a = pb;
#else
    asm("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
#endif
}

// Pins the order of the statements around it (the compiler otherwise moves matrix instructions across the hand-issued reads)
#ifdef LM_EMU_BUILD
#define LM_SCHED_FENCE() \
    do {                 \
    } while (0)
#else
#define LM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// 16-byte LDS reads that the compiler's waitcnt pass cannot see (so it does not drain an in-flight LDS-DMA
// in front of them), with an explicit counted wait that names every destination register.
#ifdef LM_EMU_BUILD
#define LM_OPAQUE3(a, b, c) \
    do {                    \
    } while (0)
#define LM_LDS_WAIT6(N, a, b, c, d, e, f) \
    do {                                  \
    } while (0)
#define LM_LDS_WAIT3(N, a, b, c) \
    do {                         \
    } while (0)
#define LM_LDS_READ128(dst, ptr, OFF) (dst) = *reinterpret_cast<const lm_h16x8*>(reinterpret_cast<const char*>(ptr) + (OFF))
#define LM_LDS_WAIT8(N, a, b, c, d, e, f, g, h) \
    do {                                        \
    } while (0)
#define LM_LDS_WAIT5(N, a, b, c, d, e) \
    do {                               \
    } while (0)
#define LM_LDS_WAIT2(N, a, b) \
    do {                      \
    } while (0)
#define LM_LDS_WAIT1(N, a) \
    do {                   \
    } while (0)
#define LM_LDS_WRITE2_64(ptr, lo8, hi8)                                  \
    do {                                                                 \
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ptr)) = (lo8); \
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ptr) + 16) = (hi8); \
    } while (0)
#else
// two 8-byte LDS stores of one lane, 16 bytes apart, as ONE hand-issued instruction (counted in lgkmcnt like the reads)
typedef unsigned lm_u32x2 __attribute__((ext_vector_type(2)));
#define LM_LDS_WRITE2_64(ptr, lo8, hi8)                                                                                      \
    do {                                                                                                                     \
        const lm_u32x2 d0_ = {(lo8).x, (lo8).y}, d1_ = {(hi8).x, (hi8).y};                                                   \
        asm volatile("ds_write2_b64 %0, %1, %2 offset1:2"                                                                    \
                     :                                                                                                       \
                     : "v"((unsigned)(size_t)(__attribute__((address_space(3))) char*)(ptr)), "v"(d0_), "v"(d1_)            \
                     : "memory");                                                                                            \
    } while (0)
#define LM_LDS_WAIT2(N, a, b)                                                                \
    do {                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#define LM_LDS_WAIT1(N, a)                                                                   \
    do {                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "i"(N) : "memory");                 \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#define LM_LDS_WAIT5(N, a, b, c, d, e)                                                                            \
    do {                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "i"(N) : "memory");  \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#define LM_OPAQUE3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#define LM_LDS_WAIT6(N, a, b, c, d, e, f)                                                                         \
    do {                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "i"(N) : "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#define LM_LDS_WAIT3(N, a, b, c)                                                                                  \
    do {                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "i"(N) : "memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#define LM_LDS_READ128(dst, ptr, OFF)                                                                             \
    asm volatile("ds_read_b128 %0, %1 offset:%2"                                                                  \
                 : "=v"(dst)                                                                                      \
                 : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const char*)(ptr)), "i"(OFF)          \
                 : "memory")
#define LM_LDS_WAIT8(N, a, b, c, d, e, f, g, h)                                                                   \
    do {                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(%8)"                                                                      \
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)                     \
                     : "i"(N)                                                                                     \
                     : "memory");                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#endif
