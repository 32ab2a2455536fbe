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
