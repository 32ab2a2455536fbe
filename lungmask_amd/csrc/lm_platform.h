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
