// Micro-benchmark: what the matrix pipes SUSTAIN for the operand statistics and accumulate pattern of conv_igemm_h3p --
// the measured ceiling the conv kernel's MFMA fraction is quoted against ("prove where the ceiling is").
// v_mfma_f32_32x32x16_f16 only, 4 accumulators in rotation, operands held in registers (no LDS, no DMA, no barriers), run
// for ~1 s per case so that the power/clock governor reaches steady state.  Three operand fills: zeros (the DVFS
// give-back case), a sign-constant small-integer fill, and full-range uniform [-1, 1) halves (what the network sees).
// Reports TFLOP/s and the sustained shader clock (s_memtime ticks / wall time).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o tools/ubench/mfma_rate && tools/ubench/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float urand(unsigned x) {
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    return (x & 0xffffff) / 8388608.0f - 1.0f;
}

__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int fill) {
    h8 a[4], b[4];  // 4 "weight" and 4 "activation" fragments, as one tap of the conv kernel holds
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            const unsigned id = (blockIdx.x * 512u + threadIdx.x) * 64u + j * 8 + i;
            float va = 0.f, vb = 0.f;
            if (fill == 1) { va = (float)(1 + (id & 3)); vb = (float)(1 + ((id >> 2) & 3)); }
            if (fill == 2) { va = urand(id); vb = urand(id ^ 0x9e3779b9u); }
            a[j][i] = (_Float16)va;
            b[j][i] = (_Float16)vb;
        }
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {  // hi*hi, hi*lo, lo*hi of the split scheme: same accumulator every 4th instruction
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[g & 1], b[0 ^ (g >> 1)], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[g & 1], b[1 ^ (g >> 1)], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 + (g & 1)], b[2 ^ (g >> 1)], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 + (g & 1)], b[3 ^ (g >> 1)], c3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out;
    unsigned long long* cyc;
    const int blocks = 256;
    (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    (void)hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const char* names[3] = {"zeros", "small integers, one sign", "uniform [-1,1) halves"};
    for (int threads : {256, 512}) {
        for (int fill = 0; fill < 3; ++fill) {
            int iters = 20000;
            float ms = 0;
            for (int pass = 0; pass < 2; ++pass) {  // pass 0 calibrates the iteration count for ~1 s
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, fill);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (pass == 0) iters = (int)(iters * (1000.0 / ms));
            }
            unsigned long long c[256];
            (void)hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
            double avg = 0;
            for (int i = 0; i < blocks; ++i) avg += (double)c[i] / blocks;
            const double flop = (double)blocks * (threads / 64) * iters * 12.0 * 32768.0;
            printf("%d waves/SIMD  %-26s %8.1f TFLOP/s  %7.1f ms  clock %.3f GHz  (%.1f cycles per MFMA per SIMD)\n", threads / 256, names[fill], flop / ms / 1e9, ms,
                   avg / (ms * 1e6), avg / ((double)iters * 12.0 * (threads / 256)));
        }
    }
    return 0;
}
