// Micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate for the accumulate patterns of conv_igemm_h3*.
// hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int PATTERN>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    h8 a0, a1, b0, b1;
    for (int i = 0; i < 8; ++i) { a0[i] = (_Float16)(threadIdx.x * 0.001f + i); a1[i] = a0[i] + (_Float16)1; b0[i] = a0[i] * (_Float16)0.5f; b1[i] = a1[i] * (_Float16)0.25f; }
    f16v m0 = {0}, m1 = {0}, c0 = {0}, c1 = {0};
    for (int it = 0; it < iters; ++it) {
        if (PATTERN == 0) {  // as written in the kernel: dependent pairs back to back
            m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, m0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, m1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c1, 0, 0, 0);
        } else {  // interleaved: the two uses of c0/c1 three instructions apart
            m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, m0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, m1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += m0[i] + m1[i] + c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int PATTERN>
void run(int threads, int blocks_per_cu, const char* name) {
    float* out;
    int blocks = 256 * blocks_per_cu, iters = 20000;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<PATTERN><<<blocks, threads>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<PATTERN><<<blocks, threads>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double flop = (double)blocks * (threads / 64) * iters * 6.0 * 32768.0;
    printf("%-28s threads=%4d blocks/CU=%d  %.1f TFLOP/s  (%.2f ms)\n", name, threads, blocks_per_cu, flop / ms / 1e9, ms);
    hipFree(out);
}

int main() {
    run<0>(256, 1, "dependent-pairs 1 wave/SIMD");
    run<1>(256, 1, "interleaved     1 wave/SIMD");
    run<0>(512, 1, "dependent-pairs 2 waves/SIMD");
    run<1>(512, 1, "interleaved     2 waves/SIMD");
    run<0>(1024, 1, "dependent-pairs 4 waves/SIMD");
    run<1>(1024, 1, "interleaved     4 waves/SIMD");
    return 0;
}
