// Micro-test: are f16 DENORMAL operands of v_mfma_f32_32x32x16_f16 honoured (not flushed), and does the f32 -> f16
// conversion under the default HIP float mode produce them?  (Prerequisite of storing the `lo` halves unscaled.)
// hipcc --offload-arch=gfx950 -O3 mfma_denorm.hip -o mfma_denorm && ./mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k(float* out, float tiny) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)tiny;  // 2^-20: an f16 denormal (min normal 2^-14, min denormal 2^-24)
        b[i] = (_Float16)1.0f;
    }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = c[0];              // expect 16 * 2^-20 = 1.52587890625e-05 when denormals are honoured, 0 when flushed
        out[1] = (float)a[0];       // the conversion itself
        _Float16 n = (_Float16)(tiny * 64.0f);  // 2^-14: normal
        out[2] = (float)n;
    }
}

int main() {
    float* out;
    hipMalloc(&out, 64);
    k<<<1, 64>>>(out, 9.5367431640625e-07f);
    float h[3];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("mfma(16 x denormal 2^-20 * 1) = %.10e (expect 1.5258789062e-05)\n", h[0]);
    printf("f32->f16->f32 of 2^-20 = %.10e (expect 9.5367431641e-07), of 2^-14 = %.10e\n", h[1], h[2]);
    return 0;
}
