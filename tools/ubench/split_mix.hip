// Micro-test: lm_split4's one-instruction remainder (v_fma_mixlo/mixhi_f16: f32 v * 1.0 - f16 hi, rounded once to f16) against the
// convert / subtract / convert sequence it replaces, bit for bit, over values of every exponent the network can produce
// (f16 denormal remainders, negative values, exact halves, zero, values beyond the f16 range included).
// hipcc --offload-arch=gfx950 -O3 -I lungmask_amd/csrc -I include tools/ubench/split_mix.hip -o tools/ubench/split_mix && tools/ubench/split_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "lm_platform.h"

__device__ __forceinline__ void split4_ref(float v0, float v1, float v2, float v3, uint2* hi, uint2* lo) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {v0, v1, v2, v3};
    const h4 h = __builtin_convertvector(v, h4);
    const f4 r = v - __builtin_convertvector(h, f4);
    const h4 l = __builtin_convertvector(r, h4);
    __builtin_memcpy(hi, &h, 8);
    __builtin_memcpy(lo, &l, 8);
}

__global__ void k(const float4* in, uint4* a, uint4* b, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = in[i];
    uint2 h0, l0, h1, l1;
    lm_split4(v.x, v.y, v.z, v.w, &h0, &l0);
    split4_ref(v.x, v.y, v.z, v.w, &h1, &l1);
    a[i] = make_uint4(h0.x, h0.y, l0.x, l0.y);
    b[i] = make_uint4(h1.x, h1.y, l1.x, l1.y);
}

int main() {
    const size_t n = 1u << 22;
    std::vector<float> h(4 * n);
    srand(7);
    for (size_t i = 0; i < 4 * n; ++i) {
        const int e = (rand() % 48) - 30;  // 2^-30 .. 2^17
        const float m = 1.0f + (float)rand() / (float)RAND_MAX;
        float v = ldexpf(m, e) * ((rand() & 1) ? -1.f : 1.f);
        const int r = rand() % 64;
        if (r == 0) v = 0.f;
        if (r == 1) v = ldexpf((float)(rand() % 2048), e - 10);  // exactly representable in f16 (remainder 0)
        if (r == 2) v = ldexpf((float)(2 * (rand() % 1024) + 1), e - 11);  // halfway cases of the hi rounding
        h[i] = v;
    }
    float4* din;
    uint4 *da, *db;
    hipMalloc(&din, n * 16);
    hipMalloc(&da, n * 16);
    hipMalloc(&db, n * 16);
    hipMemcpy(din, h.data(), n * 16, hipMemcpyHostToDevice);
    k<<<(unsigned)((n + 255) / 256), 256>>>(din, da, db, n);
    std::vector<unsigned> a(4 * n), b(4 * n);
    hipMemcpy(a.data(), da, n * 16, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 16, hipMemcpyDeviceToHost);
    size_t bad = 0, denorm = 0;
    for (size_t i = 0; i < 4 * n; ++i) {
        if (a[i] != b[i]) {
            if (bad < 5) printf("mismatch at %zu: %08x vs %08x (inputs around %g)\n", i, a[i], b[i], h[i & ~(size_t)3]);
            ++bad;
        }
        if ((i & 3) >= 2)  // lo words: count f16 denormal remainders seen (exponent field 0, mantissa != 0)
            for (int s = 0; s < 32; s += 16) denorm += (((b[i] >> s) & 0x7c00u) == 0 && ((b[i] >> s) & 0x3ffu) != 0);
    }
    printf("split_mix: %zu values, %zu denormal remainders among them, %zu mismatching words\n", 4 * n, denorm, bad);
    return bad ? 1 : 0;
}
