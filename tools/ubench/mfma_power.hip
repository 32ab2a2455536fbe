// Micro-benchmark: how the operand BIT PATTERNS of the split-f16 scheme move the sustained matrix rate.
// MI355X clocks to a power budget; a pure v_mfma_f32_32x32x16_f16 stream runs at 2.37 GHz on zeros but at 1.60 GHz on
// full-range random halves (tools/ubench/mfma_rate.hip).  Two thirds of the conv kernel's matrix instructions have a
// REMAINDER operand (lo = f16(v - f16(v))), of which the 1e-3 parity bar needs only the leading ~5 bits -- this probe
// measures what zeroing the trailing mantissa bits of lo (or other operand statistics) buys in TFLOP/s.
// Same instruction pattern as one tap of conv_igemm_h3p: 4 accumulators; w_hi*a_hi, w_hi*a_lo, w_lo*a_hi.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o tools/ubench/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float urand(unsigned x) {
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    return (x & 0xffffff) / 8388608.0f - 1.0f;
}
__device__ __forceinline__ _Float16 keep_bits(_Float16 h, int k) {  // keep the k leading bits of the 10-bit mantissa field
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    u &= (unsigned short)(0xffffu << (10 - k));
    __builtin_memcpy(&h, &u, 2);
    return h;
}

struct Fill {
    int hi_bits;   // mantissa bits kept in the hi halves (10 = all)
    int lo_bits;   // mantissa bits kept in the lo halves (10 = all, -1: lo = 0)
    int relu_pct;  // % of activation values replaced by one constant (BatchNorm shift behind a ReLU)
    int zero;      // everything zero
    int lo_as_hi;  // 1: the "lo" operands are independent full-range values (three hi*hi products: the mfma_rate case)
    int zero_pct;  // % of activation values that are exactly zero (a ReLU output stored WITHOUT the BatchNorm shift)
    int order;     // 0: accumulators in rotation (shipped); 1: the three products of an accumulator back to back; 2: operands swapped
};

__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, Fill f) {
    h8 whi[2], wlo[2], ahi[2], alo[2];
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 8; ++i) {
            const unsigned id = (blockIdx.x * 512u + threadIdx.x) * 64u + j * 8 + i;
            float w = urand(id), a = urand(id ^ 0x9e3779b9u);
            if ((int)(urand(id ^ 0x51ed270bu) * 50.f + 50.f) < f.relu_pct) a = 0.117f;
            if ((int)(urand(id ^ 0x2545f491u) * 50.f + 50.f) < f.zero_pct) a = 0.f;
            if (f.zero) w = a = 0.f;
            _Float16 wh = keep_bits((_Float16)w, f.hi_bits), ah = keep_bits((_Float16)a, f.hi_bits);
            _Float16 wl = (_Float16)(w - (float)wh), al = (_Float16)(a - (float)ah);
            if (f.lo_bits >= 0) { wl = keep_bits(wl, f.lo_bits); al = keep_bits(al, f.lo_bits); }
            else { wl = (_Float16)0.f; al = (_Float16)0.f; }
            if (f.lo_as_hi) { wl = (_Float16)urand(id ^ 0x1234567u); al = (_Float16)urand(id ^ 0x7654321u); }
            whi[j][i] = wh; wlo[j][i] = wl; ahi[j][i] = ah; alo[j][i] = al;
        }
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (f.order == 0) {
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[0], ahi[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[0], ahi[1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[1], ahi[0], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[1], ahi[1], c3, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[0], alo[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[0], alo[1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[1], alo[0], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[1], alo[1], c3, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[0], ahi[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[0], ahi[1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[1], ahi[0], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[1], ahi[1], c3, 0, 0, 0);
        }
    } else if (f.order == 1) {
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[0], ahi[0], c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[0], alo[0], c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[0], ahi[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[0], ahi[1], c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[0], alo[1], c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[0], ahi[1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[1], ahi[0], c2, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[1], alo[0], c2, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[1], ahi[0], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[1], ahi[1], c3, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[1], alo[1], c3, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[1], ahi[1], c3, 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[0], whi[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[1], whi[0], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[0], whi[1], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[1], whi[1], c3, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[0], whi[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[1], whi[0], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[0], whi[1], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[1], whi[1], c3, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[0], wlo[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[1], wlo[0], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[0], wlo[1], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[1], wlo[1], c3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the same operand statistics through v_mfma_f32_16x16x32_f16: half the accumulator traffic per MAC, twice the operand traffic
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k16(float* out, unsigned long long* cyc, int iters, Fill f) {
    h8 whi[2], wlo[2], ahi[2], alo[2];
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 8; ++i) {
            const unsigned id = (blockIdx.x * 512u + threadIdx.x) * 64u + j * 8 + i;
            float w = urand(id), a = urand(id ^ 0x9e3779b9u);
            if ((int)(urand(id ^ 0x2545f491u) * 50.f + 50.f) < f.zero_pct) a = 0.f;
            _Float16 wh = (_Float16)w, ah = (_Float16)a;
            whi[j][i] = wh; wlo[j][i] = (_Float16)(w - (float)wh); ahi[j][i] = ah; alo[j][i] = (_Float16)(a - (float)ah);
        }
    f4v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0}, c4 = {0}, c5 = {0}, c6 = {0}, c7 = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[0], ahi[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[0], ahi[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[1], ahi[0], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[1], ahi[1], c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[0], alo[0], c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[0], alo[1], c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[1], alo[0], c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[1], alo[1], c7, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[0], ahi[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[0], ahi[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[1], ahi[0], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[1], ahi[1], c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + c4[i] + c5[i] + c6[i] + c7[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out;
    unsigned long long* cyc;
    const int blocks = 256, threads = 512;
    (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    (void)hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    struct { const char* name; Fill f; } cases[] = {
        {"zeros", {10, 10, 0, 1, 0, 0, 0}},
        {"three independent full-range products", {10, 10, 0, 0, 1, 0, 0}},
        {"split, lo = all 10 mantissa bits (shipped)", {10, 10, 0, 0, 0, 0, 0}},
        {"split, lo keeps 8 mantissa bits", {10, 8, 0, 0, 0, 0, 0}},
        {"split, lo keeps 6 mantissa bits", {10, 6, 0, 0, 0, 0, 0}},
        {"split, lo keeps 4 mantissa bits", {10, 4, 0, 0, 0, 0, 0}},
        {"split, lo keeps 2 mantissa bits", {10, 2, 0, 0, 0, 0, 0}},
        {"split, lo = 0", {10, -1, 0, 0, 0, 0, 0}},
        {"split, hi keeps 7 bits (bf16-like), lo all", {7, 10, 0, 0, 0, 0, 0}},
        {"split (lo all), 50% of activations constant", {10, 10, 50, 0, 0, 0, 0}},
        {"split (lo 4 bits), 50% of activations constant", {10, 4, 50, 0, 0, 0, 0}},
        {"split (lo all), 50% of activations ZERO", {10, 10, 0, 0, 0, 50, 0}},
        {"split (lo all), 70% of activations ZERO", {10, 10, 0, 0, 0, 70, 0}},
        {"split (lo 6 bits), 50% of activations ZERO", {10, 6, 0, 0, 0, 50, 0}},
        {"split (lo all), 3 products of an accumulator back to back", {10, 10, 0, 0, 0, 0, 1}},
        {"split (lo all), activations as the A operand", {10, 10, 0, 0, 0, 0, 2}},
    };
    for (auto& c : cases) {
        int iters = 4000;
        float ms = 0;
        for (int pass = 0; pass < 2; ++pass) {  // pass 0 calibrates the iteration count for ~0.6 s
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, c.f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (pass == 0) iters = (int)(iters * (600.0 / ms));
        }
        unsigned long long cy[256];
        (void)hipMemcpy(cy, cyc, sizeof cy, hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < blocks; ++i) avg += (double)cy[i] / blocks;
        const double flop = (double)blocks * (threads / 64) * iters * 12.0 * 32768.0;
        printf("%-62s %8.1f TFLOP/s  clock %.3f GHz\n", c.name, flop / ms / 1e9, avg / (ms * 1e6));  // s_memtime ticks per wall time
    }
    for (int zp : {0, 50}) {
        Fill f{10, 10, 0, 0, 0, zp, 0};
        int iters = 8000;
        float ms = 0;
        for (int pass = 0; pass < 2; ++pass) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k16, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (pass == 0) iters = (int)(iters * (600.0 / ms));
        }
        const double flop = (double)blocks * (threads / 64) * iters * 12.0 * 16384.0;
        printf("v_mfma_f32_16x16x32_f16, split (lo all), %2d%% of activations zero          %8.1f TFLOP/s\n", zp, flop / ms / 1e9);
    }
    return 0;
}
