// Micro-benchmark: what HBM write / read / copy bandwidth a plain streaming kernel reaches on this chip (16-byte accesses, grid-stride,
// several grid sizes, ordinary vs non-temporal stores) -- the ceiling the network's write-bound kernels (first conv, upsample) are
// measured against.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/write_bw.hip -o tools/ubench/write_bw && tools/ubench/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void k_write(u4* p, size_t n, unsigned v) {
    const u4 w = {v, v + 1, v + 2, v + 3};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(w, p + i);
        else p[i] = w;
    }
}
__global__ void k_read(const u4* p, size_t n, unsigned* out) {
    u4 a = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a[0] + a[1] + a[2] + a[3] == 0x12345678u) *out = 1;
}
template <bool NT>
__global__ void k_copy(const u4* s, u4* d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u4 w = s[i];
        if (NT) __builtin_nontemporal_store(w, d + i);
        else d[i] = w;
    }
}
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    u4 *a, *b;
    unsigned* flag;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&flag, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto&& launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        return ms / 5;
    };
    for (int blocks : {256, 1024, 4096, 16384, 65536}) {
        const float w = timeit([&] { k_write<false><<<blocks, 256>>>(a, n, 7u); });
        const float wn = timeit([&] { k_write<true><<<blocks, 256>>>(a, n, 7u); });
        const float r = timeit([&] { k_read<<<blocks, 256>>>(a, n, flag); });
        const float c = timeit([&] { k_copy<false><<<blocks, 256>>>(a, b, n); });
        const float cn = timeit([&] { k_copy<true><<<blocks, 256>>>(a, b, n); });
        printf("grid %6d x 256: write %.2f TB/s  write(nt) %.2f  read %.2f  copy %.2f (read+write bytes)  copy(nt) %.2f\n", blocks, bytes / w / 1e9, bytes / wn / 1e9,
               bytes / r / 1e9, 2.0 * bytes / c / 1e9, 2.0 * bytes / cn / 1e9);
    }
    return 0;
}
