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
// store patterns of a "split" producer: every lane owns one 32-byte group (16 B of hi halves, 16 B of lo halves)
//   mode 0: two instructions, each lane 16 B at a 32 B stride (hi pieces, then the lo pieces that fill the gaps) -- what the
//           first conv / upsample kernels do;  mode 1: lane pairs exchange so that an instruction writes whole 32 B sectors
//           (even groups, then odd groups);  mode 2: contiguous (16 B per lane, lane-linear), two instructions
template <int MODE, bool NT>
__global__ void k_write_split(u4* p, size_t ngroups, unsigned v) {
    const u4 hi = {v, v + 1, v + 2, v + 3}, lo = {v + 4, v + 5, v + 6, v + 7};
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * blockDim.x) {
        u4 *a, *b;
        if (MODE == 0) { a = p + 2 * g; b = p + 2 * g + 1; }
        else if (MODE == 1) { const size_t e = g & ~(size_t)1; a = p + 2 * e + (g & 1); b = p + 2 * (e + 1) + (g & 1); }
        else { const size_t w = g & ~(size_t)63, l = g & 63; a = p + 2 * w + l; b = p + 2 * w + 64 + l; }
        if (NT) { __builtin_nontemporal_store(hi, a); __builtin_nontemporal_store(lo, b); }
        else { *a = hi; *b = lo; }
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
    for (int blocks : {256, 4096, 65536}) {
        float t[6];
        t[0] = timeit([&] { k_write_split<0, false><<<blocks, 256>>>(a, n / 2, 7u); });
        t[1] = timeit([&] { k_write_split<0, true><<<blocks, 256>>>(a, n / 2, 7u); });
        t[2] = timeit([&] { k_write_split<1, false><<<blocks, 256>>>(a, n / 2, 7u); });
        t[3] = timeit([&] { k_write_split<1, true><<<blocks, 256>>>(a, n / 2, 7u); });
        t[4] = timeit([&] { k_write_split<2, false><<<blocks, 256>>>(a, n / 2, 7u); });
        t[5] = timeit([&] { k_write_split<2, true><<<blocks, 256>>>(a, n / 2, 7u); });
        printf("grid %6d x 256: 16 B at 32 B stride x2: %.2f TB/s (nt %.2f) | 32 B sectors x2: %.2f (nt %.2f) | lane-linear x2: %.2f (nt %.2f)\n", blocks, bytes / t[0] / 1e9,
               bytes / t[1] / 1e9, bytes / t[2] / 1e9, bytes / t[3] / 1e9, bytes / t[4] / 1e9, bytes / t[5] / 1e9);
    }
    return 0;
}
