// Which compute units does a CU-masked stream (hipExtStreamCreateWithCUMask) run on?  Prints, per XCD, the set of (SE, CU) ids that
// executed workgroups of a kernel launched on streams whose mask enables bits [0, N) or [N, 256).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/cumask_probe.hip -o tools/ubench/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) << 16 | (hwid & 0xffff);
}
static void run(const char* name, hipStream_t s, unsigned* d, int blocks) {
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, s, d, 200000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(blocks);
    hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus[16];
    for (unsigned v : h) cus[v >> 16].insert(((v >> 13) & 7) << 8 | ((v >> 12) & 1) << 4 | ((v >> 8) & 0xf));  // se, sh, cu
    printf("%-28s:", name);
    int tot = 0;
    for (int x = 0; x < 8; ++x) { printf(" xcc%d=%zu", x, cus[x].size()); tot += (int)cus[x].size(); }
    printf("  total %d CUs\n", tot);
}
int main(int argc, char** argv) {
    const int n_bw = argc > 1 ? atoi(argv[1]) : 32;
    unsigned* d; hipMalloc(&d, 4096 * 4);
    hipStream_t s0; hipStreamCreate(&s0);
    run("unmasked", s0, d, 2048);
    unsigned mA[8], mB[8];
    for (int w = 0; w < 8; ++w) { mA[w] = 0; mB[w] = 0; }
    for (int i = 0; i < 256; ++i) (i < 256 - n_bw ? mA : mB)[i >> 5] |= 1u << (i & 31);
    hipStream_t sa, sb;
    printf("create A: %s\n", hipGetErrorString(hipExtStreamCreateWithCUMask(&sa, 8, mA)));
    printf("create B: %s\n", hipGetErrorString(hipExtStreamCreateWithCUMask(&sb, 8, mB)));
    run("mask bits [0,256-n)", sa, d, 2048);
    run("mask bits [256-n,256)", sb, d, 2048);
    // every 8th bit pattern: bits i with (i % 8) >= 7 -> B
    for (int w = 0; w < 8; ++w) { mA[w] = 0; mB[w] = 0; }
    for (int i = 0; i < 256; ++i) ((i & 7) != 7 ? mA : mB)[i >> 5] |= 1u << (i & 31);
    hipStream_t sc, sd;
    hipExtStreamCreateWithCUMask(&sc, 8, mA); hipExtStreamCreateWithCUMask(&sd, 8, mB);
    run("mask bits i%8!=7", sc, d, 2048);
    run("mask bits i%8==7", sd, d, 2048);
    return 0;
}
