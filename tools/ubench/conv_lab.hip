// Lab harness for the persistent split-f16 conv kernel (NOT part of the product): runs conv_igemm_h3p on every 3x3 layer
// shape of the U-Net (batch 20, uniform-random split operands -- the guide's rule 25: never benchmark on zeros) and prints,
// per layer, the algorithmic TFLOP/s from HIP events and where the waves' cycles went (the kernel is compiled here with
// -DLM_H3_TRACE: per-wave shader-clock sums between marks -- main tap pipeline, chunk barrier, epilogue, item switch).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DLM_H3_TRACE -I lungmask_amd/csrc tools/ubench/conv_lab.hip -o tools/ubench/conv_lab
#include "../../lungmask_amd/csrc/nn_kernels_h3.hip"

#include <cstdio>
#include <vector>

// v = hi + lo with hi uniform in [-1, 1) and lo a remainder-sized half (|lo| <= 2^-11 |hi|-ish): the operand statistics of the
// split scheme (full-range signs and mantissas: no DVFS give-back from constant data)
__global__ void fill_split(char* p, size_t groups, unsigned seed) {
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
        _Float16 h[8], l[8];
        for (int k = 0; k < 8; ++k) {
            unsigned x = (unsigned)(g * 8 + k) * 2654435761u + seed;
            x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
            const float v = ((x & 0xffffff) / 8388608.0f - 1.0f) * 0.25f;
            h[k] = (_Float16)v;
            l[k] = (_Float16)(v - (float)h[k]);
        }
        memcpy(p + g * 32, h, 16);
        memcpy(p + g * 32 + 16, l, 16);
    }
}
__global__ void fill_f32(float* p, size_t n, float a, float b, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = a + (b - a) * ((x & 0xffffff) / 16777216.0f);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 20;
    const int reps = argc > 2 ? atoi(argv[2]) : 6;
    struct Shape { int H, Cin, Cout; bool pool; };
    const Shape shapes[] = {{256, 64, 64, true},   {128, 64, 128, false},  {128, 128, 128, true}, {64, 128, 256, false}, {64, 256, 256, true},
                            {32, 256, 512, false}, {32, 512, 512, true},   {16, 512, 1024, false}, {16, 1024, 1024, false},
                            {32, 1024, 512, false}, {64, 512, 256, false}, {128, 256, 128, false}, {256, 128, 64, false}};
    unsigned* trace = nullptr;
    CK(hipMalloc(&trace, 256 * 8 * 8 * sizeof(unsigned)));
#if defined(LM_H3_TRACE) || defined(LM_H3_TIMELINE)
    CK(hipMemcpyToSymbol(HIP_SYMBOL(lm_h3_trace_ptr), &trace, sizeof(trace)));
#endif
    char* zeros = nullptr;
    CK(hipMalloc(&zeros, 256));
    CK(hipMemset(zeros, 0, 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double tot_ms = 0, tot_flop = 0;
    printf("%-22s %9s %9s | %6s %6s %6s %6s %6s | %9s %7s\n", "layer", "ms", "TFLOP/s", "main%", "barr%", "tail%", "epi%", "swit%", "cyc/wave", "GHz");
    for (const Shape& s : shapes) {
        const size_t npx = (size_t)B * s.H * s.H;
        char *in, *out, *w, *pool = nullptr;
        float *bias, *bs, *bt;
        CK(hipMalloc(&in, npx * s.Cin * 4));
        CK(hipMalloc(&out, npx * s.Cout * 4));
        CK(hipMalloc(&w, (size_t)9 * s.Cout * s.Cin * 4));
        if (s.pool) CK(hipMalloc(&pool, npx / 4 * s.Cout * 4));
        CK(hipMalloc(&bias, s.Cout * 4));
        CK(hipMalloc(&bs, s.Cout * 4));
        CK(hipMalloc(&bt, s.Cout * 4));
        fill_split<<<1024, 256>>>(in, npx * s.Cin / 8, 1u);
        fill_split<<<1024, 256>>>(w, (size_t)9 * s.Cout * s.Cin / 8, 2u);
        fill_f32<<<4, 256>>>(bias, s.Cout, -0.1f, 0.1f, 3u);
        fill_f32<<<4, 256>>>(bs, s.Cout, 0.75f, 1.25f, 4u);
        fill_f32<<<4, 256>>>(bt, s.Cout, -0.1f, 0.1f, 5u);
        float* corr = nullptr;  // the product runs every 3x3 conv with a border-correction table (deferred BatchNorm shift)
        CK(hipMalloc(&corr, 16 * s.Cout * 4));
        fill_f32<<<16, 256>>>(corr, 16 * (size_t)s.Cout, -0.05f, 0.05f, 6u);
        CK(hipMemset(corr, 0, s.Cout * 4));  // mask 0 (interior pixel) carries no correction, as in the product
        lm::ConvParamsH3 p{};
        p.in = in; p.in_cstride = s.Cin; p.in_coff = 0; p.w = w; p.acc_scale = 1.f; p.bias = bias; p.bn_s = bs; p.bn_t = bt;
        p.out = out; p.out_cstride = s.Cout; p.out_coff = 0; p.pool = pool; p.pool_cstride = s.Cout; p.pool_coff = 0; p.zeros = zeros;
        p.B = B; p.H = s.H; p.W = s.H; p.Cin = s.Cin; p.Cout = s.Cout;
        if (!getenv("LM_LAB_NO_BORDER")) p.border_corr = corr;
        for (int i = 0; i < 2; ++i) CK(lm::launch_conv3x3_h3(p, 0));
        CK(hipDeviceSynchronize());
        if (getenv("LM_LAB_VERIFY")) {  // the persistent kernel against the simple 4-wave kernel: outputs must agree bit for bit
            const size_t ob = npx * s.Cout * 4, pb = s.pool ? npx / 4 * s.Cout * 4 : 0;
            std::vector<unsigned char> o1(ob), o2(ob), p1(pb), p2(pb);
            CK(hipMemcpy(o1.data(), out, ob, hipMemcpyDeviceToHost));
            if (pb) CK(hipMemcpy(p1.data(), pool, pb, hipMemcpyDeviceToHost));
            CK(hipMemset(out, 0xee, ob));
            if (pb) CK(hipMemset(pool, 0xee, pb));
            const int tiles = ((p.W + 15) / 16) * ((p.H + 15) / 16);
            hipLaunchKernelGGL((lm::conv_igemm_h3<9>), dim3((unsigned)(tiles * p.B), (unsigned)(p.Cout / 64)), dim3(256), (lm::H3Smem<9>::BYTES), 0, p);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(o2.data(), out, ob, hipMemcpyDeviceToHost));
            if (pb) CK(hipMemcpy(p2.data(), pool, pb, hipMemcpyDeviceToHost));
            size_t bad = 0, first = (size_t)-1, badp = 0;
            for (size_t i = 0; i < ob; ++i) if (o1[i] != o2[i]) { if (!bad) first = i; ++bad; }
            for (size_t i = 0; i < pb; ++i) badp += p1[i] != p2[i];
            printf("verify H%d_Ci%d_Co%d: %zu differing output bytes of %zu (first at byte %zu = pixel %zu, byte %zu of its %d), pool: %zu of %zu\n", s.H, s.Cin, s.Cout, bad, ob,
                   first, first == (size_t)-1 ? 0 : first / (s.Cout * 4), first == (size_t)-1 ? 0 : first % (s.Cout * 4), s.Cout * 4, badp, pb);
        }
        CK(hipMemset(trace, 0, 256 * 8 * 8 * sizeof(unsigned)));
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) CK(lm::launch_conv3x3_h3(p, 0));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        std::vector<unsigned> t(256 * 8 * 8);
        CK(hipMemcpy(t.data(), trace, t.size() * 4, hipMemcpyDeviceToHost));
#ifdef LM_H3_TIMELINE
        {   // one more launch, alone, for the timeline of workgroups 0 and LM_TL_WG2: [2][8][256] (id << 28 | clock)
            CK(hipMemset(trace, 0xff, 2 * 8 * 256 * sizeof(unsigned)));
            CK(lm::launch_conv3x3_h3(p, 0));
            CK(hipDeviceSynchronize());
            std::vector<unsigned> tl(2 * 8 * 256);
            CK(hipMemcpy(tl.data(), trace, tl.size() * 4, hipMemcpyDeviceToHost));
            char fn[128];
            snprintf(fn, sizeof fn, "%s/tl_H%d_Ci%d_Co%d.txt", getenv("LM_TL_DIR") ? getenv("LM_TL_DIR") : ".", s.H, s.Cin, s.Cout);
            if (FILE* f = fopen(fn, "w")) {
                for (int wg = 0; wg < 2; ++wg)
                    for (int w = 0; w < 8; ++w) {
                        fprintf(f, "wg%d wave%d:", wg, w);
                        for (int k = 0; k < 256 && tl[(wg * 8 + w) * 256 + k] != 0xffffffffu; ++k)
                            fprintf(f, " %u:%u", tl[(wg * 8 + w) * 256 + k] >> 28, tl[(wg * 8 + w) * 256 + k] & 0xfffffffu);
                        fprintf(f, "\n");
                    }
                fclose(f);
            }
            printf("%-22s %9.4f %9.1f (timeline build)\n", fn, ms, 2.0 * npx * s.Cout * s.Cin * 9 / ms / 1e9);
            (void)hipFree(in); (void)hipFree(out); (void)hipFree(w); if (pool) (void)hipFree(pool);
            (void)hipFree(bias); (void)hipFree(bs); (void)hipFree(bt); (void)hipFree(corr);
            continue;
        }
#endif
        double sum[6] = {0, 0, 0, 0, 0, 0};
        int nw = 0;
        for (int i = 0; i < 256 * 8; ++i) {
            double tt = 0;
            for (int k = 0; k < 5; ++k) tt += t[i * 8 + k];
            if (tt == 0) continue;
            ++nw;
            for (int k = 0; k < 5; ++k) sum[k] += t[i * 8 + k];
        }
        double all = sum[0] + sum[1] + sum[2] + sum[3] + sum[4];
        const double flop = 2.0 * npx * s.Cout * s.Cin * 9;
        char name[64];
        snprintf(name, sizeof name, "H%d_Ci%d_Co%d", s.H, s.Cin, s.Cout);
        printf("%-22s %9.4f %9.1f | %6.1f %6.1f %6.1f %6.1f %6.1f | %9.0f %7.3f\n", name, ms, flop / ms / 1e9, 100 * sum[0] / all, 100 * sum[1] / all, 100 * sum[2] / all,
               100 * sum[3] / all, 100 * sum[4] / all, nw ? all / nw : 0.0, nw ? all / nw / (ms * 1e6) : 0.0);
        tot_ms += ms;
        tot_flop += flop;
        (void)hipFree(in); (void)hipFree(out); (void)hipFree(w); if (pool) (void)hipFree(pool);
        (void)hipFree(bias); (void)hipFree(bs); (void)hipFree(bt); (void)hipFree(corr);
    }
    // weights: the 64->64 and Cin->Cin layers appear twice in the network (13 shapes, 17 launches)
    printf("sum over the 13 shapes: %.3f ms, %.1f TFLOP/s\n", tot_ms, tot_flop / tot_ms / 1e9);
    return 0;
}
