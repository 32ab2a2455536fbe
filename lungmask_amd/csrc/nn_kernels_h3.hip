// Split-f16 ("3-product") U-Net kernels: fp32-class accuracy on the 16-bit matrix cores.
//
// Every fp32 value v is carried as a pair of halves
//      hi = f16(v),   lo = f16(v - hi)                   =>  v = hi + lo  to ~2^-22 relative
// (lo is stored UNSCALED: for activations of O(1) it is a normal f16 number, for tiny ones an f16 denormal with 2^-25
// absolute error -- conversions and the matrix instructions honour denormals; weights are pre-multiplied by a power
// of two per layer so that their remainders are normal too, and 2^-k comes back in the epilogue).  A product a*w is
//      a_hi*w_hi  +  a_hi*w_lo + a_lo*w_hi                 (the lo*lo term, 2^-22 relative, is dropped)
// -- three v_mfma_f32_32x32x16_f16 per 16-deep k-block into ONE fp32 accumulator, i.e. an effective peak of
// 2.5 PFLOP/s / 3 = 833 TFLOP/s against 157 TFLOP/s for the exact-fp32 matrix op.
//
// Storage ("split NHWC"): channels in groups of 8; a group is 32 bytes = 8 hi halves then 8 lo halves.
// A pixel with C channels is C/8 groups = 4*C bytes -- the same footprint and strides as fp32 NHWC,
// so the buffers, channel offsets and the concat-by-offset trick of the fp32 path carry over.  One
// 16-byte load is exactly one MFMA operand fragment (8 consecutive k for one row/column).
// Weights are packed [tap][cout][cin groups] in the same group format.
//
// conv_igemm_h3<TAPS>: D[cout][pixel] (weights are the MFMA "A" operand, activations "B": the
// accumulator then holds 4 consecutive output channels per register quad, which is what the split
// layout wants for 8-byte stores).  256 threads = 4 waves, tile 64 couts x 256 pixels (16x16), each wave
// 64 couts x 64 pixels = 2x2 MFMA tiles x {main, corr}.  Per 16-channel chunk the halo tile and the
// weight slab go global -> LDS by DMA (global_load_lds_dwordx4, no VGPR round trip); the LDS image is
// lane-linear, so the bank swizzle (16-byte slot ^= (row>>2)&3) is applied on the SOURCE address and on
// the fragment reads.  Out-of-image halo pixels are DMA'd from a zero page.
#include <cstdlib>

#include "nn_kernels.h"

#if defined(LM_H3_TRACE) || defined(LM_H3_TIMELINE)  // lab builds only (tools/ubench/conv_lab.hip)
__device__ unsigned* lm_h3_trace_ptr = nullptr;
#endif

namespace lm {

namespace {

constexpr int TH = 16, TW = 16, TN = 64, KC = 16;

// Staging through LDS reads back, as 16-byte units, bytes that were written as 8-byte units: these accesses
// must be exempt from type-based alias analysis or the loads may be hoisted above the stores.
typedef uint2 __attribute__((may_alias)) uint2_a;
typedef uint4 __attribute__((may_alias)) uint4_a;

__device__ __forceinline__ void split_store4(char* group_base, int half_off_bytes, float v0, float v1, float v2, float v3) {
    // writes 4 consecutive channels (hi at +0, lo at +16 of the 32-byte group); half_off_bytes = (c & 7) * 2
    uint2 ph, plo;
    lm_split4(v0, v1, v2, v3, &ph, &plo);
    *reinterpret_cast<uint2*>(group_base + half_off_bytes) = ph;
    *reinterpret_cast<uint2*>(group_base + 16 + half_off_bytes) = plo;
}

template <int TAPS>
struct H3Smem {
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO;
    static constexpr int A_ROWS = PH * PW;      // halo pixels
    static constexpr int W_ROWS = TAPS * TN;    // (tap, cout)
    static constexpr int ROW_BYTES = KC * 4;    // 2 groups x 32 B = 4 slots of 16 B
    static constexpr int A_BYTES = ((A_ROWS * ROW_BYTES + 1023) / 1024) * 1024;  // whole 1 KiB DMA pieces
    static constexpr int W_BYTES = W_ROWS * ROW_BYTES;
    static constexpr int BYTES = A_BYTES + W_BYTES;
};

}  // namespace

template <int TAPS>
__global__ __launch_bounds__(256, 2) void conv_igemm_h3(ConvParamsH3 p) {
    using SM = H3Smem<TAPS>;
    constexpr int HALO = SM::HALO, PW = SM::PW, A_ROWS = SM::A_ROWS, W_ROWS = SM::W_ROWS;
    LM_DYN_SMEM(smem);
    char* As = smem;
    char* Ws = smem + SM::A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH, n0 = blockIdx.y * TN;

    lm_f32x16 accm[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accm[i][j][r] = 0.f;
            }

    const int li = lane & 31, kb = lane >> 5;
    const int pr = li >> 4, pc = li & 15;
    const char* __restrict__ in_b = p.in + ((size_t)b * p.H * p.W * p.in_cstride + p.in_coff) * 4;
    const size_t w_row_bytes = (size_t)p.Cin * 4;

    for (int c0 = 0; c0 < p.Cin; c0 += KC) {
        __syncthreads();  // all fragment reads of the previous chunk are done
        // ---- DMA the activation halo tile: A_ROWS rows x 4 slots; physical slot ps holds logical slot ps ^ ((row>>2)&3)
        for (int piece = wave; piece * 64 < A_ROWS * 4; piece += 4) {
            const int idx = piece * 64 + lane;
            if (idx < A_ROWS * 4) {
                const int row = idx >> 2, ls = (idx & 3) ^ ((row >> 2) & 3);
                const int py = row / PW, px = row - py * PW;
                const int gy = y0 + py - HALO, gx = x0 + px - HALO;
                const char* src = p.zeros;
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                    src = in_b + ((size_t)gy * p.W + gx) * p.in_cstride * 4 + (size_t)((c0 >> 3) + (ls >> 1)) * 32 + (ls & 1) * 16;
                lm_global_load_lds16(src, As + piece * 1024);
            }
        }
        // ---- DMA the weight slab: W_ROWS rows (tap, cout) x 4 slots, same swizzle
        for (int piece = wave; piece < W_ROWS * 4 / 64; piece += 4) {
            const int idx = piece * 64 + lane;
            const int row = idx >> 2, ls = (idx & 3) ^ ((row >> 2) & 3);
            const int tap = row / TN, n = row - tap * TN;
            const char* src = p.w + ((size_t)tap * p.Cout + n0 + n) * w_row_bytes + (size_t)((c0 >> 3) + (ls >> 1)) * 32 + (ls & 1) * 16;
            lm_global_load_lds16(src, Ws + piece * 1024);
        }
        lm_dma_wait_all();
        __syncthreads();  // the DMA'd tile is complete for every wave
#pragma unroll 1
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = (TAPS == 9) ? tap / 3 : 0, dx = (TAPS == 9) ? tap - 3 * dy : 0;
            lm_h16x8 whi[2], wlo[2], ahi[2], alo[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = tap * TN + 32 * mt + li;
                const int sw = (row >> 2) & 3;
                whi[mt] = *reinterpret_cast<const lm_h16x8*>(Ws + row * 64 + ((2 * kb) ^ sw) * 16);
                wlo[mt] = *reinterpret_cast<const lm_h16x8*>(Ws + row * 64 + ((2 * kb + 1) ^ sw) * 16);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int row = (4 * wave + 2 * nt + pr + dy) * PW + pc + dx;
                const int sw = (row >> 2) & 3;
                ahi[nt] = *reinterpret_cast<const lm_h16x8*>(As + row * 64 + ((2 * kb) ^ sw) * 16);
                alo[nt] = *reinterpret_cast<const lm_h16x8*>(As + row * 64 + ((2 * kb + 1) ^ sw) * 16);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    accm[mt][nt] = lm_mfma_f32_32x32x16_f16(whi[mt], ahi[nt], accm[mt][nt]);
                    accm[mt][nt] = lm_mfma_f32_32x32x16_f16(whi[mt], alo[nt], accm[mt][nt]);
                    accm[mt][nt] = lm_mfma_f32_32x32x16_f16(wlo[mt], ahi[nt], accm[mt][nt]);
                }
        }
    }

    // ---- epilogue.  D[cout][pixel]: lane = pixel (li) of the N-tile, register quad g = 4 consecutive couts.
    const bool bn = p.bn_s != nullptr;
    float amax = 0.f;  // f16 range guard (ConvParamsH3::range_flag)
    const int Hp = p.H >> 1, Wp = p.W >> 1;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int y = y0 + 4 * wave + 2 * nt + pr, x = x0 + pc;
        const bool inside = y < p.H && x < p.W;
        char* orow = p.out + ((((size_t)b * p.H + y) * p.W + x) * p.out_cstride + p.out_coff) * 4;
        const bool pool_lane = p.pool != nullptr && pr == 0 && (pc & 1) == 0;
        char* prow = p.pool ? p.pool + ((((size_t)b * Hp + (y >> 1)) * Wp + (x >> 1)) * p.pool_cstride + p.pool_coff) * 4 : nullptr;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cb = n0 + 32 * mt + 8 * g + 4 * kb;  // first of 4 consecutive output channels
                float4 bias = *reinterpret_cast<const float4*>(p.bias + cb);
                if (TAPS == 9 && p.border_corr != nullptr) {  // deferred-shift input: taps outside the image saw 0, not -T
                    const int mask = (y == 0 ? 1 : 0) | (y == p.H - 1 ? 2 : 0) | (x == 0 ? 4 : 0) | (x == p.W - 1 ? 8 : 0);
                    const float4 c = *reinterpret_cast<const float4*>(p.border_corr + (size_t)mask * p.Cout + cb);
                    bias.x -= c.x;
                    bias.y -= c.y;
                    bias.z -= c.z;
                    bias.w -= c.w;
                }
                float v[4];
                v[0] = fmaf(accm[mt][nt][4 * g + 0], p.acc_scale, bias.x);
                v[1] = fmaf(accm[mt][nt][4 * g + 1], p.acc_scale, bias.y);
                v[2] = fmaf(accm[mt][nt][4 * g + 2], p.acc_scale, bias.z);
                v[3] = fmaf(accm[mt][nt][4 * g + 3], p.acc_scale, bias.w);
                if (bn) {
                    const float4 s = *reinterpret_cast<const float4*>(p.bn_s + cb);
                    const float4 sh = *reinterpret_cast<const float4*>(p.bn_t + cb);
                    v[0] = fmaf(fmaxf(v[0], 0.f), s.x, sh.x);
                    v[1] = fmaf(fmaxf(v[1], 0.f), s.y, sh.y);
                    v[2] = fmaf(fmaxf(v[2], 0.f), s.z, sh.z);
                    v[3] = fmaf(fmaxf(v[3], 0.f), s.w, sh.w);
                }
                amax = fmaxf(fmaxf(amax, fabsf(v[0])), fmaxf(fmaxf(fabsf(v[1]), fabsf(v[2])), fabsf(v[3])));
                const int grp = cb >> 3;  // 8-channel group index inside the output tensor slice
                if (inside) split_store4(orow + (size_t)grp * 32, (cb & 7) * 2, v[0], v[1], v[2], v[3]);
                if (p.pool != nullptr) {  // avg_pool2d(2): partners are lane^1 (x+1) and lane^16 (y+1)
                    float q[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float h = v[k] + __shfl_xor(v[k], 1);
                        q[k] = 0.25f * (h + __shfl_xor(h, 16));
                    }
                    if (pool_lane && y + 1 < p.H && x + 1 < p.W) split_store4(prow + (size_t)grp * 32, (cb & 7) * 2, q[0], q[1], q[2], q[3]);
                }
            }
        }
    }
    if (p.range_flag != nullptr && !(amax < kF16Guard)) atomicOr(p.range_flag, 1u);
}

// ---------------------------------------------------------------------------------------------
// Persistent variant for W % 32 == 0 and the 16x16 bottleneck (second geometry).
//
// One 512-thread workgroup per CU (8 waves = 2 per SIMD, <= 256 VGPRs) walks a list of work items
// (64 couts x 16 rows x 32 cols of one slice) and software-pipelines ACROSS chunks and items:
//   * LDS is double buffered (2 x 75 KiB); 16-channel chunk c of an item lives in buffer c & 1 (Cin/16 is even).
//   * global -> LDS by buffer-descriptor LDS-DMA (lm_dma16): branch-free, out-of-image halo lanes carry an
//     out-of-range offset and the hardware writes zeros.  Per item each lane's source offsets are computed ONCE
//     (voff); per chunk a DMA piece is "set M0, issue" with the chunk offset in the scalar offset operand.
//   * the nine taps of a chunk are a register pipeline (fragment reads of tap t+1 in flight under the MFMAs of tap
//     t, two fragment sets) that runs THROUGH the chunk barrier: the single barrier of a chunk sits in front of
//     the LAST tap's MFMAs (all of this wave's LDS reads of the chunk have returned; its DMAs of the next chunk
//     have landed), the first fragment reads of the next chunk are issued right behind it, and the 12 MFMAs of the
//     last tap cover barrier skew and read latency.  The DMA pieces of chunk c+2 (or of the next item's chunk 0)
//     are issued one at a time between the MFMAs of the first three taps of chunk c+1.
//   * an item ends with its epilogue (staged through the buffer of its last chunk); the next item's first chunk
//     and epilogue constants are already resident.  The barrier behind the epilogue only orders LDS accesses
//     (lm_barrier_lds): the epilogue's global stores stay in flight.
// Wave w = row pair w: both 32-cout M-tiles against two 32-pixel N-tiles (rows 2w, 2w+1): 64 accumulator
// registers; 8 fragment reads feed 12 MFMAs per tap.  Fragment reads are hand-issued ds_read_b128 with immediate
// offsets (the bank swizzle depends on the halo COLUMN only, so tap shifts are plain byte offsets); both
// fragment streams are bank-conflict free.  The 2x2 average pool is in-lane (two N-tiles) + one lane^1 exchange.
namespace {
// G16 = false: one slice, tile 16 rows x 32 cols.  G16 = true (16-pixel-wide levels): TWO slices per item, each
// a 16x16 tile with its own halo; waves 0-3 work on the first slice, 4-7 on the second.
template <int TAPS, bool G16>
struct H3WSmem {
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int TWW = G16 ? 16 : 32;
    static constexpr int NSL = G16 ? 2 : 1;
    static constexpr int PW = TWW + 2 * HALO, PH = TH + 2 * HALO;
    static constexpr int SL_ROWS = PH * PW;  // halo pixels of one slice
    static constexpr int A_ROWS = NSL * SL_ROWS;
    static constexpr int NTSTEP = G16 ? 2 : 1;  // halo rows between the two N-tiles of a wave
    static constexpr int W_ROWS = TAPS * TN;
    static constexpr int A_PIECES = (A_ROWS * 4 + 63) / 64, W_PIECES = W_ROWS * 4 / 64;
    static constexpr int A_BYTES = A_PIECES * 1024, W_BYTES = W_PIECES * 1024;
    static constexpr int BUF_BYTES = A_BYTES + W_BYTES;
    static constexpr int NW = 8;
    static constexpr int A_PER_WAVE = (A_PIECES + NW - 1) / NW, W_PER_WAVE = (W_PIECES + NW - 1) / NW;
};

__device__ __forceinline__ float4 as_float4(const lm_h16x8& v) {
    float4 r;
    memcpy(&r, &v, 16);
    return r;
}
}  // namespace

// Lab builds (tools/ubench/conv_lab.hip) can ablate one stream of the kernel to price it: -DLM_H3_ABLATE=1 issues no fragment
// reads after an item's first chunk (the registers keep its operands), =2 stages nothing after an item's second chunk.
#ifdef LM_H3_ABLATE
#define LM_ABL_READS(ci) (LM_H3_ABLATE == 1 && (ci) > 0)
#define LM_ABL_DMA(c0) (LM_H3_ABLATE == 2 && (c0) > KC)
#else
#define LM_ABL_READS(ci) false
#define LM_ABL_DMA(c0) false
#endif
// fragment reads of tap (DY, DX) from the chunk buffer at AS into the set F: whi0 whi1 wlo0 wlo1 a0hi a0lo a1hi a1lo
#define H3P_READS(F, AS, DY, DX)                                                              \
    if (!abl_r) {                                                                             \
        const int a_lo_ = a_off[DX] ^ 16;                                                     \
        LM_LDS_READ128(F[4], (AS) + a_off[DX], (DY) * ROWB);                                  \
        LM_LDS_READ128(F[5], (AS) + a_lo_, (DY) * ROWB);                                      \
        LM_LDS_READ128(F[6], (AS) + a_off[DX], ((DY) + NTSTEP) * ROWB);                       \
        LM_LDS_READ128(F[7], (AS) + a_lo_, ((DY) + NTSTEP) * ROWB);                           \
        LM_LDS_READ128(F[0], (AS) + w_off, (3 * (DY) + (DX)) * (TN * 64));                    \
        LM_LDS_READ128(F[1], (AS) + w_off, (3 * (DY) + (DX)) * (TN * 64) + 2048);             \
        LM_LDS_READ128(F[2], (AS) + w_off_lo, (3 * (DY) + (DX)) * (TN * 64));                 \
        LM_LDS_READ128(F[3], (AS) + w_off_lo, (3 * (DY) + (DX)) * (TN * 64) + 2048);          \
    } else                                                                                    \
        (void)0
#define H3P_MFMA3(F, I0, I1, I2)                                                              \
    do {                                                                                      \
        H3P_MFMA1(F, I0);                                                                     \
        H3P_MFMA1(F, I1);                                                                     \
        H3P_MFMA1(F, I2);                                                                     \
    } while (0)
// the 12 matrix instructions of one tap, numbered 0..11: (M-tile, N-tile) fastest, then hi*hi, hi*lo, lo*hi
#define H3P_MFMA1(F, I)                                                                                                           \
    accm[((I) >> 1) & 1][(I) & 1] = lm_mfma_f32_32x32x16_f16(F[((I) < 8 ? 0 : 2) + (((I) >> 1) & 1)],                             \
                                                            F[4 + 2 * ((I) & 1) + (((I) >> 2) == 1 ? 1 : 0)], accm[((I) >> 1) & 1][(I) & 1])
// (PS: the producer slot that runs beside these matrix instructions -- prod_step(PS, m), six micro-steps, one in front of every
// pair; nothing at all when the instantiation has no producer)
#define H3P_MFMA2(F, I0, I1) \
    do {                     \
        H3P_MFMA1(F, I0);    \
        H3P_MFMA1(F, I1);    \
    } while (0)
#define H3P_MFMAS(F, PS)       \
    do {                       \
        prod_step(PS, 0);      \
        H3P_MFMA2(F, 0, 1);    \
        prod_step(PS, 1);      \
        H3P_MFMA2(F, 2, 3);    \
        prod_step(PS, 2);      \
        H3P_MFMA2(F, 4, 5);    \
        prod_step(PS, 3);      \
        H3P_MFMA2(F, 6, 7);    \
        prod_step(PS, 4);      \
        H3P_MFMA2(F, 8, 9);    \
        prod_step(PS, 5);      \
        H3P_MFMA2(F, 10, 11);  \
    } while (0)
// the same with four DMA slots (K0 .. K0+3) spread between the matrix instructions (behind every third one)
#define H3P_MFMAS_D(F, K0, PS) \
    do {                       \
        prod_step(PS, 0);      \
        H3P_MFMA2(F, 0, 1);    \
        prod_step(PS, 1);      \
        H3P_MFMA1(F, 2);       \
        dma_slot((K0) + 0);    \
        H3P_MFMA1(F, 3);       \
        prod_step(PS, 2);      \
        H3P_MFMA2(F, 4, 5);    \
        dma_slot((K0) + 1);    \
        prod_step(PS, 3);      \
        H3P_MFMA2(F, 6, 7);    \
        prod_step(PS, 4);      \
        H3P_MFMA1(F, 8);       \
        dma_slot((K0) + 2);    \
        H3P_MFMA1(F, 9);       \
        prod_step(PS, 5);      \
        H3P_MFMA2(F, 10, 11);  \
        dma_slot((K0) + 3);    \
    } while (0)
#define H3P_WAITF(N, F) LM_LDS_WAIT8(N, F[0], F[1], F[2], F[3], F[4], F[5], F[6], F[7])
// one pipeline step: issue the reads of the NEXT tap into FN, wait for the current set FC, run its MFMAs
// (PS: the producer slot of the step -- prod_step, 1 + the tap whose matrix instructions it runs beside; slot 0 lies beside the last
// tap of the previous chunk, right behind the barrier that freed the buffer being filled)
#define H3P_STEP(FC, FN, AS, NDY, NDX, PS) \
    do {                                   \
        H3P_READS(FN, AS, NDY, NDX);       \
        H3P_WAITF(8, FC);                  \
        H3P_MFMAS(FC, PS);                 \
    } while (0)
#define H3P_STEP_D(FC, FN, AS, NDY, NDX, K0, PS) \
    do {                                         \
        H3P_READS(FN, AS, NDY, NDX);             \
        H3P_WAITF(8, FC);                        \
        H3P_MFMAS_D(FC, K0, PS);                 \
    } while (0)
// taps 0..7 of the chunk in buffer AS (tap 0 already in flight in FA); leaves tap 8 in flight in FA
#define H3P_CHUNK_STEPS(FA, FB, AS)             \
    do {                                        \
        H3P_STEP_D(FA, FB, AS, 0, 1, 0, 1);     \
        H3P_STEP_D(FB, FA, AS, 0, 2, 4, 2);     \
        H3P_STEP_D(FA, FB, AS, 1, 0, 8, 3);     \
        H3P_STEP(FB, FA, AS, 1, 1, 4);          \
        H3P_STEP(FA, FB, AS, 1, 2, 5);          \
        H3P_STEP(FB, FA, AS, 2, 0, 6);          \
        H3P_STEP(FA, FB, AS, 2, 1, 7);          \
        H3P_STEP(FB, FA, AS, 2, 2, 8);          \
    } while (0)

// HEAD: the fused-head form of the epilogue (last decoder conv; 1: labels only, 2: labels + log-probabilities) -- separate
// instantiations, so that the 16 other launches of a forward do not carry its registers.
// epilogue constants (bias, BN scale, BN shift) of 4 consecutive channels from the item's staged arrays
#if LM_H3_FOLD_SCALE  // the bias only: scale and shift of the layer travel with its consumers' weights and biases
#define H3P_EPI_READS(E, CL) LM_LDS_READ128(E[0], ep + (CL) * 4, 0)
#define H3P_EPI_WAIT(NEXT, E) LM_LDS_WAIT1((NEXT) ? 1 : 0, E[0])
#else
#define H3P_EPI_READS(E, CL)                        \
    do {                                            \
        LM_LDS_READ128(E[0], ep + (CL) * 4, 0);     \
        LM_LDS_READ128(E[1], ep + (CL) * 4, TN * 4); \
        LM_LDS_READ128(E[2], ep + (CL) * 4, 2 * TN * 4); \
    } while (0)
#define H3P_EPI_WAIT(NEXT, E) LM_LDS_WAIT3((NEXT) ? 3 : 0, E[0], E[1], E[2])
#endif
#define H3P_EPI_CL(MG) (32 * ((MG) >> 2) + 8 * ((MG) & 3) + 4 * kb)

// ---- PROD = 1: the first layer of the network inside this kernel's loader (ConvParamsH3::fc_x) -------------------------------
// The 64-channel input tensor of down_path.0's second conv is the first conv's output: one input channel, 9 multiply-adds, ReLU and
// a scale per value (resunet.py:93-95) -- 16 MiB per slice written and read back for 37.7 MMAC.  In this form the tensor never
// exists: per work item the 20 x 36 fp32 patch of the network input under the item's halo tile (grown by the first conv's own halo)
// is DMA'd to LDS once (2.9 KB instead of 4 x 39 KB of activation chunks), and the activation image of 16-channel chunk c + 1 is
// computed on the vector ALU -- the operation order of first_conv_h3_kernel, so the values are bit-identical -- and written to the
// LDS buffer the DMA would have filled, in slices that run beside the matrix instructions of chunk c.  A task is (halo pixel,
// 8-channel group): 612 x 2 per chunk.  Thread t takes pixel t for both groups (rounds 0 and 1: one read of the pixel's 3 x 3 input
// neighbourhood serves both) and, for t < 200, pixel 512 + t % 100 for group t / 100 (round 2); a round is cut in two halves of 4
// channels, a half runs beside the 12 matrix instructions of one tap in six micro-steps (LDS reads of the weights one step ahead
// of the packed fused multiply-adds that use them), one in front of every matrix-instruction pair.
namespace {
constexpr int FC_PW = 36, FC_PH = 20, FC_PATCH = FC_PW * FC_PH;
constexpr int FC_CONST = 9 * 64 + 64 + 64;  // w[9][64] | bias[64] | bn scale[64]
// LM_PROD_DPP (needs LM_H3_FOLD_SCALE: no scale row): the constants live in LDS as 10 rows (taps 0..8, bias) of 64 + 4 floats -- the
// padding puts the 16-byte pieces of consecutive rows 4 banks apart -- and a half-task fetches ALL its 40 constants with ONE
// ds_read_b128: lane l reads row min(l & 15, 9) at its 4 channels, so register j of the result holds, in lane k of every row of 16
// lanes, constant k of channel j, and the multiply-adds take it from there by DPP row broadcast (lm_fmac_rowbcast).  Ten broadcast
// ds_read_b128 per half-task (1 KiB of LDS bandwidth each) become one.
#ifndef LM_PROD_DPP
#define LM_PROD_DPP (LM_H3_FOLD_SCALE ? 1 : 0)
#endif
constexpr int FC_ROW = 68;
static_assert(!LM_PROD_DPP || (LM_H3_FOLD_SCALE && 10 * FC_ROW <= FC_CONST), "the DPP producer's constant table");
constexpr int FC_TASKS = 18 * 34 * 2;

}  // namespace

// KS (3x3 form): the split-K instantiation -- ConvParamsH3::ksplit parts per output tile, every part a work item of its own that sums
// Cin / ksplit input channels in a chain that starts at zero and leaves its raw fp32 accumulators in kpart[ks];
// splitk_reduce3_h3_kernel adds the parts in index order and runs the conv's epilogue.  The accumulator chain is what the split-f16
// arithmetic's error comes from (3 K / 16 roundings at a magnitude that grows along the chain: the error of a layer goes with
// K / sqrt(parts)); this is the "precise" tier of the accuracy guard (nn_engine.hip), a separate instantiation so that the 17
// launches of the fast tier do not carry a line of it.
template <int TAPS, bool G16, int HEAD = 0, int PROD = 0, bool KS = false>
__global__ __launch_bounds__(512) void conv_igemm_h3p(ConvParamsH3 p, int n_ptiles, int n_items, int xcd_order) {
    using SM = H3WSmem<TAPS, G16>;
    static_assert(!KS || (TAPS == 9 && HEAD == 0 && PROD == 0), "the split-K instantiation is the plain 3x3 form");
    constexpr bool KSPLIT = TAPS == 1 || KS;  // instantiations that can run with ConvParamsH3::ksplit > 1
    static_assert(PROD == 0 || (TAPS == 9 && !G16 && HEAD == 0), "the loader-side producers belong to the 32-wide 3x3 form");
    constexpr int HALO = SM::HALO, PW = SM::PW, TWW = SM::TWW, NW = SM::NW, ROWB = PW * 64, NTSTEP = SM::NTSTEP;
    // Bank swizzle of the activation tile: 16-byte slot ^= (halo column >> ASWZ) & 3.  A ds_read_b128 lane group of 16
    // lanes spans 16 consecutive-ish columns of ONE row in the 32-wide geometry (>> 2 is conflict free for all three dx)
    // but 8 + 8 columns of TWO rows in the 16-wide one, where the 18-pixel halo rows lie half the banks apart: there >> 1
    // is the conflict-free choice (enumerated over lane groups x dx x hi/lo; with >> 2 SQ_LDS_BANK_CONFLICT was 23 % of
    // the LDS cycles, now 1 %).  The 1x1 form has no halo (rows a whole number of bank sets apart): >> 2 for both widths.
    constexpr int ASWZ = (G16 && TAPS == 9) ? 1 : 2;
    constexpr int HSTR = 144;                    // staged pixel stride of an epilogue pass: 128 B (32 channels) of split data + 16 B pad
    constexpr int STAGE_BYTES = NW * 64 * HSTR;  // both rows of every wave, half of the item's channels
    // The epilogue staging area reuses the DMA buffer of the last chunk when it fits (3x3: 75 KiB), else it is extra.
    // A pipeline stage of the 1x1 form holds TWO chunk images (32 channels per barrier: with 16 the kernel is bound by the
    // barrier + DMA round trip, not by anything it computes); the 3x3 form holds one.
    constexpr int SUB = TAPS == 1 ? 2 : 1;
    constexpr int STAGE_EXTRA = STAGE_BYTES <= SUB * SM::BUF_BYTES ? 0 : STAGE_BYTES;
    __shared__ __attribute__((aligned(1024))) char lds[2 * SUB * SM::BUF_BYTES + STAGE_EXTRA];
    __shared__ __attribute__((aligned(16))) float epi[2][3][TN];  // bias, bn scale, bn shift of the item (double buffered)
    __shared__ __attribute__((aligned(16))) float hw[HEAD ? kMaxClasses * 64 + kMaxClasses : 4];  // fused head: weights, bias
    // fused first layer: input patches of the running and the next item, the first conv's constants (with the two chunk buffers and
    // the epilogue constants this fills the CU's 160 KB to within ~100 bytes)
    __shared__ __attribute__((aligned(16))) float fcx[PROD == 1 ? 2 * FC_PATCH : 4];
    __shared__ __attribute__((aligned(16))) float fcc[PROD == 1 ? FC_CONST : 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = lm_uniform(tid >> 6);
    const int li = lane & 31, kb = lane >> 5;
    // wave -> pixels.  G32: rows 2w, 2w+1 of the tile, lane li = column.  G16: slice w>>2, rows 4(w&3)..+3 as two
    // N-tiles of 2 rows x 16 cols (lane li = 16*row + column).
    const int wsl = G16 ? (wave >> 2) : 0;                       // slice of the item this wave works on
    const int wrow = G16 ? 4 * (wave & 3) + (li >> 4) : 2 * wave;  // first tile row of this lane (N-tile 0)
    const int wcol = G16 ? (li & 15) : li;

    // ---- fragment byte offsets inside a buffer (item invariant)
    int a_off[TAPS == 9 ? 3 : 1];
#pragma unroll
    for (int dx = 0; dx < (TAPS == 9 ? 3 : 1); ++dx) {
        const int px = wcol + dx;
        a_off[dx] = (wsl * SM::SL_ROWS + wrow * PW + px) * 64 + ((2 * kb) ^ ((px >> ASWZ) & 3)) * 16;
    }
    const int w_off = SM::A_BYTES + li * 64 + ((2 * kb) ^ ((li >> 2) & 3)) * 16;  // second M-tile: +2048
    const int w_off_lo = w_off ^ 16;

    // ---- DMA lane geometry (item invariant): which halo pixel / weight row this lane feeds
    unsigned relA[SM::A_PER_WAVE];  // source offset of this lane's 16 bytes relative to the halo tile's top-left pixel
    int pyx[SM::A_PER_WAVE];        // py | px << 8 | slice << 16, or -1 when the lane has nothing to do for that piece
#pragma unroll
    for (int j = 0; j < SM::A_PER_WAVE; ++j) {
        const int piece = wave + NW * j, idx = piece * 64 + lane;
        pyx[j] = -1;
        relA[j] = 0;
        if (PROD != 1 && piece < SM::A_PIECES && idx < SM::A_ROWS * 4) {
            const int row = idx >> 2;
            const int sl = row / SM::SL_ROWS, rr = row - sl * SM::SL_ROWS;
            const int py = rr / PW, px = rr - py * PW;
            const int ls = (idx & 3) ^ ((px >> ASWZ) & 3);  // logical 16-byte slot behind this lane's physical one
            relA[j] = (unsigned)(((sl * p.H + py) * p.W + px) * p.in_cstride * 4 + (ls >> 1) * 32 + (ls & 1) * 16);
            pyx[j] = py | (px << 8) | (sl << 16);
        }
    }
    unsigned voffW;  // weight row of this lane in piece `wave`; piece wave + 8 j lies two taps further per j
    {
        const int idx = wave * 64 + lane;
        const int row = idx >> 2, ls = (idx & 3) ^ ((row >> 2) & 3);
        const int tap = row / TN, n = row - tap * TN;
        voffW = (unsigned)((tap * p.Cout + n) * p.Cin * 4 + (ls >> 1) * 32 + (ls & 1) * 16);
    }
    const unsigned w_piece_stride = (unsigned)(NW * 64 / 4 / TN) * (unsigned)p.Cout * (unsigned)p.Cin * 4u;
    const unsigned slice_bytes = (unsigned)p.H * (unsigned)p.W * (unsigned)p.in_cstride * 4u;
    const lm_rsrc rsrcA = PROD == 1 ? lm_make_rsrc(p.fc_x, (size_t)p.B * p.H * p.W * 4)  // the network input: fp32 [B][H][W]
                                    : lm_make_rsrc(p.in + (size_t)p.in_coff * 4, (size_t)p.B * slice_bytes - (size_t)p.in_coff * 4);
    const lm_rsrc rsrcW = lm_make_rsrc(p.w, (size_t)TAPS * p.Cout * p.Cin * 4);
    const int tiles_x = p.W / TWW;
    const int nchunks = p.Cin / KC / ((KSPLIT && p.ksplit > 1) ? p.ksplit : 1);  // per item; even (checked by the launcher)
    // Conv -> ReLU -> BatchNorm for every 3x3 conv of the network, bias only for the decoder's 1x1 convs (resunet.py:93-105,
    // :131-133): a property of the instantiation, not a run-time select per value (the launcher sends anything else to the
    // simple kernel)
    constexpr bool bn = TAPS == 9;

    // Work-item order.  `it` runs over a (padded) index space; workgroup g takes it = g, g + grid, ... and lives on XCD
    // g % 8 (workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MB L2).
    //   xcd_order = 0: cout tile major -- every concurrent workgroup streams a DIFFERENT activation tile, and each tile
    //     is fetched from HBM once per cout tile (Cout/64 times).
    //   xcd_order = 1: it -> XCD x = it % 8, then cout tile fastest, then pixel tile (pt = 8 * ptl + x): the 32
    //     workgroups that run together on one XCD work on a few pixel tiles x ALL cout tiles, so an activation tile is
    //     fetched once and hit in that XCD's L2 by the other cout tiles.  Returns false for the padding of the last row.
    const int n_ct = p.Cout / TN;
    const int tiles_y = (p.H + TH - 1) / TH;
    // every tile count of this network is a power of two: shifts instead of integer divisions on the item-switch path
    const bool pow2 = ((n_ct & (n_ct - 1)) | (tiles_x & (tiles_x - 1)) | (tiles_y & (tiles_y - 1))) == 0;
    const int sh_ct = 31 - __clz(n_ct), sh_tx = 31 - __clz(tiles_x), sh_ty = 31 - __clz(tiles_y);
    // Split-K (the 1x1 form and the KS instantiation of the 3x3 form, ConvParamsH3::ksplit > 1): the cout-tile index also counts the K split -- item (ks, ct, pt) sums input
    // channels [ks, ks + 1) * Cin / ksplit and leaves its raw fp32 accumulators in kpart[ks]; splitk_reduce_h3_kernel adds the parts
    // in fixed order.  kc = first input channel of the item's range.
    const int ksplit = (KSPLIT && p.ksplit > 1) ? p.ksplit : 1;
    auto decode = [&](int it, int& b, int& y0, int& x0, int& n0, int& kc) -> bool {
        int ct, pt;
        if (xcd_order) {
            const int x = it & 7, s = it >> 3;
            ct = pow2 ? (s & (n_ct - 1)) : s % n_ct;
            pt = (pow2 ? (s >> sh_ct) : s / n_ct) * 8 + x;
        } else {
            ct = it / n_ptiles;
            pt = it - ct * n_ptiles;
        }
        kc = 0;
        if (KSPLIT && ksplit > 1) {  // (cout-tile-major order only: the launcher sees to it)
            const int ks = ct / n_ct;
            ct -= ks * n_ct;
            kc = ks * (p.Cin / ksplit);
        }
        const bool valid = pt < n_ptiles;
        const int tx = pow2 ? (pt & (tiles_x - 1)) : pt % tiles_x;
        pt = pow2 ? (pt >> sh_tx) : pt / tiles_x;
        const int ty = pow2 ? (pt & (tiles_y - 1)) : pt % tiles_y;
        b = (pow2 ? (pt >> sh_ty) : pt / tiles_y) * SM::NSL;  // first slice of the item
        y0 = ty * TH;
        x0 = tx * TWW;
        n0 = ct * TN;
        return valid;
    };
    // per-lane source offsets of the halo tile of item (b, y0, x0), relative to slice b of rsrcA
    auto item_voffs = [&](int b, int y0, int x0, unsigned* voff) __attribute__((always_inline)) {
        if constexpr (PROD == 1) return;  // no activation DMA: the loader computes the tile
        const int ioff = ((y0 - HALO) * p.W + (x0 - HALO)) * p.in_cstride * 4;  // may be negative; the sum below is not
#pragma unroll
        for (int j = 0; j < SM::A_PER_WAVE; ++j) {
            const int gy = y0 + (pyx[j] & 0xff) - HALO, gx = x0 + ((pyx[j] >> 8) & 0xff) - HALO;
            const bool inb = pyx[j] >= 0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && b + (pyx[j] >> 16) < p.B;
            voff[j] = inb ? (unsigned)((int)relA[j] + ioff) : LM_DMA_OOB;
        }
    };

    unsigned voffC[SM::A_PER_WAVE], voffN[SM::A_PER_WAVE];  // this item's / the next item's source offsets
    // ---- the pending stage: the chunk whose DMA pieces the slots of the running taps issue
    bool d_next = false;  // the stage belongs to the NEXT item (its source offsets are voffN)
    unsigned d_soffA = 0, d_soffW = 0;
    char* d_buf = lds;
    int d_nA = 0, d_nW = 0, d_epi = -1, d_n0 = 0;  // piece counts (0: nothing to stage); epi buffer to fill or -1
    // ---- the producer's stage (PROD != 0): the chunk image the vector ALU computes beside the running taps
    bool pr_on = false;  // wave-uniform
    int pr_c0 = 0, pr_y0 = 0, pr_x0 = 0, pr_par = 0;
    char* pr_buf = lds;
    lm_f32x2 pr_x[5];      // a pixel's 3 x 3 input neighbourhood (taps 2 i, 2 i + 1), kept for the half-tasks that share it
    lm_f32x4 pr_w[LM_PROD_DPP ? 1 : 5];  // weight / constant reads in flight (issued one micro-step ahead of their use)
    float pr_s[4] = {0.f, 0.f, 0.f, 0.f};  // LM_PROD_DPP: the half-task's four channel accumulators as single registers
    lm_f32x2 pr_a[2];      // the half-task's four channel accumulators
    unsigned pr_max = 0u;  // f16 range guard of the values the producer writes
    // this thread's two halo pixels (item invariant): py | px << 8, byte offset of the pixel's group-0 hi slot in a chunk image
    int pr_pyx[2] = {0, 0}, pr_woff[2] = {0, 0};
    if constexpr (PROD == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#if LM_PROD_DPP  // round 2: group g = tid / 112 (seven whole rows of 16 lanes per group: the constants are per row), pixel 512 + tid % 112
            const int pxl = i == 0 ? tid : min(512 + (tid % 112), 611);
#else
            const int pxl = i == 0 ? tid : 512 + (tid % 100);
#endif
            const int py = pxl / 34, px = pxl - 34 * py;
            pr_pyx[i] = py | (px << 8);
            pr_woff[i] = pxl * 64 + (((px >> 2) & 3) << 4);  // logical slot 0 at physical slot (px >> 2) & 3; slot s at ^ (s << 4)
        }
    }
    int it = blockIdx.x;
    int b, y0, x0, n0, kc0 = 0;
    int nb = 0, ny0 = 0, nx0 = 0, nn0 = 0, nkc0 = 0;
    int epar = 0;
    auto set_dma = [&](bool next_item, int b, int n0, int c0, int par, bool on, int epar_or_neg) __attribute__((always_inline)) {
        if constexpr (PROD == 1) {
            pr_on = on;
            pr_c0 = c0;
            pr_buf = lds + par * SM::BUF_BYTES;
            pr_y0 = next_item ? ny0 : y0;
            pr_x0 = next_item ? nx0 : x0;
            pr_par = next_item ? (epar ^ 1) : epar;  // the patch buffer of the item the chunk belongs to
        }
        d_next = next_item;
        d_soffA = (unsigned)b * slice_bytes + (unsigned)c0 * 4u;
        d_soffW = ((unsigned)n0 * (unsigned)p.Cin + (unsigned)c0) * 4u;
        d_buf = lds + par * SM::BUF_BYTES;
        on = on && !LM_ABL_DMA(c0);
        d_nA = (on && PROD != 1) ? SM::A_PIECES : 0;
        d_nW = on ? SM::W_PIECES : 0;
        d_epi = on ? epar_or_neg : -1;
        d_n0 = n0;
    };
    // slot k of the stage: k = 2 j -> activation piece wave + 8 j, k = 2 j + 1 -> weight piece wave + 8 j, the last slot
    // -> the item's epilogue constants (64 floats per array = one 4-byte DMA per wave).  Every condition is wave-uniform.
    constexpr int N_SLOTS = 12;
    // this lane's element of the constant array its wave stages (a per-lane pointer: no kernel-argument reload in the tap loop)
    const float* const epi_src = (wave == 0 ? p.bias : (wave == 1 ? p.bn_s : p.bn_t)) + lane;
    static_assert(2 * SM::A_PER_WAVE <= N_SLOTS && 2 * SM::W_PER_WAVE + 1 <= N_SLOTS, "DMA slots");
    auto dma_slot = [&](int k) __attribute__((always_inline)) {
        const int j = k >> 1;
        if (k == N_SLOTS - 1) {
            if (d_epi >= 0 && wave < (bn ? 3 : 1)) lm_dma4_global(epi_src + d_n0, &epi[d_epi][wave][0]);
        } else if ((k & 1) == 0) {
            if (j < SM::A_PER_WAVE && wave + NW * j < d_nA) {
                const int jj = j < SM::A_PER_WAVE ? j : 0;
                const unsigned vc = voffC[jj], vn = voffN[jj];  // (both read as VALUES: a select between the arrays themselves sends them to scratch)
                lm_dma16(rsrcA, d_next ? vn : vc, d_soffA, d_buf + (wave + NW * j) * 1024);
            }
        } else {
            if (j < SM::W_PER_WAVE && wave + NW * j < d_nW)
                lm_dma16(rsrcW, voffW, d_soffW + (unsigned)j * w_piece_stride, d_buf + SM::A_BYTES + (wave + NW * j) * 1024);
        }
    };

    // Producer slot S of the stage (S = 0: beside the last tap of the previous chunk, 1 + t: beside tap t), micro-step m = 0..5 in
    // front of the m-th matrix-instruction pair.  Half h of round r runs in slot PROD_S0 + 2 r + h; everything a slot writes is
    // published by the chunk barrier behind tap 7.  The statement order is pinned (LM_SCHED_FENCE): reads one micro-step -- two
    // matrix instructions of this wave, two of its SIMD partner -- ahead of their use.
#ifndef LM_PROD_S0
#define LM_PROD_S0 1
#endif
    constexpr int PROD_S0 = LM_PROD_S0;
    // (slot 0 does not exist for the stage that fills chunk 1 of an item -- it starts behind the item switch, not behind a tap 8 --
    // and slot 8 lies behind the barrier that publishes the image)
    static_assert(PROD_S0 >= 1 && PROD_S0 + 5 <= 7, "producer slots must lie beside taps 0..6 of the running chunk");
    auto prod_step = [&](int S, int m) __attribute__((always_inline)) {
        if constexpr (PROD == 1) {
            const int hs = S - PROD_S0;
            if (hs < 0 || hs >= 6) return;
            const int r = hs >> 1, half = hs & 1, pi = r == 2 ? 1 : 0;
            if (!pr_on) return;
            LM_SCHED_FENCE();
#if LM_PROD_DPP
            if (r < 2 || wave < 4) {  // (whole waves: a row broadcast needs its source lanes active; idle lanes compute and do not store)
                const int grp = r == 2 ? (tid >= 112 ? 1 : 0) : r;
                const int py = pr_pyx[pi] & 0xff, px = pr_pyx[pi] >> 8;
                float& a0 = pr_s[0];
                float& a1 = pr_s[1];
                float& a2 = pr_s[2];
                float& a3 = pr_s[3];
                lm_f32x4& W = pr_w[0];
#define PROD_TAP(K, X)                          \
    do {                                        \
        lm_fmac_rowbcast<K>(a0, W[0], (X));     \
        lm_fmac_rowbcast<K>(a1, W[1], (X));     \
        lm_fmac_rowbcast<K>(a2, W[2], (X));     \
        lm_fmac_rowbcast<K>(a3, W[3], (X));     \
    } while (0)
                if (m == 0) {
                    if (half == 0 && r != 1) {  // the pixel's 3 x 3 input neighbourhood (the patch starts one more pixel up and left)
                        const float* xp = fcx + pr_par * FC_PATCH + py * FC_PW + px;
                        pr_x[0] = lm_f32x2{xp[0], xp[1]};
                        pr_x[1] = lm_f32x2{xp[2], xp[FC_PW]};
                        pr_x[2] = lm_f32x2{xp[FC_PW + 1], xp[FC_PW + 2]};
                        pr_x[3] = lm_f32x2{xp[2 * FC_PW], xp[2 * FC_PW + 1]};
                        pr_x[4] = lm_f32x2{xp[2 * FC_PW + 2], 0.f};
                    }
                    // all 40 constants of this half-task's four channels: lane l fetches row min(l & 15, 9)
                    W = *reinterpret_cast<const lm_f32x4*>(fcc + min(lane & 15, 9) * FC_ROW + pr_c0 + 8 * grp + 4 * half);
                } else if (m == 1) {
                    // first_conv_h3_kernel's chain: bias, then taps 0..8
                    a0 = lm_mov_rowbcast<9>(W[0]);
                    a1 = lm_mov_rowbcast<9>(W[1]);
                    a2 = lm_mov_rowbcast<9>(W[2]);
                    a3 = lm_mov_rowbcast<9>(W[3]);
                    PROD_TAP(0, pr_x[0][0]);
                    PROD_TAP(1, pr_x[0][1]);
                } else if (m == 2) {
                    PROD_TAP(2, pr_x[1][0]);
                    PROD_TAP(3, pr_x[1][1]);
                } else if (m == 3) {
                    PROD_TAP(4, pr_x[2][0]);
                    PROD_TAP(5, pr_x[2][1]);
                } else if (m == 4) {
                    PROD_TAP(6, pr_x[3][0]);
                    PROD_TAP(7, pr_x[3][1]);
                } else {
                    PROD_TAP(8, pr_x[4][0]);
                    const float v0 = fmaxf(a0, 0.f), v1 = fmaxf(a1, 0.f), v2 = fmaxf(a2, 0.f), v3 = fmaxf(a3, 0.f);
                    // zero padding of THIS conv: halo pixels outside the image are 0 (one unsigned compare per axis)
                    const bool inside = (unsigned)(pr_y0 - 1 + py) < (unsigned)p.H && (unsigned)(pr_x0 - 1 + px) < (unsigned)p.W;
                    uint2 ph, plo;
                    lm_split4(v0, v1, v2, v3, &ph, &plo);
                    if (!inside) {
                        ph.x = ph.y = 0u;
                        plo.x = plo.y = 0u;
                    }
                    if (r < 2 || (tid < 224 && (tid % 112) < 100)) {
                        pr_max = lm_pk_absmax_u16(lm_pk_absmax_u16(pr_max, ph.x), ph.y);
                        const int off = (pr_woff[pi] ^ (grp << 5)) + half * 8;  // group g: logical slots 2 g (hi), 2 g + 1 (lo)
                        *reinterpret_cast<uint2_a*>(pr_buf + off) = ph;
                        *reinterpret_cast<uint2_a*>(pr_buf + (off ^ 16)) = plo;
                    }
                }
#undef PROD_TAP
            }
#else
            if (r < 2 || tid < 200) {
                const int grp = r == 2 ? (tid >= 100 ? 1 : 0) : r;
                const int py = pr_pyx[pi] & 0xff, px = pr_pyx[pi] >> 8;
                const float* cw = fcc + pr_c0 + 8 * grp + 4 * half;  // this half-task's 4 channels: w[k] at + 64 k, bias + 576, scale + 640
#if defined(LM_FC_ABL) && LM_FC_ABL == 1  // lab ablation: no LDS reads of the weights (timing only, results are garbage)
                auto ldw = [&](int i) { return lm_f32x4{0.5f + i, 0.25f, -0.5f, 0.125f * i}; };
#else
                auto ldw = [&](int i) { return *reinterpret_cast<const lm_f32x4*>(cw + i * 64); };
#endif
                auto fma2 = [&](int k, const lm_f32x4& w) __attribute__((always_inline)) {  // tap k on the four channels
#if defined(LM_FC_ABL) && LM_FC_ABL == 2  // lab ablation: no multiply-adds (timing only)
                    if (k > 0) return;
#endif
#ifndef LM_PROD_SCALAR_FMA
#define LM_PROD_SCALAR_FMA 1
#endif
#if LM_PROD_SCALAR_FMA  // four v_fma_f32 instead of two v_pk_fma_f32 (same bits; 0 = the packed form of round 4, the A/B arm of profiles/r05a_*)
                    {
                        const float xk = pr_x[k >> 1][k & 1];
                        float a0 = pr_a[0][0], a1 = pr_a[0][1], a2 = pr_a[1][0], a3 = pr_a[1][1];
                        lm_fma_f32_single(a0, xk, w[0]);
                        lm_fma_f32_single(a1, xk, w[1]);
                        lm_fma_f32_single(a2, xk, w[2]);
                        lm_fma_f32_single(a3, xk, w[3]);
                        pr_a[0] = lm_f32x2{a0, a1};
                        pr_a[1] = lm_f32x2{a2, a3};
                        return;
                    }
#endif
                    const lm_f32x2 w01 = {w[0], w[1]}, w23 = {w[2], w[3]};
                    if (k & 1) {
                        lm_pk_fma_bcast<1>(pr_a[0], pr_x[k >> 1], w01);
                        lm_pk_fma_bcast<1>(pr_a[1], pr_x[k >> 1], w23);
                    } else {
                        lm_pk_fma_bcast<0>(pr_a[0], pr_x[k >> 1], w01);
                        lm_pk_fma_bcast<0>(pr_a[1], pr_x[k >> 1], w23);
                    }
                };
                if (m == 0) {
                    if (half == 0 && r != 1) {  // the pixel's 3 x 3 input neighbourhood (the patch starts one more pixel up and left)
                        const float* xp = fcx + pr_par * FC_PATCH + py * FC_PW + px;
                        pr_x[0] = lm_f32x2{xp[0], xp[1]};
                        pr_x[1] = lm_f32x2{xp[2], xp[FC_PW]};
                        pr_x[2] = lm_f32x2{xp[FC_PW + 1], xp[FC_PW + 2]};
                        pr_x[3] = lm_f32x2{xp[2 * FC_PW], xp[2 * FC_PW + 1]};
                        pr_x[4] = lm_f32x2{xp[2 * FC_PW + 2], 0.f};
                    }
                    pr_w[0] = ldw(9);  // bias
                    pr_w[1] = ldw(0);
                    pr_w[2] = ldw(1);
                } else if (m == 1) {
                    pr_w[3] = ldw(2);
                    pr_w[4] = ldw(3);
                    // first_conv_h3_kernel's chain: bias, then taps 0..8
                    pr_a[0] = lm_f32x2{pr_w[0][0], pr_w[0][1]};
                    pr_a[1] = lm_f32x2{pr_w[0][2], pr_w[0][3]};
                    fma2(0, pr_w[1]);
                    fma2(1, pr_w[2]);
                } else if (m == 2) {
                    pr_w[0] = ldw(4);
                    pr_w[1] = ldw(5);
                    fma2(2, pr_w[3]);
                    fma2(3, pr_w[4]);
                } else if (m == 3) {
                    pr_w[2] = ldw(6);
                    pr_w[3] = ldw(7);
                    fma2(4, pr_w[0]);
                    fma2(5, pr_w[1]);
                } else if (m == 4) {
                    pr_w[0] = ldw(8);
#if !LM_H3_FOLD_SCALE
                    pr_w[1] = ldw(10);  // BatchNorm scale
#endif
                    fma2(6, pr_w[2]);
                    fma2(7, pr_w[3]);
                } else {
                    fma2(8, pr_w[0]);
                    // ReLU, BatchNorm scale; the shift is deferred to this conv's bias and border table (fmaf(., s, 0): what the
                    // stand-alone kernel evaluates with its zero shift array)
#if LM_H3_FOLD_SCALE  // (weights and bias carry the layer's 2^E, the scale travels with this conv's weights: fmaf(., 1, 0) of the stand-alone kernel)
                    const float v0 = fmaxf(pr_a[0][0], 0.f), v1 = fmaxf(pr_a[0][1], 0.f), v2 = fmaxf(pr_a[1][0], 0.f), v3 = fmaxf(pr_a[1][1], 0.f);
#else
                    const float v0 = fmaf(fmaxf(pr_a[0][0], 0.f), pr_w[1][0], 0.f), v1 = fmaf(fmaxf(pr_a[0][1], 0.f), pr_w[1][1], 0.f);
                    const float v2 = fmaf(fmaxf(pr_a[1][0], 0.f), pr_w[1][2], 0.f), v3 = fmaf(fmaxf(pr_a[1][1], 0.f), pr_w[1][3], 0.f);
#endif
                    // zero padding of THIS conv: halo pixels outside the image are 0 (one unsigned compare per axis)
                    const bool inside = (unsigned)(pr_y0 - 1 + py) < (unsigned)p.H && (unsigned)(pr_x0 - 1 + px) < (unsigned)p.W;
                    uint2 ph, plo;
                    lm_split4(v0, v1, v2, v3, &ph, &plo);
                    if (!inside) {
                        ph.x = ph.y = 0u;
                        plo.x = plo.y = 0u;
                    }
                    pr_max = lm_pk_absmax_u16(lm_pk_absmax_u16(pr_max, ph.x), ph.y);
                    const int off = (pr_woff[pi] ^ (grp << 5)) + half * 8;  // group g: logical slots 2 g (hi), 2 g + 1 (lo)
                    *reinterpret_cast<uint2_a*>(pr_buf + off) = ph;
                    *reinterpret_cast<uint2_a*>(pr_buf + (off ^ 16)) = plo;
                }
            }
#endif
            LM_SCHED_FENCE();
        }
    };
    // the 20 x 36 input patch of item (b, y0, x0) -> fcx[par]: 12 DMA pieces of 64 floats (the last one 16), 1-2 per wave
    auto fc_patch_dma = [&](int b, int y0, int x0, int par) __attribute__((always_inline)) {
        if constexpr (PROD == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int piece = wave + NW * j, idx = piece * 64 + lane;
                if (piece < (FC_PATCH + 63) / 64 && idx < FC_PATCH) {  // (the partial piece runs with the other lanes masked off)
                    const int py = idx / FC_PW, px = idx - py * FC_PW;
                    const int gy = y0 - 2 + py, gx = x0 - 2 + px;
                    const bool inb = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                    lm_dma4(rsrcA, inb ? (unsigned)(((b * p.H + gy) * p.W + gx) * 4) : LM_DMA_OOB, 0u, fcx + par * FC_PATCH + piece * 64);
                }
            }
        }
    };

    lm_f32x16 accm[2][2];  // [M-tile][N-tile = row]: all three products of the split scheme accumulate here
    while (it < n_items && !decode(it, b, y0, x0, n0, kc0)) it += gridDim.x;
    if (it >= n_items) return;
    if (!bn && tid < 2 * TN) {  // no BatchNorm (decoder 1x1): identity constants, never overwritten
        epi[0][1 + tid / TN][tid % TN] = tid < TN ? 1.f : 0.f;
        epi[1][1 + tid / TN][tid % TN] = tid < TN ? 1.f : 0.f;
    }
    static_assert(HEAD == 0 || (TAPS == 9 && !G16), "the fused head belongs to the 32-wide 3x3 form");
    if constexpr (HEAD != 0) {
        for (int i = tid; i < p.head_C * 64; i += 512) hw[i] = p.head_w[i];
        if (tid < p.head_C) hw[kMaxClasses * 64 + tid] = p.head_b[tid];
    }
    item_voffs(b, y0, x0, voffC);
    char* const buf0 = lds;
    char* const buf1 = lds + SUB * SM::BUF_BYTES;
    // prologue: stage 0 of the first item, all pieces at once (set_dma's buffer argument counts chunk images)
    set_dma(false, b, n0, kc0, 0, true, epar);
#pragma unroll
    for (int k = 0; k < N_SLOTS; ++k) dma_slot(k);
    if constexpr (PROD == 1) {  // the first item's chunk 0 is computed here, with nothing to run beside
#if LM_PROD_DPP
        for (int i = tid; i < 640; i += 512) fcc[(i >> 6) * FC_ROW + (i & 63)] = p.fc_c[i];
#else
        for (int i = tid; i < FC_CONST; i += 512) fcc[i] = p.fc_c[i];
#endif
        fc_patch_dma(b, y0, x0, 0);
        lm_barrier_dma();
#pragma unroll
        for (int S = PROD_S0; S < PROD_S0 + 6; ++S)
#pragma unroll
            for (int m = 0; m < 6; ++m) prod_step(S, m);
    }
    if (SUB == 2) {
        set_dma(false, b, n0, kc0 + KC, 1, true, -1);
#pragma unroll
        for (int k = 0; k < N_SLOTS; ++k) dma_slot(k);
    }
    lm_barrier_dma();
    LM_TRACE_INIT();
    while (true) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    accm[i][j][r] = 0.f;
                }
        int nit = it + gridDim.x;
        while (nit < n_items && !decode(nit, nb, ny0, nx0, nn0, nkc0)) nit += gridDim.x;
        const bool have_next = nit < n_items;
        item_voffs(nb, ny0, nx0, voffN);
        // (fused first layer) the next item's input patch: its buffer was last read while the previous item's chunk 2 ran, the
        // barrier of this item's chunk 0 publishes it, and it is first read beside this item's last chunk
        if (have_next) fc_patch_dma(nb, ny0, nx0, epar ^ 1);
        // chunk 0 of this item is resident in buffer 0 (and visible: a barrier lies behind its DMA wait)
        lm_h16x8 f[8], g[8];  // two fragment sets: whi0 whi1 wlo0 wlo1 a0hi a0lo a1hi a1lo
        bool abl_r = false;   // lab ablation (constant false in the product)
        if constexpr (TAPS == 9) {
            const int kb0 = KS ? kc0 : 0, nkb0 = KS ? nkc0 : 0;  // first input channel of this / the next item's K range (0 without the split)
            set_dma(false, b, n0, kb0 + KC, 1, true, -1);  // chunk 1 -> buffer 1, issued from the slots of chunk 0
            H3P_READS(f, buf0, 0, 0);
            for (int ci = 0; ci < nchunks; ci += 2) {
                abl_r = LM_ABL_READS(ci);
                // ---- even chunk ci (buffer 0), fragments start in f
                H3P_CHUNK_STEPS(f, g, buf0);
                H3P_WAITF(0, f);
                LM_TRACE_MARK(0);
                lm_barrier_dma();  // everyone has read buffer 0 for the last time; chunk ci + 1 is complete in buffer 1
                LM_TRACE_MARK(1);
                H3P_READS(g, buf1, 0, 0);
                if (ci + 2 < nchunks) set_dma(false, b, n0, kb0 + (ci + 2) * KC, 0, true, -1);
                else set_dma(true, nb, nn0, nkb0, 0, have_next, epar ^ 1);  // the next item's chunk 0 + epilogue constants
                H3P_MFMAS(f, 0);
                // ---- odd chunk ci + 1 (buffer 1), fragments start in g
                H3P_CHUNK_STEPS(g, f, buf1);
                H3P_WAITF(0, g);
                LM_TRACE_MARK(0);
                lm_barrier_dma();
                LM_TRACE_MARK(1);
                if (ci + 2 < nchunks) {
                    H3P_READS(f, buf0, 0, 0);
                    set_dma(false, b, n0, kb0 + (ci + 3) * KC, 1, true, -1);
                } else {
                    set_dma(false, b, n0, 0, 1, false, -1);  // buffer 1 becomes the epilogue's staging area
                }
                H3P_MFMAS(g, 0);
            }
        } else {
            // 1x1: stage si = chunks 2 si, 2 si + 1 (two chunk images side by side), resident in stage buffer si & 1; the number of
            // stages is even (checked by the launcher), so stage 0 of every item lives in buffer 0
            const int nstages = nchunks / 2;
            for (int si = 0; si < nstages; ++si) {
                const char* as = (si & 1) ? buf1 : buf0;
                const int nxt = ((si + 1) & 1) * 2;  // first chunk image of the other stage buffer
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {  // stage the next two chunks (or the next item's first two) into the other buffer
                    if (si + 1 < nstages) set_dma(false, b, n0, kc0 + (2 * (si + 1) + sub) * KC, nxt + sub, true, -1);
                    else set_dma(true, nb, nn0, nkc0 + sub * KC, sub, have_next, sub == 0 ? (epar ^ 1) : -1);
#pragma unroll
                    for (int k = 0; k < N_SLOTS; ++k) dma_slot(k);
                }
                H3P_READS(f, as, 0, 0);
                H3P_READS(g, as + SM::BUF_BYTES, 0, 0);
                H3P_WAITF(8, f);
                H3P_MFMAS(f, -1);
                H3P_WAITF(0, g);
                H3P_MFMAS(g, -1);
                lm_barrier_dma();
            }
        }
        LM_TRACE_MARK(2);
        // ---- epilogue of this item (the first chunk of the next item is already resident in buffer 0)
        {
            // All waves are done with the buffer of the last chunk (buffer 1): it becomes the staging area that turns the
            // accumulator layout (lane = pixel, 8 bytes per 4 couts: 64 scattered lines per store) into
            // 128-byte pixel lines written 16 bytes per lane (the scattered form cost ~10 us per tile in the TA).
            char* const hstage = (STAGE_EXTRA ? lds + 2 * SUB * SM::BUF_BYTES : buf1) + wave * (64 * HSTR);
            unsigned gmax = 0u;  // running max of the |hi| halves this lane writes (f16 range guard, lm_pk_absmax_u16)
            const int bs = b + wsl;                      // slice this wave writes
            const int yb = y0 + (G16 ? 4 * (wave & 3) : 2 * wave);  // first image row of the wave's N-tile 0
            const int Hp = p.H >> 1, Wp = p.W >> 1;
            const char* ep = reinterpret_cast<const char*>(&epi[epar][0][0]);
            bool part_done = false;
            if constexpr (KSPLIT) {
                if (KS || ksplit > 1) {
                    // ---- split-K: this item's raw accumulators -> kpart[ks][slice][pixel][cout] (fp32, dense), 16 bytes per lane and
                    // 4-channel quad; scale, bias and the split happen in the reduction
                    const int ks = kc0 / (p.Cin / ksplit);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int yl = G16 ? yb + 2 * nt + (li >> 4) : yb + nt, xl = G16 ? wcol : x0 + li;
                        const bool ok = bs < p.B && yl < p.H;
                        float* dst = p.kpart + ((((size_t)ks * p.B + bs) * p.H + yl) * p.W + xl) * p.Cout + n0 + 4 * kb;
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4)
                                if (ok)
                                    *reinterpret_cast<float4*>(dst + 32 * mt + 8 * g4) =
                                        make_float4(accm[mt][nt][4 * g4], accm[mt][nt][4 * g4 + 1], accm[mt][nt][4 * g4 + 2], accm[mt][nt][4 * g4 + 3]);
                    }
                    part_done = true;
                }
            }
            if constexpr (KS) {  // (nothing but the partial sums leaves this instantiation)
            } else if (part_done) {
            } else if constexpr (HEAD != 0) {
                // ---- fused head: this item holds ALL 64 channels of its pixels (n0 == 0).  Per pixel the two lanes kb = 0/1
                // own channels 8q + 4kb + k (q = mg, k = 0..3).  The head is evaluated on the fp32 values themselves -- the last
                // conv's output is never rounded to a 22-bit hi/lo pair on this path (round 3 did so to stay bit-identical with
                // launch_head_h3 on the stored tensor; that kernel, which only serves widths the persistent kernel does not, is now
                // the one that may differ on near-tie pixels) -- in launch_head_h3's summation order: each 8-channel block one fma
                // chain (k = 0..7 from 0), continued across the lane pair with one shuffle, the blocks added pairwise (q ^ 4,
                // q ^ 2, q ^ 1) in registers.  HEAD == 2 also writes the log-softmax of the logits (resunet.py:70): labels and
                // log-probabilities of a forward come from the same numbers.
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int yl = yb + nt;
                    const bool tile_ok = bs < p.B && yl < p.H;
                    const int bmask = (yl == 0 ? 1 : 0) | (yl == p.H - 1 ? 2 : 0) | (x0 + li == 0 ? 4 : 0) | (x0 + li == p.W - 1 ? 8 : 0);
                    const bool border = p.border_corr != nullptr && __any(bmask != 0);  // wave-uniform
                    float vv[8][4];
                    // deferred-shift input: the taps outside the image saw 0, not -T (ConvParamsH3::border_corr).  All eight
                    // channel groups' corrections are fetched in ONE round trip up front (the fragment registers are free
                    // here); fetched one group at a time inside the loop they were eight serial cache latencies per row for
                    // every wave of every item that touches the image border -- all items from the 64 x 64 level down.
                    float4 cb[8];
                    if (border) {
#pragma unroll
                        for (int mg = 0; mg < 8; ++mg) cb[mg] = *reinterpret_cast<const float4*>(p.border_corr + (size_t)bmask * p.Cout + n0 + H3P_EPI_CL(mg));
                    }
                    lm_h16x8 ec[2][3];  // the constants of channel group mg + 1 are fetched under the arithmetic of group mg
                    H3P_EPI_READS(ec[0], H3P_EPI_CL(0));
#pragma unroll
                    for (int mg = 0; mg < 8; ++mg) {
                        const int mt = mg >> 2, g4 = mg & 3;
                        const int cl = H3P_EPI_CL(mg);
                        if (mg + 1 < 8) {
                            H3P_EPI_READS(ec[(mg + 1) & 1], H3P_EPI_CL(mg + 1));
                            H3P_EPI_WAIT(true, ec[mg & 1]);
                        } else {
                            H3P_EPI_WAIT(false, ec[mg & 1]);
                        }
                        const float4 bias = as_float4(ec[mg & 1][0]);
                        float bb[4] = {bias.x, bias.y, bias.z, bias.w};
#if !LM_H3_FOLD_SCALE
                        const float4 s = as_float4(ec[mg & 1][1]), sh = as_float4(ec[mg & 1][2]);
                        const float ss[4] = {s.x, s.y, s.z, s.w}, tt[4] = {sh.x, sh.y, sh.z, sh.w};
#endif
                        if (border) {
                            const float4 c = cb[mg];
                            bb[0] -= c.x;
                            bb[1] -= c.y;
                            bb[2] -= c.z;
                            bb[3] -= c.w;
                        }
                        float v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float t = fmaf(accm[mt][nt][4 * g4 + k], p.acc_scale, bb[k]);
#if LM_H3_FOLD_SCALE
                            if (bn) t = fmaxf(t, 0.f);
#else
                            if (bn) t = fmaf(fmaxf(t, 0.f), ss[k], tt[k]);
#endif
                            v[k] = t;
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) vv[mg][k] = v[k];
                    }
                    float best = 0.f;
                    int arg = 0;
                    float lgs[HEAD == 2 ? kMaxClasses : 1];
                    auto logit = [&](int c) __attribute__((always_inline)) -> float {
                        float sq[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float4 w4 = *reinterpret_cast<const float4*>(&hw[c * 64 + 8 * q + 4 * kb]);
                            // chain from 0: right for the kb = 0 lanes ...
                            float s0 = fmaf(vv[q][0], w4.x, 0.f);
                            s0 = fmaf(vv[q][1], w4.y, s0);
                            s0 = fmaf(vv[q][2], w4.z, s0);
                            s0 = fmaf(vv[q][3], w4.w, s0);
                            // ... continued by the kb = 1 lane of the pixel from its partner's partial sum
                            float s1 = __shfl_xor(s0, 32);
                            s1 = fmaf(vv[q][0], w4.x, s1);
                            s1 = fmaf(vv[q][1], w4.y, s1);
                            s1 = fmaf(vv[q][2], w4.z, s1);
                            s1 = fmaf(vv[q][3], w4.w, s1);
                            sq[q] = s1;
                        }
                        return (((sq[0] + sq[4]) + (sq[2] + sq[6])) + ((sq[1] + sq[5]) + (sq[3] + sq[7]))) + hw[kMaxClasses * 64 + c];
                    };
                    if constexpr (HEAD == 2) {  // (unrolled: the logits stay in registers for the log-softmax)
#pragma unroll
                        for (int c = 0; c < kMaxClasses; ++c) {
                            if (c < p.head_C) {  // wave-uniform
                                const float lg = logit(c);
                                lgs[c] = lg;
                                if (c == 0 || lg > best) {
                                    best = lg;
                                    arg = c;
                                }
                            }
                        }
                    } else {
#pragma unroll 1
                        for (int c = 0; c < p.head_C; ++c) {
                            const float lg = logit(c);
                            if (c == 0 || lg > best) {
                                best = lg;
                                arg = c;
                            }
                        }
                    }
                    if (tile_ok && kb == 1) p.head_labels[((size_t)bs * p.H + yl) * p.W + x0 + li] = (uint8_t)arg;
                    if constexpr (HEAD == 2) {  // log_softmax over the classes, launch_head_h3's formula: lg - (best + log(sum exp(lg - best)))
                        float se = 0.f;
#pragma unroll
                        for (int c = 0; c < kMaxClasses; ++c)
                            if (c < p.head_C) se += expf(lgs[c] - best);
                        const float lse = best + logf(se);
                        if (tile_ok && kb == 1) {
                            const size_t HW = (size_t)p.H * p.W, yx = (size_t)yl * p.W + x0 + li;
#pragma unroll
                            for (int c = 0; c < kMaxClasses; ++c)
                                if (c < p.head_C) p.head_logp[((size_t)bs * p.head_C + c) * HW + yx] = lgs[c] - lse;
                        }
                    }
                }
            } else {
            // Stored output, one pass per M-tile (32 of the item's 64 channels): per 4-channel group the epilogue constants are
            // read ONCE and both rows of the wave go through the arithmetic as two independent dependency chains (the chain
            // bias -> ReLU -> scale -> split -> stage is ~8 dependent VALU results long and the hand-placed waits keep the
            // compiler from overlapping consecutive groups: row after row it ran at ~370 cycles per group); the pooled value
            // of a group is complete as soon as both rows are, so no per-row accumulator array lives across the passes.
            // Staging: 64 pixels x (128 B + 16 B pad) per wave and pass.
#ifndef LM_EPI_DIRECT
#define LM_EPI_DIRECT 0
#endif
            // LM_EPI_DIRECT: no LDS staging -- the lane pair of a pixel exchanges halves (lm_permlane32_swap) and each lane stores
            // 16 bytes (8 hi halves or 8 lo halves of one channel group) straight from registers: 32 bytes per pixel and store
            // instruction instead of whole 128-byte lines, but no ds_write / ds_read / fence per group.
            constexpr bool EPI_DIRECT = LM_EPI_DIRECT != 0;
            int yl2[2], bmask2[2];
            bool ok2[2], border2[2];
            char* obase[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                // image row of this lane's pixel in N-tile nt, and whether the wave's 32 pixels of this N-tile exist
                yl2[nt] = G16 ? yb + 2 * nt + (li >> 4) : yb + nt;
                ok2[nt] = bs < p.B && (G16 ? yb + 2 * nt + 1 < p.H : yb + nt < p.H);
                const int xl = G16 ? wcol : x0 + li;
                bmask2[nt] = (yl2[nt] == 0 ? 1 : 0) | (yl2[nt] == p.H - 1 ? 2 : 0) | (xl == 0 ? 4 : 0) | (xl == p.W - 1 ? 8 : 0);
                border2[nt] = TAPS == 9 && p.border_corr != nullptr && __any(bmask2[nt] != 0);  // wave-uniform
                // G32: 32 consecutive pixels of one image row; G16 (W == 16): two consecutive 16-pixel rows = 32 consecutive pixels
                const int y_first = G16 ? yb + 2 * nt : yb + nt;
                obase[nt] = p.out + ((((size_t)bs * p.H + y_first) * p.W + x0) * p.out_cstride + p.out_coff + n0) * 4;
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                // border corrections (deferred-shift input: the taps outside the image saw 0, not -T, ConvParamsH3::border_corr):
                // the loads of channel group g4 + 1 are in flight under the arithmetic of group g4 (fetched where they were
                // used they were a serial cache latency per group for every wave of every item that touches the image border
                // -- all items from the 64 x 64 level down)
                float4 cb[2][2];
                auto load_cb = [&](int g4) __attribute__((always_inline)) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        if (border2[nt])
                            cb[g4 & 1][nt] = *reinterpret_cast<const float4*>(p.border_corr + (size_t)bmask2[nt] * p.Cout + n0 + H3P_EPI_CL(4 * mt + g4));
                };
                load_cb(0);
                float qs[4][4];     // pooled groups of this pass (32-wide geometry)
                lm_h16x8 ec[2][3];  // the constants of channel group g4 + 1 are fetched under the arithmetic of group g4
                H3P_EPI_READS(ec[0], H3P_EPI_CL(4 * mt));
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int mg = 4 * mt + g4;
                    const int cl = H3P_EPI_CL(mg);  // first of 4 consecutive local output channels
                    if (g4 + 1 < 4) {
                        H3P_EPI_READS(ec[(g4 + 1) & 1], H3P_EPI_CL(mg + 1));
                        H3P_EPI_WAIT(true, ec[g4 & 1]);
                    } else {
                        H3P_EPI_WAIT(false, ec[g4 & 1]);
                    }
                    const float4 bias = as_float4(ec[g4 & 1][0]);
#if !LM_H3_FOLD_SCALE
                    const float4 s = as_float4(ec[g4 & 1][1]), sh = as_float4(ec[g4 & 1][2]);
                    const float ss[4] = {s.x, s.y, s.z, s.w}, tt[4] = {sh.x, sh.y, sh.z, sh.w};
#endif
                    if (g4 + 1 < 4) load_cb(g4 + 1);
                    float v[2][4];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        float bb[4] = {bias.x, bias.y, bias.z, bias.w};
                        if (border2[nt]) {
                            const float4 c = cb[g4 & 1][nt];
                            bb[0] -= c.x;
                            bb[1] -= c.y;
                            bb[2] -= c.z;
                            bb[3] -= c.w;
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float t = fmaf(accm[mt][nt][4 * g4 + k], p.acc_scale, bb[k]);
#if LM_H3_FOLD_SCALE
                            if (bn) t = fmaxf(t, 0.f);
#else
                            if (bn) t = fmaf(fmaxf(t, 0.f), ss[k], tt[k]);
#endif
                            v[nt][k] = t;
                        }
                        uint2 ph, plo;
                        lm_split4(v[nt][0], v[nt][1], v[nt][2], v[nt][3], &ph, &plo);
                        gmax = lm_pk_absmax_u16(lm_pk_absmax_u16(gmax, ph.x), ph.y);
                        if constexpr (EPI_DIRECT) {
                            lm_permlane32_swap(ph.x, plo.x);
                            lm_permlane32_swap(ph.y, plo.y);
                            if (ok2[nt]) {
                                const uint4 val = {ph.x, ph.y, plo.x, plo.y};  // kb = 0: the group's 8 hi halves, kb = 1: its 8 lo halves
                                char* dst = obase[nt] + (size_t)li * p.out_cstride * 4 + mt * 128 + g4 * 32 + kb * 16;
                                if (p.stream_out) lm_store16_stream(dst, val);
                                else *reinterpret_cast<uint4_a*>(dst) = val;
                            }
                        } else {
                            char* d = hstage + (nt * 32 + li) * HSTR + ((cl & 31) >> 3) * 32 + (cl & 7) * 2;
                            *reinterpret_cast<uint2_a*>(d) = ph;
                            *reinterpret_cast<uint2_a*>(d + 16) = plo;
                        }
                        if (G16 && p.pool != nullptr) {  // both pool partners are in this N-tile: lane^16 (y+1) and lane^1 (x+1)
                            float q[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float h = v[nt][k] + __shfl_xor(v[nt][k], 16);
                                q[k] = 0.25f * (h + __shfl_xor(h, 1));
                            }
                            if (ok2[nt] && (li & 17) == 0) {
                                const int cg = n0 + cl;
                                char* prow = p.pool + ((((size_t)bs * Hp + (yl2[nt] >> 1)) * Wp + (wcol >> 1)) * p.pool_cstride + p.pool_coff) * 4;
                                split_store4(prow + (size_t)(cg >> 3) * 32, (cg & 7) * 2, q[0], q[1], q[2], q[3]);
                            }
                        }
                    }
                    if (!G16 && p.pool != nullptr) {  // avg_pool2d(2): rows yb, yb+1 summed first, then x+1 = lane^1
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float pl = v[0][k] + v[1][k];
                            qs[g4][k] = 0.25f * (pl + lm_lane_xor1(pl));
                        }
                    }
                }
                LM_TRACE_SUB(5);
                // the wave's own 64 pixels x 128 B are now in LDS (same-wave LDS ops are ordered): stream them out, a 128-byte
                // line per pixel and pass, 16 bytes per lane
                if constexpr (!EPI_DIRECT) {
                lm_wave_lds_fence();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = i * 64 + lane, px = q >> 3, part = q & 7;
                    if (ok2[i >> 2]) {
                        const uint4 val = *reinterpret_cast<const uint4_a*>(hstage + px * HSTR + part * 16);
                        char* dst = obase[i >> 2] + (size_t)(px & 31) * p.out_cstride * 4 + mt * 128 + part * 16;
                        if (p.stream_out) lm_store16_stream(dst, val);
                        else *reinterpret_cast<uint4_a*>(dst) = val;
                    }
                }
                lm_wave_lds_fence();  // the staging rows are rewritten by the pooled values / the next pass
                }
                LM_TRACE_SUB(6);
                if (EPI_DIRECT && !G16 && p.pool != nullptr) {  // pooled row of this pass: the even lanes' pixel pairs, 16 bytes per lane
                    char* prow = p.pool + ((((size_t)bs * Hp + (yb >> 1)) * Wp + (x0 >> 1)) * p.pool_cstride + p.pool_coff + n0) * 4;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        uint2 ph, plo;
                        lm_split4(qs[g4][0], qs[g4][1], qs[g4][2], qs[g4][3], &ph, &plo);
                        lm_permlane32_swap(ph.x, plo.x);
                        lm_permlane32_swap(ph.y, plo.y);
                        if ((li & 1) == 0 && bs < p.B && yb + 1 < p.H) {
                            const uint4 val = {ph.x, ph.y, plo.x, plo.y};
                            *reinterpret_cast<uint4_a*>(prow + (size_t)(li >> 1) * p.pool_cstride * 4 + mt * 128 + g4 * 32 + kb * 16) = val;
                        }
                    }
                }
                if (!EPI_DIRECT && !G16 && p.pool != nullptr) {
                    // The wave's pooled output of this pass is ONE row of 16 pixels x 32 channels.  It goes through the staging rows
                    // like the full-resolution output (the even lanes' 8-byte pieces straight to memory were 16 scattered stores
                    // at a 256-byte stride per wave -- ~4.7 k cycles per item in the in-kernel timeline, four layers of the network).
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int cl = H3P_EPI_CL(4 * mt + g4);
                        uint2 ph, plo;
                        lm_split4(qs[g4][0], qs[g4][1], qs[g4][2], qs[g4][3], &ph, &plo);
                        if ((li & 1) == 0) {
                            char* d = hstage + (li >> 1) * HSTR + ((cl & 31) >> 3) * 32 + (cl & 7) * 2;
                            *reinterpret_cast<uint2_a*>(d) = ph;
                            *reinterpret_cast<uint2_a*>(d + 16) = plo;
                        }
                    }
                    lm_wave_lds_fence();
                    if (bs < p.B && yb + 1 < p.H) {
                        char* prow = p.pool + ((((size_t)bs * Hp + (yb >> 1)) * Wp + (x0 >> 1)) * p.pool_cstride + p.pool_coff + n0) * 4;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int q = i * 64 + lane, px = q >> 3, part = q & 7;
                            const uint4 val = *reinterpret_cast<const uint4_a*>(hstage + px * HSTR + part * 16);
                            *reinterpret_cast<uint4_a*>(prow + (size_t)px * p.pool_cstride * 4 + mt * 128 + part * 16) = val;
                        }
                    }
                    lm_wave_lds_fence();
                }
            }
            }  // stored-output epilogue
            if (p.range_flag != nullptr && lm_pk_out_of_f16_guard(gmax)) atomicOr(p.range_flag, 1u);
        }
        LM_TRACE_MARK(3);
        if (!have_next) break;
        it = nit;
        b = nb;
        y0 = ny0;
        x0 = nx0;
        n0 = nn0;
        kc0 = nkc0;
#pragma unroll
        for (int j = 0; j < SM::A_PER_WAVE; ++j) voffC[j] = voffN[j];
        epar ^= 1;
        // the staging area (buffer 1 for the 3x3 form) is rewritten by the DMA of the new item's chunk 1: every wave must have
        // finished its staging reads.  No DMA is outstanding here, so the global stores of the epilogue need not be waited for.
        lm_barrier_lds();
        LM_TRACE_MARK(4);
    }
    if (PROD != 0 && p.range_flag != nullptr && lm_pk_out_of_f16_guard(pr_max)) atomicOr(p.range_flag, 1u);
    LM_TRACE_FLUSH();
}

// The persistent kernel addresses a tensor through a raw buffer descriptor with 32-bit offsets whose top bit marks an
// out-of-image lane: it serves tensors below 2 GiB with an even number of 16-channel chunks (every level of the network;
// batches are cut into sub-batches that fit); anything else takes the simple kernel.
static bool h3_persistent_ok(const ConvParamsH3& p, int taps) {
    // LM_H3_FALLBACK=1 forces the simple 4-wave kernel everywhere (it normally only serves odd widths): test hook
    static const bool wide_ok = [] { const char* e = getenv("LM_H3_FALLBACK"); return !(e && e[0] == '1'); }();
    const size_t slice_bytes = (size_t)p.H * p.W * p.in_cstride * 4;
    return wide_ok && (p.W % 32 == 0 || p.W == 16) && (p.Cin / KC) % (taps == 1 ? 4 : 2) == 0 && 2 * slice_bytes < 0x7fffffffull &&
           (size_t)taps * p.Cout * p.Cin * 4 < 0x7fffffffull && (p.bn_s != nullptr) == (taps == 9);
}

template <int TAPS>
static hipError_t launch_conv_h3_t(const ConvParamsH3& p, hipStream_t stream) {
    if (p.Cin % KC != 0 || p.Cout % TN != 0 || (p.in_cstride & 7) || (p.in_coff & 7) || (p.out_cstride & 7) || (p.out_coff & 7)) return hipErrorInvalidValue;
    if (p.pool != nullptr && (((p.H | p.W) & 1) || (p.pool_cstride & 7) || (p.pool_coff & 7))) return hipErrorInvalidValue;
    if (h3_persistent_ok(p, TAPS)) {
        const bool g16 = p.W == 16;
        // LM_H3_ORDER: 0 cout-major everywhere, 1 XCD-aware pixel-tile-major everywhere, default: per layer (see the kernel)
        static const int order_env = [] { const char* e = getenv("LM_H3_ORDER"); return e ? atoi(e) : -1; }();
        static const int n_cu = [] {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            return n > 0 ? n : 256;
        }();
        // sub-batches whose input tensor stays below 2 GiB (an even number of slices: the 16-wide geometry pairs them)
        const size_t slice_bytes = (size_t)p.H * p.W * p.in_cstride * 4;
        const int max_b = std::max(2, (int)(0x7fffffffull / slice_bytes) & ~1);
        for (int b0 = 0; b0 < p.B; b0 += max_b) {
            ConvParamsH3 pd = p;
            pd.B = std::min(max_b, p.B - b0);
            pd.in = p.in + (size_t)b0 * slice_bytes;
            pd.out = p.out + (size_t)b0 * p.H * p.W * p.out_cstride * 4;
            if (p.pool) pd.pool = p.pool + (size_t)b0 * (p.H / 2) * (p.W / 2) * p.pool_cstride * 4;
            if (p.head_labels) pd.head_labels = p.head_labels + (size_t)b0 * p.H * p.W;
            if (p.head_logp) pd.head_logp = p.head_logp + (size_t)b0 * p.head_C * p.H * p.W;
            if (p.fc_x) pd.fc_x = p.fc_x + (size_t)b0 * p.H * p.W;
            const int n_ptiles = g16 ? ((p.H + TH - 1) / TH) * ((pd.B + 1) / 2) : (p.W / 32) * ((p.H + TH - 1) / TH) * pd.B;
            const int n_ct = p.Cout / TN;
            const int ks = p.ksplit > 1 ? p.ksplit : 1;  // split-K items: cout-tile-major order, ks outermost
            const int xcd_order = ks > 1 ? 0 : (order_env >= 0 ? order_env : (n_ct >= 2 && n_ptiles >= 64 ? 1 : 0));
            const int n_items = xcd_order ? 8 * ((n_ptiles + 7) / 8) * n_ct : n_ptiles * n_ct * ks;
            // LM_H3_GRID: lab hook, caps the number of persistent workgroups
            static const int grid_cap = [] { const char* e = getenv("LM_H3_GRID"); return e ? atoi(e) : 0; }();
            const unsigned blocks = (unsigned)std::min(n_items, grid_cap > 0 ? std::min(grid_cap, n_cu) : n_cu);
            if (TAPS == 9 && ks > 1) {  // the split-K instantiation (launch_conv3x3_h3 has checked the shape)
                if (g16) LM_LAUNCH((conv_igemm_h3p<9, true, 0, 0, true>), dim3(blocks), dim3(512), 0, stream, pd, n_ptiles, n_items, xcd_order);
                else LM_LAUNCH((conv_igemm_h3p<9, false, 0, 0, true>), dim3(blocks), dim3(512), 0, stream, pd, n_ptiles, n_items, xcd_order);
            } else if (g16)
                LM_LAUNCH((conv_igemm_h3p<TAPS, true>), dim3(blocks), dim3(512), 0, stream, pd, n_ptiles, n_items, xcd_order);
            else if (TAPS == 9 && pd.head_labels != nullptr && pd.head_logp != nullptr)
                LM_LAUNCH((conv_igemm_h3p<9, false, 2>), dim3(blocks), dim3(512), 0, stream, pd, n_ptiles, n_items, xcd_order);
            else if (TAPS == 9 && pd.head_labels != nullptr)
                LM_LAUNCH((conv_igemm_h3p<9, false, 1>), dim3(blocks), dim3(512), 0, stream, pd, n_ptiles, n_items, xcd_order);
            else if (TAPS == 9 && pd.fc_x != nullptr)
                LM_LAUNCH((conv_igemm_h3p<9, false, 0, 1>), dim3(blocks), dim3(512), 0, stream, pd, n_ptiles, n_items, xcd_order);
            else
                LM_LAUNCH((conv_igemm_h3p<TAPS, false>), dim3(blocks), dim3(512), 0, stream, pd, n_ptiles, n_items, xcd_order);
            const hipError_t err = hipGetLastError();
            if (err != hipSuccess) return err;
        }
        return hipSuccess;
    }
    const int tiles = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH);
    dim3 grid((unsigned)(tiles * p.B), (unsigned)(p.Cout / TN));
    LM_LAUNCH((conv_igemm_h3<TAPS>), grid, dim3(256), (H3Smem<TAPS>::BYTES), stream, p);
    return hipGetLastError();
}

bool conv3x3_h3_can_fuse_head(const ConvParamsH3& p) {
    static const bool allow = [] { const char* e = getenv("LM_H3_FUSE_HEAD"); return !(e && e[0] == '0'); }();  // A/B hook
    return allow && p.Cout == TN && p.W % 32 == 0 && h3_persistent_ok(p, 9);
}

bool conv3x3_h3_can_fuse_first(const ConvParamsH3& p) {
    static const bool allow = [] { const char* e = getenv("LM_H3_FUSE_FIRST"); return !(e && e[0] == '0'); }();  // A/B hook
    return allow && p.Cin == 64 && p.W % 32 == 0 && p.head_labels == nullptr && h3_persistent_ok(p, 9);
}

// Reduction of a split-K 3x3 conv (the KS instantiation): parts added in index order, then the conv's epilogue -- acc * 2^-k + (bias -
// the border correction of a deferred-shift input), ReLU, BatchNorm scale / shift (ones / zeros when the consumers carry them), the
// hi / lo split, the f16 range guard, and the 2 x 2 average pool of the encoder's second convs.  POOL: thread = (2 x 2 pixel block,
// 8-channel group), else (pixel, group).
template <bool POOL>
__global__ __launch_bounds__(256) void splitk_reduce3_h3_kernel(ConvParamsH3 p) {
    const size_t G = (size_t)p.Cout >> 3;
    const int Wq = POOL ? p.W >> 1 : p.W, Hq = POOL ? p.H >> 1 : p.H;
    const size_t nq = (size_t)p.B * Hq * Wq, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nq * G) return;
    const size_t q = e / G;
    const int g = (int)(e - q * G);
    const int xq = (int)(q % Wq), yq = (int)((q / Wq) % Hq), b = (int)(q / ((size_t)Wq * Hq));
    const size_t npix = (size_t)p.B * p.H * p.W;
    float bias[8], sc[8], sh[8], pl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bias[k] = p.bias[8 * g + k];
        sc[k] = p.bn_s[8 * g + k];
        sh[k] = p.bn_t[8 * g + k];
        pl[k] = 0.f;
    }
    unsigned gmax = 0u;
    constexpr int NP = POOL ? 4 : 1;
    float vv[NP][8];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int y = POOL ? 2 * yq + (i & 1) : yq, x = POOL ? 2 * xq + (i >> 1) : xq;  // (rows first: the pool's summation order below)
        const size_t pix = ((size_t)b * p.H + y) * p.W + x;
        float a[8];
        for (int s = 0; s < p.ksplit; ++s) {
            const float4* src = reinterpret_cast<const float4*>(p.kpart + ((size_t)s * npix + pix) * p.Cout + 8 * g);
            const float4 v0 = src[0], v1 = src[1];
            if (s == 0) {
                a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
            } else {
                a[0] += v0.x; a[1] += v0.y; a[2] += v0.z; a[3] += v0.w; a[4] += v1.x; a[5] += v1.y; a[6] += v1.z; a[7] += v1.w;
            }
        }
        const int mask = (y == 0 ? 1 : 0) | (y == p.H - 1 ? 2 : 0) | (x == 0 ? 4 : 0) | (x == p.W - 1 ? 8 : 0);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float bb = bias[k];
            if (p.border_corr != nullptr && mask) bb -= p.border_corr[(size_t)mask * p.Cout + 8 * g + k];
            v[k] = fmaf(fmaxf(fmaf(a[k], p.acc_scale, bb), 0.f), sc[k], sh[k]);
            vv[i][k] = v[k];
        }
        uint2 h0, l0, h1, l1;
        lm_split4(v[0], v[1], v[2], v[3], &h0, &l0);
        lm_split4(v[4], v[5], v[6], v[7], &h1, &l1);
        gmax = lm_pk_absmax_u16(lm_pk_absmax_u16(lm_pk_absmax_u16(lm_pk_absmax_u16(gmax, h0.x), h0.y), h1.x), h1.y);
        char* dst = p.out + (pix * p.out_cstride + p.out_coff) * 4 + (size_t)g * 32;
        *reinterpret_cast<uint4*>(dst) = uint4{h0.x, h0.y, h1.x, h1.y};
        *reinterpret_cast<uint4*>(dst + 16) = uint4{l0.x, l0.y, l1.x, l1.y};
    }
    if constexpr (POOL) {  // avg_pool2d(2): rows y, y + 1 summed first, then x + 1 (the persistent kernel's order)
#pragma unroll
        for (int k = 0; k < 8; ++k) pl[k] = 0.25f * ((vv[0][k] + vv[1][k]) + (vv[2][k] + vv[3][k]));
        uint2 h0, l0, h1, l1;
        lm_split4(pl[0], pl[1], pl[2], pl[3], &h0, &l0);
        lm_split4(pl[4], pl[5], pl[6], pl[7], &h1, &l1);
        char* dst = p.pool + ((((size_t)b * Hq + yq) * Wq + xq) * p.pool_cstride + p.pool_coff) * 4 + (size_t)g * 32;
        *reinterpret_cast<uint4*>(dst) = uint4{h0.x, h0.y, h1.x, h1.y};
        *reinterpret_cast<uint4*>(dst + 16) = uint4{l0.x, l0.y, l1.x, l1.y};
    }
    if (p.range_flag != nullptr && lm_pk_out_of_f16_guard(gmax)) atomicOr(p.range_flag, 1u);
}

// The K split launch_conv3x3_h3 uses for this shape when no accumulator chain may run over more than max_k = 9 * channels products
// (1: none; the engine sizes kpart with it): parts of an even number (>= 2) of 16-channel chunks, plain 3x3 launches on the
// persistent kernel only (no fused head / first layer), the whole batch in one launch.
int conv3x3_h3_ksplit(const ConvParamsH3& p, int max_k) {
    if (max_k <= 0 || p.head_labels != nullptr || p.fc_x != nullptr || p.bn_s == nullptr || p.bn_t == nullptr || !h3_persistent_ok(p, 9)) return 1;
    if ((size_t)p.B * p.H * p.W * p.in_cstride * 4 >= 0x7fffffffull) return 1;  // (a launch cut into sub-batches keeps the single chain)
    int S = 1;
    while (9 * (p.Cin / S) > max_k && (p.Cin / KC) % (2 * S * 2) == 0 && p.Cin / KC / (2 * S) >= 2 && S < 16) S *= 2;
    return S;
}

hipError_t launch_conv3x3_h3(const ConvParamsH3& p, hipStream_t stream) {
    if (p.fc_x != nullptr && (!conv3x3_h3_can_fuse_first(p) || !p.fc_c)) return hipErrorInvalidValue;
    if (p.head_labels != nullptr && (!conv3x3_h3_can_fuse_head(p) || p.head_C < 1 || p.head_C > kMaxClasses || !p.head_w || !p.head_b))
        return hipErrorInvalidValue;
    if (p.head_logp != nullptr && p.head_labels == nullptr) return hipErrorInvalidValue;
    if (p.ksplit <= 1) return launch_conv_h3_t<9>(p, stream);
    if (p.kpart == nullptr || conv3x3_h3_ksplit(p, 9 * p.Cin / p.ksplit) != p.ksplit || (p.pool != nullptr && ((p.H | p.W) & 1))) return hipErrorInvalidValue;
    const hipError_t err = launch_conv_h3_t<9>(p, stream);
    if (err != hipSuccess) return err;
    if (p.pool != nullptr) {
        const size_t n = (size_t)p.B * (p.H / 2) * (p.W / 2) * (p.Cout >> 3);
        LM_LAUNCH((splitk_reduce3_h3_kernel<true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p);
    } else {
        const size_t n = (size_t)p.B * p.H * p.W * (p.Cout >> 3);
        LM_LAUNCH((splitk_reduce3_h3_kernel<false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p);
    }
    return hipGetLastError();
}
// Reduction of a split-K 1x1 conv: thread = (pixel, 8-channel group); parts added in index order, then the conv epilogue of the 1x1
// form (acc * 2^-k + bias, no ReLU / BatchNorm), the hi / lo split and the f16 range guard.
__global__ __launch_bounds__(256) void splitk_reduce_h3_kernel(const float* __restrict__ part, int S, size_t npix, int Cout, float acc_scale,
                                                               const float* __restrict__ bias, char* out, int out_cstride, int out_coff, unsigned* range_flag) {
    const size_t G = (size_t)Cout >> 3, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= npix * G) return;
    const size_t pix = e / G;
    const int g = (int)(e - pix * G);
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    for (int s = 0; s < S; ++s) {
        const float4* src = reinterpret_cast<const float4*>(part + ((size_t)s * npix + pix) * Cout + 8 * g);
        const float4 v0 = src[0], v1 = src[1];
        if (s == 0) {
            a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
        } else {
            a[0] += v0.x; a[1] += v0.y; a[2] += v0.z; a[3] += v0.w; a[4] += v1.x; a[5] += v1.y; a[6] += v1.z; a[7] += v1.w;
        }
    }
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = fmaf(a[k], acc_scale, bias[8 * g + k]);
    uint2 h0, l0, h1, l1;
    lm_split4(v[0], v[1], v[2], v[3], &h0, &l0);
    lm_split4(v[4], v[5], v[6], v[7], &h1, &l1);
    const unsigned gmax = lm_pk_absmax_u16(lm_pk_absmax_u16(lm_pk_absmax_u16(lm_pk_absmax_u16(0u, h0.x), h0.y), h1.x), h1.y);
    char* dst = out + (pix * out_cstride + out_coff) * 4 + (size_t)g * 32;
    *reinterpret_cast<uint4*>(dst) = uint4{h0.x, h0.y, h1.x, h1.y};
    *reinterpret_cast<uint4*>(dst + 16) = uint4{l0.x, l0.y, l1.x, l1.y};
    if (range_flag != nullptr && lm_pk_out_of_f16_guard(gmax)) atomicOr(range_flag, 1u);
}

int conv1x1_h3_ksplit(const ConvParamsH3& p) {
    static const bool allow = [] { const char* e = getenv("LM_H3_SPLITK"); return !(e && e[0] == '0'); }();  // A/B hook
    if (!allow || !h3_persistent_ok(p, 1) || p.pool != nullptr) return 1;
    if ((size_t)p.B * p.H * p.W * p.in_cstride * 4 >= 0x7fffffffull) return 1;  // (a launch cut into sub-batches keeps the single chain)
    const bool g16 = p.W == 16;
    const int n_ptiles = g16 ? ((p.H + TH - 1) / TH) * ((p.B + 1) / 2) : (p.W / 32) * ((p.H + TH - 1) / TH) * p.B;
    const int items = n_ptiles * (p.Cout / TN), nstages = p.Cin / (2 * KC);
    int S = 1;
    // as many parts as keep the item count within two rounds of the chip and leave every part an even number (>= 4) of stages
    while (items * S * 2 <= 512 && nstages % (S * 2 * 2) == 0 && nstages / (S * 2) >= 4) S *= 2;
    return S;
}

hipError_t launch_conv1x1_h3(const ConvParamsH3& p, hipStream_t stream) {
    if (p.ksplit <= 1) return launch_conv_h3_t<1>(p, stream);
    if (p.ksplit != conv1x1_h3_ksplit(p) || p.kpart == nullptr) return hipErrorInvalidValue;
    const hipError_t err = launch_conv_h3_t<1>(p, stream);
    if (err != hipSuccess) return err;
    const size_t npix = (size_t)p.B * p.H * p.W, n = npix * (p.Cout >> 3);
    LM_LAUNCH(splitk_reduce_h3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p.kpart, p.ksplit, npix, p.Cout, p.acc_scale, p.bias, p.out,
              p.out_cstride, p.out_coff, p.range_flag);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// First layer (Cin = 1), fp32 arithmetic on the VALU, split output.  thread = (pixel, group of 8 output channels) with the
// group fastest: the 8 threads of a pixel write its 256 bytes as 16 x 16-byte stores (one 32-byte hi8|lo8 group each), a wave
// writes 8 consecutive pixels = 2 KiB contiguous.  (Round 1 had lane = channel and stored single halves.)
__global__ __launch_bounds__(256) void first_conv_h3_kernel(FirstConvParams p) {
    __shared__ float tile[18 * 18];
    const int tid = threadIdx.x, grp = tid & 7, pl = tid >> 3;  // channel group, pixel lane 0..31
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    for (int idx = tid; idx < 18 * 18; idx += 256) {
        const int py = idx / 18, px = idx - py * 18;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        float v = 0.f;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) v = p.in[((size_t)b * p.H + gy) * p.W + gx];
        tile[idx] = v;
    }
    float wr[9][8], bias[8], s[8], sh[8];  // this thread's 8 channels: weights [tap][channel] (p.w is [9][64])
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wr[k][c] = p.w[k * 64 + 8 * grp + c];
        bias[c] = p.bias[8 * grp + c];
        s[c] = p.bn_s[8 * grp + c];
        sh[c] = p.bn_t[8 * grp + c];
    }
    __syncthreads();
    char* out = reinterpret_cast<char*>(p.out);
    unsigned gmax = 0u;  // f16 range guard (running max of |hi| bit patterns)
    for (int pass = 0; pass < 8; ++pass) {  // 8 passes x 32 pixels = the 16 x 16 tile
        const int pix = pass * 32 + pl, r = pix >> 4, c0 = pix & 15;
        float in9[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) in9[k] = tile[(r + k / 3) * 18 + c0 + (k % 3)];
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float a = bias[c];
#pragma unroll
            for (int k = 0; k < 9; ++k) a = fmaf(in9[k], wr[k][c], a);  // same chain as the per-channel form: bias, then taps 0..8
            v[c] = fmaf(fmaxf(a, 0.f), s[c], sh[c]);
        }
        const int y = y0 + r, x = x0 + c0;
        if (y < p.H && x < p.W) {
            char* g = out + ((((size_t)b * p.H + y) * p.W + x) * p.out_cstride + p.out_coff) * 4 + (size_t)grp * 32;
            uint2 h0, l0, h1, l1;
            lm_split4(v[0], v[1], v[2], v[3], &h0, &l0);
            lm_split4(v[4], v[5], v[6], v[7], &h1, &l1);
            gmax = lm_pk_absmax_u16(lm_pk_absmax_u16(lm_pk_absmax_u16(lm_pk_absmax_u16(gmax, h0.x), h0.y), h1.x), h1.y);
            const uint4 hi = {h0.x, h0.y, h1.x, h1.y}, lo = {l0.x, l0.y, l1.x, l1.y};
            lm_store16_stream(g, hi);
            lm_store16_stream(g + 16, lo);
        }
    }
    if (p.range_flag != nullptr && lm_pk_out_of_f16_guard(gmax)) atomicOr(p.range_flag, 1u);
}
hipError_t launch_first_conv_h3(const FirstConvParams& p, hipStream_t stream) {
    const int tiles = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH);
    LM_LAUNCH(first_conv_h3_kernel, dim3((unsigned)(tiles * p.B)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

namespace {
__device__ __forceinline__ void load_group(const char* g, float* v) {  // 32-byte group -> 8 floats
    lm_h16 hh[8], ll[8];
    const uint4 a = *reinterpret_cast<const uint4*>(g), c = *reinterpret_cast<const uint4*>(g + 16);
    memcpy(hh, &a, 16);
    memcpy(ll, &c, 16);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = lm_h2f(ll[k]) + lm_h2f(hh[k]);
}
}  // namespace

// Bilinear x2 (align_corners=False) on split tensors.  thread = (LOW-resolution pixel (i, j), 8-channel group) and writes the 2 x 2
// output pixels (2i..2i+1, 2j..2j+1): the 3 x 3 low-resolution neighbourhood is loaded and converted once for four outputs (a
// thread per output pixel loaded and converted 4 groups per output: the kernel was bound by its conversions, not by HBM), the
// interpolation is separable -- per source row the two horizontal phases, then the two vertical ones -- with exactly the weights,
// operands and operation order of the per-output form (out = wya * (wxa * a00 + wxb * a01) + wyb * (wxa * a10 + wxb * a11), edge
// rows / columns with weights (1, 0) on the clamped index), so the results are bit-identical to it.  One block row per
// low-resolution image row; a wave writes 2 rows x 64 / G x 2 pixels of 4 * C contiguous bytes each.
__global__ __launch_bounds__(256) void upsample2x_h3_kernel(UpsampleParams p) {
    const unsigned G = (unsigned)p.C >> 3;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;  // element of the low-resolution row: j * G + g
    if (e >= (unsigned)p.w * G) return;
    const int j = (int)(e / G), g = (int)(e - (unsigned)j * G);
    const int i = (int)blockIdx.y, b = (int)blockIdx.z;
    const int W2 = 2 * p.w, H2 = 2 * p.h;
    const char* in = reinterpret_cast<const char*>(p.in);
    char* out = reinterpret_cast<char*>(p.out);
    // horizontal taps of the even output column 2j: (xa, xb, wxa, wxb); of the odd one 2j+1: (j, min(j+1, w-1), 0.75, 0.25)
    const int exa = j == 0 ? 0 : j - 1, exb = j;
    const float ewa = j == 0 ? 1.f : 0.25f, ewb = j == 0 ? 0.f : 0.75f;
    const int oxb = min(j + 1, p.w - 1);
    const int rows[3] = {max(i - 1, 0), i, min(i + 1, p.h - 1)};
    const char* base = in + (size_t)b * p.h * p.w * p.C * 4 + (size_t)g * 32;
    float hE[3][8], hO[3][8];  // horizontally interpolated source rows i-1, i, i+1: even / odd output column
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (r == 0 && i == 0) continue;  // (row i-1 is not used by the first image row: weights (1, 0) on row 0)
        const char* rowp = base + (size_t)rows[r] * p.w * p.C * 4;
        float am[8], ac[8], ap[8];
        load_group(rowp + (size_t)exa * p.C * 4, am);
        load_group(rowp + (size_t)j * p.C * 4, ac);
        load_group(rowp + (size_t)oxb * p.C * 4, ap);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            hE[r][k] = ewa * am[k] + ewb * (j == 0 ? am[k] : ac[k]);  // j == 0: both taps are pixel 0 (exa == exb == 0)
            hO[r][k] = 0.75f * ac[k] + 0.25f * ap[k];
        }
    }
    auto store = [&](int y, int x, const float* o) __attribute__((always_inline)) {
        char* dst = out + ((((size_t)b * H2 + y) * W2 + x) * p.out_cstride + p.out_coff) * 4 + (size_t)g * 32;
        uint2 h0, l0, h1, l1;
        lm_split4(o[0], o[1], o[2], o[3], &h0, &l0);
        lm_split4(o[4], o[5], o[6], o[7], &h1, &l1);
        const uint4 hi = {h0.x, h0.y, h1.x, h1.y}, lo = {l0.x, l0.y, l1.x, l1.y};
        lm_store16_stream(dst, hi);
        lm_store16_stream(dst + 16, lo);
    };
    float o[8];
    // even output row 2i: (ya, yb, wya, wyb) = i == 0 ? (0, 0, 1, 0) : (i-1, i, 0.25, 0.75)
    const float eya = i == 0 ? 1.f : 0.25f, eyb = i == 0 ? 0.f : 0.75f;
    const int ra = i == 0 ? 1 : 0;  // index into hE / hO of the row "ya"
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = eya * hE[ra][k] + eyb * hE[1][k];
    store(2 * i, 2 * j, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = eya * hO[ra][k] + eyb * hO[1][k];
    store(2 * i, 2 * j + 1, o);
    // odd output row 2i+1: (i, min(i+1, h-1), 0.75, 0.25)
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = 0.75f * hE[1][k] + 0.25f * hE[2][k];
    store(2 * i + 1, 2 * j, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = 0.75f * hO[1][k] + 0.25f * hO[2][k];
    store(2 * i + 1, 2 * j + 1, o);
}

hipError_t launch_upsample2x_h3(const UpsampleParams& p, hipStream_t stream) {
    if ((p.C & 7) || (p.out_cstride & 7) || (p.out_coff & 7)) return hipErrorInvalidValue;
    const unsigned row_elems = (unsigned)p.w * (unsigned)(p.C >> 3);
    if (p.B > 65535 || p.h > 65535) return hipErrorInvalidValue;
    LM_LAUNCH(upsample2x_h3_kernel, dim3((row_elems + 255) / 256, (unsigned)p.h, (unsigned)p.B), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// Head on a split tensor: 8 lanes share one pixel (one 32-byte group of the 64 channels each).
__global__ __launch_bounds__(256) void head_h3_kernel(HeadParams p) {
    __shared__ float wsm[kMaxClasses * 64];
    __shared__ float bsm[kMaxClasses];
    const int tid = threadIdx.x;
    for (int i = tid; i < p.C * 64; i += 256) wsm[i] = p.w[i];
    if (tid < p.C) bsm[tid] = p.bias[tid];
    __syncthreads();
    const int q = tid & 7;
    const size_t npix = (size_t)p.B * p.H * p.W;
    const size_t HW = (size_t)p.H * p.W;
    const size_t ngroups = (npix + 31) / 32;  // 32 pixels per 256-thread block-iteration
    const char* in = reinterpret_cast<const char*>(p.in);
    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t pix = g * 32 + (tid >> 3);
        const bool valid = pix < npix;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        if (valid) load_group(in + pix * 256 + (size_t)q * 32, v);
        float part[kMaxClasses];
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) {
            float s = 0.f;
            if (c < p.C) {
#pragma unroll
                for (int k = 0; k < 8; ++k) s = fmaf(v[k], wsm[c * 64 + 8 * q + k], s);
            }
            part[c] = s;
        }
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) {
            if (c < p.C) {  // wave-uniform
#pragma unroll
                for (int m = 4; m >= 1; m >>= 1) part[c] += __shfl_xor(part[c], m);
            }
        }
        if (valid && q == 0) {
            float best = part[0] + bsm[0];
            int arg = 0;
            float lg[kMaxClasses];
            lg[0] = best;
#pragma unroll
            for (int c = 1; c < kMaxClasses; ++c) {
                if (c < p.C) {
                    lg[c] = part[c] + bsm[c];
                    if (lg[c] > best) { best = lg[c]; arg = c; }
                }
            }
            if (p.labels) p.labels[pix] = (uint8_t)arg;
            if (p.logp) {
                float se = 0.f;
#pragma unroll
                for (int c = 0; c < kMaxClasses; ++c)
                    if (c < p.C) se += expf(lg[c] - best);
                const float lse = best + logf(se);
                const size_t b = pix / HW, yx = pix - b * HW;
#pragma unroll
                for (int c = 0; c < kMaxClasses; ++c)
                    if (c < p.C) p.logp[(b * p.C + c) * HW + yx] = lg[c] - lse;
            }
        }
    }
}

hipError_t launch_head_h3(const HeadParams& p, hipStream_t stream) {
    if (p.C < 1 || p.C > kMaxClasses) return hipErrorInvalidValue;
    const size_t npix = (size_t)p.B * p.H * p.W;
    const unsigned blocks = (unsigned)std::min<size_t>((npix + 31) / 32, 256 * 32);
    LM_LAUNCH(head_h3_kernel, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace lm
