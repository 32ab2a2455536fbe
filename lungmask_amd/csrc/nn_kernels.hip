// U-Net forward kernels for gfx950 (MI355X).  Hand-written; no library calls.
//
//  * conv_igemm_f32<TAPS,KC>: im2col-free implicit GEMM for the 3x3 (TAPS=9) and
//    1x1 (TAPS=1) convolutions on the exact-f32 matrix core op
//    v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain, so logits stay fp32-class).
//    M = pixels, N = output channels, K = taps*Cin.  A 16x16-pixel x 64-channel
//    output tile per 256-thread workgroup; each of the 4 waves owns a 4x16 pixel
//    strip = 2 M-tiles x 2 N-tiles of 32x32 (64 accumulator VGPRs).  Per
//    KC-channel chunk the (16+2)^2 halo tile and the [taps][KC][64] weight slab
//    are staged once in LDS and the 9 taps are 9 shifted LDS reads of the same
//    tile -- nothing is materialised.  Epilogue fuses bias + ReLU + BatchNorm
//    (eval affine) + optional avg_pool2d(2) (resunet.py:93-105, :64) and writes
//    NHWC with a channel offset, so torch.cat (resunet.py:147) costs nothing.
//  * first_conv_kernel: Cin=1 3x3 conv (K=9: bandwidth-bound, VALU).
//  * upsample2x_kernel: bilinear x2, align_corners=False (resunet.py:132).
//  * head_kernel: 1x1 conv 64->C + log_softmax + first-max argmax
//    (resunet.py:69-70, mask.py:184-186).
#include "nn_kernels.h"

namespace lm {

constexpr int TH = 16, TW = 16, TN = 64;

template <int TAPS, int KC>
struct ConvSmem {
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO;
    static constexpr int S = KC + 1;  // odd per-pixel stride: A-fragment reads hit 32 distinct banks
    static constexpr int A_FLOATS = ((PH * PW * S) + 3) & ~3;
    static constexpr int W_FLOATS = TAPS * KC * TN;
    static constexpr int BYTES = (A_FLOATS + W_FLOATS) * 4;
};

template <int TAPS, int KC>
__global__ __launch_bounds__(256) void conv_igemm_f32(ConvParams p) {
    using SM = ConvSmem<TAPS, KC>;
    constexpr int HALO = SM::HALO, PW = SM::PW, PH = SM::PH, S = SM::S;
    LM_DYN_SMEM(smem);
    float* As = reinterpret_cast<float*>(smem);
    float* Ws = As + SM::A_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH, n0 = blockIdx.y * TN;

    lm_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int li = lane & 31, kx = lane >> 5;
    const int pr = li >> 4, pc = li & 15;
    const float* __restrict__ in_b = p.in + (size_t)b * p.H * p.W * p.in_cstride + p.in_coff;

    for (int c0 = 0; c0 < p.Cin; c0 += KC) {
        // Chunked accumulation: every 16-channel chunk (TAPS x 8 matrix steps of two products each) sums into a fresh accumulator
        // that is added to the running one in chunk order.  The matrix op is a sequential fp32 chain -- one rounding per two
        // products, up to 4608 in a row for K = 9216 -- and as ONE chain it was less accurate than torch's own blocked fp32 GEMM
        // (6.4e-4 against the reference's 1.6e-4 distance to float64 on the Appendix-D head, round 3); 72 steps per chunk + one
        // addition per chunk bound the error growth by sqrt(72) + sqrt(K / 144) instead of sqrt(K / 2).
        lm_f32x16 cacc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) cacc[i][j][r] = 0.f;
        __syncthreads();
        // ---- stage the activation halo tile: [PH*PW pixels][KC channels], zero outside the image
        for (int idx = tid; idx < PH * PW * (KC / 4); idx += 256) {
            const int q = idx % (KC / 4), pix = idx / (KC / 4);
            const int py = pix / PW, px = pix - py * PW;
            const int gy = y0 + py - HALO, gx = x0 + px - HALO;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                v = *reinterpret_cast<const float4*>(in_b + ((size_t)gy * p.W + gx) * p.in_cstride + c0 + 4 * q);
            float* d = As + pix * S + 4 * q;
            d[0] = v.x;
            d[1] = v.y;
            d[2] = v.z;
            d[3] = v.w;
        }
        // ---- stage the weight slab: [TAPS][KC][TN]
        for (int idx = tid; idx < TAPS * KC * (TN / 4); idx += 256) {
            const int j = idx % (TN / 4), rk = idx / (TN / 4);
            const int tap = rk / KC, k = rk - tap * KC;
            const float4 v = *reinterpret_cast<const float4*>(p.w + ((size_t)tap * p.Cin + c0 + k) * p.Cout + n0 + 4 * j);
            *reinterpret_cast<float4*>(Ws + rk * TN + 4 * j) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = (TAPS == 9) ? tap / 3 : 0, dx = (TAPS == 9) ? tap - 3 * dy : 0;
            const float* a0p = As + ((4 * wave + pr + dy) * PW + pc + dx) * S + kx;
            const float* a1p = a0p + 2 * PW * S;
            const float* bp = Ws + (tap * KC + kx) * TN + li;
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                const float a0 = a0p[2 * kk], a1 = a1p[2 * kk];
                const float b0 = bp[2 * kk * TN], b1 = bp[2 * kk * TN + 32];
                cacc[0][0] = lm_mfma_f32_32x32x2(a0, b0, cacc[0][0]);
                cacc[0][1] = lm_mfma_f32_32x32x2(a0, b1, cacc[0][1]);
                cacc[1][0] = lm_mfma_f32_32x32x2(a1, b0, cacc[1][0]);
                cacc[1][1] = lm_mfma_f32_32x32x2(a1, b1, cacc[1][1]);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += cacc[i][j][r];
    }

    // ---- epilogue: bias (+ ReLU + BN affine) (+ 2x2 average pool), NHWC stores (128 B per half-wave)
    const bool bn = p.bn_s != nullptr;
    const int Hp = p.H >> 1, Wp = p.W >> 1;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + 32 * nt + li;
            const float bias = p.bias[n];
            const float s = bn ? p.bn_s[n] : 1.f, sh = bn ? p.bn_t[n] : 0.f;
            float vals[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * kx;
                const int y = y0 + 4 * wave + 2 * mt + (i >> 4), x = x0 + (i & 15);
                float v = acc[mt][nt][r] + bias;
                if (bn) v = fmaf(fmaxf(v, 0.f), s, sh);
                vals[r] = v;
                if (y < p.H && x < p.W)
                    p.out[(((size_t)b * p.H + y) * p.W + x) * p.out_cstride + p.out_coff + n] = v;
            }
            if (p.pool != nullptr) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = 2 * rr;  // regs 0,2,4,6: even column, first row of the M-tile
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * kx;
                    const int y = y0 + 4 * wave + 2 * mt, x = x0 + (i & 15);
                    if (y + 1 < p.H && x + 1 < p.W) {
                        const float v = 0.25f * (((vals[r] + vals[r ^ 1]) + vals[r ^ 8]) + vals[r ^ 9]);
                        p.pool[(((size_t)b * Hp + (y >> 1)) * Wp + (x >> 1)) * p.pool_cstride + p.pool_coff + n] = v;
                    }
                }
            }
        }
    }
}

constexpr int kKC = 16;

template <int TAPS>
static hipError_t launch_conv(const ConvParams& p, hipStream_t stream) {
    if (p.Cin % kKC != 0 || p.Cout % TN != 0 || (p.in_cstride & 3) || (p.in_coff & 3)) return hipErrorInvalidValue;
    if (p.pool != nullptr && ((p.H | p.W) & 1)) return hipErrorInvalidValue;
    const int tiles = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH);
    dim3 grid((unsigned)(tiles * p.B), (unsigned)(p.Cout / TN));
    LM_LAUNCH((conv_igemm_f32<TAPS, kKC>), grid, dim3(256), (ConvSmem<TAPS, kKC>::BYTES), stream, p);
    return hipGetLastError();
}

hipError_t launch_conv3x3(const ConvParams& p, hipStream_t stream) { return launch_conv<9>(p, stream); }
hipError_t launch_conv1x1(const ConvParams& p, hipStream_t stream) { return launch_conv<1>(p, stream); }

// ---------------------------------------------------------------------------------------------
// First layer: Cin = 1 (down_path.0.block.0).  lane = output channel, wave = 4 rows of the tile.
__global__ __launch_bounds__(256) void first_conv_kernel(FirstConvParams p) {
    __shared__ float tile[18 * 18];
    __shared__ float wsm[9 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    for (int idx = tid; idx < 9 * 64; idx += 256) wsm[idx] = p.w[idx];
    __syncthreads();
    float wr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = wsm[k * 64 + lane];
    const float bias = p.bias[lane], s = p.bn_s[lane], sh = p.bn_t[lane];
    for (int pix = 0; pix < 64; ++pix) {
        const int r = 4 * wave + (pix >> 4), c = pix & 15;
        float v = bias;
#pragma unroll
        for (int k = 0; k < 9; ++k) v = fmaf(tile[(r + k / 3) * 18 + c + (k % 3)], wr[k], v);
        v = fmaf(fmaxf(v, 0.f), s, sh);
        const int y = y0 + r, x = x0 + c;
        if (y < p.H && x < p.W) p.out[(((size_t)b * p.H + y) * p.W + x) * p.out_cstride + p.out_coff + lane] = v;
    }
}

hipError_t launch_first_conv(const FirstConvParams& p, hipStream_t stream) {
    const int tiles = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH);
    LM_LAUNCH(first_conv_kernel, dim3((unsigned)(tiles * p.B)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Bilinear x2, align_corners=False: out[2i] = .25 in[i-1] + .75 in[i], out[2i+1] = .75 in[i] + .25 in[i+1],
// source index clamped (out[0] = in[0], out[2h-1] = in[h-1]).
__global__ __launch_bounds__(256) void upsample2x_kernel(UpsampleParams p) {
    const int C4 = p.C >> 2;
    const int H2 = 2 * p.h, W2 = 2 * p.w;
    const size_t total = (size_t)p.B * H2 * W2 * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        size_t rest = idx / C4;
        const int x = (int)(rest % W2);
        rest /= W2;
        const int y = (int)(rest % H2);
        const int b = (int)(rest / H2);
        int ya, yb, xa, xb;
        float wya, wyb, wxa, wxb;
        {
            const int i = y >> 1;
            if (y & 1) { ya = i; yb = min(i + 1, p.h - 1); wya = 0.75f; wyb = 0.25f; }
            else if (i == 0) { ya = 0; yb = 0; wya = 1.f; wyb = 0.f; }
            else { ya = i - 1; yb = i; wya = 0.25f; wyb = 0.75f; }
            const int j = x >> 1;
            if (x & 1) { xa = j; xb = min(j + 1, p.w - 1); wxa = 0.75f; wxb = 0.25f; }
            else if (j == 0) { xa = 0; xb = 0; wxa = 1.f; wxb = 0.f; }
            else { xa = j - 1; xb = j; wxa = 0.25f; wxb = 0.75f; }
        }
        const float* base = p.in + (size_t)b * p.h * p.w * p.C + 4 * c4;
        const float4 a00 = *reinterpret_cast<const float4*>(base + ((size_t)ya * p.w + xa) * p.C);
        const float4 a01 = *reinterpret_cast<const float4*>(base + ((size_t)ya * p.w + xb) * p.C);
        const float4 a10 = *reinterpret_cast<const float4*>(base + ((size_t)yb * p.w + xa) * p.C);
        const float4 a11 = *reinterpret_cast<const float4*>(base + ((size_t)yb * p.w + xb) * p.C);
        float4 o;
        o.x = wya * (wxa * a00.x + wxb * a01.x) + wyb * (wxa * a10.x + wxb * a11.x);
        o.y = wya * (wxa * a00.y + wxb * a01.y) + wyb * (wxa * a10.y + wxb * a11.y);
        o.z = wya * (wxa * a00.z + wxb * a01.z) + wyb * (wxa * a10.z + wxb * a11.z);
        o.w = wya * (wxa * a00.w + wxb * a01.w) + wyb * (wxa * a10.w + wxb * a11.w);
        *reinterpret_cast<float4*>(p.out + (((size_t)b * H2 + y) * W2 + x) * p.out_cstride + p.out_coff + 4 * c4) = o;
    }
}

hipError_t launch_upsample2x(const UpsampleParams& p, hipStream_t stream) {
    if ((p.C & 3) || (p.out_cstride & 3) || (p.out_coff & 3)) return hipErrorInvalidValue;
    const size_t total = (size_t)p.B * 4 * p.h * p.w * (p.C >> 2);
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
    LM_LAUNCH(upsample2x_kernel, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Head: logits = last(x) (1x1, 64 -> C); log_softmax over C; label = first index of the maximum.
// 16 lanes share one pixel (one float4 of the 64 channels each = a 256 B coalesced row).
__global__ __launch_bounds__(256) void head_kernel(HeadParams p) {
    __shared__ float wsm[kMaxClasses * 64];
    __shared__ float bsm[kMaxClasses];
    const int tid = threadIdx.x;
    for (int i = tid; i < p.C * 64; i += 256) wsm[i] = p.w[i];
    if (tid < p.C) bsm[tid] = p.bias[tid];
    __syncthreads();
    const int q = tid & 15;
    float wr[kMaxClasses][4];
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) wr[c][k] = (c < p.C) ? wsm[c * 64 + 4 * q + k] : 0.f;
    const size_t npix = (size_t)p.B * p.H * p.W;
    const size_t HW = (size_t)p.H * p.W;
    const size_t ngroups = (npix + 15) / 16;  // 16 pixels per 256-thread block-iteration
    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t pix = g * 16 + (tid >> 4);
        const bool valid = pix < npix;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v = *reinterpret_cast<const float4*>(p.in + pix * 64 + 4 * q);
        float part[kMaxClasses];
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) {
            float s = v.x * wr[c][0];
            s = fmaf(v.y, wr[c][1], s);
            s = fmaf(v.z, wr[c][2], s);
            s = fmaf(v.w, wr[c][3], s);
            part[c] = s;
        }
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) {
            if (c < p.C) {  // wave-uniform
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) part[c] += __shfl_xor(part[c], m);
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

hipError_t launch_head(const HeadParams& p, hipStream_t stream) {
    if (p.C < 1 || p.C > kMaxClasses) return hipErrorInvalidValue;
    const size_t npix = (size_t)p.B * p.H * p.W;
    const unsigned blocks = (unsigned)std::min<size_t>((npix + 15) / 16, 256 * 32);
    LM_LAUNCH(head_kernel, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace lm
