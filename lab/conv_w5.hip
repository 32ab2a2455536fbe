// LAB (not part of the product): conv_w5's main loop (one wave per SIMD, 4 rows per wave) with the EPILOGUE OF ITEM i RUNNING
// INSIDE THE MAIN LOOP OF ITEM i + 1: a second accumulator set (the lone wave may use 512 registers), the epilogue cut into 32
// units of one (channel group, row) each -- bias / ReLU / BatchNorm / split, a v_permlane32_swap exchange so that every lane holds
// 16 contiguous bytes, one store straight from registers (no LDS staging, no fences) -- and one unit issued behind every eighth
// matrix instruction of the next item's first two chunks.
// (conv_w5.hip is the same kernel with the epilogue in its usual place.)
// --- original header of conv_w5.hip:
// the persistent 3x3 conv with ONE wave per SIMD -- 256 threads, wave w owns FOUR rows of the
// 16 x 32 tile (2 M-tiles x 4 N-tiles = 128 accumulator registers of the 512 a lone wave may use).  Question asked: does the
// main loop get cheaper when a wave's weight fragments serve four rows instead of two and its activation fragments are reused
// across the three vertical taps (per chunk 72 + 36 = 108 fragment reads per SIMD instead of 144), without a partner wave to
// cover its stalls?  Same LDS image, DMA, swizzle and epilogue structure as conv_igemm_h3p (nn_kernels_h3.hip); taps run
// column-major (dx outer) so that the six halo rows of one column offset stay in registers for three taps -- the fp32 summation
// order differs from the product kernel, outputs are compared with a tolerance.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I lungmask_amd/csrc -I include tools/ubench/conv_w5.hip -o tools/ubench/conv_w5
#include "../../lungmask_amd/csrc/nn_kernels_h3.hip"

#include <cmath>
#include <cstring>
#include <cstdio>
#include <vector>

namespace lm {

// DIRECT: the epilogue stores straight from registers -- a lane pair (kb = 0 / 1 of one pixel) exchanges halves with
// v_permlane32_swap so that each lane holds 16 contiguous bytes (hi8 / lo8 of an 8-channel group) -- instead of through LDS staging
template <bool POOLT>
__global__ __launch_bounds__(256) void conv_igemm_w5(ConvParamsH3 p, int n_ptiles, int n_items, int xcd_order) {
    using SM = H3WSmem<9, false>;
    constexpr int PW = SM::PW, ROWB = PW * 64, NWV = 4;
    constexpr int A_PER_WAVE = (SM::A_PIECES + NWV - 1) / NWV, W_PER_WAVE = (SM::W_PIECES + NWV - 1) / NWV;
    constexpr int HSTR = 144;
    static_assert(NWV * 128 * HSTR <= SM::BUF_BYTES, "staging fits the freed chunk buffer");
    __shared__ __attribute__((aligned(1024))) char lds[2 * SM::BUF_BYTES];
    __shared__ __attribute__((aligned(16))) float epi[2][3][TN];

    const int tid = threadIdx.x, lane = tid & 63, wave = lm_uniform(tid >> 6);
    const int li = lane & 31, kb = lane >> 5;
    const int wrow = 4 * wave;

    int a_off[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int px = li + dx;
        a_off[dx] = (wrow * PW + px) * 64 + ((2 * kb) ^ ((px >> 2) & 3)) * 16;
    }
    const int w_off = SM::A_BYTES + li * 64 + ((2 * kb) ^ ((li >> 2) & 3)) * 16;
    const int w_off_lo = w_off ^ 16;

    unsigned relA[A_PER_WAVE];
    int pyx[A_PER_WAVE];
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
        const int piece = wave + NWV * j, idx = piece * 64 + lane;
        pyx[j] = -1;
        relA[j] = 0;
        if (piece < SM::A_PIECES && idx < SM::A_ROWS * 4) {
            const int row = idx >> 2;
            const int py = row / PW, px = row - py * PW;
            const int ls = (idx & 3) ^ ((px >> 2) & 3);
            relA[j] = (unsigned)((py * p.W + px) * p.in_cstride * 4 + (ls >> 1) * 32 + (ls & 1) * 16);
            pyx[j] = py | (px << 8);
        }
    }
    unsigned voffW;
    {
        const int idx = wave * 64 + lane;
        const int row = idx >> 2, ls = (idx & 3) ^ ((row >> 2) & 3);
        const int tap = row / TN, n = row - tap * TN;
        voffW = (unsigned)((tap * p.Cout + n) * p.Cin * 4 + (ls >> 1) * 32 + (ls & 1) * 16);
    }
    const unsigned w_piece_stride = (unsigned)(NWV * 64 / 4 / TN) * (unsigned)p.Cout * (unsigned)p.Cin * 4u;  // one tap per j
    const unsigned slice_bytes = (unsigned)p.H * (unsigned)p.W * (unsigned)p.in_cstride * 4u;
    const lm_rsrc rsrcA = lm_make_rsrc(p.in + (size_t)p.in_coff * 4, (size_t)p.B * slice_bytes - (size_t)p.in_coff * 4);
    const lm_rsrc rsrcW = lm_make_rsrc(p.w, (size_t)9 * p.Cout * p.Cin * 4);
    const int tiles_x = p.W / 32;
    const int nchunks = p.Cin / KC;
    const int n_ct = p.Cout / TN;
    const int tiles_y = (p.H + TH - 1) / TH;
    auto decode = [&](int it, int& b, int& y0, int& x0, int& n0) -> bool {
        int ct, pt;
        if (xcd_order) {
            const int x = it & 7, s = it >> 3;
            ct = s % n_ct;
            pt = (s / n_ct) * 8 + x;
        } else {
            ct = it / n_ptiles;
            pt = it - ct * n_ptiles;
        }
        const bool valid = pt < n_ptiles;
        const int tx = pt % tiles_x;
        pt /= tiles_x;
        const int ty = pt % tiles_y;
        b = pt / tiles_y;
        y0 = ty * TH;
        x0 = tx * 32;
        n0 = ct * TN;
        return valid;
    };
    auto item_voffs = [&](int b, int y0, int x0, unsigned* voff) __attribute__((always_inline)) {
        const int ioff = ((y0 - 1) * p.W + (x0 - 1)) * p.in_cstride * 4;
#pragma unroll
        for (int j = 0; j < A_PER_WAVE; ++j) {
            const int gy = y0 + (pyx[j] & 0xff) - 1, gx = x0 + ((pyx[j] >> 8) & 0xff) - 1;
            const bool inb = pyx[j] >= 0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && b < p.B;
            voff[j] = inb ? (unsigned)((int)relA[j] + ioff) : LM_DMA_OOB;
        }
    };
    unsigned voffC[A_PER_WAVE], voffN[A_PER_WAVE];
    bool d_next = false;
    unsigned d_soffA = 0, d_soffW = 0;
    char* d_buf = lds;
    int d_nA = 0, d_nW = 0, d_epi = -1, d_n0 = 0;
    auto set_dma = [&](bool next_item, int b, int n0, int c0, int par, bool on, int epar_or_neg) __attribute__((always_inline)) {
        d_next = next_item;
        d_soffA = (unsigned)b * slice_bytes + (unsigned)c0 * 4u;
        d_soffW = ((unsigned)n0 * (unsigned)p.Cin + (unsigned)c0) * 4u;
        d_buf = lds + par * SM::BUF_BYTES;
        d_nA = on ? SM::A_PIECES : 0;
        d_nW = on ? SM::W_PIECES : 0;
        d_epi = on ? epar_or_neg : -1;
        d_n0 = n0;
    };
    constexpr int N_SLOTS = 21;  // 10 activation pieces, 9 weight pieces (one spare), epilogue constants
    static_assert(2 * A_PER_WAVE <= N_SLOTS - 1 && 2 * W_PER_WAVE <= N_SLOTS - 1, "DMA slots");
    const float* const epi_src = (wave == 0 ? p.bias : (wave == 1 ? p.bn_s : p.bn_t)) + lane;
    auto dma_slot = [&](int k) __attribute__((always_inline)) {
        const int j = k >> 1;
        if (k == N_SLOTS - 1) {
            if (d_epi >= 0 && wave < 3) lm_dma4_global(epi_src + d_n0, &epi[d_epi][wave][0]);
        } else if ((k & 1) == 0) {
            if (j < A_PER_WAVE && wave + NWV * j < d_nA) {
                const unsigned vc = voffC[j < A_PER_WAVE ? j : 0], vn = voffN[j < A_PER_WAVE ? j : 0];
                lm_dma16(rsrcA, d_next ? vn : vc, d_soffA, d_buf + (wave + NWV * j) * 1024);
            }
        } else {
            if (j < W_PER_WAVE && wave + NWV * j < d_nW)
                lm_dma16(rsrcW, voffW, d_soffW + (unsigned)j * w_piece_stride, d_buf + SM::A_BYTES + (wave + NWV * j) * 1024);
        }
    };

    lm_f32x16 acc[2][4], accO[2][4];
    int o_b = 0, o_y0 = 0, o_x0 = 0, o_n0 = 0, o_epar = 0;
    bool pending = false;
    float4 u_bias, u_s, u_sh;  // constants of the running (M-tile, channel group): loaded by its first unit, used by four
    float u_prev[4];           // the previous row's values of the group (2x2 average pool)
    const int Hp = p.H >> 1, Wp = p.W >> 1;
    // unit U = 16 * mt + 4 * g4 + nt of the deferred epilogue: one 4-channel group of one row of this wave
#define W5_EPI_UNIT(U)                                                                                                      \
    do {                                                                                                                   \
        if ((U) >= 0 && (U) < 32) {                                                                                        \
            constexpr int u_ = (U) >= 0 && (U) < 32 ? (U) : 0;                                                              \
            constexpr int mt_ = (u_ >> 4) & 1, g4_ = (u_ >> 2) & 3, nt_ = u_ & 3;                                           \
            const int cl_ = 32 * mt_ + 8 * g4_ + 4 * kb;                                                                   \
            if (nt_ == 0) {                                                                                                \
                const char* ep_ = reinterpret_cast<const char*>(&epi[o_epar][0][0]);                                       \
                u_bias = *reinterpret_cast<const float4*>(ep_ + cl_ * 4);                                                  \
                u_s = *reinterpret_cast<const float4*>(ep_ + TN * 4 + cl_ * 4);                                            \
                u_sh = *reinterpret_cast<const float4*>(ep_ + 2 * TN * 4 + cl_ * 4);                                       \
            }                                                                                                              \
            const int yl_ = o_y0 + wrow + nt_, xl_ = o_x0 + li;                                                            \
            float bb_[4] = {u_bias.x, u_bias.y, u_bias.z, u_bias.w};                                                       \
            if (p.border_corr != nullptr) {                                                                                \
                const int bm_ = (yl_ == 0 ? 1 : 0) | (yl_ == p.H - 1 ? 2 : 0) | (xl_ == 0 ? 4 : 0) | (xl_ == p.W - 1 ? 8 : 0); \
                if (__any(bm_ != 0)) {                                                                                     \
                    const float4 c_ = *reinterpret_cast<const float4*>(p.border_corr + (size_t)bm_ * p.Cout + o_n0 + cl_); \
                    bb_[0] -= c_.x; bb_[1] -= c_.y; bb_[2] -= c_.z; bb_[3] -= c_.w;                                        \
                }                                                                                                          \
            }                                                                                                              \
            const float ss_[4] = {u_s.x, u_s.y, u_s.z, u_s.w}, tt_[4] = {u_sh.x, u_sh.y, u_sh.z, u_sh.w};                   \
            float v_[4];                                                                                                   \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                \
                const float t_ = fmaf(accO[mt_][nt_][4 * g4_ + k], p.acc_scale, bb_[k]);                                   \
                v_[k] = fmaf(fmaxf(t_, 0.f), ss_[k], tt_[k]);                                                              \
            }                                                                                                              \
            uint2 ph_, plo_;                                                                                               \
            lm_split4(v_[0], v_[1], v_[2], v_[3], &ph_, &plo_);                                                            \
            {                                                                                                              \
                const auto r0_ = __builtin_amdgcn_permlane32_swap(ph_.x, plo_.x, false, false);                            \
                const auto r1_ = __builtin_amdgcn_permlane32_swap(ph_.y, plo_.y, false, false);                            \
                const uint4 val_ = {(unsigned)r0_[0], (unsigned)r1_[0], (unsigned)r0_[1], (unsigned)r1_[1]};               \
                if (o_b < p.B && yl_ < p.H) {                                                                              \
                    char* dst_ = p.out + ((((size_t)o_b * p.H + yl_) * p.W + xl_) * p.out_cstride + p.out_coff + o_n0) * 4 + (4 * mt_ + g4_) * 32 + kb * 16; \
                    *reinterpret_cast<uint4*>(dst_) = val_;                                                                \
                }                                                                                                          \
            }                                                                                                              \
            if (POOLT) {                                                                                                   \
                if ((nt_ & 1) == 0) {                                                                                      \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) u_prev[k] = v_[k];                                       \
                } else {                                                                                                   \
                    float q_[4];                                                                                           \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                        \
                        const float pl_ = u_prev[k] + v_[k];                                                               \
                        q_[k] = 0.25f * (pl_ + lm_lane_xor1(pl_));                                                         \
                    }                                                                                                      \
                    lm_split4(q_[0], q_[1], q_[2], q_[3], &ph_, &plo_);                                                    \
                    const auto r0_ = __builtin_amdgcn_permlane32_swap(ph_.x, plo_.x, false, false);                        \
                    const auto r1_ = __builtin_amdgcn_permlane32_swap(ph_.y, plo_.y, false, false);                        \
                    const uint4 val_ = {(unsigned)r0_[0], (unsigned)r1_[0], (unsigned)r0_[1], (unsigned)r1_[1]};           \
                    if ((li & 1) == 0 && o_b < p.B && yl_ < p.H) {                                                         \
                        char* pd_ = p.pool + ((((size_t)o_b * Hp + (yl_ >> 1)) * Wp + (xl_ >> 1)) * p.pool_cstride + p.pool_coff + o_n0) * 4 + (4 * mt_ + g4_) * 32 + kb * 16; \
                        *reinterpret_cast<uint4*>(pd_) = val_;                                                             \
                    }                                                                                                      \
                }                                                                                                          \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
    int it = blockIdx.x;
    int b, y0, x0, n0;
    while (it < n_items && !decode(it, b, y0, x0, n0)) it += gridDim.x;
    if (it >= n_items) return;
    item_voffs(b, y0, x0, voffC);
    int epar = 0;
    char* const buf0 = lds;
    char* const buf1 = lds + SM::BUF_BYTES;
    set_dma(false, b, n0, 0, 0, true, epar);
#pragma unroll
    for (int k = 0; k < N_SLOTS; ++k) dma_slot(k);
    lm_barrier_dma();

    // fragment sets: activations of one column offset = halo rows wrow .. wrow+5, hi and lo (12 fragments), double buffered;
    // weights of one tap = whi0 whi1 wlo0 wlo1, double buffered
    lm_h16x8 fa[2][12], fw[2][4];
#define W4_READS_A(S, AS, DX)                                                   \
    do {                                                                        \
        const int lo_ = a_off[DX] ^ 16;                                         \
        _Pragma("unroll") for (int h = 0; h < 6; ++h) {                         \
            LM_LDS_READ128(fa[S][2 * h], (AS) + a_off[DX], h * ROWB);           \
            LM_LDS_READ128(fa[S][2 * h + 1], (AS) + lo_, h * ROWB);             \
        }                                                                       \
    } while (0)
#define W4_READS_W(S, AS, TAP)                                                  \
    do {                                                                        \
        LM_LDS_READ128(fw[S][0], (AS) + w_off, (TAP) * (TN * 64));              \
        LM_LDS_READ128(fw[S][1], (AS) + w_off, (TAP) * (TN * 64) + 2048);       \
        LM_LDS_READ128(fw[S][2], (AS) + w_off_lo, (TAP) * (TN * 64));           \
        LM_LDS_READ128(fw[S][3], (AS) + w_off_lo, (TAP) * (TN * 64) + 2048);    \
    } while (0)
#define W4_WAIT_ALL()                                                                                                      \
    do {                                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < 12; ++q_) { asm volatile("" : "+v"(fa[0][q_]), "+v"(fa[1][q_])); }         \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) { asm volatile("" : "+v"(fw[0][q_]), "+v"(fw[1][q_])); }          \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)
    // 24 matrix instructions of tap (DY, column set SA, weight set SW), K0 = first of three DMA slots spread over them (or -1)
#define W4_GROUP(SA, SW, DY, PR)                                                                                            \
    do {                                                                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {                                                                 \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                             \
                acc[mt][nt] = lm_mfma_f32_32x32x16_f16(fw[SW][((PR) == 2 ? 2 : 0) + mt], fa[SA][2 * ((DY) + nt) + ((PR) == 1 ? 1 : 0)], acc[mt][nt]); \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
#define W4_SLOT(K, E)                                              \
    do {                                                           \
        if ((K) >= 0) dma_slot(K);                                 \
        if ((E) >= 0 && pending) { W5_EPI_UNIT(E); }               \
    } while (0)
#define W4_MFMAS(SA, SW, DY, K0, E0)                               \
    do {                                                           \
        W4_GROUP(SA, SW, DY, 0);                                   \
        W4_SLOT((K0) >= 0 ? (K0) : -1, (E0) >= 0 ? (E0) : -1);     \
        W4_GROUP(SA, SW, DY, 1);                                   \
        W4_SLOT((K0) >= 0 ? (K0) + 1 : -1, (E0) >= 0 ? (E0) + 1 : -1); \
        W4_GROUP(SA, SW, DY, 2);                                   \
        W4_SLOT((K0) >= 0 ? (K0) + 2 : -1, (E0) >= 0 ? (E0) + 2 : -1); \
        __builtin_amdgcn_sched_barrier(0);                         \
    } while (0)

    while (true) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        int nit = it + gridDim.x;
        int nb = 0, ny0 = 0, nx0 = 0, nn0 = 0;
        while (nit < n_items && !decode(nit, nb, ny0, nx0, nn0)) nit += gridDim.x;
        const bool have_next = nit < n_items;
        item_voffs(nb, ny0, nx0, voffN);
        // chunk 0 is resident in buffer 0; first fragments
        W4_READS_A(0, buf0, 0);
        W4_READS_W(0, buf0, 0);
        // Register-set parities: a chunk has 9 taps and 3 column groups, both odd, so the weight set of (chunk c, tap t) is
        // (c + t) & 1 and the activation set of (chunk c, column dx) is (c + dx) & 1 -- the set a prefetch writes is never the one
        // the running matrix instructions read.  Chunks are unrolled in pairs (the chunk count is even) to keep every index static.
#define W4_TAP(PAR, T, AS, AN, LAST, EPI)                                                                                       \
    do {                                                                                                                   \
        constexpr int dxi_ = (T) / 3, dyi_ = (T) - 3 * dxi_;                                                               \
        W4_WAIT_ALL(); /* every read issued during the previous tap has landed (it had 768 cycles) */                     \
        if ((T) == 8) {                                                                                                    \
            /* all reads of this chunk are done; the DMAs of the next chunk (issued during taps 0..6) must have landed */ \
            lm_barrier_dma();                                                                                              \
            if (!(LAST)) {                                                                                                 \
                W4_READS_A(((PAR) + 1) & 1, AN, 0);                                                                        \
                W4_READS_W(((PAR) + 9) & 1, AN, 0);                                                                        \
            }                                                                                                              \
        } else {                                                                                                           \
            if (dyi_ == 0 && dxi_ < 2) W4_READS_A(((PAR) + dxi_ + 1) & 1, AS, dxi_ + 1);                                   \
            constexpr int tn_ = (T) + 1, ndx_ = tn_ / 3, ndy_ = tn_ - 3 * ndx_;                                            \
            W4_READS_W(((PAR) + tn_) & 1, AS, 3 * ndy_ + ndx_);                                                            \
        }                                                                                                                  \
        W4_MFMAS(((PAR) + dxi_) & 1, ((PAR) + (T)) & 1, dyi_, (T) < 7 ? 3 * (T) : -1, (EPI) > 0 ? ((EPI) - 1) * 27 + 3 * (T) : -1); \
    } while (0)
#define W4_CHUNK(PAR, AS, AN, LAST, EPI)    \
    do {                               \
        W4_TAP(PAR, 0, AS, AN, LAST, EPI);  \
        W4_TAP(PAR, 1, AS, AN, LAST, EPI);  \
        W4_TAP(PAR, 2, AS, AN, LAST, EPI);  \
        W4_TAP(PAR, 3, AS, AN, LAST, EPI);  \
        W4_TAP(PAR, 4, AS, AN, LAST, EPI);  \
        W4_TAP(PAR, 5, AS, AN, LAST, EPI);  \
        W4_TAP(PAR, 6, AS, AN, LAST, EPI);  \
        W4_TAP(PAR, 7, AS, AN, LAST, EPI);  \
        W4_TAP(PAR, 8, AS, AN, LAST, EPI);  \
    } while (0)
        {   // first pair of chunks: carries the units of the PREVIOUS item's epilogue (54 slots for 32 units)
            set_dma(false, b, n0, KC, 1, true, -1);
            W4_CHUNK(0, buf0, buf1, false, 1);
            const bool last = 2 >= nchunks;
            if (!last) set_dma(false, b, n0, 2 * KC, 0, true, -1);
            else set_dma(true, nb, nn0, 0, 0, have_next, epar ^ 1);
            W4_CHUNK(1, buf1, buf0, last, 2);
        }
        for (int ci = 2; ci < nchunks; ci += 2) {
            set_dma(false, b, n0, (ci + 1) * KC, 1, true, -1);
            W4_CHUNK(0, buf0, buf1, false, 0);
            const bool last = ci + 2 >= nchunks;
            if (!last) set_dma(false, b, n0, (ci + 2) * KC, 0, true, -1);
            else set_dma(true, nb, nn0, 0, 0, have_next, epar ^ 1);  // the next item's chunk 0 + epilogue constants
            W4_CHUNK(1, buf1, buf0, last, 0);
        }
        // ---- this item's epilogue is DEFERRED: its accumulators move to the second set, its units run inside the next item's
        // first two chunks (or right here when there is no next item)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) accO[i][j] = acc[i][j];
        o_b = b;
        o_y0 = y0;
        o_x0 = x0;
        o_n0 = n0;
        o_epar = epar;
        pending = true;
        if (!have_next) break;
        it = nit;
        b = nb;
        y0 = ny0;
        x0 = nx0;
        n0 = nn0;
#pragma unroll
        for (int j = 0; j < A_PER_WAVE; ++j) voffC[j] = voffN[j];
        epar ^= 1;
    }
    // the last item's epilogue, serially
#define W5_U4(U) W5_EPI_UNIT(U); W5_EPI_UNIT((U) + 1); W5_EPI_UNIT((U) + 2); W5_EPI_UNIT((U) + 3)
    W5_U4(0); W5_U4(4); W5_U4(8); W5_U4(12); W5_U4(16); W5_U4(20); W5_U4(24); W5_U4(28);
}

static hipError_t launch_w5(const ConvParamsH3& p, hipStream_t stream, bool = false) {
    const int n_ptiles = (p.W / 32) * ((p.H + TH - 1) / TH) * p.B;
    const int n_ct = p.Cout / TN;
    const int xcd_order = (n_ct >= 2 && n_ptiles >= 64) ? 1 : 0;
    const int n_items = xcd_order ? 8 * ((n_ptiles + 7) / 8) * n_ct : n_ptiles * n_ct;
    const unsigned blocks = (unsigned)std::min(n_items, 256);
    if (p.pool) hipLaunchKernelGGL((conv_igemm_w5<true>), dim3(blocks), dim3(256), 0, stream, p, n_ptiles, n_items, xcd_order);
    else hipLaunchKernelGGL((conv_igemm_w5<false>), dim3(blocks), dim3(256), 0, stream, p, n_ptiles, n_items, xcd_order);
    return hipGetLastError();
}

}  // namespace lm

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
        __builtin_memcpy(p + g * 32, h, 16);
        __builtin_memcpy(p + g * 32 + 16, l, 16);
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

static double max_diff(const std::vector<unsigned char>& a, const std::vector<unsigned char>& b) {  // split tensors: groups of 8 hi + 8 lo halves
    double m = 0;
    for (size_t g = 0; g + 32 <= a.size(); g += 32) {
        _Float16 ha[16], hb[16];
        memcpy(ha, &a[g], 32);
        memcpy(hb, &b[g], 32);
        for (int k = 0; k < 8; ++k) m = std::max(m, std::fabs(((double)ha[k] + (double)ha[8 + k]) - ((double)hb[k] + (double)hb[8 + k])));
    }
    return m;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 20;
    const int reps = argc > 2 ? atoi(argv[2]) : 6;
    struct Shape { int H, Cin, Cout; bool pool; };
    const Shape shapes[] = {{256, 64, 64, true},   {128, 64, 128, false},  {128, 128, 128, true}, {64, 128, 256, false}, {64, 256, 256, true},
                            {32, 256, 512, false}, {32, 512, 512, true},   {32, 1024, 512, false}, {64, 512, 256, false}, {128, 256, 128, false},
                            {256, 128, 64, false}};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    char* zeros = nullptr;
    CK(hipMalloc(&zeros, 256));
    CK(hipMemset(zeros, 0, 256));
    printf("%-22s %10s %10s %8s | %s\n", "layer", "h3p ms", "w4 ms", "w4/h3p", "max |diff| out / pool (values ~ +-10)");
    double th = 0, tw = 0;
    for (const Shape& s : shapes) {
        const size_t npx = (size_t)B * s.H * s.H;
        char *in, *out, *out2, *w, *pool = nullptr, *pool2 = nullptr;
        float *bias, *bs, *bt, *corr;
        const size_t ob = npx * s.Cout * 4, pb = s.pool ? npx / 4 * s.Cout * 4 : 0;
        CK(hipMalloc(&in, npx * s.Cin * 4));
        CK(hipMalloc(&out, ob));
        CK(hipMalloc(&out2, ob));
        CK(hipMalloc(&w, (size_t)9 * s.Cout * s.Cin * 4));
        if (s.pool) { CK(hipMalloc(&pool, pb)); CK(hipMalloc(&pool2, pb)); }
        CK(hipMalloc(&bias, s.Cout * 4)); CK(hipMalloc(&bs, s.Cout * 4)); CK(hipMalloc(&bt, s.Cout * 4)); CK(hipMalloc(&corr, 16 * s.Cout * 4));
        fill_split<<<1024, 256>>>(in, npx * s.Cin / 8, 1u);
        fill_split<<<1024, 256>>>(w, (size_t)9 * s.Cout * s.Cin / 8, 2u);
        fill_f32<<<4, 256>>>(bias, s.Cout, -0.1f, 0.1f, 3u);
        fill_f32<<<4, 256>>>(bs, s.Cout, 0.75f, 1.25f, 4u);
        fill_f32<<<4, 256>>>(bt, s.Cout, -0.1f, 0.1f, 5u);
        fill_f32<<<16, 256>>>(corr, 16 * (size_t)s.Cout, -0.05f, 0.05f, 6u);
        CK(hipMemset(corr, 0, s.Cout * 4));
        lm::ConvParamsH3 p{};
        p.in = in; p.in_cstride = s.Cin; p.in_coff = 0; p.w = w; p.acc_scale = 1.f; p.bias = bias; p.bn_s = bs; p.bn_t = bt;
        p.out = out; p.out_cstride = s.Cout; p.out_coff = 0; p.pool = pool; p.pool_cstride = s.Cout; p.pool_coff = 0; p.zeros = zeros;
        p.B = B; p.H = s.H; p.W = s.H; p.Cin = s.Cin; p.Cout = s.Cout; p.border_corr = getenv("W4_NO_BORDER") ? nullptr : corr;
        p.stream_out = !getenv("W4_NO_NT") && npx * s.Cout * 4.0 > 128.0 * 1048576.0;
        lm::ConvParamsH3 q = p;
        q.out = out2; q.pool = pool2;
        CK(hipMemset(out, 0xee, ob)); CK(hipMemset(out2, 0xdd, ob));
        CK(lm::launch_conv3x3_h3(p, 0));
        const bool direct = getenv("W4_DIRECT") != nullptr;
        CK(lm::launch_w5(q, 0, direct));
        CK(hipDeviceSynchronize());
        std::vector<unsigned char> o1(ob), o2(ob), p1(pb), p2(pb);
        CK(hipMemcpy(o1.data(), out, ob, hipMemcpyDeviceToHost));
        CK(hipMemcpy(o2.data(), out2, ob, hipMemcpyDeviceToHost));
        if (pb) { CK(hipMemcpy(p1.data(), pool, pb, hipMemcpyDeviceToHost)); CK(hipMemcpy(p2.data(), pool2, pb, hipMemcpyDeviceToHost)); }
        const double dmax = max_diff(o1, o2), dpool = pb ? max_diff(p1, p2) : 0.0;
        float ms[2];
        for (int which = 0; which < 2; ++which) {
            for (int i = 0; i < 2; ++i) CK(which ? lm::launch_w5(q, 0, direct) : lm::launch_conv3x3_h3(p, 0));
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) CK(which ? lm::launch_w5(q, 0, direct) : lm::launch_conv3x3_h3(p, 0));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[which], e0, e1));
            ms[which] /= reps;
        }
        char name[64];
        snprintf(name, sizeof name, "H%d_Ci%d_Co%d%s", s.H, s.Cin, s.Cout, s.pool ? "+pool" : "");
        printf("%-22s %10.4f %10.4f %8.3f | %.3e / %.3e\n", name, ms[0], ms[1], ms[1] / ms[0], dmax, dpool);
        th += ms[0]; tw += ms[1];
        (void)hipFree(in); (void)hipFree(out); (void)hipFree(out2); (void)hipFree(w); if (pool) { (void)hipFree(pool); (void)hipFree(pool2); }
        (void)hipFree(bias); (void)hipFree(bs); (void)hipFree(bt); (void)hipFree(corr);
    }
    printf("sum: h3p %.3f ms, w4 %.3f ms (%.3f)\n", th, tw, tw / th);
    return 0;
}
