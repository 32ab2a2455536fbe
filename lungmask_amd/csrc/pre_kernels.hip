// Pre-processing kernels: replace lungmask/utils.py preprocess (:32-52),
// simple_bodymask (:55-82), crop_and_resize (:85-111), reshape_mask (:114-129)
// and the HU normalisation of mask.py:167-168.  Integer results are bit-exact
// against scipy.ndimage / skimage semantics:
//
//  * bodymask_bbox_kernel: one 128-thread workgroup per slice, thread = one row
//    of the 128x128 work image, rows bit-packed (128 bit) in LDS.  nearest zoom
//    -> threshold -> closing -> fill-holes -> 2x erosion -> largest 4-connected
//    component -> 2x dilation -> (implicit) nearest zoom back -> bbox of the
//    first 8-connected region.  The zoom back to full resolution is a pull-back
//    through monotone index maps, so the connected components of the full-res
//    mask are those of the 128^2 mask restricted to the source rows/columns
//    that are actually hit: everything stays in the 2 KiB bit image.
//  * resample_norm_kernel: ndimage.zoom(order=1) of the cropped slice to
//    256x256 with float64 coordinates/weights, round-half-away to int16, then
//    (x+1024)/1624 in float64 -> float32.
//  * reshape_mask_kernel: ndimage.zoom(order=0) of the label slice to the bbox
//    size and paste into the zeroed full-size slice.
//
// Compile with -ffp-contract=off: the float64 expressions must not be fused.
#include "pre_kernels.h"

namespace lm {

namespace {

struct U128 {
    unsigned long long lo, hi;
};
__device__ __forceinline__ U128 mk(unsigned long long lo, unsigned long long hi) { U128 r; r.lo = lo; r.hi = hi; return r; }
__device__ __forceinline__ U128 operator|(U128 a, U128 b) { return mk(a.lo | b.lo, a.hi | b.hi); }
__device__ __forceinline__ U128 operator&(U128 a, U128 b) { return mk(a.lo & b.lo, a.hi & b.hi); }
__device__ __forceinline__ U128 operator~(U128 a) { return mk(~a.lo, ~a.hi); }
__device__ __forceinline__ bool nz(U128 a) { return (a.lo | a.hi) != 0ull; }
__device__ __forceinline__ bool eq(U128 a, U128 b) { return a.lo == b.lo && a.hi == b.hi; }
// bit i = column i; shl moves towards higher column index
template <int N>
__device__ __forceinline__ U128 shl(U128 a) {
    if constexpr (N == 64) return mk(0ull, a.lo);
    else return mk(a.lo << N, (a.hi << N) | (a.lo >> (64 - N)));
}
template <int N>
__device__ __forceinline__ U128 shr(U128 a) {
    if constexpr (N == 64) return mk(a.hi, 0ull);
    else return mk((a.lo >> N) | (a.hi << (64 - N)), a.hi >> N);
}
__device__ __forceinline__ int popc(U128 a) { return __popcll(a.lo) + __popcll(a.hi); }
__device__ __forceinline__ int first_bit(U128 a) { return a.lo ? __ffsll((long long)a.lo) - 1 : 64 + __ffsll((long long)a.hi) - 1; }
__device__ __forceinline__ int last_bit(U128 a) { return a.hi ? 127 - __clzll((long long)a.hi) : 63 - __clzll((long long)a.lo); }
__device__ __forceinline__ U128 bit(int i) { return i < 64 ? mk(1ull << i, 0ull) : mk(0ull, 1ull << (i - 64)); }

// flood `s` along the runs of `m` inside one row (both directions), Kogge-Stone
__device__ __forceinline__ U128 hfill(U128 s, U128 m) {
    U128 g = s & m, p = m;
    g = g | (p & shl<1>(g));  p = p & shl<1>(p);
    g = g | (p & shl<2>(g));  p = p & shl<2>(p);
    g = g | (p & shl<4>(g));  p = p & shl<4>(p);
    g = g | (p & shl<8>(g));  p = p & shl<8>(p);
    g = g | (p & shl<16>(g)); p = p & shl<16>(p);
    g = g | (p & shl<32>(g)); p = p & shl<32>(p);
    g = g | (p & shl<64>(g));
    p = m;
    g = g | (p & shr<1>(g));  p = p & shr<1>(p);
    g = g | (p & shr<2>(g));  p = p & shr<2>(p);
    g = g | (p & shr<4>(g));  p = p & shr<4>(p);
    g = g | (p & shr<8>(g));  p = p & shr<8>(p);
    g = g | (p & shr<16>(g)); p = p & shr<16>(p);
    g = g | (p & shr<32>(g)); p = p & shr<32>(p);
    g = g | (p & shr<64>(g));
    return g;
}

constexpr int G = 128;  // work-image side (utils.py:68)

struct Planes {
    U128 a[G], t[G], s[G], best[G], rem[G];
};

__device__ __forceinline__ U128 row_or_zero(const U128* pl, int r) { return (r >= 0 && r < G) ? pl[r] : mk(0ull, 0ull); }

// binary_dilation / binary_erosion with the 4-connected cross, border_value = 0 (in place on plane `a`)
__device__ void dilate_cross(U128* a, U128* tmp, int r) {
    const U128 c = a[r];
    tmp[r] = c | row_or_zero(a, r - 1) | row_or_zero(a, r + 1) | shl<1>(c) | shr<1>(c);
    __syncthreads();
    a[r] = tmp[r];
    __syncthreads();
}
__device__ void erode_cross(U128* a, U128* tmp, int r) {
    const U128 c = a[r];
    tmp[r] = c & row_or_zero(a, r - 1) & row_or_zero(a, r + 1) & shl<1>(c) & shr<1>(c);
    __syncthreads();
    a[r] = tmp[r];
    __syncthreads();
}

// Flood the seeds in plane `s` through mask plane `m` (4- or 8-connected).  All 128 threads participate.
template <bool CONN8>
__device__ void flood(U128* s, const U128* m, int r, int* flag) {
    s[r] = hfill(s[r], m[r]);
    __syncthreads();
    for (;;) {
        if (r == 0) *flag = 0;
        __syncthreads();
        U128 n = row_or_zero(s, r - 1) | row_or_zero(s, r + 1);
        if (CONN8) n = n | shl<1>(n) | shr<1>(n);
        const U128 cur = s[r];
        const U128 add = n & m[r] & ~cur;
        U128 nw = cur;
        if (nz(add)) {
            nw = hfill(cur | add, m[r]);
            *flag = 1;
        }
        __syncthreads();
        s[r] = nw;
        const int again = *flag;
        __syncthreads();
        if (!again) break;
    }
}

__device__ __forceinline__ double zoom_factor(int in, int out) { return out > 1 ? (double)(in - 1) / (double)(out - 1) : 1.0; }

// nearest-neighbour source index of ndimage.zoom(order=0): -1 when the coordinate falls outside [0, in-1]
__device__ __forceinline__ int nn_index(int o, double zf, int in) {
    const double cc = (double)o * zf;
    if (cc < 0.0 || cc > (double)(in - 1)) return -1;
    return (int)floor(cc + 0.5);
}

template <class T>
__device__ __forceinline__ bool above_threshold(T v) { return (double)v > -500.0; }

}  // namespace

template <class T>
__global__ __launch_bounds__(128) void bodymask_bbox_kernel(BodyMaskParams p) {
    __shared__ Planes pl;
    __shared__ int sh_flag, sh_first, sh_area, sh_best_area;
    __shared__ unsigned long long used_r[2], used_c[2];
    __shared__ int rmin[G], rmax[G], cmin[G], cmax[G], rowmap[G], colmap[G];
    __shared__ int bb[4];

    const int r = threadIdx.x;  // row of the work image
    const int lane = r & 63, wave = r >> 6;
    const int z = blockIdx.x;
    const T* __restrict__ src = reinterpret_cast<const T*>(p.vol) + (size_t)z * p.H * p.W;

    // ---- 1. ndimage.zoom(img, 128/shape, order=0) > -500   (out-of-range -> cval 0 -> True)
    {
        const double zr = zoom_factor(p.H, G), zc = zoom_factor(p.W, G);
        const int c = 64 * wave + lane;
        const int sc = nn_index(c, zc, p.W);
        // 32 rows per round, every load of a round in flight together (always from a valid address, selected afterwards): one
        // dependent load per row was 128 serial memory latencies -- the whole 0.24 ms this kernel took per volume
        constexpr int RB = 32;
        const size_t col = (size_t)(sc >= 0 ? sc : 0);
        for (int r0 = 0; r0 < G; r0 += RB) {
            T v[RB];
            int srs[RB];
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                srs[k] = nn_index(r0 + k, zr, p.H);
                v[k] = src[(size_t)(srs[k] >= 0 ? srs[k] : 0) * p.W + col];
            }
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const bool on = (srs[k] >= 0 && sc >= 0) ? above_threshold(v[k]) : true;  // out of range -> cval 0 > -500
                const unsigned long long m = __ballot(on);
                if (lane == 0) {
                    if (wave == 0) pl.a[r0 + k].lo = m; else pl.a[r0 + k].hi = m;
                }
            }
        }
    }
    __syncthreads();
    // ---- 2. binary_closing (cross, 1 iteration)
    dilate_cross(pl.a, pl.t, r);
    erode_cross(pl.a, pl.t, r);
    // ---- 3. binary_fill_holes(structure=ones(3,3)): 8-connected flood of the background from outside the image
    {
        const U128 bg = ~pl.a[r];
        pl.t[r] = bg;
        pl.s[r] = (r == 0 || r == G - 1) ? bg : (bg & (bit(0) | bit(G - 1)));
        __syncthreads();
        flood<true>(pl.s, pl.t, r, &sh_flag);
        pl.a[r] = ~pl.s[r];
        __syncthreads();
    }
    // ---- 4. binary_erosion(iterations=2)
    erode_cross(pl.a, pl.t, r);
    erode_cross(pl.a, pl.t, r);
    // ---- 5. largest 4-connected region; first one (raster order of first pixel) on equal areas (np.argmax)
    pl.rem[r] = pl.a[r];
    pl.best[r] = mk(0ull, 0ull);
    if (r == 0) sh_best_area = 0;
    __syncthreads();
    for (;;) {
        if (r == 0) { sh_first = G; sh_area = 0; }
        __syncthreads();
        if (nz(pl.rem[r])) atomicMin(&sh_first, r);
        __syncthreads();
        const int fr = sh_first;
        if (fr >= G) break;
        pl.s[r] = (r == fr) ? bit(first_bit(pl.rem[r])) : mk(0ull, 0ull);
        __syncthreads();
        flood<false>(pl.s, pl.a, r, &sh_flag);
        const U128 comp = pl.s[r];
        const int cnt = popc(comp);
        if (cnt) atomicAdd(&sh_area, cnt);
        __syncthreads();
        const bool better = sh_area > sh_best_area;
        __syncthreads();
        if (better) {
            pl.best[r] = comp;
            if (r == 0) sh_best_area = sh_area;
        }
        pl.rem[r] = pl.rem[r] & ~comp;
        __syncthreads();
    }
    // ---- 6. keep it and binary_dilation(iterations=2)   (no region: mask stays empty)
    pl.a[r] = pl.best[r];
    __syncthreads();
    if (sh_best_area > 0) {
        dilate_cross(pl.a, pl.t, r);
        dilate_cross(pl.a, pl.t, r);
    }
    // ---- 7. ndimage.zoom(mask, shape/128, order=0) as index maps; which source rows/cols are hit, and by whom
    const double zr = zoom_factor(G, p.H), zc = zoom_factor(G, p.W);
    rmin[r] = 0x7fffffff; rmax[r] = -1; cmin[r] = 0x7fffffff; cmax[r] = -1;
    if (r < 2) { used_r[r] = 0ull; used_c[r] = 0ull; }
    __syncthreads();
    for (int y = r; y < p.H; y += G) {
        const int sr = nn_index(y, zr, G);
        if (sr >= 0) { atomicMin(&rmin[sr], y); atomicMax(&rmax[sr], y); atomicOr(&used_r[sr >> 6], 1ull << (sr & 63)); }
    }
    for (int x = r; x < p.W; x += G) {
        const int sc = nn_index(x, zc, G);
        if (sc >= 0) { atomicMin(&cmin[sc], x); atomicMax(&cmax[sc], x); atomicOr(&used_c[sc >> 6], 1ull << (sc & 63)); }
    }
    __syncthreads();
    if (p.bmask != nullptr) {  // test seam: the full-resolution body mask of utils.py:82
        uint8_t* dst = p.bmask + (size_t)z * p.H * p.W;
        for (int y = 0; y < p.H; ++y) {
            const int sr = nn_index(y, zr, G);
            for (int x = r; x < p.W; x += G) {
                const int sc = nn_index(x, zc, G);
                uint8_t v = 0;
                if (sr >= 0 && sc >= 0) v = nz(pl.a[sr] & bit(sc)) ? 1 : 0;
                dst[(size_t)y * p.W + x] = v;
            }
        }
    }
    // compressed image M'[k][l] = M[rowmap[k]][colmap[l]] over the used rows / columns
    const U128 ur = mk(used_r[0], used_r[1]), uc = mk(used_c[0], used_c[1]);
    {
        const U128 below = (r == 0) ? mk(0ull, 0ull) : (r <= 64 ? mk(r == 64 ? ~0ull : ((1ull << r) - 1), 0ull) : mk(~0ull, (1ull << (r - 64)) - 1));
        if (nz(ur & bit(r))) rowmap[popc(ur & below)] = r;
        if (nz(uc & bit(r))) colmap[popc(uc & below)] = r;
    }
    __syncthreads();
    const int nR = popc(ur), nC = popc(uc);
    {
        U128 row = mk(0ull, 0ull);
        if (r < nR) {
            const U128 srcrow = pl.a[rowmap[r]];
            for (int l = 0; l < nC; ++l)
                if (nz(srcrow & bit(colmap[l]))) row = row | bit(l);
        }
        pl.t[r] = row;
    }
    __syncthreads();
    // ---- 8. skimage.measure.label(bmask) (8-connected) -> regionprops[0].bbox: the component of the first pixel
    if (r == 0) sh_first = G;
    __syncthreads();
    if (nz(pl.t[r])) atomicMin(&sh_first, r);
    __syncthreads();
    const int fr = sh_first;
    if (fr >= G) {
        if (r == 0) {
            int* o = p.bbox + 4 * (size_t)z;
            o[0] = 0; o[1] = 0; o[2] = p.H; o[3] = p.W;  // utils.py:106
        }
        return;
    }
    pl.s[r] = (r == fr) ? bit(first_bit(pl.t[r])) : mk(0ull, 0ull);
    __syncthreads();
    flood<true>(pl.s, pl.t, r, &sh_flag);
    if (r == 0) { bb[0] = G; bb[1] = G; bb[2] = -1; bb[3] = -1; }
    __syncthreads();
    if (nz(pl.s[r])) {
        atomicMin(&bb[0], r);
        atomicMax(&bb[2], r);
        atomicMin(&bb[1], first_bit(pl.s[r]));
        atomicMax(&bb[3], last_bit(pl.s[r]));
    }
    __syncthreads();
    if (r == 0) {
        int* o = p.bbox + 4 * (size_t)z;
        o[0] = rmin[rowmap[bb[0]]];
        o[1] = cmin[colmap[bb[1]]];
        o[2] = rmax[rowmap[bb[2]]] + 1;
        o[3] = cmax[colmap[bb[3]]] + 1;
    }
}

// ---------------------------------------------------------------------------------------------
// ndimage.zoom(order=1) + normalisation.  One thread per output pixel.
template <class T>
__global__ __launch_bounds__(256) void resample_norm_kernel(ResampleParams p) {
    const size_t total = (size_t)p.N * p.OH * p.OW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % p.OW);
        const int oy = (int)((idx / p.OW) % p.OH);
        const int z = (int)(idx / ((size_t)p.OW * p.OH));
        const int* bb = p.bbox + 4 * (size_t)z;
        const int y0 = bb[0], x0 = bb[1], h = bb[2] - bb[0], w = bb[3] - bb[1];
        const T* __restrict__ src = reinterpret_cast<const T*>(p.vol) + (size_t)z * p.H * p.W;
        const double cr = (double)oy * zoom_factor(h, p.OH), cc = (double)ox * zoom_factor(w, p.OW);
        double v = 0.0;
        if (!(cr < 0.0 || cr > (double)(h - 1) || cc < 0.0 || cc > (double)(w - 1))) {
            const int sr = (int)floor(cr), sc = (int)floor(cc);
            const double tr = cr - (double)sr, tc = cc - (double)sc;
            const double wr0 = 1.0 - tr, wr1 = 1.0 - wr0, wc0 = 1.0 - tc, wc1 = 1.0 - wc0;
            const int sr1 = min(sr + 1, h - 1), sc1 = min(sc + 1, w - 1);
            auto at = [&](int yy, int xx) -> double {
                double a = (double)src[(size_t)(y0 + yy) * p.W + (x0 + xx)];
                a = a < -1024.0 ? -1024.0 : (a > 600.0 ? 600.0 : a);  // np.clip(-1024, 600), utils.py:45
                return a;
            };
            double acc = 0.0;
            acc += (at(sr, sc) * wr0) * wc0;
            acc += (at(sr, sc1) * wr0) * wc1;
            acc += (at(sr1, sc) * wr1) * wc0;
            acc += (at(sr1, sc1) * wr1) * wc1;
            v = acc;
        }
        if (sizeof(T) == 4 && (T)0.5 != (T)0) {
            // float32 volume: zoom output is float32 (double result rounded once), then float32 arithmetic in numpy
            const float vf = (float)v;
            if (p.out_f32) p.out_f32[idx] = (vf + 1024.0f) / 1624.0f;
        } else if ((T)0.5 != (T)0) {
            // float64 volume: everything in float64, one final cast (mask.py:178-181)
            if (p.out_f32) p.out_f32[idx] = (float)((v + 1024.0) / 1624.0);
        } else {
            // integer output dtype: round half away from zero, then truncate (scipy ni_interpolation)
            const double rv = v > 0.0 ? v + 0.5 : v - 0.5;
            const int16_t q = (int16_t)(int)rv;
            if (p.out_i16) p.out_i16[idx] = q;
            // mask.py:167-168: clip at 600 (no-op here), (x + 1024) / 1624 in float64 -> float32 (mask.py:178-181)
            if (p.out_f32) p.out_f32[idx] = (float)((double)((int)q + 1024) / 1624.0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// reshape_mask (utils.py:114-129) for a stack of slices: ndimage.zoom(order=0) to the bbox size, paste.
__global__ __launch_bounds__(256) void reshape_mask_kernel(ReshapeParams p) {
    const size_t total = (size_t)p.N * p.H * p.W;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % p.W);
        const int y = (int)((idx / p.W) % p.H);
        const int z = (int)(idx / ((size_t)p.W * p.H));
        const int* bb = p.bbox + 4 * (size_t)z;
        uint8_t v = 0;
        if (y >= bb[0] && y < bb[2] && x >= bb[1] && x < bb[3]) {
            const int hh = bb[2] - bb[0], ww = bb[3] - bb[1];
            const int sr = nn_index(y - bb[0], zoom_factor(p.MH, hh), p.MH);
            const int sc = nn_index(x - bb[1], zoom_factor(p.MW, ww), p.MW);
            if (sr >= 0 && sc >= 0) v = p.mask[((size_t)z * p.MH + sr) * p.MW + sc];
        }
        p.out[idx] = v;
    }
}

// The same for rows that are a multiple of 4 long and at most RS_MAXW wide (the hot path: 512): the nearest-neighbour column map
// of a slice is computed once per workgroup (float64, as above) and shared by its RS_ROWS rows, a thread writes 4 voxels with one
// 32-bit store: bound by the 1 B/voxel written, not by three 64-bit divisions + two float64 maps + a byte store per voxel.
constexpr int RS_ROWS = 16, RS_MAXW = 2048;
__global__ __launch_bounds__(256) void reshape_mask_rows_kernel(ReshapeParams p) {
    __shared__ short cmap[RS_MAXW];
    __shared__ int rmap[RS_ROWS];
    const int z = blockIdx.y, y0 = blockIdx.x * RS_ROWS;
    const int* bb = p.bbox + 4 * (size_t)z;
    const int b0 = bb[0], b1 = bb[1], b2 = bb[2], b3 = bb[3];
    const double zr = zoom_factor(p.MH, b2 - b0), zc = zoom_factor(p.MW, b3 - b1);
    for (int x = threadIdx.x; x < p.W; x += 256) cmap[x] = (x >= b1 && x < b3) ? (short)nn_index(x - b1, zc, p.MW) : (short)-1;
    if (threadIdx.x < RS_ROWS) {
        const int y = y0 + threadIdx.x;
        rmap[threadIdx.x] = (y >= b0 && y < b2) ? nn_index(y - b0, zr, p.MH) : -1;
    }
    __syncthreads();
    const uint8_t* __restrict__ src = p.mask + (size_t)z * p.MH * p.MW;
    uint8_t* __restrict__ dst = p.out + ((size_t)z * p.H + y0) * p.W;
    const int quads = p.W >> 2, rows = min(RS_ROWS, p.H - y0);
    for (int i = threadIdx.x; i < rows * quads; i += 256) {
        const int r = i / quads, q = i - r * quads;
        const int sr = rmap[r];
        unsigned word = 0;
        if (sr >= 0) {
            const uint8_t* row = src + (size_t)sr * p.MW;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sc = cmap[4 * q + j];
                if (sc >= 0) word |= (unsigned)row[sc] << (8 * j);
            }
        }
        *reinterpret_cast<unsigned*>(dst + (size_t)r * p.W + 4 * q) = word;
    }
}

hipError_t launch_bodymask_bbox(const BodyMaskParams& p, hipStream_t stream) {
    if (p.N <= 0) return hipSuccess;
    switch (p.dtype) {
        case LM_I16: LM_LAUNCH((bodymask_bbox_kernel<int16_t>), dim3((unsigned)p.N), dim3(128), 0, stream, p); break;
        case LM_I32: LM_LAUNCH((bodymask_bbox_kernel<int32_t>), dim3((unsigned)p.N), dim3(128), 0, stream, p); break;
        case LM_I64: LM_LAUNCH((bodymask_bbox_kernel<int64_t>), dim3((unsigned)p.N), dim3(128), 0, stream, p); break;
        case LM_F32: LM_LAUNCH((bodymask_bbox_kernel<float>), dim3((unsigned)p.N), dim3(128), 0, stream, p); break;
        case LM_F64: LM_LAUNCH((bodymask_bbox_kernel<double>), dim3((unsigned)p.N), dim3(128), 0, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_resample_norm(const ResampleParams& p, hipStream_t stream) {
    if (p.N <= 0) return hipSuccess;
    const size_t total = (size_t)p.N * p.OH * p.OW;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
    switch (p.dtype) {
        case LM_I16: LM_LAUNCH((resample_norm_kernel<int16_t>), dim3(blocks), dim3(256), 0, stream, p); break;
        case LM_I32: LM_LAUNCH((resample_norm_kernel<int32_t>), dim3(blocks), dim3(256), 0, stream, p); break;
        case LM_I64: LM_LAUNCH((resample_norm_kernel<int64_t>), dim3(blocks), dim3(256), 0, stream, p); break;
        case LM_F32: LM_LAUNCH((resample_norm_kernel<float>), dim3(blocks), dim3(256), 0, stream, p); break;
        case LM_F64: LM_LAUNCH((resample_norm_kernel<double>), dim3(blocks), dim3(256), 0, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// one wave per image row: 16 bytes per lane and step, a ballot per step; rows with a label raise the extent
__global__ __launch_bounds__(256) void label_extent_init_kernel(int* ext, int N, int H) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ext[0] = N;
        ext[1] = 0;
        ext[2] = H;
        ext[3] = 0;
    }
}
__global__ __launch_bounds__(256) void label_extent_kernel(const uint8_t* __restrict__ vol, int N, int H, int W, int* ext) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long rows = (long long)N * H;
    int zlo = N, zhi = 0, ylo = H, yhi = 0;
    for (long long r = (long long)blockIdx.x * 4 + wv; r < rows; r += (long long)gridDim.x * 4) {
        const uint8_t* row = vol + (size_t)r * W;
        bool any = false;
        if ((W & 15) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
            for (int x = lane * 16; x < W; x += 64 * 16) {
                const uint4 v = *reinterpret_cast<const uint4*>(row + x);
                any |= (v.x | v.y | v.z | v.w) != 0u;
            }
        } else {
            for (int x = lane; x < W; x += 64) any |= row[x] != 0;
        }
        if (__any(any)) {
            const int z = (int)(r / H), y = (int)(r - (long long)z * H);
            zlo = min(zlo, z);
            zhi = max(zhi, z + 1);
            ylo = min(ylo, y);
            yhi = max(yhi, y + 1);
        }
    }
    if (lane == 0 && zhi > zlo) {
        atomicMin(&ext[0], zlo);
        atomicMax(&ext[1], zhi);
        atomicMin(&ext[2], ylo);
        atomicMax(&ext[3], yhi);
    }
}

hipError_t launch_label_extent(const uint8_t* vol, int N, int H, int W, int* ext_dev, hipStream_t stream) {
    LM_LAUNCH(label_extent_init_kernel, dim3(1), dim3(256), 0, stream, ext_dev, N, H);
    if (N <= 0 || H <= 0 || W <= 0) return hipGetLastError();
    const long long rows = (long long)N * H;
    const unsigned blocks = (unsigned)std::min<long long>((rows + 3) / 4, 256 * 16);
    LM_LAUNCH(label_extent_kernel, dim3(blocks), dim3(256), 0, stream, vol, N, H, W, ext_dev);
    return hipGetLastError();
}

hipError_t launch_reshape_mask(const ReshapeParams& p, hipStream_t stream) {
    if (p.N <= 0) return hipSuccess;
    if (p.W % 4 == 0 && p.W <= RS_MAXW && p.MW < 32768 && p.N < 65536 && (reinterpret_cast<uintptr_t>(p.out) & 3) == 0) {
        LM_LAUNCH(reshape_mask_rows_kernel, dim3((unsigned)((p.H + RS_ROWS - 1) / RS_ROWS), (unsigned)p.N), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    const size_t total = (size_t)p.N * p.H * p.W;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 32);
    LM_LAUNCH(reshape_mask_kernel, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// reorient: one output element per thread (grid-stride); writes are coalesced, reads follow the
// source strides (coalesced too whenever the fastest axis is kept, e.g. pure flips).
template <typename T>
__global__ void reorient_kernel(ReorientParams p) {
    const T* __restrict__ in = (const T*)p.in;
    T* __restrict__ out = (T*)p.out;
    const size_t total = (size_t)p.n0 * p.n1 * p.n2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int i2 = (int)(i % (size_t)p.n2);
        const size_t r = i / (size_t)p.n2;
        const int i1 = (int)(r % (size_t)p.n1);
        const int i0 = (int)(r / (size_t)p.n1);
        out[i] = in[p.base + i0 * p.s0 + i1 * p.s1 + i2 * p.s2];
    }
}

hipError_t launch_reorient(const ReorientParams& p, hipStream_t stream) {
    const size_t total = (size_t)p.n0 * p.n1 * p.n2;
    if (total == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 64);
    switch (p.elem) {
        case 1: LM_LAUNCH(reorient_kernel<uint8_t>, dim3(blocks), dim3(256), 0, stream, p); break;
        case 2: LM_LAUNCH(reorient_kernel<uint16_t>, dim3(blocks), dim3(256), 0, stream, p); break;
        case 4: LM_LAUNCH(reorient_kernel<uint32_t>, dim3(blocks), dim3(256), 0, stream, p); break;
        case 8: LM_LAUNCH(reorient_kernel<uint64_t>, dim3(blocks), dim3(256), 0, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lm
