// Volume post-processing kernels (integer/byte work, HBM-bound).
//
// Connected components: union-find on an int32 parent volume with atomicMin
// hooking (root = smallest linear index = first voxel in raster order, which is
// exactly the key skimage.measure.label numbers regions by).  26-connectivity
// needs the 13 raster-earlier neighbours, 6-connectivity 3.  Stale (cached)
// parent reads are harmless: parents only ever decrease along ancestor chains
// and every hook is an L2 atomic that re-validates "was still a root".
#include "post_kernels.h"

namespace lm {

namespace {

constexpr int TPB = 256;
constexpr int ITEMS = 16;  // consecutive voxels per thread in order-preserving passes
constexpr int BLOCK_VOX = TPB * ITEMS;

inline unsigned grid_for(size_t n, int per_block = TPB, unsigned cap = 256 * 32) {
    size_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (unsigned)std::min<size_t>(b, cap);
}

// (x, y, z) of a flat voxel index.  Every volume here has < 2^31 voxels (post_engine.hip refuses larger ones: parents are ints), so
// two 32-bit divisions do what three 64-bit ones (~100 instructions each on this ISA) did.
__device__ __forceinline__ void split3(size_t flat, int H, int W, int& x, int& y, int& z) {
    const unsigned f = (unsigned)flat, q = f / (unsigned)W;
    x = (int)(f - q * (unsigned)W);
    z = (int)(q / (unsigned)H);
    y = (int)(q - (unsigned)z * (unsigned)H);
}

__device__ __forceinline__ int find_root(const int* P, int x) {
    const volatile int* vp = P;
    int p;
    while ((p = vp[x]) != x) x = p;
    return x;
}

// find with path halving: every second node on the way up is re-pointed at its grandparent.  The store races with other walkers
// and with the hooking atomicMin, harmlessly: a non-root entry only ever receives (proper) ancestors of its node, so the forest
// stays a forest over the same components with parent < child, and a root entry (P[x] == x) is never written here.
__device__ __forceinline__ int find_root_halving(int* P, int x) {
    volatile int* vp = P;
    int p = vp[x];
    while (p != x) {
        const int g = vp[p];
        if (g != p) vp[x] = g;
        x = p;
        p = g;
    }
    return x;
}

__device__ __forceinline__ void unite(int* P, int a, int b) {
    for (;;) {
        a = find_root_halving(P, a);
        b = find_root_halving(P, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&P[a], b);
        if (old == a) return;  // a was still a root: hooked under b
        a = old;               // a had been re-parented meanwhile: keep joining its (old) parent with b
    }
}

// Initial labelling: every voxel points at the first voxel of its x-run inside its 64-voxel segment
// (one coalesced pass, no atomics): row-runs are connected before the first union.
__global__ __launch_bounds__(TPB) void ccl_init_runs_kernel(const uint8_t* __restrict__ lab, int* __restrict__ P, Dims d) {
    const size_t nvox = d.nvox();
    const size_t nseg = (nvox + 63) / 64;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t seg = wave; seg < nseg; seg += nwaves) {
        const size_t v = seg * 64 + lane;
        const int L = v < nvox ? (int)lab[v] : 0;
        const int prevL = __shfl_up(L, 1);
        const bool same = lane > 0 && L != 0 && prevL == L && (v % d.W) != 0;
        const unsigned long long heads = __ballot(!same);
        const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
        const int start = 63 - __clzll((long long)(heads & upto));
        if (v < nvox) P[v] = L ? (int)(seg * 64 + start) : -1;
    }
}

// Hook the row-runs together.  Only unions that are not already implied are issued:
//  * the x-neighbour only across a 64-voxel segment boundary (runs are pre-connected inside a segment);
//  * for each earlier row (y-1 in this slice; y-1, y, y+1 in the previous slice): the voxel straight
//    across unless the left neighbours of both are in the same runs (then the pair one step to the left
//    implies it); the diagonal ones only when the straight one differs and this voxel's own left/right
//    neighbour does not already carry the connection.
template <bool C26>
__global__ __launch_bounds__(TPB) void ccl_merge_kernel(const uint8_t* __restrict__ lab, int* P, Dims d) {
    const size_t nvox = d.nvox();
    const size_t HW = (size_t)d.H * d.W;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) {
        const uint8_t L = lab[v];
        if (!L) continue;
        int x, y, z;
        split3(v, d.H, d.W, x, y, z);
        const bool left_same = x > 0 && lab[v - 1] == L;
        const bool right_same = x + 1 < d.W && lab[v + 1] == L;
        if (left_same && (v & 63) == 0) unite(P, (int)v, (int)(v - 1));
        auto row = [&](size_t u) {  // u = index of the voxel straight across in an earlier row
            if (lab[u] == L) {
                if (!(left_same && lab[u - 1] == L)) unite(P, (int)v, (int)u);
            } else if (C26) {
                if (x > 0 && !left_same && lab[u - 1] == L) unite(P, (int)v, (int)(u - 1));
                if (x + 1 < d.W && !right_same && lab[u + 1] == L) unite(P, (int)v, (int)(u + 1));
            }
        };
        if (y > 0) row(v - d.W);
        if (z > 0) {
            row(v - HW);
            if (C26) {
                if (y > 0) row(v - HW - d.W);
                if (y + 1 < d.H) row(v - HW + d.W);
            }
        }
    }
}

// ---- the same labelling for rows whose length is a multiple of 4 (every volume of the hot path) ----------------------------
// One wave per 256-voxel piece of a row, FOUR voxels per lane from one 32-bit load per row involved; the voxels left and right of a
// lane's four come from the neighbour lanes' registers, (z, y) of the piece are wave-uniform: 1.25 loads and no division per voxel
// instead of ~12 byte loads and three 64-bit divisions.  The label volumes are mostly solid (a lung, or the background around
// one), so what matters is the number of unions that reach the (one) big tree:
//  * the initial labelling connects the runs inside a whole piece (not a 64-voxel segment), so a row of <= 256 voxels needs no
//    union along x at all;
//  * the two diagonal rows of the previous slice, (y-1, z-1) and (y+1, z-1), are skipped for a voxel whose neighbour straight
//    behind, b = (x, y, z-1), has its label: v ~ b is made by the straight row, and every voxel of those two rows that touches v
//    is an in-plane 8-neighbour of b, i.e. already joined to b by the previous slice's own in-plane unions;
//  * the row straight behind is skipped for a voxel whose neighbours above, (x, y-1, z), and above-behind, (x, y-1, z-1), both
//    have its label (see the kernel).
// A solid row then issues ONE union (with the row above) instead of 4 + one per 64 voxels.
__global__ __launch_bounds__(TPB) void ccl_init_rows_kernel(const uint8_t* __restrict__ lab, int* __restrict__ P, Dims d) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * (unsigned)TPB + threadIdx.x) >> 6, nwaves = (gridDim.x * (unsigned)TPB) >> 6;
    const unsigned ppr = ((unsigned)d.W + 255u) >> 8;  // pieces per row
    const unsigned npieces = (unsigned)d.N * (unsigned)d.H * ppr;
    for (unsigned piece = wave; piece < npieces; piece += nwaves) {
        const unsigned row = piece / ppr;
        const int x = (int)((piece - row * ppr) << 8) + lane * 4;
        const bool in = x < d.W;
        const int v = (int)row * d.W + x;
        const unsigned w = in ? *reinterpret_cast<const unsigned*>(lab + v) : 0u;
        const unsigned up1 = __shfl_up(w, 1);
        const unsigned prev = lane ? (up1 >> 24) : 0u;  // a piece starts a run
        // position (0..255 in the piece) of the last run head among this lane's four voxels; voxel 0 of lane 0 is one
        int L[4], last = -1;
        bool head[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            L[j] = (int)((w >> (8 * j)) & 0xffu);
            const int before = j ? L[j - 1] : (int)prev;
            head[j] = !(L[j] != 0 && before == L[j]);
            if (head[j]) last = lane * 4 + j;
        }
        int carry = __shfl_up(last, 1);  // exclusive running maximum over the lanes before this one
        if (lane == 0) carry = -1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(carry, off);
            if (lane >= off) carry = max(carry, o);
        }
        if (in) {
            int cur = carry, out[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (head[j]) cur = lane * 4 + j;
                out[j] = L[j] ? (v - lane * 4 + cur) : -1;
            }
            int4 o4;
            o4.x = out[0], o4.y = out[1], o4.z = out[2], o4.w = out[3];
            *reinterpret_cast<int4*>(P + v) = o4;
        }
    }
}

template <bool C26>
__global__ __launch_bounds__(TPB) void ccl_merge_rows_kernel(const uint8_t* __restrict__ lab, int* P, Dims d) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * (unsigned)TPB + threadIdx.x) >> 6, nwaves = (gridDim.x * (unsigned)TPB) >> 6;
    const unsigned ppr = ((unsigned)d.W + 255u) >> 8;
    const unsigned npieces = (unsigned)d.N * (unsigned)d.H * ppr;
    const int W = d.W, HW = d.H * d.W;
    for (unsigned piece = wave; piece < npieces; piece += nwaves) {
        const unsigned row = piece / ppr;
        const int z = (int)(row / (unsigned)d.H), y = (int)(row - (unsigned)z * (unsigned)d.H);
        const int x = (int)((piece - row * ppr) << 8) + lane * 4;
        const bool in = x < W;
        const int v = (int)row * W + x;
        const unsigned w = in ? *reinterpret_cast<const unsigned*>(lab + v) : 0u;
        if (__ballot(w != 0) == 0) continue;
        // bytes 0 and 5 of `mine`: the voxels left and right of this lane's four (0 outside the row: a label is never 0)
        unsigned wl = __shfl_up(w, 1) >> 24, wr = __shfl_down(w, 1) & 0xffu;
        if (lane == 0) wl = x > 0 ? lab[v - 1] : 0u;
        if (lane == 63) wr = (in && x + 4 < W) ? lab[v + 4] : 0u;
        const unsigned long long mine = ((unsigned long long)wr << 40) | ((unsigned long long)w << 8) | wl;
        // A union is a chain of 3-5 dependent L2 accesses, and a lane meets at most a few: they are queued (two slots, then inline) and
        // made after the scan, all lanes together -- one or two latency episodes per wave instead of one per (row, voxel) site.
        int qa0 = 0, qb0 = 0, qa1 = 0, qb1 = 0, nq = 0;
        auto push = [&](int a, int b) {
            if (nq == 0) qa0 = a, qb0 = b;
            else if (nq == 1) qa1 = a, qb1 = b;
            else unite(P, a, b);
            ++nq;
        };
        if (lane == 0 && (w & 0xffu) != 0 && (w & 0xffu) == wl) push(v, v - 1);  // runs are pre-connected inside a piece only
        // a row's word for this lane plus the voxels left and right of it (as bytes 0 and 5); off: wave-uniform, a multiple of 4
        auto load_row = [&](int off, unsigned& uw) {
            const int u = v + off;
            uw = in ? *reinterpret_cast<const unsigned*>(lab + u) : 0u;
            unsigned ul = __shfl_up(uw, 1) >> 24, ur = __shfl_down(uw, 1) & 0xffu;
            if (lane == 0) ul = x > 0 ? lab[u - 1] : 0u;
            if (lane == 63) ur = (in && x + 4 < W) ? lab[u + 4] : 0u;
            return ((unsigned long long)ur << 40) | ((unsigned long long)uw << 8) | ul;
        };
        // unions of this lane's voxels with the row at distance off; voxel j is skipped when byte j of `implied` is its label
        auto join_row = [&](int off, unsigned long long theirs, unsigned implied) {
            if (w == 0) return;
            const int u = v + off;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned L = (unsigned)(mine >> (8 * j + 8)) & 0xffu;
                if (!L || ((implied >> (8 * j)) & 0xffu) == L) continue;
                const bool left_same = ((unsigned)(mine >> (8 * j)) & 0xffu) == L;
                const bool right_same = ((unsigned)(mine >> (8 * j + 16)) & 0xffu) == L;
                const unsigned uc = (unsigned)(theirs >> (8 * j + 8)) & 0xffu, ulj = (unsigned)(theirs >> (8 * j)) & 0xffu,
                               urj = (unsigned)(theirs >> (8 * j + 16)) & 0xffu;
                if (uc == L) {
                    if (!(left_same && ulj == L)) push(v + j, u + j);
                } else if (C26) {
                    if (!left_same && ulj == L) push(v + j, u + j - 1);
                    if (!right_same && urj == L) push(v + j, u + j + 1);
                }
            }
        };
        unsigned wa = 0, wb = 0, wab = 0, wbb = 0;  // above (y-1, z), behind (y, z-1), above-behind (y-1, z-1), below-behind (y+1, z-1)
        unsigned long long ra = 0, rb = 0, rab = 0, rbb = 0;
        if (y > 0) ra = load_row(-W, wa);
        if (z > 0) {
            rb = load_row(-HW, wb);
            if (y > 0) rab = load_row(-HW - W, wab);
            if (C26 && y + 1 < d.H) rbb = load_row(-HW + W, wbb);
        }
        if (y > 0) join_row(-W, ra, 0u);
        if (z > 0) {
            // v ~ (x, y, z-1) is implied when the voxels above v and above-behind v both carry v's label: v ~ above (this row's own
            // union), above ~ above-behind (made by the row above), above-behind ~ behind (made in the previous slice)
            unsigned implied = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned a = (wa >> (8 * j)) & 0xffu;
                if (a == ((wab >> (8 * j)) & 0xffu)) implied |= a << (8 * j);
            }
            join_row(-HW, rb, implied);
            if (C26) {
                if (y > 0) join_row(-HW - W, rab, wb);
                if (y + 1 < d.H) join_row(-HW + W, rbb, wb);
            }
        }
        if (nq > 0) unite(P, qa0, qb0);
        if (nq > 1) unite(P, qa1, qb1);
    }
}

__global__ __launch_bounds__(TPB) void ccl_flatten_kernel(int* P, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) {
        if (P[v] >= 0) P[v] = find_root(P, (int)v);
    }
}

// ---- dense ids in raster order ------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void count_roots_kernel(const int* __restrict__ P, int* __restrict__ blockcnt, size_t nvox) {
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const size_t base = ((size_t)blockIdx.x * TPB + threadIdx.x) * ITEMS;
    int c = 0;
    for (int i = 0; i < ITEMS; ++i) {
        const size_t v = base + i;
        if (v < nvox && P[v] == (int)v) ++c;
    }
    if (c) atomicAdd(&total, c);
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}

// exclusive scan of blockcnt[0..nb) in place by ONE 1024-thread block; blockcnt[nb] and *total_dev get the sum
__global__ __launch_bounds__(1024) void scan_blockcnt_kernel(int* blockcnt, int nb, int* total_dev) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int chunk = (nb + 1023) / 1024;
    const int lo = t * chunk, hi = min(nb, lo + chunk);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += blockcnt[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int add = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int run = part[t] - s;  // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) {
        const int c = blockcnt[i];
        blockcnt[i] = run;
        run += c;
    }
    if (t == 1023) {
        blockcnt[nb] = part[1023];
        *total_dev = part[1023];
    }
}

__global__ __launch_bounds__(TPB) void assign_rank_kernel(const int* __restrict__ P, int* __restrict__ rank, const int* __restrict__ blockoff, size_t nvox) {
    __shared__ int part[TPB];
    const int t = threadIdx.x;
    // (a volume has ~10^3 roots in ~10^4 blocks: most blocks hold none and have nothing to number; blockoff[nb] is the total)
    if (blockoff[blockIdx.x + 1] == blockoff[blockIdx.x]) return;
    const size_t base = ((size_t)blockIdx.x * TPB + t) * ITEMS;
    int c = 0;
    for (int i = 0; i < ITEMS; ++i) {
        const size_t v = base + i;
        if (v < nvox && P[v] == (int)v) ++c;
    }
    part[t] = c;
    __syncthreads();
    for (int off = 1; off < TPB; off <<= 1) {
        const int add = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int id = blockoff[blockIdx.x] + part[t] - c;  // ids are 1-based: pre-increment below
    for (int i = 0; i < ITEMS; ++i) {
        const size_t v = base + i;
        if (v < nvox && P[v] == (int)v) rank[v] = ++id;
    }
}

// the same on a forest that was not flattened: the (short, path-halved) chain to the root is walked here, which spares the
// labelling its own read-modify-write pass over the parent volume when nothing else needs flat parents
__global__ __launch_bounds__(TPB) void relabel_find_kernel(const int* __restrict__ P, const int* __restrict__ rank, int* __restrict__ ids, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) {
        const int r = P[v];
        ids[v] = r >= 0 ? rank[find_root(P, r)] : 0;
    }
}

__global__ __launch_bounds__(TPB) void relabel_kernel(const int* __restrict__ P, const int* __restrict__ rank, int* __restrict__ ids, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) {
        const int r = P[v];
        ids[v] = r >= 0 ? rank[r] : 0;
    }
}

// ---- per-region statistics ----------------------------------------------------------------------
// Histogram of a key volume (region sizes).  lane = voxel; the first lane of every run of equal keys owns
// the run length.  Runs are first accumulated in a per-workgroup LDS hash (each workgroup walks a CONTIGUOUS
// range of the volume, so a large region costs one global atomic per workgroup instead of one per run:
// same-address L2 atomics serialise at ~15/us, which was 4.5 ms per pass on a 300-slice volume).
constexpr int HS = 1024;
constexpr int H_EMPTY = -2;

__device__ __forceinline__ void hist_insert(int* hkey, int* hcnt, int key, int len, int* table) {
    unsigned h = ((unsigned)key * 2654435761u) >> 22;  // 10 bits
    for (int probe = 0; probe < 8; ++probe) {
        const int old = atomicCAS(&hkey[h], H_EMPTY, key);
        if (old == H_EMPTY || old == key) {
            atomicAdd(&hcnt[h], len);
            return;
        }
        h = (h + 1) & (HS - 1);
    }
    atomicAdd(&table[key], len);  // table full around this slot: straight to memory
}

// key(v) for v in the block's contiguous range; `counted` says whether the key is histogrammed.
template <class KeyFn>
__device__ __forceinline__ void block_run_histogram(size_t nvox, int* table, KeyFn keyfn) {
    __shared__ int hkey[HS];
    __shared__ int hcnt[HS];
    for (int i = threadIdx.x; i < HS; i += blockDim.x) {
        hkey[i] = H_EMPTY;
        hcnt[i] = 0;
    }
    __syncthreads();
    const size_t nseg = (nvox + 63) / 64;
    const size_t per = (nseg + gridDim.x - 1) / gridDim.x;
    const size_t seg0 = (size_t)blockIdx.x * per;
    const size_t seg1 = seg0 + per < nseg ? seg0 + per : nseg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (size_t seg = seg0 + wave; seg < seg1; seg += nw) {
        const size_t v = seg * 64 + lane;
        bool counted = false;
        const int key = keyfn(v, &counted);
        const int prev = __shfl_up(key, 1);
        const bool head = lane == 0 || prev != key;
        const unsigned long long heads = __ballot(head);
        if (head && counted) {
            const unsigned long long higher = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int len = higher ? __ffsll((long long)higher) : 64 - lane;
            hist_insert(hkey, hcnt, key, len, table);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HS; i += blockDim.x)
        if (hkey[i] != H_EMPTY && hcnt[i]) atomicAdd(&table[hkey[i]], hcnt[i]);
}

// `cap`: regions with an id beyond it are ignored (the host sizes the tables from the previous volume before it knows this
// volume's region count, and repeats the pass when the guess was too small).
__global__ __launch_bounds__(TPB) void region_stats_kernel(const int* __restrict__ ids, const uint8_t* __restrict__ lab, int* area,
                                                           uint8_t* labval, size_t nvox, int cap) {
    block_run_histogram(nvox, area, [&](size_t v, bool* counted) {
        int id = v < nvox ? ids[v] : 0;
        if (id > cap) id = 0;
        *counted = id != 0;
        if (id && ((v & 63) == 0 || ids[v - 1] != id)) labval[id] = lab[v];
        return id;
    });
}

// Boundary voxel classes.  Each workgroup walks a contiguous range of the volume in tiles of TPB voxels and merges
// identical records (same atom, same neighbour set) in an LDS hash table before anything is written: slots are
// claimed with a CAS on a tag word, the claimant writes the 7-int key, and -- after a barrier -- the other threads
// that met the same tag compare the FULL key (no probabilistic matching, no spinning).  A record that is still
// unplaced after RROUNDS probes is written out on its own with count 1, so the result is exact in every case.
constexpr int RH = 512;     // hash slots per workgroup
constexpr int RROUNDS = 4;  // probes before a record bypasses the table

__global__ __launch_bounds__(TPB) void boundary_records_kernel(const int* __restrict__ ids, Dims d, const int* __restrict__ halo_lo,
                                                               const int* __restrict__ halo_hi, BoundaryRec* recs, unsigned* count, unsigned cap,
                                                               size_t per_block) {
    __shared__ int hkey[RH][7];
    __shared__ unsigned htag[RH];
    __shared__ int hcnt[RH];
    __shared__ int any_flag[3];
    const int tid = threadIdx.x;
    for (int i = tid; i < RH; i += TPB) {
        htag[i] = 0u;
        hcnt[i] = 0;
    }
    if (tid < 3) any_flag[tid] = 0;
    __syncthreads();
    const size_t nvox = d.nvox();
    const size_t HW = (size_t)d.H * d.W;
    const size_t v0 = (size_t)blockIdx.x * per_block;
    const size_t v1 = v0 + per_block < nvox ? v0 + per_block : nvox;
    int tile = 0;
    for (size_t base = v0; base < v1; base += TPB, ++tile) {  // block-uniform trip count
        const size_t v = base + tid;
        int a = 0;
        int nb[6] = {0, 0, 0, 0, 0, 0};  // descending, distinct, zero padded
        if (v < v1) a = ids[v];
        if (a) {
            int x, y, z;
        split3(v, d.H, d.W, x, y, z);
            auto add = [&](int b) {
                if (b == 0 || b == a) return;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    if (b == nb[i]) return;
                    if (b > nb[i]) {
                        const int t = nb[i];
                        nb[i] = b;
                        b = t;
                        if (b == 0) return;
                    }
                }
            };
            if (x > 0) add(ids[v - 1]);
            if (x + 1 < d.W) add(ids[v + 1]);
            if (y > 0) add(ids[v - d.W]);
            if (y + 1 < d.H) add(ids[v + d.W]);
            if (z > 0) add(ids[v - HW]);
            else if (halo_lo) { const int h = halo_lo[v]; add(h ? (h | HALO_LO) : 0); }
            if (z + 1 < d.N) add(ids[v + HW]);
            else if (halo_hi) { const int h = halo_hi[v - (size_t)z * HW]; add(h ? (h | HALO_HI) : 0); }
        }
        bool active = a != 0 && nb[0] != 0;
        // does any thread of the workgroup hold a record in this tile?  (three flags in rotation: the one being
        // cleared was last read two barriers ago)
        const int f = tile % 3;
        if (tid == 0) any_flag[(tile + 1) % 3] = 0;
        if (active) any_flag[f] = 1;
        __syncthreads();
        if (!any_flag[f]) continue;
        unsigned h = (unsigned)a * 0x9E3779B1u;
#pragma unroll
        for (int i = 0; i < 6; ++i) h = (h ^ (unsigned)nb[i]) * 0x85EBCA77u + (h >> 15);
        const unsigned tagv = h | 0x80000000u;
        int slot = (int)((h >> 7) & (RH - 1));
        for (int round = 0; round < RROUNDS; ++round) {  // block-uniform
            int state = 0;                               // 1: wrote the key of `slot`, 2: must compare with the key of `slot`
            if (active) {
                const unsigned old = atomicCAS(&htag[slot], 0u, tagv);
                if (old == 0u) {
                    hkey[slot][0] = a;
#pragma unroll
                    for (int i = 0; i < 6; ++i) hkey[slot][1 + i] = nb[i];
                    state = 1;
                } else if (old == tagv) {
                    state = 2;
                } else {
                    slot = (slot + 1) & (RH - 1);
                }
            }
            __syncthreads();
            if (state == 2) {
                bool same = hkey[slot][0] == a;
#pragma unroll
                for (int i = 0; i < 6; ++i) same = same && hkey[slot][1 + i] == nb[i];
                if (same) state = 1;
                else slot = (slot + 1) & (RH - 1);
            }
            if (state == 1) {
                atomicAdd(&hcnt[slot], 1);
                active = false;
            }
        }
        if (active) {  // table crowded around this hash: the record leaves on its own
            const unsigned o = atomicAdd(count, 1u);
            if (o < cap) {
                BoundaryRec r;
                r.atom = a;
#pragma unroll
                for (int i = 0; i < 6; ++i) r.nb[i] = nb[i];
                r.count = 1;
                recs[o] = r;
            }
        }
    }
    __syncthreads();
    for (int sl = tid; sl < RH; sl += TPB) {
        if (htag[sl] == 0u || hcnt[sl] == 0) continue;
        const unsigned o = atomicAdd(count, 1u);
        if (o < cap) {
            BoundaryRec r;
            r.atom = hkey[sl][0];
            for (int i = 0; i < 6; ++i) r.nb[i] = hkey[sl][1 + i];
            r.count = hcnt[sl];
            recs[o] = r;
        }
    }
}

__global__ __launch_bounds__(TPB) void apply_lut_kernel(const int* __restrict__ ids, const uint8_t* __restrict__ lut, uint8_t* __restrict__ out, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) out[v] = lut[ids[v]];
}

// ---- largest component per label ----------------------------------------------------------------
__global__ __launch_bounds__(TPB) void area_by_root_kernel(const int* __restrict__ P, int* area_by_root, size_t nvox) {
    block_run_histogram(nvox, area_by_root, [&](size_t v, bool* counted) {
        const int r = v < nvox ? P[v] : -1;
        *counted = r >= 0;
        return r;
    });
}

__global__ __launch_bounds__(TPB) void label_max_kernel(const int* __restrict__ P, const uint8_t* __restrict__ lab, const int* __restrict__ area_by_root,
                                                        unsigned long long* best, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) {
        if (P[v] == (int)v) {
            const unsigned long long key = ((unsigned long long)(unsigned)area_by_root[v] << 32) | (unsigned long long)(unsigned)v;
            atomicMax(&best[lab[v]], key);
        }
    }
}

__global__ __launch_bounds__(TPB) void complement_kernel(const int* __restrict__ P, int keep_root, uint8_t* __restrict__ bg, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) bg[v] = (P[v] != keep_root) ? 1 : 0;
}

__global__ __launch_bounds__(TPB) void flag_faces_kernel(const int* __restrict__ BP, int* flags, Dims d) {
    const size_t nvox = d.nvox();
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) {
        int x, y, z;
        split3(v, d.H, d.W, x, y, z);
        if (x == 0 || y == 0 || z == 0 || x == d.W - 1 || y == d.H - 1 || z == d.N - 1) {
            const int r = BP[v];
            if (r >= 0) flags[r] = 1;
        }
    }
}

__global__ __launch_bounds__(TPB) void threshold_roots_kernel(const int* __restrict__ BP, int* flags, int threshold, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x)
        if (BP[v] == (int)v) flags[v] = flags[v] >= threshold ? 1 : 0;
}

__global__ __launch_bounds__(TPB) void fill_write_kernel(const int* __restrict__ P, int keep_root, const int* __restrict__ BP, const int* __restrict__ flags,
                                                         uint8_t label, uint8_t* out, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) {
        bool on = P[v] == keep_root;
        if (!on) {
            const int r = BP[v];
            on = r >= 0 && flags[r] == 0;
        }
        if (on) out[v] = label;
    }
}

// ---- hole filling confined to the kept component's bounding box (post_kernels.h: Box) -----------------------------
__global__ __launch_bounds__(TPB) void component_bboxes_kernel(const int* __restrict__ P, const uint8_t* __restrict__ lab, const int* __restrict__ keep_root,
                                                               int* bbox, Dims d) {
    __shared__ int kr[256];
    __shared__ int sb[256][6];  // per label: mins then maxs, of this workgroup's voxels
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        kr[i] = keep_root[i];
        sb[i][0] = sb[i][1] = sb[i][2] = 0x7fffffff;
        sb[i][3] = sb[i][4] = sb[i][5] = -1;
    }
    __syncthreads();
    // One wave per 64-voxel piece of a row: (z, y) are wave-uniform and the x extent of a label inside the piece comes from a
    // ballot, so ONE lane per (piece, label) touches the LDS table instead of every voxel (the kept components are most of the
    // foreground: six LDS compares per voxel were the cost of this pass).
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * (unsigned)TPB + threadIdx.x) >> 6, nwaves = (gridDim.x * (unsigned)TPB) >> 6;
    const unsigned ppr = ((unsigned)d.W + 63u) >> 6;
    const unsigned npieces = (unsigned)d.N * (unsigned)d.H * ppr;
    for (unsigned piece = wave; piece < npieces; piece += nwaves) {
        const unsigned row = piece / ppr;
        const int z = (int)(row / (unsigned)d.H), y = (int)(row - (unsigned)z * (unsigned)d.H);
        const int x0 = (int)((piece - row * ppr) << 6), x = x0 + lane;
        const size_t v = (size_t)row * d.W + x;
        const bool in = x < d.W;
        const int L = in ? (int)lab[v] : 0;
        const int root = in ? P[v] : -1;  // (unconditional: both loads of the piece in flight together)
        const bool match = L != 0 && root == kr[L];
        unsigned long long todo = __ballot(match);
        while (todo) {  // wave-uniform: one round per distinct label among the matching lanes
            const int src = __ffsll((long long)todo) - 1;
            const int Ls = __shfl(L, src);
            const unsigned long long same = __ballot(match && L == Ls);
            if (lane == src) {
                const int xa = x0 + __ffsll((long long)same) - 1, xb = x0 + 63 - __clzll((long long)same);
                if (z < sb[Ls][0]) atomicMin(&sb[Ls][0], z);
                if (y < sb[Ls][1]) atomicMin(&sb[Ls][1], y);
                if (xa < sb[Ls][2]) atomicMin(&sb[Ls][2], xa);
                if (z > sb[Ls][3]) atomicMax(&sb[Ls][3], z);
                if (y > sb[Ls][4]) atomicMax(&sb[Ls][4], y);
                if (xb > sb[Ls][5]) atomicMax(&sb[Ls][5], xb);
            }
            todo &= ~same;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        if (sb[i][3] < 0) continue;
        atomicMin(&bbox[6 * i + 0], sb[i][0]);
        atomicMin(&bbox[6 * i + 1], sb[i][1]);
        atomicMin(&bbox[6 * i + 2], sb[i][2]);
        atomicMax(&bbox[6 * i + 3], sb[i][3]);
        atomicMax(&bbox[6 * i + 4], sb[i][4]);
        atomicMax(&bbox[6 * i + 5], sb[i][5]);
    }
}

__global__ __launch_bounds__(TPB) void complement_box_kernel(const int* __restrict__ P, int keep_root, Dims d, Box box, uint8_t* __restrict__ bg) {
    const size_t n = box.d.nvox();
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
        int x, y, z;
        split3(c, box.d.H, box.d.W, x, y, z);
        const size_t v = ((size_t)(box.z0 + z) * d.H + (box.y0 + y)) * d.W + (box.x0 + x);
        bg[c] = (P[v] != keep_root) ? 1 : 0;
    }
}

__global__ __launch_bounds__(TPB) void fill_write_box_kernel(const int* __restrict__ P, int keep_root, const int* __restrict__ BP, const int* __restrict__ flags,
                                                             uint8_t label, uint8_t* out, Dims d, Box box) {
    const size_t n = box.d.nvox();
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
        int x, y, z;
        split3(c, box.d.H, box.d.W, x, y, z);
        const size_t v = ((size_t)(box.z0 + z) * d.H + (box.y0 + y)) * d.W + (box.x0 + x);
        bool on = P[v] == keep_root;
        if (!on) {
            const int r = BP[c];
            on = r >= 0 && flags[r] == 0;
        }
        if (on) out[v] = label;
    }
}

// ---- slab-sharded mode --------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void atom_first_kernel(const int* __restrict__ P, const int* __restrict__ rank, int* first, int voxel_base, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x)
        if (P[v] == (int)v) first[rank[v]] = voxel_base + (int)v;
}

__global__ __launch_bounds__(TPB) void atom_face_flags_kernel(const int* __restrict__ ids, Dims d, bool zlo_face, bool zhi_face, int* flags) {
    const size_t nvox = d.nvox();
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) {
        int x, y, z;
        split3(v, d.H, d.W, x, y, z);
        if (x == 0 || y == 0 || x == d.W - 1 || y == d.H - 1 || (zlo_face && z == 0) || (zhi_face && z == d.N - 1)) {
            const int a = ids[v];
            if (a) flags[a] = 1;
        }
    }
}

__global__ __launch_bounds__(TPB) void face_edges_kernel(const int* __restrict__ ids_a, const int* __restrict__ lab_a, const int* __restrict__ ids_b,
                                                         const int* __restrict__ lab_b, int H, int W, bool conn26, int* edges, unsigned* count,
                                                         unsigned cap) {
    const size_t HW = (size_t)H * W;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < HW; v += (size_t)gridDim.x * blockDim.x) {
        const int a = ids_a[v];
        if (!a) continue;
        const int x = (int)(v % W), y = (int)(v / W);
        const int la = lab_a ? lab_a[v] : 1;
        const int a_left = x > 0 ? ids_a[v - 1] : 0;
        const int r = conn26 ? 1 : 0;
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -r; dx <= r; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const size_t u = (size_t)yy * W + xx;
                const int b = ids_b[u];
                if (!b || (lab_b && lab_b[u] != la)) continue;
                if (a_left == a && xx > 0 && ids_b[u - 1] == b) continue;  // the voxel to the left emits the same pair
                const unsigned slot = atomicAdd(count, 1u);
                if (slot < cap) {
                    edges[2 * (size_t)slot] = a;
                    edges[2 * (size_t)slot + 1] = b;
                }
            }
        }
    }
}

__global__ __launch_bounds__(TPB) void widen_u8_kernel(const uint8_t* __restrict__ in, int* __restrict__ out, size_t n) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) out[v] = in[v];
}

__global__ __launch_bounds__(TPB) void lut_complement_kernel(const int* __restrict__ ids, const uint8_t* __restrict__ keeplut, uint8_t label,
                                                             uint8_t* __restrict__ bg, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) bg[v] = keeplut[ids[v]] != label ? 1 : 0;
}

__global__ __launch_bounds__(TPB) void fill_write_lut_kernel(const int* __restrict__ ids2, const uint8_t* __restrict__ keeplut, const int* __restrict__ ids3,
                                                             const uint8_t* __restrict__ holelut, uint8_t label, uint8_t* out, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x)
        if (keeplut[ids2[v]] == label || holelut[ids3[v]]) out[v] = label;
}

__global__ __launch_bounds__(TPB) void volume_max_kernel(const uint8_t* __restrict__ a, unsigned* mx, size_t nvox) {
    unsigned m = 0;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) m = max(m, (unsigned)a[v]);
    if (m) atomicMax(mx, m);
}

// ---- utils.bbox_3D (utils.py:361-387) and the last statement of keep_largest_connected_component (utils.py:403) as stand-alone seams
// box[6] = {zmin, ymin, xmin, zmax, ymax, xmax} over the non-zero voxels (preset by the launcher: mins INT_MAX, maxs -1).  Eight voxels
// per 64-bit load; coordinates are only worked out for words that hold a non-zero byte; one set of global atomics per workgroup, and
// only where the (stale but monotone) value in memory does not already cover the workgroup's.
__global__ __launch_bounds__(TPB) void mask_bbox_kernel(const uint8_t* __restrict__ m, int* box, Dims d) {
    __shared__ int red[6][TPB / 64];
    const size_t nvox = d.nvox();
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    auto take = [&](size_t v) {
        int x, y, z;
        split3(v, d.H, d.W, x, y, z);
        lo[0] = min(lo[0], z), lo[1] = min(lo[1], y), lo[2] = min(lo[2], x);
        hi[0] = max(hi[0], z), hi[1] = max(hi[1], y), hi[2] = max(hi[2], x);
    };
    const size_t nword = ((reinterpret_cast<uintptr_t>(m) & 7) == 0) ? nvox / 8 : 0;
    const unsigned long long* mw = reinterpret_cast<const unsigned long long*>(m);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nword; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long wv = mw[i];
        if (!wv) continue;
        for (int k = 0; k < 8; ++k)
            if ((wv >> (8 * k)) & 0xffull) take(i * 8 + k);
    }
    for (size_t v = nword * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x)
        if (m[v]) take(v);
    for (int k = 0; k < 3; ++k)
        for (int off = 32; off; off >>= 1) {
            lo[k] = min(lo[k], __shfl_xor(lo[k], off));
            hi[k] = max(hi[k], __shfl_xor(hi[k], off));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 3; ++k) red[k][wave] = lo[k], red[3 + k][wave] = hi[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        int v = red[k][0];
        for (int w2 = 1; w2 < TPB / 64; ++w2) v = k < 3 ? min(v, red[k][w2]) : max(v, red[k][w2]);
        if (k < 3) {
            if (v < *(volatile int*)&box[k]) atomicMin(&box[k], v);
        } else if (v > *(volatile int*)&box[k]) atomicMax(&box[k], v);
    }
}

__global__ __launch_bounds__(64) void mask_bbox_init_kernel(int* box) {
    if (threadIdx.x < 6) box[threadIdx.x] = threadIdx.x < 3 ? 0x7fffffff : -1;
}

// out[v] = (parent[v] == keep_root): `mask == max_region` of utils.py:403
__global__ __launch_bounds__(TPB) void component_mask_kernel(const int* __restrict__ P, int keep_root, uint8_t* __restrict__ out, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) out[v] = (P[v] == keep_root) ? 1 : 0;
}

__global__ __launch_bounds__(TPB) void fuse_kernel(uint8_t* res_l, const uint8_t* __restrict__ res_r, uint8_t spare, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (size_t)gridDim.x * blockDim.x) {
        uint8_t l = res_l[v];
        const uint8_t r = res_r[v];
        if (l == 0 && r > 0) l = spare;  // mask.py:229
        if (r == 0) l = 0;               // mask.py:230
        res_l[v] = l;
    }
}

// ---- the second labelling on the region graph (post_engine.hip: postprocess) -------------------------------------------------
// After the merge loop every voxel's mapped label is lut[region]; the components of the mapped volume are therefore unions of
// first-pass regions, and which regions belong together follows from the region ADJACENCY: the 6-adjacency is in the boundary
// records already, the rest of the 26-adjacency (pairs of regions that only touch diagonally) comes from diag_pairs_kernel, the
// regions' areas and bounding boxes from region_stats_box_kernel.  The host then finds the largest component of every label on a
// graph of ~10^3 nodes, and no second voxel-level labelling / area / bounding-box pass runs at all.

// region_stats_kernel + the bounding box of every region: box[id][6] = {zmin, ymin, xmin, zmax, ymax, xmax} (preset by the
// caller: mins INT_MAX, maxs -1).  Runs of equal ids inside a 64-voxel segment (broken at row starts) go through a per-workgroup
// LDS table -- a large region costs one set of global atomics per workgroup, not per run -- and a global atomic is only issued
// when the (possibly stale, but monotone) value in memory does not already cover the workgroup's.
__global__ __launch_bounds__(TPB) void region_stats_box_kernel(const int* __restrict__ ids, const uint8_t* __restrict__ lab, int* area, uint8_t* labval,
                                                               int* box, Dims d, int cap) {
    __shared__ int hkey[HS];
    __shared__ int hcnt[HS];
    __shared__ int hb[6][HS];
    for (int i = threadIdx.x; i < HS; i += blockDim.x) {
        hkey[i] = H_EMPTY;
        hcnt[i] = 0;
        hb[0][i] = hb[1][i] = hb[2][i] = 0x7fffffff;
        hb[3][i] = hb[4][i] = hb[5][i] = -1;
    }
    __syncthreads();
    const size_t nvox = d.nvox();
    const size_t nseg = (nvox + 63) / 64;
    const size_t per = (nseg + gridDim.x - 1) / gridDim.x;
    const size_t seg0 = (size_t)blockIdx.x * per;
    const size_t seg1 = seg0 + per < nseg ? seg0 + per : nseg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    auto gmin = [](int* p, int v) { if (v < *(volatile int*)p) atomicMin(p, v); };
    auto gmax = [](int* p, int v) { if (v > *(volatile int*)p) atomicMax(p, v); };
    for (size_t seg = seg0 + wave; seg < seg1; seg += nw) {
        const size_t v = seg * 64 + lane;
        int id = v < nvox ? ids[v] : 0;
        if (id > cap) id = 0;
        const unsigned q = (unsigned)v / (unsigned)d.W;
        const int x = (int)((unsigned)v - q * (unsigned)d.W);
        const int prev = __shfl_up(id, 1);
        const bool head = lane == 0 || prev != id || x == 0;
        const unsigned long long heads = __ballot(head);
        if (head && id) {
            const unsigned long long higher = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int len = higher ? __ffsll((long long)higher) : 64 - lane;
            const int z = (int)(q / (unsigned)d.H), y = (int)(q - (unsigned)z * (unsigned)d.H);
            if (area != nullptr) labval[id] = lab[v];
            unsigned h = ((unsigned)id * 2654435761u) >> 22;
            int slot = -1;
            for (int probe = 0; probe < 8; ++probe) {
                const int old = atomicCAS(&hkey[h], H_EMPTY, id);
                if (old == H_EMPTY || old == id) {
                    slot = (int)h;
                    break;
                }
                h = (h + 1) & (HS - 1);
            }
            if (slot >= 0) {
                atomicAdd(&hcnt[slot], len);
                if (z < hb[0][slot]) atomicMin(&hb[0][slot], z);
                if (y < hb[1][slot]) atomicMin(&hb[1][slot], y);
                if (x < hb[2][slot]) atomicMin(&hb[2][slot], x);
                if (z > hb[3][slot]) atomicMax(&hb[3][slot], z);
                if (y > hb[4][slot]) atomicMax(&hb[4][slot], y);
                if (x + len - 1 > hb[5][slot]) atomicMax(&hb[5][slot], x + len - 1);
            } else {  // table full around this slot: straight to memory
                if (area != nullptr) atomicAdd(&area[id], len);
                int* b = box + 6 * (size_t)id;
                gmin(b + 0, z); gmin(b + 1, y); gmin(b + 2, x);
                gmax(b + 3, z); gmax(b + 4, y); gmax(b + 5, x + len - 1);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HS; i += blockDim.x) {
        if (hkey[i] == H_EMPTY || !hcnt[i]) continue;
        if (area != nullptr) atomicAdd(&area[hkey[i]], hcnt[i]);
        int* b = box + 6 * (size_t)hkey[i];
        gmin(b + 0, hb[0][i]); gmin(b + 1, hb[1][i]); gmin(b + 2, hb[2][i]);
        gmax(b + 3, hb[3][i]); gmax(b + 4, hb[4][i]); gmax(b + 5, hb[5][i]);
    }
}

__global__ __launch_bounds__(TPB) void region_box_init_kernel(int* box, size_t n_regions) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 6 * n_regions; i += (size_t)gridDim.x * blockDim.x) box[i] = (i % 6) < 3 ? 0x7fffffff : -1;
}

// Pairs of regions that touch DIAGONALLY (26- but not 6-adjacent voxels with two different non-zero labels: two voxels of one
// label that touch belong to one region).  Each pair of voxels is seen from its raster-later member; pairs are deduplicated per
// workgroup in an LDS table (a key that finds its 8 probe slots taken goes out on its own) and appended to `pairs` as
// (smaller id << 32 | larger id); duplicates across workgroups are harmless (the host unites).  Voxel-wise: any row length.
constexpr int PHS = 1024;
__device__ __forceinline__ void pair_emit(unsigned long long* hk, int a, int b, unsigned long long* pairs, unsigned* count, unsigned cap) {
    if (a == b || a == 0 || b == 0) return;
    const unsigned long long key = a < b ? ((unsigned long long)(unsigned)a << 32) | (unsigned)b : ((unsigned long long)(unsigned)b << 32) | (unsigned)a;
    unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 54);  // 10 bits
    for (int probe = 0; probe < 8; ++probe) {
        const unsigned long long old = atomicCAS(&hk[h], 0ull, key);
        if (old == 0ull || old == key) return;
        h = (h + 1) & (PHS - 1);
    }
    const unsigned o = atomicAdd(count, 1u);
    if (o < cap) pairs[o] = key;
}

__global__ __launch_bounds__(TPB) void diag_pairs_kernel(const uint8_t* __restrict__ lab, const int* __restrict__ ids, Dims d, unsigned long long* pairs,
                                                         unsigned* count, unsigned cap, size_t per_block) {
    __shared__ unsigned long long hk[PHS];
    for (int i = threadIdx.x; i < PHS; i += blockDim.x) hk[i] = 0ull;
    __syncthreads();
    const size_t nvox = d.nvox();
    const int W = d.W, HW = d.H * d.W;
    const size_t v0 = (size_t)blockIdx.x * per_block;
    const size_t v1 = v0 + per_block < nvox ? v0 + per_block : nvox;
    for (size_t v = v0 + threadIdx.x; v < v1; v += blockDim.x) {
        const unsigned L = lab[v];
        if (!L) continue;
        int x, y, z;
        split3(v, d.H, d.W, x, y, z);
        // the ten raster-earlier neighbours that are not 6-neighbours: (x +- 1, y - 1, z) and, in slice z - 1, all of the 3 x 3
        // block around (x, y) except its centre
        int mine = 0;
        auto look = [&](long long u) {
            const unsigned Lu = lab[u];
            if (Lu && Lu != L) {
                if (!mine) mine = ids[v];
                pair_emit(hk, mine, ids[u], pairs, count, cap);
            }
        };
        if (y > 0) {
            if (x > 0) look((long long)v - W - 1);
            if (x + 1 < W) look((long long)v - W + 1);
        }
        if (z > 0) {
            for (int dy = -1; dy <= 1; ++dy) {
                if (y + dy < 0 || y + dy >= d.H) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    if ((dy == 0 && dx == 0) || x + dx < 0 || x + dx >= W) continue;
                    look((long long)v - HW + dy * W + dx);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PHS; i += blockDim.x) {
        if (!hk[i]) continue;
        const unsigned o = atomicAdd(count, 1u);
        if (o < cap) pairs[o] = hk[i];
    }
}

// The same for rows that are a multiple of 4 long (every volume of the hot path), in the style of ccl_merge_rows_kernel: one wave per
// 256-voxel piece of a row, four voxels per lane from one 32-bit load per row involved, neighbour bytes from the neighbour lanes;
// the label words decide, region ids are only loaded where two different labels meet.
__global__ __launch_bounds__(TPB) void diag_pairs_rows_kernel(const uint8_t* __restrict__ lab, const int* __restrict__ ids, Dims d, unsigned long long* pairs,
                                                              unsigned* count, unsigned cap, unsigned pieces_per_block) {
    __shared__ unsigned long long hk[PHS];
    for (int i = threadIdx.x; i < PHS; i += blockDim.x) hk[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned ppr = ((unsigned)d.W + 255u) >> 8;
    const unsigned npieces = (unsigned)d.N * (unsigned)d.H * ppr;
    const int W = d.W, HW = d.H * d.W;
    const unsigned p0 = blockIdx.x * pieces_per_block, p1 = min(npieces, p0 + pieces_per_block);
    for (unsigned piece = p0 + wave; piece < p1; piece += TPB / 64) {
        const unsigned row = piece / ppr;
        const int z = (int)(row / (unsigned)d.H), y = (int)(row - (unsigned)z * (unsigned)d.H);
        const int x = (int)((piece - row * ppr) << 8) + lane * 4;
        const bool in = x < W;
        const int v = (int)row * W + x;
        const unsigned w = in ? *reinterpret_cast<const unsigned*>(lab + v) : 0u;
        if (__ballot(w != 0) == 0 || (y == 0 && z == 0)) continue;
        // a row's word for this lane plus the voxels left and right of it (bytes 0 and 5); off: wave-uniform, a multiple of 4
        auto load_row = [&](int off) {
            const int u = v + off;
            const unsigned uw = in ? *reinterpret_cast<const unsigned*>(lab + u) : 0u;
            unsigned ul = __shfl_up(uw, 1) >> 24, ur = __shfl_down(uw, 1) & 0xffu;
            if (lane == 0) ul = x > 0 ? lab[u - 1] : 0u;
            if (lane == 63) ur = (in && x + 4 < W) ? lab[u + 4] : 0u;
            return ((unsigned long long)ur << 40) | ((unsigned long long)uw << 8) | ul;
        };
        // voxel j of this lane against the row at `off`: the left / right neighbour there, and (centre) the one straight across
        auto scan_row = [&](int off, unsigned long long theirs, bool centre) {
            if (w == 0) return;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned L = (w >> (8 * j)) & 0xffu;
                if (!L) continue;
                const unsigned ul = (unsigned)(theirs >> (8 * j)) & 0xffu, uc = (unsigned)(theirs >> (8 * j + 8)) & 0xffu, ur = (unsigned)(theirs >> (8 * j + 16)) & 0xffu;
                const bool hl = ul && ul != L, hc = centre && uc && uc != L, hr = ur && ur != L;
                if (!(hl || hc || hr)) continue;
                const int a = ids[v + j];
                if (hl) pair_emit(hk, a, ids[v + off + j - 1], pairs, count, cap);
                if (hc) pair_emit(hk, a, ids[v + off + j], pairs, count, cap);
                if (hr) pair_emit(hk, a, ids[v + off + j + 1], pairs, count, cap);
            }
        };
        unsigned long long ra = 0, rb = 0, rab = 0, rbb = 0;  // above (y-1, z), behind (y, z-1), above-behind, below-behind
        if (y > 0) ra = load_row(-W);
        if (z > 0) {
            rb = load_row(-HW);
            if (y > 0) rab = load_row(-HW - W);
            if (y + 1 < d.H) rbb = load_row(-HW + W);
        }
        if (y > 0) scan_row(-W, ra, false);
        if (z > 0) {
            scan_row(-HW, rb, false);
            if (y > 0) scan_row(-HW - W, rab, true);
            if (y + 1 < d.H) scan_row(-HW + W, rbb, true);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PHS; i += blockDim.x) {
        if (!hk[i]) continue;
        const unsigned o = atomicAdd(count, 1u);
        if (o < cap) pairs[o] = hk[i];
    }
}

// box kernels of the hole fill driven by the region table instead of a second labelling: "in the kept component of `label`" ==
// keeplut[ids[v]] == label
__global__ __launch_bounds__(TPB) void complement_lut_box_kernel(const int* __restrict__ ids, const uint8_t* __restrict__ keeplut, uint8_t label, Dims d, Box box,
                                                                 uint8_t* __restrict__ bg) {
    const size_t n = box.d.nvox();
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
        int x, y, z;
        split3(c, box.d.H, box.d.W, x, y, z);
        const size_t v = ((size_t)(box.z0 + z) * d.H + (box.y0 + y)) * d.W + (box.x0 + x);
        bg[c] = keeplut[ids[v]] != label ? 1 : 0;
    }
}

__global__ __launch_bounds__(TPB) void fill_write_lut_box_kernel(const int* __restrict__ ids, const uint8_t* __restrict__ keeplut, const int* __restrict__ BP,
                                                                 const int* __restrict__ flags, uint8_t label, uint8_t* out, Dims d, Box box) {
    const size_t n = box.d.nvox();
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
        int x, y, z;
        split3(c, box.d.H, box.d.W, x, y, z);
        const size_t v = ((size_t)(box.z0 + z) * d.H + (box.y0 + y)) * d.W + (box.x0 + x);
        bool on = keeplut[ids[v]] == label;
        if (!on) {
            const int r = BP[c];
            on = r >= 0 && flags[r] == 0;
        }
        if (on) out[v] = label;
    }
}

}  // namespace

hipError_t region_stats_box(const int* ids, const uint8_t* lab, int* area, uint8_t* labval, int* box, Dims d, hipStream_t s, int cap) {
    LM_LAUNCH(region_box_init_kernel, dim3(grid_for((size_t)6 * ((size_t)cap + 1))), dim3(TPB), 0, s, box, (size_t)cap + 1);
    LM_LAUNCH(region_stats_box_kernel, dim3(grid_for(d.nvox(), 64 * 64, 2048)), dim3(TPB), 0, s, ids, lab, area, labval, box, d, cap);
    return hipGetLastError();
}

hipError_t diag_pairs(const uint8_t* lab, const int* ids, Dims d, unsigned long long* pairs, unsigned* count_dev, unsigned cap, hipStream_t s) {
    const size_t nvox = d.nvox();
    if (nvox == 0) return hipSuccess;
    if (d.W % 4 == 0 && (reinterpret_cast<uintptr_t>(lab) & 3) == 0 && nvox < (size_t)0x7fffffff) {
        const size_t pieces = (size_t)d.N * d.H * ((d.W + 255) / 256);
        // contiguous ranges of pieces per workgroup (the longer the range, the more voxels share a table), 2048 workgroups when the volume has them (1024: 0.27 instead of 0.17 ms for 12 % fewer pairs, profiles/history/r05e_bench.json)
        const unsigned ppb = (unsigned)std::max<size_t>((pieces + 2047) / 2048, 16);
        LM_LAUNCH(diag_pairs_rows_kernel, dim3((unsigned)((pieces + ppb - 1) / ppb)), dim3(TPB), 0, s, lab, ids, d, pairs, count_dev, cap, ppb);
    } else {
        size_t per = (nvox + 2047) / 2048;
        per = std::max<size_t>((per + TPB - 1) / TPB * TPB, 8 * TPB);
        LM_LAUNCH(diag_pairs_kernel, dim3((unsigned)((nvox + per - 1) / per)), dim3(TPB), 0, s, lab, ids, d, pairs, count_dev, cap, per);
    }
    return hipGetLastError();
}

hipError_t complement_of_lut_box(const int* ids, const uint8_t* keeplut, uint8_t label, Dims d, Box box, uint8_t* bg, hipStream_t s) {
    LM_LAUNCH(complement_lut_box_kernel, dim3(grid_for(box.d.nvox())), dim3(TPB), 0, s, ids, keeplut, label, d, box, bg);
    return hipGetLastError();
}

hipError_t fill_write_lut_box(const int* ids, const uint8_t* keeplut, const int* bgparent, const int* flags, uint8_t label, uint8_t* out, Dims d, Box box,
                              hipStream_t s) {
    LM_LAUNCH(fill_write_lut_box_kernel, dim3(grid_for(box.d.nvox())), dim3(TPB), 0, s, ids, keeplut, bgparent, flags, label, out, d, box);
    return hipGetLastError();
}

hipError_t ccl_label(const uint8_t* lab, int* parent, Dims d, bool conn26, hipStream_t s, bool flatten) {
    const size_t n = d.nvox();
    if (n == 0) return hipSuccess;
    if (d.W % 4 == 0 && (reinterpret_cast<uintptr_t>(lab) & 3) == 0 && (reinterpret_cast<uintptr_t>(parent) & 15) == 0 && n < (size_t)0x7fffffff) {
        const size_t pieces = (size_t)d.N * d.H * ((d.W + 255) / 256);
        const dim3 grid(grid_for(pieces, TPB / 64));
        LM_LAUNCH(ccl_init_rows_kernel, grid, dim3(TPB), 0, s, lab, parent, d);
        if (conn26)
            LM_LAUNCH((ccl_merge_rows_kernel<true>), grid, dim3(TPB), 0, s, lab, parent, d);
        else
            LM_LAUNCH((ccl_merge_rows_kernel<false>), grid, dim3(TPB), 0, s, lab, parent, d);
    } else {
        LM_LAUNCH(ccl_init_runs_kernel, dim3(grid_for(n)), dim3(TPB), 0, s, lab, parent, d);
        if (conn26)
            LM_LAUNCH((ccl_merge_kernel<true>), dim3(grid_for(n)), dim3(TPB), 0, s, lab, parent, d);
        else
            LM_LAUNCH((ccl_merge_kernel<false>), dim3(grid_for(n)), dim3(TPB), 0, s, lab, parent, d);
    }
    if (flatten) LM_LAUNCH(ccl_flatten_kernel, dim3(grid_for(n)), dim3(TPB), 0, s, parent, n);
    return hipGetLastError();
}

size_t rank_blocks(size_t nvox) { return (nvox + BLOCK_VOX - 1) / BLOCK_VOX; }

hipError_t ccl_rank(const int* parent, int* rank, int* ids, int* blockcnt, int* total_dev, size_t nvox, hipStream_t s, bool flat) {
    const size_t nb = rank_blocks(nvox);
    if (nb == 0) return hipSuccess;
    LM_LAUNCH(count_roots_kernel, dim3((unsigned)nb), dim3(TPB), 0, s, parent, blockcnt, nvox);
    LM_LAUNCH(scan_blockcnt_kernel, dim3(1), dim3(1024), 0, s, blockcnt, (int)nb, total_dev);
    LM_LAUNCH(assign_rank_kernel, dim3((unsigned)nb), dim3(TPB), 0, s, parent, rank, (const int*)blockcnt, nvox);
    if (flat) LM_LAUNCH(relabel_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, parent, (const int*)rank, ids, nvox);
    else LM_LAUNCH(relabel_find_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, parent, (const int*)rank, ids, nvox);
    return hipGetLastError();
}

hipError_t region_stats(const int* ids, const uint8_t* lab, int* area, uint8_t* labval, size_t nvox, hipStream_t s, int cap) {
    LM_LAUNCH(region_stats_kernel, dim3(grid_for(nvox, 64 * 64, 2048)), dim3(TPB), 0, s, ids, lab, area, labval, nvox, cap);
    return hipGetLastError();
}

hipError_t boundary_records(const int* ids, Dims d, BoundaryRec* recs, unsigned* count_dev, unsigned cap, hipStream_t s) {
    return boundary_records_halo(ids, d, nullptr, nullptr, recs, count_dev, cap, s);
}

hipError_t boundary_records_halo(const int* ids, Dims d, const int* halo_lo, const int* halo_hi, BoundaryRec* recs, unsigned* count_dev,
                                 unsigned cap, hipStream_t s) {
    const size_t nvox = d.nvox();
    if (nvox == 0) return hipSuccess;
    // contiguous ranges of >= 8 tiles per workgroup (the longer the range, the more voxels share a table)
    size_t per = (nvox + 2047) / 2048;
    per = std::max<size_t>((per + TPB - 1) / TPB * TPB, 8 * TPB);
    const unsigned blocks = (unsigned)((nvox + per - 1) / per);
    LM_LAUNCH(boundary_records_kernel, dim3(blocks), dim3(TPB), 0, s, ids, d, halo_lo, halo_hi, recs, count_dev, cap, per);
    return hipGetLastError();
}

hipError_t atom_first(const int* parent, const int* rank, int* first, int voxel_base, size_t nvox, hipStream_t s) {
    LM_LAUNCH(atom_first_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, parent, rank, first, voxel_base, nvox);
    return hipGetLastError();
}

hipError_t atom_face_flags(const int* ids, Dims d, bool zlo_face, bool zhi_face, int* flags, hipStream_t s) {
    LM_LAUNCH(atom_face_flags_kernel, dim3(grid_for(d.nvox())), dim3(TPB), 0, s, ids, d, zlo_face, zhi_face, flags);
    return hipGetLastError();
}

hipError_t face_edges(const int* ids_a, const int* lab_a, const int* ids_b, const int* lab_b, int H, int W, bool conn26, int* edges,
                      unsigned* count_dev, unsigned cap, hipStream_t s) {
    LM_LAUNCH(face_edges_kernel, dim3(grid_for((size_t)H * W)), dim3(TPB), 0, s, ids_a, lab_a, ids_b, lab_b, H, W, conn26, edges, count_dev, cap);
    return hipGetLastError();
}

hipError_t widen_u8(const uint8_t* in, int* out, size_t n, hipStream_t s) {
    LM_LAUNCH(widen_u8_kernel, dim3(grid_for(n)), dim3(TPB), 0, s, in, out, n);
    return hipGetLastError();
}

hipError_t lut_complement(const int* ids, const uint8_t* keeplut, uint8_t label, uint8_t* bg, size_t nvox, hipStream_t s) {
    LM_LAUNCH(lut_complement_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, ids, keeplut, label, bg, nvox);
    return hipGetLastError();
}

hipError_t fill_write_lut(const int* ids2, const uint8_t* keeplut, const int* ids3, const uint8_t* holelut, uint8_t label, uint8_t* out,
                          size_t nvox, hipStream_t s) {
    LM_LAUNCH(fill_write_lut_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, ids2, keeplut, ids3, holelut, label, out, nvox);
    return hipGetLastError();
}

hipError_t apply_lut(const int* ids, const uint8_t* lut, uint8_t* out, size_t nvox, hipStream_t s) {
    LM_LAUNCH(apply_lut_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, ids, lut, out, nvox);
    return hipGetLastError();
}

hipError_t component_max(const int* parent, const uint8_t* lab, int* area_by_root, unsigned long long* best, size_t nvox, hipStream_t s) {
    hipError_t e = hipMemsetAsync(area_by_root, 0, nvox * sizeof(int), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(best, 0, 256 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    LM_LAUNCH(area_by_root_kernel, dim3(grid_for(nvox, 64 * 64, 2048)), dim3(TPB), 0, s, parent, area_by_root, nvox);
    LM_LAUNCH(label_max_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, parent, lab, (const int*)area_by_root, best, nvox);
    return hipGetLastError();
}

hipError_t complement_of_component(const int* parent, int keep_root, uint8_t* bg, size_t nvox, hipStream_t s) {
    LM_LAUNCH(complement_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, parent, keep_root, bg, nvox);
    return hipGetLastError();
}

hipError_t flag_face_components(const int* bgparent, int* flags, Dims d, hipStream_t s) {
    hipError_t e = hipMemsetAsync(flags, 0, d.nvox() * sizeof(int), s);
    if (e != hipSuccess) return e;
    LM_LAUNCH(flag_faces_kernel, dim3(grid_for(d.nvox())), dim3(TPB), 0, s, bgparent, flags, d);
    return hipGetLastError();
}

hipError_t flag_large_components(const int* bgparent, int* flags, int threshold, size_t nvox, hipStream_t s) {
    hipError_t e = hipMemsetAsync(flags, 0, nvox * sizeof(int), s);
    if (e != hipSuccess) return e;
    LM_LAUNCH(area_by_root_kernel, dim3(grid_for(nvox, 64 * 64, 2048)), dim3(TPB), 0, s, bgparent, flags, nvox);
    LM_LAUNCH(threshold_roots_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, bgparent, flags, threshold, nvox);
    return hipGetLastError();
}

hipError_t fill_write(const int* parent, int keep_root, const int* bgparent, const int* flags, uint8_t label, uint8_t* out, size_t nvox,
                      hipStream_t s) {
    LM_LAUNCH(fill_write_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, parent, keep_root, bgparent, flags, label, out, nvox);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void keep_roots_kernel(const unsigned long long* __restrict__ best, int* __restrict__ keep_root, int* __restrict__ bbox) {
    const int label = threadIdx.x;
    const unsigned long long b = best[label];
    keep_root[label] = (label && b) ? (int)(unsigned)(b & 0xffffffffull) : -1;
#pragma unroll
    for (int k = 0; k < 6; ++k) bbox[6 * label + k] = k < 3 ? 0x7fffffff : -1;
}

hipError_t keep_roots_init(const unsigned long long* best, int* keep_root, int* bbox, hipStream_t s) {
    LM_LAUNCH(keep_roots_kernel, dim3(1), dim3(256), 0, s, best, keep_root, bbox);
    return hipGetLastError();
}

hipError_t component_bboxes(const int* parent, const uint8_t* lab, const int* keep_root, int* bbox, Dims d, hipStream_t s) {
    // (2048 workgroups: each flushes its LDS table with global atomics on the same few words -- 16384 of them made the pass 5x slower)
    LM_LAUNCH(component_bboxes_kernel, dim3(grid_for(d.nvox(), 64 * 64, 2048)), dim3(TPB), 0, s, parent, lab, keep_root, bbox, d);
    return hipGetLastError();
}

hipError_t complement_of_component_box(const int* parent, int keep_root, Dims d, Box box, uint8_t* bg, hipStream_t s) {
    LM_LAUNCH(complement_box_kernel, dim3(grid_for(box.d.nvox())), dim3(TPB), 0, s, parent, keep_root, d, box, bg);
    return hipGetLastError();
}

hipError_t fill_write_box(const int* parent, int keep_root, const int* bgparent, const int* flags, uint8_t label, uint8_t* out, Dims d, Box box,
                          hipStream_t s) {
    LM_LAUNCH(fill_write_box_kernel, dim3(grid_for(box.d.nvox())), dim3(TPB), 0, s, parent, keep_root, bgparent, flags, label, out, d, box);
    return hipGetLastError();
}

hipError_t volume_max(const uint8_t* a, unsigned* max_dev, size_t nvox, hipStream_t s) {
    hipError_t e = hipMemsetAsync(max_dev, 0, sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    LM_LAUNCH(volume_max_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, a, max_dev, nvox);
    return hipGetLastError();
}

hipError_t mask_bbox(const uint8_t* mask, int* box_dev, Dims d, hipStream_t s) {
    LM_LAUNCH(mask_bbox_init_kernel, dim3(1), dim3(64), 0, s, box_dev);
    LM_LAUNCH(mask_bbox_kernel, dim3(grid_for(d.nvox(), TPB * 8 * 4, 2048)), dim3(TPB), 0, s, mask, box_dev, d);
    return hipGetLastError();
}

hipError_t component_mask(const int* parent, int keep_root, uint8_t* out, size_t nvox, hipStream_t s) {
    LM_LAUNCH(component_mask_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, parent, keep_root, out, nvox);
    return hipGetLastError();
}

hipError_t fuse_labels(uint8_t* res_l, const uint8_t* res_r, uint8_t spare, size_t nvox, hipStream_t s) {
    LM_LAUNCH(fuse_kernel, dim3(grid_for(nvox)), dim3(TPB), 0, s, res_l, res_r, spare, nvox);
    return hipGetLastError();
}

}  // namespace lm
