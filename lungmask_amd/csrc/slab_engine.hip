// Slab-sharded form of utils.postprocessing (utils.py:272-358) for the multi-GPU pipeline: every rank owns a
// contiguous block of slices ("slab") and runs the voxel passes on its own slab only; what ties the slabs
// together is small and travels through six all-gathers that the caller performs between the calls below:
//
//   begin : 26-connected multi-label CCL of the slab ("atoms"), atom table        -> faces  (labels + atom ids of
//   step 0: boundary records (halo atoms tagged), same-label 26-adjacency                   the first/last slice)
//           across the face to the next rank                                      -> table 1 (atoms, records, edges)
//   step 1: [host] union atoms -> regions numbered by first voxel (== skimage.measure.label on the whole volume),
//           records mapped to regions, the sequential merge replay (utils.py:299-342) -> LUT of this slab;
//           26-connected CCL of the mapped slab                                   -> faces 2
//   step 2: face adjacency                                                        -> table 2 (atoms, edges)
//   step 3: [host] union -> per label the largest component (utils.py:390-404); per label: 6-connected CCL of the
//           complement, atoms flagged when they touch a face of the WHOLE volume  -> faces 3 (one pair per label)
//   step 4: straight-across adjacency of background atoms                         -> table 3 (flags, edges)
//   step 5: [host] union -> holes = background components without a flagged atom (fill_voids.fill, utils.py:352);
//           write kept component + holes per label in ascending label order (utils.py:353-354).
//
// Region-graph form (round 6; LM_SLAB_GRAPH=1 -- built, exact, and NOT the default: it moves the second labelling from the devices,
// where every rank labels 1 / world of the voxels, to the host of EVERY rank, which then walks all ranks' records and pairs: 5.3 against
// 4.8 ms per rank at four ranks and 8.1 against 6.5 at eight, same box, profiles/r06i_slab_graph_vs_voxel_same_box.log): the
// second labelling -- the components of the MAPPED volume -- needs no voxel pass at all, as in the single-GPU path (post_engine.hip:
// RegionGraph).  A component of the mapped volume is a union of atoms of the FIRST labelling that are 26-adjacent and carry the same
// mapped label: the 6-adjacency is in the boundary records (halo neighbours included), the diagonal rest inside a slab comes from
// diag_pairs, and across a slab face from the 26-adjacency of ALL atom pairs of the two face planes (the planes are exchanged
// anyway; the same-label subset of those pairs is what ties atoms into regions).  Table 1 carries these pairs, the host finds every
// label's kept component right behind the merge replay, and steps 1 (second half), 2 and 3 (first half) -- map, second labelling,
// faces 2, table 2 -- do not run: four exchanges, two voxel labellings less.
//
// Exactness: regions/components are unions of atoms (two voxels adjacent across a slab face with equal labels
// are in the same region whatever the slab cut), numbering needs only each region's first voxel, areas add,
// boundary records are per voxel and are de-duplicated after the atom -> region mapping, and the merge replay is
// the same code as in the single-GPU path.  Every rank performs the identical host merge on identical tables.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <numeric>

#include "engine.h"
#include "post_kernels.h"

namespace lm {

namespace {

#define LM_K(expr)                                                    \
    do {                                                              \
        hipError_t _e = (expr);                                       \
        if (_e != hipSuccess) {                                       \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e)); \
            return LM_ERR_DEVICE;                                     \
        }                                                             \
    } while (0)

struct ProfScope {
    lm_engine* e;
    ProfScope(lm_engine* e_, const char* name, double bytes) : e(e_) { e->prof.begin(e->stream, e->prof.kind_id(name), 0, bytes); }
    ~ProfScope() { e->prof.end(e->stream); }
};

struct UnionFind {
    std::vector<int> p;
    explicit UnionFind(int n) : p(n) { std::iota(p.begin(), p.end(), 0); }
    int find(int a) {
        while (p[a] != a) {
            p[a] = p[p[a]];
            a = p[a];
        }
        return a;
    }
    void unite(int a, int b) {
        a = find(a);
        b = find(b);
        if (a != b) p[std::max(a, b)] = std::min(a, b);
    }
};

// CCL + dense ids of a u8 volume on the engine stream; returns the number of atoms.
int label_slab(lm_engine* e, const uint8_t* lab, int* parent, int* ids, Dims d, bool conn26, int* n_atoms, const char* prof_name) {
    PostWorkspace& ws = e->post;
    const size_t nvox = d.nvox();
    int* total_dev = ws.scalars.as<int>();
    {
        ProfScope ps(e, prof_name, (double)nvox * 13);
        LM_K(ccl_label(lab, parent, d, conn26, e->stream));
    }
    {
        ProfScope ps(e, "post_rank_relabel", (double)nvox * 16);
        LM_K(ccl_rank(parent, ws.rank.as<int>(), ids, ws.blockcnt.as<int>(), total_dev, nvox, e->stream));
    }
    LM_HIP(hipMemcpyAsync(n_atoms, total_dev, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    LM_HIP(hipStreamSynchronize(e->stream));
    return LM_OK;
}

// area / label value / first voxel of the atoms of (ids, lab): ws.area, ws.labval, slab.first (index 0 unused)
int atom_table(lm_engine* e, const int* ids, const uint8_t* lab, const int* parent, int n_atoms, Dims d) {
    PostWorkspace& ws = e->post;
    SlabState& st = e->slab;
    const size_t nvox = d.nvox();
    LM_TRY(ws.area.reserve(((size_t)n_atoms + 1) * 4));
    LM_TRY(ws.labval.reserve((size_t)n_atoms + 1));
    LM_TRY(st.first.reserve(((size_t)n_atoms + 1) * 4));
    LM_HIP(hipMemsetAsync(ws.area.p, 0, ((size_t)n_atoms + 1) * 4, e->stream));
    LM_HIP(hipMemsetAsync(ws.labval.p, 0, (size_t)n_atoms + 1, e->stream));
    LM_HIP(hipMemsetAsync(st.first.p, 0, ((size_t)n_atoms + 1) * 4, e->stream));
    if (n_atoms) {
        ProfScope ps(e, "post_region_stats", (double)nvox * 9);
        LM_K(region_stats(ids, lab, ws.area.as<int>(), ws.labval.as<uint8_t>(), nvox, e->stream));
        LM_K(atom_first(parent, ws.rank.as<int>(), st.first.as<int>(), st.z0 * d.H * d.W, nvox, e->stream));
    }
    return LM_OK;
}

// faces of (lab, ids): [lab first slice][ids first slice][lab last slice][ids last slice], HW ints each
int pack_faces(lm_engine* e, const uint8_t* lab, const int* ids, Dims d) {
    SlabState& st = e->slab;
    const size_t HW = (size_t)d.H * d.W, last = (size_t)(d.N - 1) * HW;
    LM_TRY(st.pack.reserve(4 * HW * 4));
    int* pk = st.pack.as<int>();
    LM_K(widen_u8(lab, pk, HW, e->stream));
    LM_HIP(hipMemcpyAsync(pk + HW, ids, HW * 4, hipMemcpyDeviceToDevice, e->stream));
    LM_K(widen_u8(lab + last, pk + 2 * HW, HW, e->stream));
    LM_HIP(hipMemcpyAsync(pk + 3 * HW, ids + last, HW * 4, hipMemcpyDeviceToDevice, e->stream));
    st.pending = (long long)(4 * HW);
    st.pending_uniform = true;
    return LM_OK;
}

// adjacency of this rank's last slice with the next rank's first slice -> st.edges (pairs), *n_edges
int cross_edges(lm_engine* e, const int* ids_a, const int* lab_a, const int* ids_b, const int* lab_b, Dims d, bool conn26, size_t edge_off,
                unsigned* n_edges) {
    SlabState& st = e->slab;
    unsigned* count_dev = e->post.scalars.as<unsigned>() + 1;
    // edge_off > 0: the caller reserved the bound for all its calls up front (a reallocation would drop earlier edges)
    unsigned cap = edge_off ? (unsigned)(st.edges.cap / 8 - edge_off) : (unsigned)std::max<size_t>(1u << 14, st.edges.cap / 8);
    for (;;) {
        if (!edge_off) LM_TRY(st.edges.reserve((size_t)cap * 8));
        LM_HIP(hipMemsetAsync(count_dev, 0, sizeof(unsigned), e->stream));
        LM_K(face_edges(ids_a, lab_a, ids_b, lab_b, d.H, d.W, conn26, st.edges.as<int>() + 2 * edge_off, count_dev, cap, e->stream));
        LM_HIP(hipMemcpyAsync(n_edges, count_dev, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
        LM_HIP(hipStreamSynchronize(e->stream));
        if (*n_edges <= cap) return LM_OK;
        if (edge_off) {
            set_error("slab: edge buffer overflow");
            return LM_ERR_INVALID;
        }
        cap = *n_edges;
    }
}

int fetch_tables(lm_engine* e, const int32_t* gathered, long long stride, const long long* lens) {
    SlabState& st = e->slab;
    st.tables.assign(st.world, std::vector<int>());
    for (int r = 0; r < st.world; ++r) {
        if (lens[r] < 0 || lens[r] > stride) {
            set_error("slab: bad table length %lld of rank %d (stride %lld)", lens[r], r, stride);
            return LM_ERR_INVALID;
        }
        st.tables[r].resize((size_t)lens[r]);
        if (lens[r]) LM_HIP(hipMemcpyAsync(st.tables[r].data(), gathered + (size_t)r * stride, (size_t)lens[r] * 4, hipMemcpyDeviceToHost, e->stream));
    }
    LM_HIP(hipStreamSynchronize(e->stream));
    return LM_OK;
}

// ---- table 1 / 2: [n, nrec, nedge, ndiag | area[n] | label[n] | first[n] | recs[nrec][8] | edges[nedge][2] | diag[ndiag][2]]
// (graph form: `edges` = ALL 26-adjacent atom pairs across the face to the next rank, whatever their labels; `diag` = pairs of this
// slab's atoms that only touch diagonally)
struct AtomTable {
    int n = 0, nrec = 0, nedge = 0, ndiag = 0;
    const int *area = nullptr, *lv = nullptr, *first = nullptr, *recs = nullptr, *edges = nullptr, *diag = nullptr;
};

int parse_atom_tables(const SlabState& st, std::vector<AtomTable>& t, std::vector<int>& base) {
    t.assign(st.world, AtomTable());
    base.assign(st.world + 1, 0);
    for (int r = 0; r < st.world; ++r) {
        const std::vector<int>& v = st.tables[r];
        if (v.size() < 4) {
            set_error("slab: table of rank %d is too short", r);
            return LM_ERR_INVALID;
        }
        AtomTable& a = t[r];
        a.n = v[0];
        a.nrec = v[1];
        a.nedge = v[2];
        a.ndiag = v[3];
        if (a.n < 0 || a.nrec < 0 || a.nedge < 0 || a.ndiag < 0 ||
            (size_t)4 + 3 * (size_t)a.n + 8 * (size_t)a.nrec + 2 * (size_t)a.nedge + 2 * (size_t)a.ndiag != v.size()) {
            set_error("slab: malformed table of rank %d", r);
            return LM_ERR_INVALID;
        }
        a.area = v.data() + 4;
        a.lv = a.area + a.n;
        a.first = a.lv + a.n;
        a.recs = a.first + a.n;
        a.edges = a.recs + 8 * (size_t)a.nrec;
        a.diag = a.edges + 2 * (size_t)a.nedge;
        base[r + 1] = base[r] + a.n;
    }
    return LM_OK;
}

// atoms (global index base[r] + id - 1) -> components; edges of rank r join an atom of r with an atom of r + 1
int unite_atoms(const SlabState& st, const std::vector<AtomTable>& t, const std::vector<int>& base, UnionFind& uf) {
    for (int r = 0; r < st.world; ++r)
        for (int j = 0; j < t[r].nedge; ++j) {
            const int a = t[r].edges[2 * j], b = t[r].edges[2 * j + 1];
            if (r + 1 >= st.world || a < 1 || a > t[r].n || b < 1 || b > t[r + 1].n) {
                set_error("slab: edge (%d,%d) of rank %d out of range", a, b, r);
                return LM_ERR_INVALID;
            }
            if (st.graph && t[r].lv == t[r].area + t[r].n && t[r].lv[a - 1] != t[r + 1].lv[b - 1]) continue;  // (all-pairs edges: regions join equal labels only)
            uf.unite(base[r] + a - 1, base[r + 1] + b - 1);
        }
    return LM_OK;
}

// step 1 host part: LUT (final label value) of this rank's atoms.  Graph form (keeplut != nullptr): also, on the atom graph, every
// label's kept component (utils.py:355-356 / :390-404) -- keeplut[atom] = its mapped label when the atom belongs to it -- and the
// labels that have one.
int merge_regions(lm_engine* e, std::vector<uint8_t>& my_lut, std::vector<uint8_t>* keeplut = nullptr, std::vector<int>* labels = nullptr) {
    SlabState& st = e->slab;
    static const bool timing = [] { const char* v = getenv("LM_POST_TIMING"); return v && v[0] == '1'; }();
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    double t_regions = 0, t_records = 0, t_replay = 0;
    std::vector<AtomTable> t;
    std::vector<int> base;
    LM_TRY(parse_atom_tables(st, t, base));
    const int G = base[st.world];
    UnionFind uf(G);
    LM_TRY(unite_atoms(st, t, base, uf));
    // regions = components, numbered by their first voxel (raster order of the whole volume)
    std::vector<int> cfirst(G, 0x7fffffff), region(G, 0);
    std::vector<long long> carea(G, 0);
    for (int r = 0; r < st.world; ++r)
        for (int a = 0; a < t[r].n; ++a) {
            const int g = base[r] + a, c = uf.find(g);
            carea[c] += t[r].area[a];
            cfirst[c] = std::min(cfirst[c], t[r].first[a]);
        }
    std::vector<int> roots;
    for (int g = 0; g < G; ++g)
        if (uf.find(g) == g) roots.push_back(g);
    std::sort(roots.begin(), roots.end(), [&](int a, int b) { return cfirst[a] < cfirst[b]; });
    const int R = (int)roots.size();
    std::vector<int> area(R + 1, 0);
    std::vector<uint8_t> lv(R + 1, 0);
    for (int i = 0; i < R; ++i) region[roots[i]] = i + 1;
    for (int r = 0; r < st.world; ++r)
        for (int a = 0; a < t[r].n; ++a) {
            const int g = base[r] + a, c = uf.find(g);
            region[g] = region[c];
            lv[region[c]] = (uint8_t)t[r].lv[a];
        }
    for (int i = 0; i < R; ++i) area[i + 1] = (int)carea[roots[i]];
    t_regions = ms_now();
    // boundary records in region ids
    size_t total_rec = 0;
    for (int r = 0; r < st.world; ++r) total_rec += (size_t)t[r].nrec;
    std::vector<BoundaryRec> recs;
    recs.reserve(total_rec);
    for (int r = 0; r < st.world; ++r)
        for (int j = 0; j < t[r].nrec; ++j) {
            const int* q = t[r].recs + 8 * (size_t)j;
            if (q[0] < 1 || q[0] > t[r].n) {
                set_error("slab: record atom out of range");
                return LM_ERR_INVALID;
            }
            BoundaryRec out;
            out.atom = region[base[r] + q[0] - 1];
            int k = 0;
            for (int i = 0; i < 6 && q[1 + i]; ++i) {
                const int code = q[1 + i], id = code & ~HALO_MASK;
                const int rr = (code & HALO_LO) ? r - 1 : ((code & HALO_HI) ? r + 1 : r);
                if (rr < 0 || rr >= st.world || id < 1 || id > t[rr].n) {
                    set_error("slab: record neighbour out of range");
                    return LM_ERR_INVALID;
                }
                const int reg = region[base[rr] + id - 1];
                if (reg == out.atom) continue;
                bool dup = false;
                for (int m = 0; m < k; ++m) dup |= out.nb[m] == reg;
                if (!dup) out.nb[k++] = reg;
            }
            if (!k) continue;
            for (int i = k; i < 6; ++i) out.nb[i] = 0;
            out.count = q[7];
            recs.push_back(out);
        }
    PostInfo& info = e->post_info;
    info = PostInfo();
    info.regions = R;
    info.boundary_records = (long long)recs.size();
    std::vector<uint8_t> lut;
    t_records = ms_now();
    replay_merge(R, area.data(), lv.data(), recs.data(), recs.size(), st.spare, st.skip_below, lut, info);
    t_replay = ms_now();
    my_lut.assign((size_t)t[st.rank].n + 1, 0);
    for (int a = 0; a < t[st.rank].n; ++a) my_lut[a + 1] = lut[region[base[st.rank] + a]];
    if (keeplut == nullptr) return LM_OK;
    // ---- the second labelling on the REGION graph (atoms of one region carry one mapped label and are connected): regions with the
    // same mapped label that are 26-adjacent -- the 6-adjacency from the records the replay just used (already in region ids), the
    // diagonal rest from the slabs' diagonal pairs and from the all-pairs face edges, both mapped atom -> region
    UnionFind uf2(R + 1);
    auto join = [&](int ra, int rb) {
        if (ra != rb && lut[ra] && lut[ra] == lut[rb]) uf2.unite(ra, rb);
    };
    for (const BoundaryRec& q : recs)
        for (int i = 0; i < 6 && q.nb[i]; ++i) join(q.atom, q.nb[i]);
    for (int r = 0; r < st.world; ++r) {
        for (int j = 0; j < t[r].ndiag; ++j) {  // diagonal-only neighbours inside the slab
            const int a = t[r].diag[2 * j], b = t[r].diag[2 * j + 1];
            if (a < 1 || a > t[r].n || b < 1 || b > t[r].n) {
                set_error("slab: diagonal pair (%d,%d) of rank %d out of range", a, b, r);
                return LM_ERR_INVALID;
            }
            join(region[base[r] + a - 1], region[base[r] + b - 1]);
        }
        for (int j = 0; j < t[r].nedge; ++j)  // across the face (ranges checked by unite_atoms)
            join(region[base[r] + t[r].edges[2 * j] - 1], region[base[r + 1] + t[r].edges[2 * j + 1] - 1]);
    }
    std::vector<int> cf2(R + 1, 0x7fffffff);
    std::vector<long long> ca2(R + 1, 0);
    long long nonzero = 0;
    for (int i = 1; i <= R; ++i) {
        if (!lut[i]) continue;
        const int c = uf2.find(i);
        ca2[c] += area[i];
        cf2[c] = std::min(cf2[c], cfirst[roots[i - 1]]);
        nonzero += area[i];
    }
    // largest area; on ties the component with the LAST first voxel (merge_components' rule, the single-GPU path's key)
    int best[256];
    for (int i = 0; i < 256; ++i) best[i] = -1;
    for (int i = 1; i <= R; ++i) {
        if (!lut[i] || uf2.find(i) != i) continue;
        const int L = lut[i], b = best[L];
        if (b < 0 || ca2[i] > ca2[b] || (ca2[i] == ca2[b] && cf2[i] > cf2[b])) best[L] = i;
    }
    // utils.py:355 `np.unique(outmask_mapped)[1:]`: without a background voxel in the mapped volume the smallest LABEL is dropped
    bool drop_smallest = R > 0 && nonzero == (long long)st.n_total * st.H * st.W;
    labels->clear();
    for (int L = 1; L < 256; ++L)
        if (best[L] >= 0) {
            if (drop_smallest) {
                drop_smallest = false;
                best[L] = -1;
                continue;
            }
            labels->push_back(L);
        }
    keeplut->assign((size_t)t[st.rank].n + 1, 0);
    for (int a = 0; a < t[st.rank].n; ++a) {
        const int reg = region[base[st.rank] + a], L = lut[reg];
        if (L && best[L] == uf2.find(reg)) (*keeplut)[a + 1] = (uint8_t)L;
    }
    if (timing && st.rank == 0)
        fprintf(stderr, "lm_slab merge (rank 0 of %d): atoms -> regions %.3f | records in region ids %.3f | merge replay %.3f | components on the region graph %.3f ms "
                "(%d atoms, %d regions, %zu records)\n", st.world, t_regions, t_records, t_replay, ms_now(), G, R, recs.size());
    return LM_OK;
}

// step 3 host part: keeplut[atom] = its label value when the atom belongs to the largest component of that label
int merge_components(lm_engine* e, std::vector<uint8_t>& keeplut, std::vector<int>& labels) {
    SlabState& st = e->slab;
    std::vector<AtomTable> t;
    std::vector<int> base;
    LM_TRY(parse_atom_tables(st, t, base));
    const int G = base[st.world];
    UnionFind uf(G);
    LM_TRY(unite_atoms(st, t, base, uf));
    std::vector<int> cfirst(G, 0x7fffffff);
    std::vector<long long> carea(G, 0);
    std::vector<int> clv(G, 0);
    for (int r = 0; r < st.world; ++r)
        for (int a = 0; a < t[r].n; ++a) {
            const int g = base[r] + a, c = uf.find(g);
            carea[c] += t[r].area[a];
            cfirst[c] = std::min(cfirst[c], t[r].first[a]);
            clv[c] = t[r].lv[a];
        }
    // largest area; on ties the component with the LAST first voxel (the single-GPU path's atomicMax key)
    int best[256];
    for (int i = 0; i < 256; ++i) best[i] = -1;
    for (int g = 0; g < G; ++g) {
        if (uf.find(g) != g) continue;
        const int L = clv[g] & 0xff, b = best[L];
        if (b < 0 || carea[g] > carea[b] || (carea[g] == carea[b] && cfirst[g] > cfirst[b])) best[L] = g;
    }
    // utils.py:355 iterates `np.unique(outmask_mapped)[1:]`: without a single background voxel in the mapped volume the value it
    // drops is the smallest LABEL (post_engine.hip: dropped_label).  The atoms of this table are the mapped volume's non-zero voxels.
    long long nonzero = 0;
    for (int g = 0; g < G; ++g)
        if (uf.find(g) == g) nonzero += carea[g];
    bool drop_smallest = G > 0 && nonzero == (long long)st.n_total * st.H * st.W;
    labels.clear();
    for (int L = 1; L < 256; ++L)
        if (best[L] >= 0) {
            if (drop_smallest) {
                drop_smallest = false;
                best[L] = -1;
                continue;
            }
            labels.push_back(L);
        }
    keeplut.assign((size_t)t[st.rank].n + 1, 0);
    for (int a = 0; a < t[st.rank].n; ++a) {
        const int c = uf.find(base[st.rank] + a), L = clv[c] & 0xff;
        if (L && best[L] == c) keeplut[a + 1] = (uint8_t)L;
    }
    return LM_OK;
}

// step 5 host part, table 3: [K,0,0,0] then per label [n, nedge | flags[n] | edges[nedge][2]] -> holelut per label
int merge_background(lm_engine* e, std::vector<std::vector<uint8_t>>& holelut) {
    SlabState& st = e->slab;
    const int K = (int)st.labels.size();
    std::vector<size_t> pos(st.world, 4);
    for (int r = 0; r < st.world; ++r)
        if (st.tables[r].size() < 4 || st.tables[r][0] != K) {
            set_error("slab: background table of rank %d does not match", r);
            return LM_ERR_INVALID;
        }
    holelut.assign(K, std::vector<uint8_t>());
    for (int k = 0; k < K; ++k) {
        std::vector<int> n(st.world), ne(st.world), base(st.world + 1, 0);
        std::vector<const int*> flags(st.world), edges(st.world);
        for (int r = 0; r < st.world; ++r) {
            const std::vector<int>& v = st.tables[r];
            if (pos[r] + 2 > v.size()) {
                set_error("slab: truncated background table");
                return LM_ERR_INVALID;
            }
            n[r] = v[pos[r]];
            ne[r] = v[pos[r] + 1];
            if (n[r] < 0 || ne[r] < 0 || pos[r] + 2 + (size_t)n[r] + 2 * (size_t)ne[r] > v.size()) {
                set_error("slab: truncated background table");
                return LM_ERR_INVALID;
            }
            flags[r] = v.data() + pos[r] + 2;
            edges[r] = flags[r] + n[r];
            pos[r] += 2 + (size_t)n[r] + 2 * (size_t)ne[r];
            base[r + 1] = base[r] + n[r];
        }
        UnionFind uf(base[st.world]);
        for (int r = 0; r < st.world; ++r)
            for (int j = 0; j < ne[r]; ++j) {
                const int a = edges[r][2 * j], b = edges[r][2 * j + 1];
                if (r + 1 >= st.world || a < 1 || a > n[r] || b < 1 || b > n[r + 1]) {
                    set_error("slab: background edge out of range");
                    return LM_ERR_INVALID;
                }
                uf.unite(base[r] + a - 1, base[r + 1] + b - 1);
            }
        std::vector<uint8_t> outside(base[st.world], 0);
        for (int r = 0; r < st.world; ++r)
            for (int a = 0; a < n[r]; ++a)
                if (flags[r][a]) outside[uf.find(base[r] + a)] = 1;
        holelut[k].assign((size_t)n[st.rank] + 1, 0);
        for (int a = 0; a < n[st.rank]; ++a) holelut[k][a + 1] = outside[uf.find(base[st.rank] + a)] ? 0 : 1;
    }
    return LM_OK;
}

int write_header(lm_engine* e, int* dst, int a, int b, int c, int d) {
    const int h[4] = {a, b, c, d};
    LM_HIP(hipMemcpyAsync(dst, h, sizeof h, hipMemcpyHostToDevice, e->stream));
    LM_HIP(hipStreamSynchronize(e->stream));  // h is on the stack
    return LM_OK;
}

// table 1 / 2 from ws.area / ws.labval / st.first (+ records, + edges)
int pack_atom_table(lm_engine* e, int n, unsigned nrec, unsigned nedge, unsigned ndiag = 0) {
    SlabState& st = e->slab;
    PostWorkspace& ws = e->post;
    const size_t len = 4 + 3 * (size_t)n + 8 * (size_t)nrec + 2 * (size_t)nedge + 2 * (size_t)ndiag;
    LM_TRY(st.pack.reserve(len * 4));
    int* pk = st.pack.as<int>();
    LM_TRY(write_header(e, pk, n, (int)nrec, (int)nedge, (int)ndiag));
    // (diag_pairs writes (smaller id << 32 | larger id) as one 64-bit word: two ints per pair, the order inside a pair is immaterial)
    if (ndiag) LM_HIP(hipMemcpyAsync(pk + 4 + 3 * (size_t)n + 8 * (size_t)nrec + 2 * (size_t)nedge, ws.pairs.p, (size_t)ndiag * 8, hipMemcpyDeviceToDevice, e->stream));
    if (n) {
        LM_HIP(hipMemcpyAsync(pk + 4, ws.area.as<int>() + 1, (size_t)n * 4, hipMemcpyDeviceToDevice, e->stream));
        LM_K(widen_u8(ws.labval.as<uint8_t>() + 1, pk + 4 + n, (size_t)n, e->stream));
        LM_HIP(hipMemcpyAsync(pk + 4 + 2 * (size_t)n, st.first.as<int>() + 1, (size_t)n * 4, hipMemcpyDeviceToDevice, e->stream));
    }
    if (nrec) LM_HIP(hipMemcpyAsync(pk + 4 + 3 * (size_t)n, ws.recs.p, (size_t)nrec * sizeof(BoundaryRec), hipMemcpyDeviceToDevice, e->stream));
    if (nedge) LM_HIP(hipMemcpyAsync(pk + 4 + 3 * (size_t)n + 8 * (size_t)nrec, st.edges.p, (size_t)nedge * 8, hipMemcpyDeviceToDevice, e->stream));
    st.pending = (long long)len;
    st.pending_uniform = false;
    return LM_OK;
}

// per kept label (st.labels; membership: keeplut[keep_ids[v]] == label): the 6-connected labelling of the component's complement
// inside the slab, atoms flagged when they touch a face of the WHOLE volume, and the first / last plane of every labelling -> faces 3
int background_faces(lm_engine* e) {
    SlabState& st = e->slab;
    PostWorkspace& ws = e->post;
    hipStream_t s = e->stream;
    const Dims d{st.n, st.H, st.W};
    const size_t nvox = d.nvox(), HW = (size_t)st.H * st.W, last = (size_t)(st.n - 1) * HW;
    const int K = (int)st.labels.size();
    st.n3.assign(K, 0);
    LM_TRY(st.ids3.reserve(std::max<size_t>((size_t)K, 1) * nvox * 4));
    for (int k = 0; k < K; ++k) {
        LM_K(lut_complement(st.keep_ids, st.keeplut.as<uint8_t>(), (uint8_t)st.labels[k], ws.bg.as<uint8_t>(), nvox, s));
        LM_TRY(label_slab(e, ws.bg.as<uint8_t>(), ws.bgparent.as<int>(), st.ids3.as<int>() + (size_t)k * nvox, d, false, &st.n3[k], "post_ccl6_background"));
    }
    size_t nflags = 0;
    for (int k = 0; k < K; ++k) nflags += (size_t)st.n3[k] + 1;
    LM_TRY(st.flags.reserve(std::max<size_t>(nflags, 1) * 4));
    LM_HIP(hipMemsetAsync(st.flags.p, 0, std::max<size_t>(nflags, 1) * 4, s));
    LM_TRY(st.pack.reserve(std::max<size_t>((size_t)K, 1) * 2 * HW * 4));
    size_t foff = 0;
    for (int k = 0; k < K; ++k) {
        const int* ids3 = st.ids3.as<int>() + (size_t)k * nvox;
        // faces of the WHOLE volume: lateral faces, the first slice of rank 0, the last slice of the last rank
        LM_K(atom_face_flags(ids3, d, st.z0 == 0, st.z0 + st.n == st.n_total, st.flags.as<int>() + foff, s));
        foff += (size_t)st.n3[k] + 1;
        LM_HIP(hipMemcpyAsync(st.pack.as<int>() + (size_t)(2 * k) * HW, ids3, HW * 4, hipMemcpyDeviceToDevice, s));
        LM_HIP(hipMemcpyAsync(st.pack.as<int>() + (size_t)(2 * k + 1) * HW, ids3 + last, HW * 4, hipMemcpyDeviceToDevice, s));
    }
    st.pending = (long long)((size_t)K * 2 * HW);
    st.pending_uniform = true;  // K is derived from the gathered tables: the same on every rank
    return LM_OK;
}

}  // namespace

static_assert(sizeof(BoundaryRec) == 32, "boundary records travel as 8 ints");

int slab_begin(lm_engine* e, uint8_t* lab, int n, int h, int w, int rank, int world, int z0, int n_total, const int* spare, int n_spare,
               int skip_below) {
    SlabState& st = e->slab;
    st.phase = -1;
    st.pending = 0;
    if (n < 1 || h < 1 || w < 1 || world < 1 || rank < 0 || rank >= world || z0 < 0 || z0 + n > n_total) {
        set_error("lm_slab_begin: bad slab geometry (n=%d z0=%d n_total=%d rank=%d world=%d)", n, z0, n_total, rank, world);
        return LM_ERR_INVALID;
    }
    if ((size_t)n_total * h * w >= 0x1fffffffull) {  // first-voxel keys are int32 and atom ids carry two tag bits
        set_error("lm_slab_begin: volume too large for 32-bit voxel indices");
        return LM_ERR_INVALID;
    }
    st.rank = rank; st.world = world; st.n = n; st.H = h; st.W = w; st.z0 = z0; st.n_total = n_total; st.skip_below = skip_below;
    st.spare.assign(spare, spare + (spare ? n_spare : 0));
    st.lab = lab;
    static const bool graph_ok = [] { const char* v = getenv("LM_SLAB_GRAPH"); return v && v[0] == '1'; }();
    st.graph = graph_ok;
    st.keep_ids = nullptr;
    const Dims d{n, h, w};
    const size_t nvox = d.nvox();
    PostWorkspace& ws = e->post;
    LM_TRY(ws.parent.reserve(nvox * 4));
    LM_TRY(ws.ids.reserve(nvox * 4));
    LM_TRY(ws.rank.reserve(nvox * 4));
    LM_TRY(ws.bgparent.reserve(nvox * 4));
    LM_TRY(ws.blockcnt.reserve((rank_blocks(nvox) + 2) * 4));
    LM_TRY(ws.mapped.reserve(nvox));
    LM_TRY(ws.bg.reserve(nvox));
    LM_TRY(ws.out.reserve(nvox));
    LM_TRY(ws.scalars.reserve(4096));
    LM_TRY(st.ids2.reserve(nvox * 4));
    LM_TRY(label_slab(e, lab, ws.parent.as<int>(), ws.ids.as<int>(), d, true, &st.n1, "post_ccl26_multilabel"));
    LM_TRY(atom_table(e, ws.ids.as<int>(), lab, ws.parent.as<int>(), st.n1, d));
    LM_TRY(pack_faces(e, lab, ws.ids.as<int>(), d));
    st.phase = 0;
    return LM_OK;
}

int slab_emit(lm_engine* e, int32_t* dst) {
    SlabState& st = e->slab;
    if (st.phase < 0 || !dst) {
        set_error("lm_slab_emit: nothing pending");
        return LM_ERR_INVALID;
    }
    if (st.pending) LM_HIP(hipMemcpyAsync(dst, st.pack.p, (size_t)st.pending * 4, hipMemcpyDeviceToDevice, e->stream));
    LM_HIP(hipStreamSynchronize(e->stream));
    return LM_OK;
}

int slab_step(lm_engine* e, const int32_t* gathered, long long stride, const long long* lens) {
    SlabState& st = e->slab;
    PostWorkspace& ws = e->post;
    if (st.phase < 0 || st.phase > 5 || !lens || (stride > 0 && !gathered)) {
        set_error("lm_slab_step: no slab post-processing in progress");
        return LM_ERR_INVALID;
    }
    const Dims d{st.n, st.H, st.W};
    const size_t nvox = d.nvox(), HW = (size_t)st.H * st.W, last = (size_t)(st.n - 1) * HW;
    const bool has_next = st.rank + 1 < st.world, has_prev = st.rank > 0;
    hipStream_t s = e->stream;
    unsigned* count_dev = ws.scalars.as<unsigned>() + 1;
    auto faces_ok = [&](size_t ints) {
        for (int r = 0; r < st.world; ++r)
            if ((size_t)lens[r] != ints) return false;
        return (size_t)stride >= ints;
    };
    switch (st.phase) {
        case 0: {  // faces 1 -> table 1
            if (!faces_ok(4 * HW)) {
                set_error("lm_slab_step(0): expected %zu ints per rank", 4 * HW);
                return LM_ERR_INVALID;
            }
            const int* halo_lo = has_prev ? gathered + (size_t)(st.rank - 1) * stride + 3 * HW : nullptr;
            const int* halo_hi = has_next ? gathered + (size_t)(st.rank + 1) * stride + HW : nullptr;
            unsigned cap = (unsigned)std::min<size_t>(nvox, std::max<size_t>(ws.recs.cap / sizeof(BoundaryRec), 1u << 18));
            unsigned nrec = 0;
            for (;;) {
                LM_TRY(ws.recs.reserve((size_t)cap * sizeof(BoundaryRec)));
                LM_HIP(hipMemsetAsync(count_dev, 0, sizeof(unsigned), s));
                {
                    ProfScope ps(e, "post_boundary_records", (double)nvox * 4);
                    LM_K(boundary_records_halo(ws.ids.as<int>(), d, halo_lo, halo_hi, ws.recs.as<BoundaryRec>(), count_dev, cap, s));
                }
                LM_HIP(hipMemcpyAsync(&nrec, count_dev, sizeof(unsigned), hipMemcpyDeviceToHost, s));
                LM_HIP(hipStreamSynchronize(s));
                if (nrec <= cap) break;
                cap = nrec;
            }
            unsigned nedge = 0, ndiag = 0;
            if (has_next) {
                const int* mine = gathered + (size_t)st.rank * stride;
                const int* next = gathered + (size_t)(st.rank + 1) * stride;
                // graph form: ALL 26-adjacent atom pairs across the face (labels ignored); the host picks the equal-label ones for the regions
                if (st.graph) LM_TRY(cross_edges(e, mine + 3 * HW, nullptr, next + HW, nullptr, d, true, 0, &nedge));
                else LM_TRY(cross_edges(e, mine + 3 * HW, mine + 2 * HW, next + HW, next, d, true, 0, &nedge));
            }
            if (st.graph) {  // pairs of this slab's atoms that only touch diagonally
                unsigned* pcount_dev = ws.scalars.as<unsigned>() + 3;
                unsigned pcap = (unsigned)std::max<size_t>(ws.pairs.cap / 8, 1u << 16);
                for (;;) {
                    LM_TRY(ws.pairs.reserve((size_t)pcap * 8));
                    LM_HIP(hipMemsetAsync(pcount_dev, 0, sizeof(unsigned), s));
                    {
                        ProfScope ps(e, "post_diag_pairs", (double)nvox * 2);
                        LM_K(diag_pairs(st.lab, ws.ids.as<int>(), d, ws.pairs.as<unsigned long long>(), pcount_dev, pcap, s));
                    }
                    LM_HIP(hipMemcpyAsync(&ndiag, pcount_dev, sizeof(unsigned), hipMemcpyDeviceToHost, s));
                    LM_HIP(hipStreamSynchronize(s));
                    if (ndiag <= pcap) break;
                    pcap = ndiag;
                }
            }
            LM_TRY(pack_atom_table(e, st.n1, nrec, nedge, ndiag));
            break;
        }
        case 1: {  // table 1 -> LUT, mapped slab, its labelling -> faces 2   (graph form: -> kept components, background labelling -> faces 3)
            if (st.graph) {
                LM_TRY(fetch_tables(e, gathered, stride, lens));
                std::vector<uint8_t> lut, keep;
                LM_TRY(merge_regions(e, lut, &keep, &st.labels));
                if ((int)keep.size() != st.n1 + 1) {
                    set_error("lm_slab_step(1): table does not match this rank's atoms");
                    return LM_ERR_INVALID;
                }
                LM_TRY(st.keeplut.reserve(keep.size()));
                LM_HIP(hipMemcpyAsync(st.keeplut.p, keep.data(), keep.size(), hipMemcpyHostToDevice, s));
                LM_HIP(hipStreamSynchronize(s));
                st.keep_ids = ws.ids.as<int>();
                LM_TRY(background_faces(e));
                st.phase = 4;  // faces 3 -> table 3 comes next: steps 2 and 3 do not exist in this form
                return LM_OK;
            }
            LM_TRY(fetch_tables(e, gathered, stride, lens));
            std::vector<uint8_t> lut;
            LM_TRY(merge_regions(e, lut));
            if ((int)lut.size() != st.n1 + 1) {
                set_error("lm_slab_step(1): table does not match this rank's atoms");
                return LM_ERR_INVALID;
            }
            LM_TRY(ws.lut.reserve(lut.size()));
            LM_HIP(hipMemcpyAsync(ws.lut.p, lut.data(), lut.size(), hipMemcpyHostToDevice, s));
            LM_HIP(hipStreamSynchronize(s));
            {
                ProfScope ps(e, "post_apply_lut", (double)nvox * 5);
                LM_K(apply_lut(ws.ids.as<int>(), ws.lut.as<uint8_t>(), ws.mapped.as<uint8_t>(), nvox, s));
            }
            LM_TRY(label_slab(e, ws.mapped.as<uint8_t>(), ws.parent.as<int>(), st.ids2.as<int>(), d, true, &st.n2, "post_ccl26_mapped"));
            LM_TRY(atom_table(e, st.ids2.as<int>(), ws.mapped.as<uint8_t>(), ws.parent.as<int>(), st.n2, d));
            LM_TRY(pack_faces(e, ws.mapped.as<uint8_t>(), st.ids2.as<int>(), d));
            break;
        }
        case 2: {  // faces 2 -> table 2
            if (!faces_ok(4 * HW)) {
                set_error("lm_slab_step(2): expected %zu ints per rank", 4 * HW);
                return LM_ERR_INVALID;
            }
            unsigned nedge = 0;
            if (has_next) {
                const int* mine = gathered + (size_t)st.rank * stride;
                const int* next = gathered + (size_t)(st.rank + 1) * stride;
                LM_TRY(cross_edges(e, mine + 3 * HW, mine + 2 * HW, next + HW, next, d, true, 0, &nedge));
            }
            LM_TRY(pack_atom_table(e, st.n2, 0, nedge));
            break;
        }
        case 3: {  // table 2 -> kept components; per label the background labelling -> faces 3
            LM_TRY(fetch_tables(e, gathered, stride, lens));
            std::vector<uint8_t> keep;
            LM_TRY(merge_components(e, keep, st.labels));
            if ((int)keep.size() != st.n2 + 1) {
                set_error("lm_slab_step(3): table does not match this rank's atoms");
                return LM_ERR_INVALID;
            }
            LM_TRY(st.keeplut.reserve(keep.size()));
            LM_HIP(hipMemcpyAsync(st.keeplut.p, keep.data(), keep.size(), hipMemcpyHostToDevice, s));
            LM_HIP(hipStreamSynchronize(s));
            st.keep_ids = st.ids2.as<int>();
            LM_TRY(background_faces(e));
            break;
        }
        case 4: {  // faces 3 -> table 3
            const int K = (int)st.labels.size();
            if (!faces_ok((size_t)K * 2 * HW)) {
                set_error("lm_slab_step(4): expected %zu ints per rank", (size_t)K * 2 * HW);
                return LM_ERR_INVALID;
            }
            std::vector<unsigned> ne(K, 0);
            size_t eoff = 0;
            LM_TRY(st.edges.reserve((std::max<size_t>((size_t)K, 1) * HW + 1) * 8));  // at most one pair per voxel and label
            for (int k = 0; k < K && has_next; ++k) {
                const int* mine = gathered + (size_t)st.rank * stride + (size_t)(2 * k + 1) * HW;
                const int* next = gathered + (size_t)(st.rank + 1) * stride + (size_t)(2 * k) * HW;
                LM_TRY(cross_edges(e, mine, nullptr, next, nullptr, d, false, eoff, &ne[k]));
                eoff += ne[k];
            }
            size_t len = 4;
            for (int k = 0; k < K; ++k) len += 2 + (size_t)st.n3[k] + 2 * (size_t)ne[k];
            LM_TRY(st.pack.reserve(len * 4));
            int* pk = st.pack.as<int>();
            LM_TRY(write_header(e, pk, K, 0, 0, 0));
            size_t pos = 4, foff = 0;
            eoff = 0;
            for (int k = 0; k < K; ++k) {
                const int h2[2] = {st.n3[k], (int)ne[k]};
                LM_HIP(hipMemcpyAsync(pk + pos, h2, sizeof h2, hipMemcpyHostToDevice, s));
                LM_HIP(hipStreamSynchronize(s));
                pos += 2;
                if (st.n3[k]) LM_HIP(hipMemcpyAsync(pk + pos, st.flags.as<int>() + foff + 1, (size_t)st.n3[k] * 4, hipMemcpyDeviceToDevice, s));
                pos += (size_t)st.n3[k];
                foff += (size_t)st.n3[k] + 1;
                if (ne[k]) LM_HIP(hipMemcpyAsync(pk + pos, st.edges.as<int>() + 2 * eoff, (size_t)ne[k] * 8, hipMemcpyDeviceToDevice, s));
                pos += 2 * (size_t)ne[k];
                eoff += ne[k];
            }
            st.pending = (long long)len;
            st.pending_uniform = false;
            break;
        }
        case 5: {  // table 3 -> holes; write the result
            LM_TRY(fetch_tables(e, gathered, stride, lens));
            std::vector<std::vector<uint8_t>> hole;
            LM_TRY(merge_background(e, hole));
            const int K = (int)st.labels.size();
            uint8_t* out = ws.out.as<uint8_t>();
            LM_HIP(hipMemsetAsync(out, 0, nvox, s));
            size_t total = 0;
            for (int k = 0; k < K; ++k) total += hole[k].size();
            LM_TRY(st.holelut.reserve(std::max<size_t>(total, 1)));
            size_t off = 0;
            for (int k = 0; k < K; ++k) {
                if ((int)hole[k].size() != st.n3[k] + 1) {
                    set_error("lm_slab_step(5): table does not match this rank's atoms");
                    return LM_ERR_INVALID;
                }
                LM_HIP(hipMemcpyAsync(st.holelut.as<uint8_t>() + off, hole[k].data(), hole[k].size(), hipMemcpyHostToDevice, s));
                off += hole[k].size();
            }
            LM_HIP(hipStreamSynchronize(s));  // `hole` is host memory
            off = 0;
            for (int k = 0; k < K; ++k) {  // ascending label order: later labels overwrite (utils.py:353-354)
                ProfScope ps(e, "post_fill_write", (double)nvox * 13);
                LM_K(fill_write_lut(st.keep_ids, st.keeplut.as<uint8_t>(), st.ids3.as<int>() + (size_t)k * nvox, st.holelut.as<uint8_t>() + off,
                                    (uint8_t)st.labels[k], out, nvox, s));
                off += hole[k].size();
            }
            LM_HIP(hipMemcpyAsync(st.lab, out, nvox, hipMemcpyDeviceToDevice, s));
            LM_HIP(hipStreamSynchronize(s));
            st.pending = 0;
            st.phase = -1;
            return 1;  // finished
        }
    }
    st.phase++;
    return LM_OK;
}

}  // namespace lm
