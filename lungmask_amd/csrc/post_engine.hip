// Host orchestration of utils.postprocessing (utils.py:272-358) and of the whole
// LMInferer.apply pipeline (mask.py:141-232) over device-resident volumes.
//
// Voxel-level work (3-D labelling, region statistics, boundary extraction, LUT
// mapping, largest component, hole filling) runs in post_kernels.hip.  The
// reference's region-merge loop (utils.py:310-339) is sequential and
// order-dependent by construction; it is replayed here on the region graph:
// every voxel that touches another region is shipped once as a BoundaryRec, and
// the shell histogram "np.unique(sub[dilated], return_counts=True)" of a
// (possibly already merged) region is recomputed from those records with a
// per-record stamp, so a voxel adjacent to several members of the region counts
// once -- exactly the 6-connected binary_dilation of utils.py:317.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <numeric>

#include <thread>

#include "engine.h"
#include "post_kernels.h"
#include "pre_kernels.h"

namespace lm {

namespace {

#define LM_K(expr)                                                              \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) {                                                 \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));           \
            return LM_ERR_DEVICE;                                               \
        }                                                                       \
    } while (0)

struct ProfScope {
    lm_engine* e;
    hipStream_t st;
    ProfScope(lm_engine* e_, const char* name, double bytes, hipStream_t st_ = nullptr) : e(e_), st(st_ ? st_ : e_->stream) {
        e->prof.begin(st, e->prof.kind_id(name), 0, bytes);
    }
    ~ProfScope() { e->prof.end(st); }
};

}  // namespace

// utils.py:303-342 on the region graph.  Returns lut[atom] = final label value (0 = removed).
void replay_merge(int R, const int* area, const uint8_t* lv, const BoundaryRec* recs, size_t nrecs, const std::vector<int>& spare,
                  int skip_below, std::vector<uint8_t>& lut, PostInfo& info) {
    // (the tables of this function and of graph_components live across calls: a dozen allocations of ~100 KB each per volume were
    // a measurable part of the 0.2 ms the replay takes)
    static thread_local std::vector<int> order;
    order.resize(R);
    std::iota(order.begin(), order.end(), 1);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return area[a] < area[b]; });  // :299
    unsigned maxsub[256];
    memset(maxsub, 0, sizeof maxsub);
    static thread_local std::vector<uint8_t> lobemap;
    static thread_local std::vector<long long> cache_area;
    lobemap.assign(R + 1, 0);
    cache_area.assign(area, area + R + 1);
    for (int r : order) {  // :303-308
        const int mi = lv[r];
        if (cache_area[r] > (long long)maxsub[mi]) {
            maxsub[mi] = (unsigned)cache_area[r];
            lobemap[r] = (uint8_t)mi;
        }
    }
    bool spare_label[256];
    memset(spare_label, 0, sizeof spare_label);
    for (int s : spare)
        if (s >= 0 && s < 256) spare_label[s] = true;
    auto id_in_spare = [&](int id) {  // utils.py:323 compares a REGION ID with the spare LABEL values
        for (int s : spare)
            if (s == id) return true;
        return false;
    };
    // adjacency: atom -> records in which it appears as a neighbour
    static thread_local std::vector<unsigned> adj_off, adj, fill;
    adj_off.assign(R + 2, 0);
    for (size_t j = 0; j < nrecs; ++j)
        for (int k = 0; k < 6 && recs[j].nb[k]; ++k) adj_off[recs[j].nb[k] + 1]++;
    for (int i = 1; i <= R + 1; ++i) adj_off[i] += adj_off[i - 1];
    adj.resize(adj_off[R + 1]);
    {
        fill.assign(adj_off.begin(), adj_off.end() - 1);
        for (size_t j = 0; j < nrecs; ++j)
            for (int k = 0; k < 6 && recs[j].nb[k]; ++k) adj[fill[recs[j].nb[k]]++] = (unsigned)j;
    }
    // current id of every atom: union-find + member lists
    static thread_local std::vector<int> uf, setid, head, tail, next, rep;
    uf.resize(R + 1), setid.resize(R + 1), head.resize(R + 1), tail.resize(R + 1), rep.resize(R + 1);
    next.assign(R + 1, 0);
    for (int i = 0; i <= R; ++i) uf[i] = setid[i] = head[i] = tail[i] = rep[i] = i;
    auto find = [&](int a) {
        while (uf[a] != a) {
            uf[a] = uf[uf[a]];
            a = uf[a];
        }
        return a;
    };
    static thread_local std::vector<int> stamp, counts, touched;
    stamp.assign(nrecs, 0);
    counts.assign(R + 1, 0);
    touched.clear();
    for (int r : order) {  // :310-339
        const int mi = lv[r];
        if (!((cache_area[r] < (long long)maxsub[mi] || spare_label[mi]) && cache_area[r] >= skip_below)) continue;
        info.processed++;
        touched.clear();
        for (int a = head[r]; a; a = next[a]) {
            for (unsigned q = adj_off[a]; q < adj_off[a + 1]; ++q) {
                const unsigned j = adj[q];
                if (stamp[j] == r) continue;
                stamp[j] = r;
                const int n = setid[find(recs[j].atom)];
                if (n == r) continue;
                if (counts[n] == 0) touched.push_back(n);
                counts[n] += recs[j].count;  // `count` voxels share this record
            }
        }
        std::sort(touched.begin(), touched.end());  // np.unique is sorted
        int mapto = r, maxmap = 0;
        long long myarea = 0;
        for (int n : touched) {
            if (counts[n] > maxmap && !id_in_spare(n)) {
                maxmap = counts[n];
                mapto = n;
                myarea = cache_area[r];
            }
        }
        for (int n : touched) counts[n] = 0;
        if (mapto != r) {  // regionmask[regionmask == r.label] = mapto
            const int rr = find(rep[r]), rm = find(rep[mapto]);
            uf[rr] = rm;
            setid[rm] = mapto;
            next[tail[mapto]] = head[r];
            tail[mapto] = tail[r];
            head[r] = 0;
            info.merged++;
        }
        const int mt = lv[mapto];
        if (cache_area[mapto] == (long long)maxsub[mt]) maxsub[mt] += (unsigned)myarea;  // :330-338
        cache_area[mapto] += myarea;                                                     // :339
    }
    lut.assign(R + 1, 0);
    for (int a = 1; a <= R; ++a) {
        uint8_t v = lobemap[setid[find(a)]];  // :341
        if (spare_label[v]) v = 0;            // :342
        lut[a] = v;
    }
}

// utils.py:355-356 / :390-404 on the region graph: the components of the mapped volume are unions of first-pass regions that are
// 26-adjacent and carry the same mapped label (the 6-adjacency is in the boundary records, the diagonal rest in `pairs`).
struct RegionGraph {
    // union-find over the regions; the root of a set is its SMALLEST id (= the region that holds the component's first voxel)
    std::vector<int> uf;
    std::vector<long long> carea;
    const std::vector<uint8_t>* lut = nullptr;
    int R = 0;
    int find(int a) {
        while (uf[a] != a) {
            uf[a] = uf[uf[a]];
            a = uf[a];
        }
        return a;
    }
    void unite(int a, int b) {
        const std::vector<uint8_t>& l = *lut;
        if (a < 1 || b < 1 || a > R || b > R || !l[a] || l[a] != l[b]) return;
        a = find(a);
        b = find(b);
        if (a == b) return;
        if (a < b) uf[b] = a;
        else uf[a] = b;
    }
    // phase A (needs only the boundary records: runs while the boxes / pairs kernels are still on the device): the 6-adjacency
    void begin(int R_, const std::vector<uint8_t>& lut_, const BoundaryRec* recs, size_t nrecs) {
        R = R_;
        lut = &lut_;
        uf.resize((size_t)R + 1);
        std::iota(uf.begin(), uf.end(), 0);
        for (size_t j = 0; j < nrecs; ++j)
            for (int k = 0; k < 6 && recs[j].nb[k]; ++k) unite(recs[j].atom, recs[j].nb[k]);
    }
    // phase B: the diagonal rest of the 26-adjacency, then for every label the largest component (area; on ties the component whose
    // first voxel comes LAST in raster order -- regions are numbered by their first voxel, so that is the component with the
    // largest smallest id: the rule of component_max's key).  keeplut[region] = its label when the region belongs to its label's
    // kept component, else 0; lbox[label] = that component's bounding box.
    void finish(const int* area, const unsigned long long* pairs, size_t npairs, const int* rbox, int dropped_label, std::vector<uint8_t>& keeplut, int lbox[256][6],
                bool kept[256]) {
        const std::vector<uint8_t>& l = *lut;
        for (size_t j = 0; j < npairs; ++j) unite((int)(pairs[j] >> 32), (int)(pairs[j] & 0xffffffffull));
        carea.assign((size_t)R + 1, 0);
        for (int a = 1; a <= R; ++a)
            if (l[a]) carea[find(a)] += area[a];
        int best[256];
        for (int i = 0; i < 256; ++i) {
            best[i] = 0;
            kept[i] = false;
            lbox[i][0] = lbox[i][1] = lbox[i][2] = 0x7fffffff;
            lbox[i][3] = lbox[i][4] = lbox[i][5] = -1;
        }
        for (int a = 1; a <= R; ++a) {
            if (!l[a] || uf[a] != a) continue;
            const int L = l[a], b = best[L];
            if (!b || carea[a] > carea[b] || (carea[a] == carea[b] && a > b)) best[L] = a;
        }
        keeplut.assign((size_t)R + 1, 0);
        for (int a = 1; a <= R; ++a) {
            const int L = l[a];
            if (!L || L == dropped_label || find(a) != best[L]) continue;
            keeplut[a] = (uint8_t)L;
            kept[L] = true;
            const int* b = rbox + 6 * (size_t)a;
            for (int k = 0; k < 3; ++k) lbox[L][k] = std::min(lbox[L][k], b[k]);
            for (int k = 3; k < 6; ++k) lbox[L][k] = std::max(lbox[L][k], b[k]);
        }
    }
};

int postprocess(lm_engine* e, uint8_t* lab, int N, int H, int W, const int* spare_p, int n_spare, int skip_below, int range_slot, bool* range_tripped) {
    if (range_tripped) *range_tripped = false;
    if (N <= 0 || H <= 0 || W <= 0) return LM_OK;
    const Dims d{N, H, W};
    const size_t nvox = d.nvox();
    if (nvox >= 0x7fffffffull) {
        set_error("volume too large for 32-bit voxel indices");
        return LM_ERR_INVALID;
    }
    hipStream_t s = e->stream;
    PostWorkspace& ws = e->post;
    PostInfo& info = e->post_info;
    info = PostInfo();
    // LM_POST_TIMING=1: host-side timestamps of this call on stderr (where the wall time of the post-processing goes: kernels,
    // the two read-backs, the merge replay) -- the call is preceded by a stream sync so that the forward is not counted
    static const bool timing = [] { const char* v = getenv("LM_POST_TIMING"); return v && v[0] == '1'; }();
    if (timing) (void)hipStreamSynchronize(s);
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    double t_enq1 = 0, t_sync1 = 0, t_replay = 0, t_enq2 = 0, t_sync2 = 0;
    std::vector<int> spare(spare_p, spare_p + (spare_p ? n_spare : 0));
    const size_t nb = rank_blocks(nvox);
    LM_TRY(ws.parent.reserve(nvox * 4));
    LM_TRY(ws.ids.reserve(nvox * 4));
    LM_TRY(ws.rank.reserve(nvox * 4));
    LM_TRY(ws.bgparent.reserve(nvox * 4));
    LM_TRY(ws.blockcnt.reserve((nb + 2) * 4));
    LM_TRY(ws.mapped.reserve(nvox));
    LM_TRY(ws.bg.reserve(nvox));
    LM_TRY(ws.out.reserve(nvox));
    LM_TRY(ws.scalars.reserve(4096));
    int* parent = ws.parent.as<int>();
    int* ids = ws.ids.as<int>();
    int* total_dev = ws.scalars.as<int>();
    unsigned* count_dev = ws.scalars.as<unsigned>() + 1;
    unsigned long long* best_dev = reinterpret_cast<unsigned long long*>(ws.scalars.as<char>() + 1024);

    // The second labelling (step 5) on the REGION GRAPH instead of the voxels (N > 1; LM_POST_GRAPH=0: the voxel form, A/B and test
    // hook): part 1 also delivers every region's bounding box and the pairs of regions that only touch diagonally, the host finds
    // the kept component of every label on the graph, and apply_lut / the second 26-connected labelling / component_max / the
    // bounding-box pass and their read-back do not run at all.  Nothing reads flat parents then, so the labelling skips that pass.
    static const bool graph_ok = [] { const char* v = getenv("LM_POST_GRAPH"); return !(v && v[0] == '0'); }();
    const bool graph = graph_ok && N > 1;
    unsigned* pcount_dev = ws.scalars.as<unsigned>() + 3;
    // ---- (1) skimage.measure.label (26-connected, multi-label), ids in raster order        utils.py:293
    {
        ProfScope ps(e, "post_ccl26_multilabel", (double)nvox * 13);
        LM_K(ccl_label(lab, parent, d, true, s, !graph));
    }
    {
        ProfScope ps(e, "post_rank_relabel", (double)nvox * 16);
        LM_K(ccl_rank(parent, ws.rank.as<int>(), ids, ws.blockcnt.as<int>(), total_dev, nvox, s, !graph));
    }
    // ---- (2)-(3) run BEFORE the host knows this volume's region count: the tables are sized from the previous volume (what it
    // needed + a margin; the kernels ignore ids / records beyond the capacity and the host repeats a pass whose guess was too
    // small), and ONE read-back delivers the region count, the record count, the f16 range flag of the forward that produced the
    // labels (range_slot) and the first `guess` entries of the three tables -- normally all of them.  (Round 2 synchronised four
    // times here: range flag, region count, record count, tables.)
    const bool want_range = range_slot >= 0 && range_tripped != nullptr && e->range_flag != nullptr && e->precision == 1 && !e->models[range_slot].force_f32;
    LM_TRY(ws.h_scalars.reserve(64));
    volatile int* hs = ws.h_scalars.as<int>();  // [0] regions, [1] records
    // LM_POST_SMALL_TABLES=1 (test hook): start every table and every read-back guess tiny, so that small test volumes take the
    // grow-and-repeat and the second-copy paths
    static const bool tiny = [] { const char* v = getenv("LM_POST_SMALL_TABLES"); return v && v[0] == '1'; }();
    int rcap = tiny ? 8 : std::max(16384, ws.last_regions + ws.last_regions / 2 + 1024);
    unsigned cap = tiny ? 8u : (unsigned)std::min<size_t>(nvox, std::max<size_t>(ws.recs.cap / sizeof(BoundaryRec), 1u << 20));
    unsigned pcap = tiny ? 8u : (unsigned)std::max<size_t>(ws.pairs.cap / 8, 1u << 18);
    int R = 0;
    unsigned nrec = 0, npair = 0;
    size_t g_p = 0, g_r_used = 0;  // what the speculative read-backs of the pairs / the boxes fetched
    std::vector<uint8_t> lut(1, 0);
    const int* area = nullptr;
    const uint8_t* lv = nullptr;
    const BoundaryRec* recs = nullptr;
    // The boxes / pairs kernels need nothing of steps 2-3.  Running them on the second forward lane's stream (idle here) BESIDE
    // region_stats / boundary_records instead of behind the first read-back was built and measured: the four latency- and atomic-bound
    // kernels slow each other down by what the overlap saves (first read-back 0.78 -> 0.88 ms, whole pass 1.62 -> 1.66 ms,
    // profiles/history/r05f_post_timing_side_stream_ab.log) -- off by default, LM_POST_SIDE=1 is the hook.
    static const bool side_ok = [] { const char* v = getenv("LM_POST_SIDE"); return v && v[0] == '1'; }();
    hipStream_t side = (graph && side_ok && e->stream2 != nullptr) ? e->stream2 : s;
    if (side != s && !ws.side_fork) {
        LM_HIP(hipEventCreateWithFlags(&ws.side_fork, hipEventDisableTiming));
        LM_HIP(hipEventCreateWithFlags(&ws.side_done, hipEventDisableTiming));
    }
    auto enqueue_graph_inputs = [&](size_t g_r) -> int {  // boxes, diagonal pairs and their speculative read-back, on `side`
        LM_TRY(ws.rbox.reserve(((size_t)rcap + 1) * 6 * 4));
        LM_TRY(ws.pairs.reserve((size_t)pcap * 8));
        {
            ProfScope ps(e, "post_region_boxes", (double)nvox * 4, side);
            LM_K(region_stats_box(ids, lab, nullptr, nullptr, ws.rbox.as<int>(), d, side, rcap));
        }
        LM_HIP(hipMemsetAsync(pcount_dev, 0, sizeof(unsigned), side));
        {
            ProfScope ps(e, "post_diag_pairs", (double)nvox * 2, side);
            LM_K(diag_pairs(lab, ids, d, ws.pairs.as<unsigned long long>(), pcount_dev, pcap, side));
        }
        g_p = tiny ? std::min<size_t>(pcap, 4) : std::min<size_t>(pcap, std::max<size_t>(16384, (size_t)ws.last_pairs + ws.last_pairs / 4 + 1024));
        LM_TRY(ws.h_rbox.reserve((g_r + 1) * 6 * 4));
        LM_TRY(ws.h_pairs.reserve(g_p * 8));
        LM_HIP(hipMemcpyAsync(const_cast<int*>(hs) + 3, pcount_dev, sizeof(unsigned), hipMemcpyDeviceToHost, side));
        LM_HIP(hipMemcpyAsync(ws.h_rbox.p, ws.rbox.p, (g_r + 1) * 6 * 4, hipMemcpyDeviceToHost, side));
        LM_HIP(hipMemcpyAsync(ws.h_pairs.p, ws.pairs.p, g_p * 8, hipMemcpyDeviceToHost, side));
        return LM_OK;
    };
    for (int attempt = 0;; ++attempt) {
        rcap = (int)std::min<size_t>((size_t)rcap, nvox);
        LM_TRY(ws.area.reserve(((size_t)rcap + 1) * 4));
        LM_TRY(ws.labval.reserve((size_t)rcap + 1));
        LM_TRY(ws.recs.reserve((size_t)cap * sizeof(BoundaryRec)));
        const size_t g_r = tiny ? (size_t)std::min(rcap, 4) : (size_t)std::min(rcap, std::max(4096, ws.last_regions + ws.last_regions / 4 + 256));
        if (side != s) {  // fork: the side stream sees the labelling, then works beside steps 2-3
            LM_HIP(hipEventRecord(ws.side_fork, s));
            LM_HIP(hipStreamWaitEvent(side, ws.side_fork, 0));
            LM_TRY(enqueue_graph_inputs(g_r));
            LM_HIP(hipEventRecord(ws.side_done, side));
        }
        // ---- (2) regionprops: area + label value                                           utils.py:298
        LM_HIP(hipMemsetAsync(ws.area.p, 0, ((size_t)rcap + 1) * 4, s));
        LM_HIP(hipMemsetAsync(ws.labval.p, 0, (size_t)rcap + 1, s));
        {
            ProfScope ps(e, "post_region_stats", (double)nvox * 5);
            LM_K(region_stats(ids, lab, ws.area.as<int>(), ws.labval.as<uint8_t>(), nvox, s, rcap));
        }
        // ---- (3) boundary voxels between regions (the only voxels the merge loop can ever count)
        LM_HIP(hipMemsetAsync(count_dev, 0, sizeof(unsigned), s));
        {
            ProfScope ps(e, "post_boundary_records", (double)nvox * 4);
            LM_K(boundary_records(ids, d, ws.recs.as<BoundaryRec>(), count_dev, cap, s));
        }
        // speculative read-back
        const size_t g_n = tiny ? std::min<size_t>(cap, 4) : std::min<size_t>(cap, std::max<size_t>(32768, (size_t)ws.last_records + ws.last_records / 4 + 1024));
        LM_TRY(ws.h_area.reserve((g_r + 1) * 4));
        LM_TRY(ws.h_labval.reserve(g_r + 1));
        LM_TRY(ws.h_recs.reserve(g_n * sizeof(BoundaryRec)));
        LM_HIP(hipMemcpyAsync(ws.h_scalars.p, total_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, s));  // total_dev, count_dev are neighbours
        if (want_range && attempt == 0) LM_HIP(hipMemcpyAsync(e->range_flag_host, e->range_flag, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        LM_HIP(hipMemcpyAsync(ws.h_area.p, ws.area.p, (g_r + 1) * 4, hipMemcpyDeviceToHost, s));
        LM_HIP(hipMemcpyAsync(ws.h_labval.p, ws.labval.p, g_r + 1, hipMemcpyDeviceToHost, s));
        LM_HIP(hipMemcpyAsync(ws.h_recs.p, ws.recs.p, g_n * sizeof(BoundaryRec), hipMemcpyDeviceToHost, s));
        if (graph) {
            // The region-graph inputs of step 5 -- every region's bounding box and the pairs of regions that only touch diagonally --
            // are not needed by the merge replay: the host waits for an event behind the FIRST read-back and replays the merge while
            // they are still running (on the side stream; without one: behind the first read-back on this stream).
            if (!ws.tables_ready) LM_HIP(hipEventCreateWithFlags(&ws.tables_ready, hipEventDisableTiming));
            LM_HIP(hipEventRecord(ws.tables_ready, s));
            if (side == s) LM_TRY(enqueue_graph_inputs(g_r));
        }
        t_enq1 = ms_now();
        if (graph) LM_HIP(hipEventSynchronize(ws.tables_ready));
        else LM_HIP(hipStreamSynchronize(s));
        t_sync1 = ms_now();
        if (want_range && attempt == 0) {
            LM_TRY(range_flag_consume(e, range_slot, range_tripped));
            if (*range_tripped) return LM_OK;
        }
        R = hs[0];
        nrec = (unsigned)hs[1];
        if (R > rcap || nrec > cap) {  // rare: a table was too small -- grow and repeat the passes
            rcap = std::max(rcap, R);
            cap = std::max(cap, nrec);
            if (graph) {  // (the boxes / pairs kernels of this attempt write tables that are about to grow)
                LM_HIP(hipStreamSynchronize(s));
                if (side != s) LM_HIP(hipStreamSynchronize(side));
            }
            continue;
        }
        if ((size_t)R > g_r || nrec > g_n) {  // the tables are complete on the device, the guess of what to fetch was short (first volume)
            LM_TRY(ws.h_area.reserve(((size_t)R + 1) * 4));
            LM_TRY(ws.h_labval.reserve((size_t)R + 1));
            LM_TRY(ws.h_recs.reserve(std::max<size_t>((size_t)nrec * sizeof(BoundaryRec), 64)));
            LM_HIP(hipMemcpyAsync(ws.h_area.p, ws.area.p, ((size_t)R + 1) * 4, hipMemcpyDeviceToHost, s));
            LM_HIP(hipMemcpyAsync(ws.h_labval.p, ws.labval.p, (size_t)R + 1, hipMemcpyDeviceToHost, s));
            if (nrec) LM_HIP(hipMemcpyAsync(ws.h_recs.p, ws.recs.p, (size_t)nrec * sizeof(BoundaryRec), hipMemcpyDeviceToHost, s));
            LM_HIP(hipStreamSynchronize(s));
        }
        g_r_used = g_r;
        break;
    }
    ws.last_regions = R;
    ws.last_records = nrec;
    info.regions = R;
    info.boundary_records = nrec;
    area = ws.h_area.as<int>();
    lv = ws.h_labval.as<uint8_t>();
    recs = ws.h_recs.as<BoundaryRec>();
    if (R > 0) {
        // ---- (4) the sequential merge on the region graph                                  utils.py:299-342
        const auto t0 = std::chrono::steady_clock::now();
        replay_merge(R, area, lv, recs, nrec, spare, skip_below, lut, info);
        info.host_replay_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    t_replay = ms_now();
    // utils.py:355 iterates `np.unique(outmask_mapped)[1:]`: it drops the SMALLEST value present -- the background 0 whenever the
    // mapped volume has a background voxel, and otherwise (a volume in which every voxel carries a kept label) the smallest LABEL,
    // which then does not appear in the result.  The mapped volume's non-zero count follows from the areas and the map.
    int dropped_label = 0;
    {
        long long nonzero = 0;
        int smallest = 256;
        for (int a = 1; a <= R; ++a)
            if (lut[a]) {
                nonzero += area[a];
                if (area[a] > 0) smallest = std::min<int>(smallest, lut[a]);
            }
        if (R > 0 && nonzero == (long long)nvox && smallest < 256) dropped_label = smallest;
    }
    if (graph) {
        // ---- (5) on the region graph: kept component of every label, its bounding box; then per label the hole fill on its box
        static thread_local RegionGraph rg;
        rg.begin(R, lut, recs, nrec);     // (still beside the boxes / pairs kernels)
        if (side != s) LM_HIP(hipEventSynchronize(ws.side_done));  // boxes + diagonal pairs
        else LM_HIP(hipStreamSynchronize(s));
        npair = (unsigned)hs[3];
        while (npair > pcap) {  // rare: the pair table was too small -- grow it and repeat that one pass
            pcap = npair;
            LM_TRY(ws.pairs.reserve((size_t)pcap * 8));
            LM_HIP(hipMemsetAsync(pcount_dev, 0, sizeof(unsigned), s));
            LM_K(diag_pairs(lab, ids, d, ws.pairs.as<unsigned long long>(), pcount_dev, pcap, s));
            LM_HIP(hipMemcpyAsync(const_cast<int*>(hs) + 3, pcount_dev, sizeof(unsigned), hipMemcpyDeviceToHost, s));
            LM_HIP(hipStreamSynchronize(s));
            npair = (unsigned)hs[3];
            g_p = 0;
        }
        if ((size_t)R > g_r_used || npair > g_p) {
            LM_TRY(ws.h_rbox.reserve(((size_t)R + 1) * 6 * 4));
            LM_TRY(ws.h_pairs.reserve(std::max<size_t>((size_t)npair * 8, 64)));
            LM_HIP(hipMemcpyAsync(ws.h_rbox.p, ws.rbox.p, ((size_t)R + 1) * 6 * 4, hipMemcpyDeviceToHost, s));
            if (npair) LM_HIP(hipMemcpyAsync(ws.h_pairs.p, ws.pairs.p, (size_t)npair * 8, hipMemcpyDeviceToHost, s));
            LM_HIP(hipStreamSynchronize(s));
        }
        ws.last_pairs = npair;
        std::vector<uint8_t> keeplut;
        int lbox[256][6];
        bool kept[256];
        rg.finish(area, ws.h_pairs.as<unsigned long long>(), npair, ws.h_rbox.as<int>(), dropped_label, keeplut, lbox, kept);
        const double t_graph = ms_now();
        keeplut.resize(std::max<size_t>(keeplut.size(), 1), 0);
        LM_TRY(ws.lut.reserve(keeplut.size()));
        LM_HIP(hipMemcpyAsync(ws.lut.p, keeplut.data(), keeplut.size(), hipMemcpyHostToDevice, s));
        uint8_t* out = ws.out.as<uint8_t>();
        LM_HIP(hipMemsetAsync(out, 0, nvox, s));
        const uint8_t* kl = ws.lut.as<uint8_t>();
        for (int label = 1; label < 256; ++label) {
            if (!kept[label]) continue;
            const int* bb = lbox[label];
            Box box;
            box.z0 = std::max(bb[0] - 1, 0);
            box.y0 = std::max(bb[1] - 1, 0);
            box.x0 = std::max(bb[2] - 1, 0) & ~3;  // x extent widened to multiples of 4 (a larger box is as exact): the row-wise labelling kernel applies
            box.d = Dims{std::min(bb[3] + 2, N) - box.z0, std::min(bb[4] + 2, H) - box.y0, std::min((bb[5] + 2 + 3) & ~3, W) - box.x0};
            const size_t nbox = box.d.nvox();
            LM_K(complement_of_lut_box(ids, kl, (uint8_t)label, d, box, ws.bg.as<uint8_t>(), s));
            {
                ProfScope ps(e, "post_ccl6_background", (double)nbox * 9);
                LM_K(ccl_label(ws.bg.as<uint8_t>(), ws.bgparent.as<int>(), box.d, false, s));
            }
            LM_K(flag_face_components(ws.bgparent.as<int>(), ws.rank.as<int>() /* free since the numbering */, box.d, s));
            {
                ProfScope ps(e, "post_fill_write", (double)nbox * 13);
                LM_K(fill_write_lut_box(ids, kl, ws.bgparent.as<int>(), ws.rank.as<int>(), (uint8_t)label, out, d, box, s));
            }
        }
        LM_HIP(hipMemcpyAsync(lab, out, nvox, hipMemcpyDeviceToDevice, s));
        // (keeplut lives on this stack frame until the copy above has read it: pageable source, the call returns after staging)
        if (timing) {
            const double t_enq3 = ms_now();
            (void)hipStreamSynchronize(s);
            fprintf(stderr, "lm_postprocess (region graph): part 1 enqueued %.3f | read-back done %.3f | merge replay done %.3f | components on the graph %.3f | "
                    "hole fills enqueued %.3f | all done %.3f ms (%d regions, %u records, %u diagonal pairs)\n", t_enq1, t_sync1, t_replay, t_graph, t_enq3, ms_now(), R, nrec, npair);
        }
        return LM_OK;
    }
    LM_TRY(ws.lut.reserve(lut.size()));
    LM_HIP(hipMemcpyAsync(ws.lut.p, lut.data(), lut.size(), hipMemcpyHostToDevice, s));
    uint8_t* mapped = ws.mapped.as<uint8_t>();
    {
        ProfScope ps(e, "post_apply_lut", (double)nvox * 5);
        LM_K(apply_lut(ids, ws.lut.as<uint8_t>(), mapped, nvox, s));
    }
    // ---- (5) per label: keep the largest 26-connected component, fill holes, write          utils.py:344-356
    {
        ProfScope ps(e, "post_ccl26_mapped", (double)nvox * 13);
        LM_K(ccl_label(mapped, parent, d, true, s));
    }
    {
        ProfScope ps(e, "post_component_max", (double)nvox * 9);
        LM_K(component_max(parent, mapped, ids /* reused: area per root */, best_dev, nvox, s));
    }
    // bounding boxes of the kept components: the hole fill of a label only has to look inside its box (post_kernels.h: Box).  The
    // roots to keep are derived from `best` on the device, so areas and boxes reach the host in one round trip.
    unsigned long long best[256];
    int keep_roots[256], bbox[256 * 6];
    if (N > 1) {
        LM_TRY(ws.bbox.reserve((256 + 256 * 6) * sizeof(int)));
        int* kr_dev = ws.bbox.as<int>();
        int* bbox_dev = kr_dev + 256;
        LM_K(keep_roots_init(best_dev, kr_dev, bbox_dev, s));
        {
            ProfScope ps(e, "post_component_bbox", (double)nvox * 5);
            LM_K(component_bboxes(parent, mapped, kr_dev, bbox_dev, d, s));
        }
        LM_HIP(hipMemcpyAsync(bbox, bbox_dev, sizeof bbox, hipMemcpyDeviceToHost, s));
    }
    LM_HIP(hipMemcpyAsync(best, best_dev, sizeof best, hipMemcpyDeviceToHost, s));
    t_enq2 = ms_now();
    LM_HIP(hipStreamSynchronize(s));
    t_sync2 = ms_now();
    uint8_t* out = ws.out.as<uint8_t>();
    LM_HIP(hipMemsetAsync(out, 0, nvox, s));
    for (int label = 0; label < 256; ++label) keep_roots[label] = (label && best[label]) ? (int)(unsigned)(best[label] & 0xffffffffull) : -1;
    for (int label = 1; label < 256; ++label) {
        if (!best[label] || label == dropped_label) continue;
        const int keep_root = keep_roots[label];
        if (N == 1) {  // skimage.morphology.area_closing(area_threshold=64): areas of whole components -> whole slice   utils.py:344-350
            LM_K(complement_of_component(parent, keep_root, ws.bg.as<uint8_t>(), nvox, s));
            {
                ProfScope ps(e, "post_ccl6_background", (double)nvox * 9);
                LM_K(ccl_label(ws.bg.as<uint8_t>(), ws.bgparent.as<int>(), d, false, s));
            }
            LM_K(flag_large_components(ws.bgparent.as<int>(), ids, 64, nvox, s));
            ProfScope ps(e, "post_fill_write", (double)nvox * 13);
            LM_K(fill_write(parent, keep_root, ws.bgparent.as<int>(), ids, (uint8_t)label, out, nvox, s));
            continue;
        }
        // fill_voids.fill: background not 6-connected to a face of the volume                    utils.py:352
        const int* bb = bbox + 6 * label;
        Box box;
        box.z0 = std::max(bb[0] - 1, 0);
        box.y0 = std::max(bb[1] - 1, 0);
        box.x0 = std::max(bb[2] - 1, 0) & ~3;  // x extent widened to multiples of 4 (a larger box is as exact): the row-wise labelling kernel applies
        box.d = Dims{std::min(bb[3] + 2, N) - box.z0, std::min(bb[4] + 2, H) - box.y0, std::min((bb[5] + 2 + 3) & ~3, W) - box.x0};
        const size_t nbox = box.d.nvox();
        LM_K(complement_of_component_box(parent, keep_root, d, box, ws.bg.as<uint8_t>(), s));
        {
            ProfScope ps(e, "post_ccl6_background", (double)nbox * 9);
            LM_K(ccl_label(ws.bg.as<uint8_t>(), ws.bgparent.as<int>(), box.d, false, s));
        }
        LM_K(flag_face_components(ws.bgparent.as<int>(), ids, box.d, s));
        {
            ProfScope ps(e, "post_fill_write", (double)nbox * 13);
            LM_K(fill_write_box(parent, keep_root, ws.bgparent.as<int>(), ids, (uint8_t)label, out, d, box, s));
        }
    }
    LM_HIP(hipMemcpyAsync(lab, out, nvox, hipMemcpyDeviceToDevice, s));
    if (timing) {
        const double t_enq3 = ms_now();
        (void)hipStreamSynchronize(s);
        fprintf(stderr, "lm_postprocess: part 1 enqueued %.3f | read-back 1 done %.3f | merge replay done %.3f | part 2 enqueued %.3f | read-back 2 done %.3f | "
                "part 3 enqueued %.3f | all done %.3f ms (%d regions, %u records)\n", t_enq1, t_sync1, t_replay, t_enq2, t_sync2, t_enq3, ms_now(), R, nrec);
    }
    return LM_OK;
}

// ------------------------------------------------------------------------------ utils.bbox_3D / keep_largest_connected_component
// utils.py:361-387: per axis the first / last index with a non-zero voxel, grown by `margin` and clipped: [zmin, zmax, ymin, ymax,
// xmin, xmax], maxima exclusive.  A mask without a non-zero voxel has no box (the reference raises IndexError at :377): all six -1.
int bbox3d(lm_engine* e, const uint8_t* mask, int N, int H, int W, int margin, int32_t out[6]) {
    for (int k = 0; k < 6; ++k) out[k] = -1;
    if (N <= 0 || H <= 0 || W <= 0) return LM_OK;
    const Dims d{N, H, W};
    if (d.nvox() >= 0x7fffffffull) {
        set_error("volume too large for 32-bit voxel indices");
        return LM_ERR_INVALID;
    }
    PostWorkspace& ws = e->post;
    LM_TRY(ws.scalars.reserve(4096));
    int* box_dev = ws.scalars.as<int>() + 16;
    {
        ProfScope ps(e, "mask_bbox", (double)d.nvox());
        LM_K(mask_bbox(mask, box_dev, d, e->stream));
    }
    int b[6];
    LM_HIP(hipMemcpyAsync(b, box_dev, sizeof b, hipMemcpyDeviceToHost, e->stream));
    LM_HIP(hipStreamSynchronize(e->stream));
    if (b[3] < 0) return LM_OK;  // empty
    const int dim[3] = {N, H, W};
    for (int k = 0; k < 3; ++k) {
        out[2 * k] = std::max(b[k] - margin, 0);                 // :378, :380
        out[2 * k + 1] = std::min(b[3 + k] + margin + 1, dim[k]);  // :379, :381
    }
    return LM_OK;
}

// utils.py:390-404: skimage.measure.label (full connectivity; voxels of different non-zero values are different regions),
// regionprops areas, `np.argsort(resizes)[-1] + 1`, `mask == max_region` -- in place, 0 / 1.  Equal areas: the reference leaves the
// choice to numpy's argsort; up to 16 regions that is an insertion sort (stable), whose last element is the LAST of the equal
// maxima -- regions are numbered by their first voxel in raster order, so that is the region whose first voxel comes last: the key
// (area << 32 | root) of component_max, the same rule postprocess() applies.  *area_out = that region's area (0: no region at
// all, the reference raises IndexError at :402; the mask is left untouched = all zero).
int keep_largest(lm_engine* e, uint8_t* mask, int N, int H, int W, long long* area_out) {
    if (area_out) *area_out = 0;
    if (N <= 0 || H <= 0 || W <= 0) return LM_OK;
    const Dims d{N, H, W};
    const size_t nvox = d.nvox();
    if (nvox >= 0x7fffffffull) {
        set_error("volume too large for 32-bit voxel indices");
        return LM_ERR_INVALID;
    }
    hipStream_t s = e->stream;
    PostWorkspace& ws = e->post;
    LM_TRY(ws.parent.reserve(nvox * 4));
    LM_TRY(ws.ids.reserve(nvox * 4));
    LM_TRY(ws.scalars.reserve(4096));
    unsigned long long* best_dev = reinterpret_cast<unsigned long long*>(ws.scalars.as<char>() + 1024);
    {
        ProfScope ps(e, "post_ccl26_multilabel", (double)nvox * 13);
        LM_K(ccl_label(mask, ws.parent.as<int>(), d, true, s));
    }
    {
        ProfScope ps(e, "post_component_max", (double)nvox * 9);
        LM_K(component_max(ws.parent.as<int>(), mask, ws.ids.as<int>(), best_dev, nvox, s));
    }
    unsigned long long best[256];
    LM_HIP(hipMemcpyAsync(best, best_dev, sizeof best, hipMemcpyDeviceToHost, s));
    LM_HIP(hipStreamSynchronize(s));
    unsigned long long top = 0;
    for (int label = 1; label < 256; ++label) top = std::max(top, best[label]);
    if (!top) return LM_OK;
    if (area_out) *area_out = (long long)(top >> 32);
    LM_K(component_mask(ws.parent.as<int>(), (int)(unsigned)(top & 0xffffffffull), mask, nvox, s));
    return LM_OK;
}

// ------------------------------------------------------------------------------ LMInferer.apply
namespace {

int inference(lm_engine* e, int slot, const void* vol, int dtype, int n, int h, int w, int batch, int vol_post, bool have_pre, uint8_t* out) {
    ApplyWorkspace& a = e->app;
    constexpr int R = 256;  // mask.py:166 resolution=[256, 256]
    if (slot < 0 || slot >= 4 || !e->models[slot].loaded) {
        set_error("model slot %d is empty", slot);
        return LM_ERR_NOMODEL;
    }
    LM_TRY(a.xf.reserve((size_t)n * R * R * 4));
    LM_TRY(a.bbox.reserve((size_t)n * 16));
    LM_TRY(a.labels.reserve((size_t)n * R * R));
    // Slices are independent up to the argmax (mask.py:166-187), so the volume is pre-processed in two pieces: the head (the
    // first batch of each forward lane) on the main stream, the tail on the copy stream while the head runs through the network
    // -- behind lm_apply_host's copy of the tail when the volume is still arriving (engine.h: head_slices / tail_ready).  The
    // batch loop is ONE loop over the whole volume: the lanes wait for the tail's pre-processing where they first need it, they
    // are never joined in between (round 2 ran head and tail as two loops: a drain of both lanes and a second body-mask latency).
    const int esz = dtype == LM_I16 ? 2 : ((dtype == LM_I32 || dtype == LM_F32) ? 4 : 8);
    const bool arriving = e->head_slices > 0 && e->head_slices < n;  // lm_apply_host is copying slices >= head_slices in
    const int lanes = (e->n_streams > 1 && e->stream2 != nullptr) ? 2 : 1;
    int head = arriving ? e->head_slices : (n > 2 * lanes * batch ? lanes * batch : n);
    if (have_pre) head = n;
    // per-kernel profiling pass (lm_profile_enable(e, 1) / 2: HIP events around every launch, bench.py's stage table): everything on
    // the main stream, so that an event pair measures its kernel and not the wait for compute units beside the forward
    if (e->prof.on && !e->prof.dominant_only && !arriving) head = n;
    // (a flag left behind by a forward that was never checked must not be attributed to this model)
    if (e->range_flag != nullptr) LM_HIP(hipMemsetAsync(e->range_flag, 0, sizeof(unsigned), e->stream));
    auto preprocess = [&](int s0, int ns, hipStream_t st) -> int {  // mask.py:166-168
        const char* v = reinterpret_cast<const char*>(vol) + (size_t)s0 * h * w * esz;
        BodyMaskParams bp{v, dtype, ns, h, w, a.bbox.as<int>() + (size_t)s0 * 4, nullptr};
        {
            ProfScope ps(e, "bodymask_bbox", (double)ns * 128 * 128 * esz, st);
            LM_K(launch_bodymask_bbox(bp, st));
        }
        ResampleParams rp{v, dtype, ns, h, w, a.bbox.as<int>() + (size_t)s0 * 4, R, R, nullptr, a.xf.as<float>() + (size_t)s0 * R * R};
        {
            ProfScope ps(e, "resample_norm", (double)ns * h * w * esz + (double)ns * R * R * 4, st);
            LM_K(launch_resample_norm(rp, st));
        }
        return LM_OK;
    };
    if (head < n && !e->copy_stream) {
        if (hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->tail_ready, hipEventDisableTiming) != hipSuccess) {
            set_error("creating the copy stream failed");
            return LM_ERR_DEVICE;
        }
    }
    if (head < n && !e->pre_fork) {
        if (hipEventCreateWithFlags(&e->pre_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e->pre_tail_done, hipEventDisableTiming) != hipSuccess) {
            set_error("creating the pre-processing events failed");
            return LM_ERR_DEVICE;
        }
    }
    if (head < n) LM_HIP(hipEventRecord(e->pre_fork, e->stream));  // everything the caller / the previous volume enqueued
    if (!have_pre) LM_TRY(preprocess(0, head, e->stream));
    // lm_apply_host with two lanes: the second lane's first batch arrives as a piece of its own (engine.h: mid_slices) and that lane's
    // stream pre-processes it itself, behind the piece's copy -- the tail (and its pre-processing on the copy stream) starts at `mid`
    const int mid = (arriving && !have_pre && lanes == 2 && e->mid_slices > head && e->mid_slices < n) ? e->mid_slices : 0;
    if (mid) {
        while (e->mid_enqueued.load(std::memory_order_acquire) == 0) std::this_thread::yield();
        if (e->mid_enqueued.load(std::memory_order_acquire) < 0) {
            set_error("lm_apply_host: copying the volume to the device failed");
            return LM_ERR_DEVICE;
        }
        LM_HIP(hipStreamWaitEvent(e->stream2, e->mid_ready, 0));
        LM_HIP(hipStreamWaitEvent(e->stream2, e->pre_fork, 0));
        LM_TRY(preprocess(head, mid - head, e->stream2));
    }
    const int tail0 = mid ? mid : head;
    auto gate = [&](hipEvent_t* ev) -> int {
        if (arriving) {
            while (e->tail_enqueued.load(std::memory_order_acquire) == 0) std::this_thread::yield();  // normally long done
            if (e->tail_enqueued.load(std::memory_order_acquire) < 0) {
                set_error("lm_apply_host: copying the volume to the device failed");
                return LM_ERR_DEVICE;
            }
        }
        // (the copy stream already holds lm_apply_host's copy of the tail, if any: stream order)
        LM_HIP(hipStreamWaitEvent(e->copy_stream, e->pre_fork, 0));
        LM_TRY(preprocess(tail0, n - tail0, e->copy_stream));
        LM_HIP(hipEventRecord(e->pre_tail_done, e->copy_stream));
        *ev = e->pre_tail_done;
        return LM_OK;
    };
    // mask.py:173-187
    if (head < n) LM_TRY(forward_batches(e, slot, a.xf.as<float>(), n, R, R, batch, a.labels.as<uint8_t>(), tail0, gate));
    else LM_TRY(forward_batches(e, slot, a.xf.as<float>(), n, R, R, batch, a.labels.as<uint8_t>()));
    // The f16 range flag of the forward passes is read back once per volume: inside the post-processing's first round trip when
    // there is one, on its own otherwise.  When it is set the model is now pinned to the exact-fp32 kernels: the whole volume
    // again (its pre-processed slices are all there).
    bool tripped = false;
    if (vol_post) LM_TRY(postprocess(e, a.labels.as<uint8_t>(), n, R, R, nullptr, 0, 3, slot, &tripped));  // mask.py:191-194
    else LM_TRY(forward_range_check(e, slot, &tripped));
    if (tripped) {
        LM_TRY(forward_guarded(e, slot, a.xf.as<float>(), n, R, R, batch, a.labels.as<uint8_t>(), nullptr));
        if (vol_post) LM_TRY(postprocess(e, a.labels.as<uint8_t>(), n, R, R, nullptr, 0, 3));
    }
    ReshapeParams rs{a.labels.as<uint8_t>(), a.bbox.as<int>(), out, n, R, R, h, w};  // mask.py:196-202
    {
        ProfScope ps(e, "reshape_mask", (double)n * ((double)R * R + (double)h * w));
        LM_K(launch_reshape_mask(rs, e->stream));
    }
    return LM_OK;
}

}  // namespace

int apply_volume(lm_engine* e, int slot, int fill_slot, const void* vol, int dtype, int n, int h, int w, int batch, int vol_post,
                 uint8_t* out) {
    if (n <= 0) return LM_OK;
    if (batch <= 0) batch = 20;
    if (dtype != LM_I16 && dtype != LM_I32 && dtype != LM_I64 && dtype != LM_F32 && dtype != LM_F64) {
        set_error("lm_apply: unsupported dtype code %d", dtype);
        return LM_ERR_INVALID;
    }
    LM_TRY(inference(e, slot, vol, dtype, n, h, w, batch, vol_post, false, out));
    if (fill_slot < 0) return LM_OK;
    // ---- LTRCLobes_R231 fusion (mask.py:223-232); the reference recomputes the identical pre-processing, we reuse it
    ApplyWorkspace& a = e->app;
    const size_t nvox = (size_t)n * h * w;
    LM_TRY(a.res_r.reserve(nvox));
    LM_TRY(inference(e, fill_slot, vol, dtype, n, h, w, batch, vol_post, true, a.res_r.as<uint8_t>()));
    LM_TRY(e->post.scalars.reserve(4096));
    unsigned* mx_dev = e->post.scalars.as<unsigned>() + 2;
    LM_K(volume_max(out, mx_dev, nvox, e->stream));
    unsigned mx = 0;
    LM_HIP(hipMemcpyAsync(&mx, mx_dev, sizeof mx, hipMemcpyDeviceToHost, e->stream));
    LM_HIP(hipStreamSynchronize(e->stream));
    const int spare = (int)((mx + 1) & 0xff);  // res_l.max() + 1 (uint8), mask.py:228
    {
        ProfScope ps(e, "fuse_labels", (double)nvox * 3);
        LM_K(fuse_labels(out, a.res_r.as<uint8_t>(), (uint8_t)spare, nvox, e->stream));
    }
    return postprocess(e, out, n, h, w, &spare, 1, 3);  // mask.py:232
}

}  // namespace lm
