// Host-side engine state behind the C ABI (include/lungmask_hip.h).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lungmask_hip.h"
#include "lm_platform.h"
#include "nn_kernels.h"

namespace lm {

void set_error(const char* fmt, ...);
const char* get_error();

#define LM_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            lm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return LM_ERR_DEVICE;                                                                     \
        }                                                                                             \
    } while (0)

#define LM_TRY(expr)                \
    do {                            \
        int _s = (expr);            \
        if (_s != LM_OK) return _s; \
    } while (0)

// Grow-only device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct ConvLayer {
    float *w = nullptr, *bias = nullptr, *bn_s = nullptr, *bn_t = nullptr;
    char* w_h3 = nullptr;  // split-f16 packing [taps][Cout][Cin/8][hi8|lo8] of w * 2^k
    float h3_acc_scale = 1.f;  // 2^-k
    int cin = 0, cout = 0, taps = 0;
    // Deferred BatchNorm shift of the split-f16 path (nn_engine.hip, "r-form"): a Conv->ReLU->BN layer stores r = relu(.)*s and
    // leaves its shift t to the consumers -- half of r is exactly zero, and zero operands cost the matrix pipes less power.
    // For THIS layer as a consumer of a tensor with per-channel shift T: bias_h3 = bias + sum_taps S[tap], S[tap][co] =
    // sum_ci w[co][ci][tap] T[ci]; corr_h3[mask][co] = sum over the taps that fall outside the image for border mask
    // (1 top, 2 bottom, 4 left, 8 right) of S[tap][co] (zero padding pads the TRUE activation, not r).
    float *bias_h3 = nullptr, *corr_h3 = nullptr;
    std::vector<float> h_bn_t;  // host copy of this layer's own shift (empty: no BatchNorm)
    // LM_H3_FOLD_SCALE: this layer's BatchNorm scale s[co] = 2^e[co] * m[co]: h_row_pow2 = the 2^e[co] its own packed weight row and
    // bias carry (exact), h_fold_s = the m[co] a consumer's weights carry for the stored channel (empty: 1)
    std::vector<float> h_fold_s, h_row_pow2;
};

// accumulator-chain limits the accuracy guard walks through (products per chain; 0 = no split: up to 9 216).  Measured on the 300-slice
// two-lane forward (profiles/r06j_ksplit_ab.log): +1 % / +4 % / +12 % time, heavy-tailed weights' rms error x 0.80 / 0.63 / 0.52
#ifdef LM_EMU_BUILD  // (the test emulator: every tier is another emulated forward)
constexpr int kChainTiers[] = {0, 1152};
constexpr int kNChainTiers = 2;
#else
constexpr int kChainTiers[] = {0, 4608, 2304, 1152};
constexpr int kNChainTiers = 4;
#endif

struct Model {
    bool loaded = false;
    int n_classes = 0;
    ConvLayer first;       // down_path.0.block.0 (Cin = 1)
    ConvLayer down[5][2];  // [0][0] unused (== first)
    ConvLayer up1x1[4];    // up_path.i.up.1
    ConvLayer upc[4][2];   // up_path.i.conv_block.block.{0,3}
    float *head_w = nullptr, *head_b = nullptr;
    float* head_b_h3 = nullptr;  // head bias + head_w . (shift of the last conv): the head reads an r-form tensor
    float* zeros_h3 = nullptr;   // 1024 zeros: the "shift" a deferred-shift producer applies
    float* ones_h3 = nullptr;    // 1024 ones: the "scale" of a producer whose scale its consumers carry (LM_H3_FOLD_SCALE)
    float* head_w_h3 = nullptr;  // head weights times the last conv's folded scale (== head_w without the fold)
    float* fc_pack = nullptr;    // first conv for the fused loader (ConvParamsH3::fc_c): w[9][64] | bias[64] | bn scale[64]
    std::vector<void*> allocs;
    // The split-f16 path stores activations as f16 pairs: a model whose activations left the f16 range (detected by the
    // kernels' range guard) is pinned to the exact-fp32 kernels from then on.
    bool force_f32 = false;
    // accuracy guard (nn_engine.hip: model_probe): split-f16 against exact-fp32 on one probe slice at load time
    bool probed = false, acc_pinned = false;
    float probe_err = -1.f;  // max |delta log-prob| of the probe; < 0: not probed (guard off, fp32 engine, or the range guard tripped first)
    // the guard's middle tiers: split-f16 with the 3x3 convs split along K so that no accumulator chain runs over chain_k products
    // (0: the fast form, one chain per output); probe_err is the chosen form's, probe_err_fast the single-chain form's
    int chain_k = 0;
    float probe_err_fast = -1.f;
    void release();
};

// Per-launch HIP-event timing on the engine stream (feeds bench.py's roofline block).
struct Profiler {
    bool on = false;
    bool per_layer = false;  // lm_profile_enable(e, 2): one entry per conv shape instead of per kernel
    bool dominant_only = false;  // lm_profile_enable(e, 3): events around the conv3x3 launches only (bench.py's timed region)
    bool skipped = false;
    bool timeline = false;  // lm_profile_enable(e, 4): keep every launch's span (lm_profile_timeline)
    int lane = 0;           // forward lane of the launches being recorded (set by forward())
    hipEvent_t t0 = nullptr;  // first recorded event since the last reset: origin of the timeline
    std::vector<lm_launch_span> spans;
    struct Rec {
        int kind;
        hipEvent_t a, b;
        double flops, bytes;
        int lane;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    std::vector<std::string> names;
    std::map<int, lm_kernel_stat> acc;
    int kind_id(const char* name);
    hipEvent_t get_event();
    void begin(hipStream_t s, int kind, double flops, double bytes);
    void end(hipStream_t s);
    void collect();
    void reset();
    void release();
};

struct NNWorkspace {
    DevBuf t1, t2, t3, cat[4], pool[4];
    DevBuf kpart;  // partial sums of a split-K 1x1 conv (ConvParamsH3::kpart)
    void release() {
        t1.release(); t2.release(); t3.release(); kpart.release();
        for (int i = 0; i < 4; ++i) { cat[i].release(); pool[i].release(); }
    }
};

// Grow-only pinned host buffer (D2H of the region table / boundary records at PCIe speed instead of through a bounce buffer).
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct PostWorkspace {
    HostBuf h_area, h_labval, h_recs, h_scalars, h_rbox, h_pairs;
    // what the previous volume needed: sizes the tables and the speculative read-back of the next one (postprocess)
    int last_regions = 0;
    unsigned last_records = 0, last_pairs = 0;
    DevBuf parent, ids, rank, bgparent, blockcnt, area, labval, recs, lut, mapped, bg, out, scalars, bbox;
    DevBuf rbox, pairs;  // region-graph form of the second labelling: bounding box per region, diagonal adjacency pairs
    hipEvent_t tables_ready = nullptr;  // behind the first read-back (the host replays the merge while the boxes / pairs kernels run)
    hipEvent_t side_fork = nullptr, side_done = nullptr;  // the boxes / pairs kernels on the second lane's (idle) stream beside part 1
    void release() {
        parent.release(); ids.release(); rank.release(); bgparent.release(); blockcnt.release(); area.release(); labval.release();
        recs.release(); lut.release(); mapped.release(); bg.release(); out.release(); scalars.release(); bbox.release();
        rbox.release(); pairs.release();
        if (tables_ready) (void)hipEventDestroy(tables_ready);
        if (side_fork) (void)hipEventDestroy(side_fork);
        if (side_done) (void)hipEventDestroy(side_done);
        tables_ready = side_fork = side_done = nullptr;
        h_area.release(); h_labval.release(); h_recs.release(); h_scalars.release(); h_rbox.release(); h_pairs.release();
    }
};

struct ApplyWorkspace {
    DevBuf vol, xf, bbox, labels, res_r, out;
    void release() { vol.release(); xf.release(); bbox.release(); labels.release(); res_r.release(); out.release(); }
};

// Slab-sharded post-processing (slab_engine.hip): state between the exchange points of lm_slab_*.
struct SlabState {
    int rank = 0, world = 1, n = 0, H = 0, W = 0, z0 = 0, n_total = 0, skip_below = 3;
    std::vector<int> spare;
    int phase = -1;          // next lm_slab_step to run (0..5); -1 = idle
    uint8_t* lab = nullptr;  // the caller's slab (device); receives the result
    int n1 = 0, n2 = 0;
    // region-graph form (slab_engine.hip): the second labelling runs on the atom graph inside the first table merge -- four exchanges
    // instead of six; keep_ids = the dense atom ids keeplut is indexed by (first labelling in the graph form, second one otherwise)
    bool graph = false;
    const int* keep_ids = nullptr;
    std::vector<int> labels, n3;  // label values with a kept component; atoms of each label's background labelling
    DevBuf ids2, ids3, first, flags, edges, pack, keeplut, holelut;
    long long pending = 0;  // ints of `pack` this rank contributes to the next exchange
    bool pending_uniform = false;  // the pending length is the same on every rank by construction (face planes)
    std::vector<std::vector<int>> tables;  // host copies of the gathered tables
    void release() {
        ids2.release(); ids3.release(); first.release(); flags.release(); edges.release(); pack.release(); keeplut.release(); holelut.release();
    }
};

// What the last lm_postprocess_dev call saw (reported by bench.py next to the timing: it is data dependent).
struct PostInfo {
    long long regions = 0, boundary_records = 0, merged = 0, processed = 0;
    double host_replay_ms = 0;
};

// One persistent helper thread per engine (lm_apply_host's second copy lane): started on first use, parked on a condition
// variable between calls, joined by lm_engine_destroy.  run() hands it ONE job; wait() blocks until that job has returned.
// start() reports failure instead of throwing (a std::thread that cannot be created must not unwind through the C ABI): the
// caller then takes its single-threaded path.
class HostHelper {
public:
    bool start() noexcept {
        if (started_) return true;
        try {
            th_ = std::thread([this] { loop(); });
        } catch (...) {
            return false;
        }
        started_ = true;
        return true;
    }
    void run(std::function<void()> job) {
        std::lock_guard<std::mutex> g(m_);
        job_ = std::move(job);
        busy_ = true;
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !busy_; });
    }
    void stop() {
        if (!started_) return;
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
            cv_.notify_all();
        }
        th_.join();
        started_ = false;
    }

private:
    void loop() {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [this] { return quit_ || (busy_ && job_); });
            if (quit_) return;
            std::function<void()> job = std::move(job_);
            job_ = nullptr;
            g.unlock();
            job();
            g.lock();
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool started_ = false, busy_ = false, quit_ = false;
};

}  // namespace lm

struct lm_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    lm::Model models[4];
    lm::NNWorkspace nn;
    // second forward lane: consecutive slice batches alternate between two streams / workspaces so that the
    // tail of one batch's kernels (tile quantisation, launch gaps, the small bandwidth-bound kernels) is filled
    // by the other batch's work
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    lm::NNWorkspace nn2;
    int n_streams = 2;
    lm::PostWorkspace post;
    lm::ApplyWorkspace app;
    lm::PostInfo post_info;
    lm::SlabState slab;
    lm::Profiler prof;
    int precision = 1;  // 1 (default): split-f16 3-product; 0: exact fp32 matrix ops (lm_set_precision)
    int fusion = 11;    // (default: bits 0 and 3; bit 2, split-K, measured slower -- profiles/history/r04n_splitk_1x1_ab.log) lm_set_fusion: bit 0 first conv in conv 2's loader, bit 1 bilinear x2 in the decoder conv's loader, bit 2 split-K 1x1, bit 3 head in the last conv's epilogue
    char* zero_page = nullptr;
    // lm_apply_host: the volume arrives in two pieces (the first two batches on the main stream, the rest on copy_stream while
    // they are computed); `tail_ready` is recorded behind the second piece, the hot path waits for it before it touches
    // slices >= head_slices.  head_slices == 0: the whole volume is already resident (lm_apply_dev).
    hipStream_t copy_stream = nullptr;
    hipEvent_t tail_ready = nullptr;
    hipEvent_t pre_fork = nullptr, pre_tail_done = nullptr;  // the tail's pre-processing on copy_stream (post_engine.hip: inference)
    int head_slices = 0;
    // set by the copying thread once tail_ready has been recorded (1) or the copy failed (-1): the hot path must not enqueue
    // its wait on an event that has not been recorded yet (that would be a no-op)
    std::atomic<int> tail_enqueued{0};
    // Round 5: with two forward lanes the head is ONE batch (the main stream's first) and the second lane's first batch -- slices
    // [head_slices, mid_slices) -- arrives as a piece of its own in front of the tail: `mid_ready` is recorded behind it and lane 1's
    // stream pre-processes it itself, so the first kernel starts after 10 MB have crossed the link instead of 21.  mid_slices == 0: no such piece.
    hipEvent_t mid_ready = nullptr;
    int mid_slices = 0;
    std::atomic<int> mid_enqueued{0};
    lm::HostHelper helper;
    unsigned* range_flag = nullptr;       // device word of the f16 range guard (ConvParamsH3::range_flag)
    unsigned* range_flag_host = nullptr;  // pinned copy
    // lm_pipe_* (capi.hip): volumes queued through one engine -- two resident input / result buffers, the copy-in of volume i + 1 and
    // the copy-back of volume i on streams of their own beside the hot path of their neighbours
    struct Pipe {
        lm::DevBuf vol[2], out[2];
        lm::HostBuf stage[2];  // page-locked staging of a pageable input volume
        bool staged[2] = {false, false};
        hipStream_t up_stream = nullptr, out_stream = nullptr;
        hipEvent_t uploaded[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
        bool copied_valid[2] = {false, false};
        void release() {
            for (int k = 0; k < 2; ++k) {
                vol[k].release();
                out[k].release();
                stage[k].release();
                staged[k] = false;
                if (uploaded[k]) (void)hipEventDestroy(uploaded[k]);
                if (done[k]) (void)hipEventDestroy(done[k]);
                if (copied[k]) (void)hipEventDestroy(copied[k]);
                uploaded[k] = done[k] = copied[k] = nullptr;
                copied_valid[k] = false;
            }
            if (up_stream) (void)hipStreamDestroy(up_stream);
            if (out_stream) (void)hipStreamDestroy(out_stream);
            up_stream = out_stream = nullptr;
        }
    } pipe;
    // lm_dist_* (dist_rccl.hip): RCCL communicator of this engine's rank; world 0 = none, world 1 = no library involved
    void* dist_comm = nullptr;
    int dist_rank = 0, dist_world = 0;
};

namespace lm {
int model_load(lm_engine* e, int slot, const lm_tensor* tensors, int n);
int forward(lm_engine* e, int slot, const float* x, int B, int H, int W, uint8_t* labels, float* logp, int lane = 0);
// n slices in batches of `batch` (mask.py:173-187), batches alternating over the engine's forward lanes
// `gate` (optional): called ONCE on the host right before the first batch that starts at or beyond slice `gate_slice` is
// enqueued; it returns an event (or nullptr) that every lane waits for before it runs such a batch.  This is how the tail of a
// volume -- still being copied in / pre-processed on another stream -- joins the batch loop without a join of the lanes.
int forward_batches(lm_engine* e, int slot, const float* x, int n, int H, int W, int batch, uint8_t* labels, int gate_slice = -1,
                    const std::function<int(hipEvent_t*)>& gate = nullptr);
// The two above + the f16 range guard: waits for the forward, and when a split-f16 forward reported activations beyond the f16
// range, pins the model to the exact-fp32 kernels and runs the forward again.  What the C ABI and lm_apply call.
int forward_guarded(lm_engine* e, int slot, const float* x, int n, int H, int W, int batch, uint8_t* labels, float* logp);
// the check alone, for callers that enqueue several forward_batches first (post_engine.hip: inference)
int forward_range_check(lm_engine* e, int slot, bool* tripped);
int model_probe(lm_engine* e, int slot);
// range_slot >= 0: the f16 range flag of the forward that produced `lab` is read back in the SAME round trip as the region
// count (instead of a synchronisation of its own before the call); when it is set, *range_tripped = true, the model of that
// slot is pinned to the exact-fp32 kernels and nothing else is done (the caller repeats forward + post-processing).
int postprocess(lm_engine* e, uint8_t* lab, int N, int H, int W, const int* spare, int n_spare, int skip_below, int range_slot = -1,
                bool* range_tripped = nullptr);
// bookkeeping of a range flag value that has already been read back into e->range_flag_host
int range_flag_consume(lm_engine* e, int slot, bool* tripped);
struct BoundaryRec;
void replay_merge(int R, const int* area, const uint8_t* lv, const BoundaryRec* recs, size_t nrecs, const std::vector<int>& spare, int skip_below,
                  std::vector<uint8_t>& lut, PostInfo& info);
int slab_begin(lm_engine* e, uint8_t* lab, int n, int h, int w, int rank, int world, int z0, int n_total, const int* spare, int n_spare, int skip_below);
int slab_emit(lm_engine* e, int32_t* dst);
int slab_step(lm_engine* e, const int32_t* gathered, long long stride, const long long* lens);
// utils.py:361-387 / :390-404 as seams of their own (post_engine.hip)
int bbox3d(lm_engine* e, const uint8_t* mask, int N, int H, int W, int margin, int32_t out[6]);
int keep_largest(lm_engine* e, uint8_t* mask, int N, int H, int W, long long* area_out);
int apply_volume(lm_engine* e, int slot, int fill_slot, const void* vol_dev, int dtype, int n, int h, int w, int batch_size,
                 int volume_postprocessing, uint8_t* out_dev);
}  // namespace lm
