// extern "C" surface of liblungmask_hip.so (include/lungmask_hip.h).
#include <chrono>
#include <cstring>
#include <thread>

#include "engine.h"
#include "post_kernels.h"
#include "pre_kernels.h"

using namespace lm;

// every entry point that launches work first makes the engine's device current (callers such as
// torch.distributed workers may have switched devices on this thread)
#define LM_DEVICE(e) LM_HIP(hipSetDevice((e)->device))

extern "C" {

const char* lm_last_error(void) { return get_error(); }
const char* lm_version(void) { return "lungmask_hip 0.1 (gfx950)"; }
int lm_is_gpu_build(void) { return LM_IS_GPU_BUILD; }

int lm_engine_create(lm_engine** out, int device_id) {
    if (!out) return LM_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    LM_HIP(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) {
        set_error("device %d not present (%d devices)", device_id, n);
        return LM_ERR_INVALID;
    }
    LM_HIP(hipSetDevice(device_id));
    lm_engine* e = new lm_engine();
    e->device = device_id;
    hipError_t err = hipStreamCreate(&e->stream);
    if (err != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(err));
        delete e;
        return LM_ERR_DEVICE;
    }
    // The second forward lane must sit on a different hardware queue than the first, or the two batches serialise.
    // Same-priority streams share a small round-robin pool of queues (whether two of them collide depends on how many
    // streams the process created before -- e.g. torch.distributed's -- which cost 15 % in one start-up order); a stream of
    // another priority class always gets a queue of its own.  Lowest priority: the lane fills gaps, it never preempts.
    hipError_t err2;
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        err2 = hipStreamCreateWithPriority(&e->stream2, hipStreamDefault, least);
    }
    if (err2 != hipSuccess || hipEventCreate(&e->ev_fork) != hipSuccess || hipEventCreate(&e->ev_join) != hipSuccess) {
        set_error("creating the second forward lane failed");
        e->stream2 = nullptr;
        e->n_streams = 1;
    }
    *out = e;
    return LM_OK;
}

void lm_engine_destroy(lm_engine* e) {
    if (!e) return;
    e->helper.stop();
    (void)lm_dist_destroy(e);
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    e->prof.release();
    for (auto& m : e->models) m.release();
    if (e->stream2) (void)hipStreamSynchronize(e->stream2);
    if (e->zero_page) (void)hipFree(e->zero_page);
    if (e->range_flag_host) (void)hipHostFree(e->range_flag_host);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->tail_ready) (void)hipEventDestroy(e->tail_ready);
    if (e->mid_ready) (void)hipEventDestroy(e->mid_ready);
    if (e->pre_fork) (void)hipEventDestroy(e->pre_fork);
    if (e->pre_tail_done) (void)hipEventDestroy(e->pre_tail_done);
    e->nn.release();
    e->nn2.release();
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->stream2) (void)hipStreamDestroy(e->stream2);
    e->post.release();
    e->slab.release();
    e->app.release();
    e->pipe.release();
    (void)hipStreamDestroy(e->stream);
    delete e;
}

int lm_engine_sync(lm_engine* e) {
    if (!e) return LM_ERR_INVALID;
    if (e->stream2) LM_HIP(hipStreamSynchronize(e->stream2));
    LM_HIP(hipStreamSynchronize(e->stream));
    return LM_OK;
}

int lm_dev_alloc(lm_engine* e, void** dev_ptr, size_t bytes) {
    if (!e || !dev_ptr) return LM_ERR_INVALID;
    LM_HIP(hipSetDevice(e->device));
    LM_HIP(hipMalloc(dev_ptr, bytes ? bytes : 16));
    return LM_OK;
}
int lm_dev_free(lm_engine* e, void* dev_ptr) {
    if (!e) return LM_ERR_INVALID;
    LM_HIP(hipStreamSynchronize(e->stream));
    LM_HIP(hipFree(dev_ptr));
    return LM_OK;
}
int lm_host_alloc(lm_engine* e, void** host_ptr, size_t bytes) {
    if (!e || !host_ptr) return LM_ERR_INVALID;
    LM_HIP(hipSetDevice(e->device));
    LM_HIP(hipHostMalloc(host_ptr, bytes ? bytes : 16, 0));
    return LM_OK;
}
int lm_host_free(lm_engine* e, void* host_ptr) {
    if (e) LM_HIP(hipStreamSynchronize(e->stream));
    if (host_ptr) LM_HIP(hipHostFree(host_ptr));
    return LM_OK;
}
// whether [p, p + bytes) is page-locked memory the runtime knows (lm_host_alloc, hipHostMalloc, a registered range)
static bool host_range_is_pinned(const void* p, size_t bytes) {
#ifdef LM_EMU_BUILD
    (void)p;
    (void)bytes;
    return false;
#else
    hipPointerAttribute_t a0{}, a1{};
    if (hipPointerGetAttributes(&a0, p) != hipSuccess) {
        (void)hipGetLastError();  // ordinary memory: "invalid value", not an error of ours
        return false;
    }
    if (bytes > 1 && hipPointerGetAttributes(&a1, reinterpret_cast<const char*>(p) + bytes - 1) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a0.type == hipMemoryTypeHost && (bytes <= 1 || a1.type == hipMemoryTypeHost);
#endif
}
int lm_copy_h2d(lm_engine* e, void* dev_dst, const void* host_src, size_t bytes) {
    if (!e) return LM_ERR_INVALID;
    LM_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, e->stream));
    LM_HIP(hipStreamSynchronize(e->stream));
    return LM_OK;
}
int lm_copy_d2h(lm_engine* e, void* host_dst, const void* dev_src, size_t bytes) {
    if (!e) return LM_ERR_INVALID;
    LM_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, e->stream));
    LM_HIP(hipStreamSynchronize(e->stream));
    return LM_OK;
}

int lm_model_load(lm_engine* e, int slot, const lm_tensor* tensors, int n_tensors) {
    if (!e) return LM_ERR_INVALID;
    LM_HIP(hipSetDevice(e->device));
    LM_TRY(model_load(e, slot, tensors, n_tensors));
    return model_probe(e, slot);  // accuracy guard (a split-f16 engine only)
}

int lm_model_probe_error(lm_engine* e, int slot, float* err_out) {
    if (!e || slot < 0 || slot >= 4 || !e->models[slot].loaded || !err_out) return LM_ERR_NOMODEL;
    *err_out = e->models[slot].probe_err;
    return e->models[slot].acc_pinned ? 1 : (e->models[slot].chain_k > 0 ? 2 : 0);
}
void* lm_engine_stream(lm_engine* e) { return e ? reinterpret_cast<void*>(e->stream) : nullptr; }

int lm_model_chain_limit(lm_engine* e, int slot) {
    if (!e || slot < 0 || slot >= 4 || !e->models[slot].loaded) return LM_ERR_NOMODEL;
    return e->models[slot].chain_k;
}

int lm_model_classes(lm_engine* e, int slot) {
    if (!e || slot < 0 || slot >= 4 || !e->models[slot].loaded) return LM_ERR_NOMODEL;
    return e->models[slot].n_classes;
}

int lm_model_precision(lm_engine* e, int slot) {
    if (!e || slot < 0 || slot >= 4 || !e->models[slot].loaded) return LM_ERR_NOMODEL;
    return (e->precision == 1 && !e->models[slot].force_f32) ? 1 : 0;
}

int lm_set_precision(lm_engine* e, int mode) {
    if (!e || (mode != 0 && mode != 1)) {
        set_error("lm_set_precision: mode must be 0 (exact fp32) or 1 (split-f16)");
        return LM_ERR_INVALID;
    }
    e->precision = mode;
    if (mode == 1) {  // models loaded while the engine was on the exact kernels have not met the accuracy guard yet
        LM_DEVICE(e);
        for (int slot = 0; slot < 4; ++slot) LM_TRY(model_probe(e, slot));
    }
    return LM_OK;
}

int lm_set_streams(lm_engine* e, int n) {
    if (!e || n < 1 || n > 2) {
        set_error("lm_set_streams: 1 or 2");
        return LM_ERR_INVALID;
    }
    e->n_streams = (n == 2 && e->stream2) ? 2 : 1;
    return LM_OK;
}

int lm_set_fusion(lm_engine* e, int mask) {
    if (!e || mask < 0 || mask > 15) {
        set_error("lm_set_fusion: mask is a combination of bits 0..3");
        return LM_ERR_INVALID;
    }
    e->fusion = mask;
    return LM_OK;
}

int lm_forward_batches_dev(lm_engine* e, int slot, const float* x_dev, int n, int h, int w, int batch_size, uint8_t* labels_dev) {
    if (!e || !x_dev || !labels_dev || n < 0) return LM_ERR_INVALID;
    LM_DEVICE(e);
    return forward_guarded(e, slot, x_dev, n, h, w, batch_size <= 0 ? 20 : batch_size, labels_dev, nullptr);
}

int lm_forward_dev(lm_engine* e, int slot, const float* x_dev, int b, int h, int w, uint8_t* labels_dev, float* logp_dev) {
    if (!e || !x_dev) return LM_ERR_INVALID;
    LM_DEVICE(e);
    return forward_guarded(e, slot, x_dev, b, h, w, 0, labels_dev, logp_dev);
}

int lm_preprocess_dev(lm_engine* e, const void* vol_dev, int dtype, int n, int h, int w, int oh, int ow, int32_t* bbox_dev,
                      float* x_f32_dev, int16_t* x_i16_dev, uint8_t* bmask_dev) {
    if (!e || !vol_dev || !bbox_dev || n < 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0) {
        set_error("lm_preprocess_dev: bad arguments");
        return LM_ERR_INVALID;
    }
    if (dtype != LM_I16 && dtype != LM_I32 && dtype != LM_I64 && dtype != LM_F32 && dtype != LM_F64) {
        set_error("lm_preprocess_dev: unsupported dtype code %d", dtype);
        return LM_ERR_INVALID;
    }
    if (x_i16_dev && (dtype == LM_F32 || dtype == LM_F64)) {
        set_error("lm_preprocess_dev: the int16 resample output only exists for integer volumes");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    BodyMaskParams bp{vol_dev, dtype, n, h, w, bbox_dev, bmask_dev};
    const double vox = (double)n * h * w;
    const int esz = dtype == LM_I16 ? 2 : ((dtype == LM_I32 || dtype == LM_F32) ? 4 : 8);
    e->prof.begin(e->stream, e->prof.kind_id("bodymask_bbox"), 0, (double)n * 128 * 128 * esz);
    hipError_t err = launch_bodymask_bbox(bp, e->stream);
    e->prof.end(e->stream);
    if (err != hipSuccess) {
        set_error("bodymask launch failed: %s", hipGetErrorString(err));
        return LM_ERR_DEVICE;
    }
    if (x_f32_dev || x_i16_dev) {
        ResampleParams rp{vol_dev, dtype, n, h, w, bbox_dev, oh, ow, x_i16_dev, x_f32_dev};
        e->prof.begin(e->stream, e->prof.kind_id("resample_norm"), 0, vox * esz + (double)n * oh * ow * 4);
        err = launch_resample_norm(rp, e->stream);
        e->prof.end(e->stream);
        if (err != hipSuccess) {
            set_error("resample launch failed: %s", hipGetErrorString(err));
            return LM_ERR_DEVICE;
        }
    }
    return LM_OK;
}

int lm_reshape_mask_dev(lm_engine* e, const uint8_t* mask_dev, const int32_t* bbox_dev, int n, int mh, int mw, int h, int w,
                        uint8_t* out_dev) {
    if (!e || !mask_dev || !bbox_dev || !out_dev || n < 0 || mh <= 0 || mw <= 0 || h <= 0 || w <= 0) {
        set_error("lm_reshape_mask_dev: bad arguments");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    ReshapeParams p{mask_dev, bbox_dev, out_dev, n, mh, mw, h, w};
    e->prof.begin(e->stream, e->prof.kind_id("reshape_mask"), 0, (double)n * ((double)mh * mw + (double)h * w));
    hipError_t err = launch_reshape_mask(p, e->stream);
    e->prof.end(e->stream);
    if (err != hipSuccess) {
        set_error("reshape_mask launch failed: %s", hipGetErrorString(err));
        return LM_ERR_DEVICE;
    }
    return LM_OK;
}

int lm_reorient_dev(lm_engine* e, const void* in_dev, void* out_dev, int elem_size, int n0, int n1, int n2, int64_t s0, int64_t s1,
                    int64_t s2, int64_t base) {
    if (!e || !in_dev || !out_dev || n0 < 0 || n1 < 0 || n2 < 0 || base < 0 ||
        (elem_size != 1 && elem_size != 2 && elem_size != 4 && elem_size != 8)) {
        set_error("lm_reorient_dev: bad arguments");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    ReorientParams p{in_dev, out_dev, elem_size, n0, n1, n2, (long long)s0, (long long)s1, (long long)s2, (long long)base};
    e->prof.begin(e->stream, e->prof.kind_id("reorient"), 0, 2.0 * elem_size * (double)n0 * n1 * n2);
    hipError_t err = launch_reorient(p, e->stream);
    e->prof.end(e->stream);
    if (err != hipSuccess) {
        set_error("reorient launch failed: %s", hipGetErrorString(err));
        return LM_ERR_DEVICE;
    }
    return LM_OK;
}

int lm_postprocess_dev(lm_engine* e, uint8_t* lab_dev, int n, int h, int w, const int* spare, int n_spare, int skip_below) {
    if (!e || !lab_dev || n < 0 || h <= 0 || w <= 0 || n_spare < 0) {
        set_error("lm_postprocess_dev: bad arguments");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    return postprocess(e, lab_dev, n, h, w, spare, n_spare, skip_below);
}

int lm_bbox3d_dev(lm_engine* e, const uint8_t* mask_dev, int n, int h, int w, int margin, int32_t bbox_out[6]) {
    if (!e || (!mask_dev && n > 0) || !bbox_out || n < 0 || h <= 0 || w <= 0 || margin < 0) {
        set_error("lm_bbox3d_dev: bad arguments");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    return bbox3d(e, mask_dev, n, h, w, margin, bbox_out);
}

int lm_keep_largest_dev(lm_engine* e, uint8_t* mask_dev, int n, int h, int w, int64_t* area_out) {
    if (!e || (!mask_dev && n > 0) || n < 0 || h <= 0 || w <= 0) {
        set_error("lm_keep_largest_dev: bad arguments");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    long long a = 0;
    const int rc = keep_largest(e, mask_dev, n, h, w, &a);
    if (area_out) *area_out = (int64_t)a;
    return rc;
}

int lm_slab_begin(lm_engine* e, uint8_t* lab_slab_dev, int n, int h, int w, int rank, int world, int z0, int n_total, const int* spare,
                  int n_spare, int skip_below) {
    if (!e || !lab_slab_dev || n_spare < 0) {
        set_error("lm_slab_begin: bad arguments");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    return slab_begin(e, lab_slab_dev, n, h, w, rank, world, z0, n_total, spare, n_spare, skip_below);
}

int64_t lm_slab_pending(lm_engine* e) { return (e && e->slab.phase >= 0) ? (int64_t)e->slab.pending : -1; }

int lm_slab_pending_uniform(lm_engine* e) { return (e && e->slab.phase >= 0 && e->slab.pending_uniform) ? 1 : 0; }

int lm_slab_emit(lm_engine* e, int32_t* dst_dev) {
    if (!e) return LM_ERR_INVALID;
    LM_DEVICE(e);
    return slab_emit(e, dst_dev);
}

int lm_slab_step(lm_engine* e, const int32_t* gathered_dev, int64_t stride, const int64_t* lens) {
    if (!e) return LM_ERR_INVALID;
    LM_DEVICE(e);
    static_assert(sizeof(long long) == sizeof(int64_t), "");
    return slab_step(e, gathered_dev, (long long)stride, reinterpret_cast<const long long*>(lens));
}

int lm_postprocess_info(lm_engine* e, int64_t info[5]) {
    if (!e || !info) return LM_ERR_INVALID;
    info[0] = e->post_info.regions;
    info[1] = e->post_info.boundary_records;
    info[2] = e->post_info.processed;
    info[3] = e->post_info.merged;
    info[4] = (int64_t)(e->post_info.host_replay_ms * 1000.0);
    return LM_OK;
}

// res_l.max() of mask.py:228 on the voxels this engine holds (a rank's slab: the pipeline combines the ranks' maxima)
int lm_label_max_dev(lm_engine* e, const uint8_t* lab_dev, size_t nvox, int* max_out) {
    if (!e || (!lab_dev && nvox) || !max_out) {
        set_error("lm_label_max_dev: bad arguments");
        return LM_ERR_INVALID;
    }
    *max_out = 0;
    if (nvox == 0) return LM_OK;
    LM_DEVICE(e);
    LM_TRY(e->post.scalars.reserve(4096));
    unsigned* mx_dev = e->post.scalars.as<unsigned>() + 2;
    hipError_t err = volume_max(lab_dev, mx_dev, nvox, e->stream);
    if (err != hipSuccess) {
        set_error("volume_max failed: %s", hipGetErrorString(err));
        return LM_ERR_DEVICE;
    }
    unsigned mx = 0;
    LM_HIP(hipMemcpyAsync(&mx, mx_dev, sizeof mx, hipMemcpyDeviceToHost, e->stream));
    LM_HIP(hipStreamSynchronize(e->stream));
    *max_out = (int)mx;
    return LM_OK;
}

// mask.py:229-230 with a spare label the caller determined (multi-GPU: max over ALL ranks' slabs + 1)
int lm_fuse_spare_dev(lm_engine* e, uint8_t* res_l_dev, const uint8_t* res_r_dev, size_t nvox, int spare) {
    if (!e || ((!res_l_dev || !res_r_dev) && nvox) || spare < 0 || spare > 255) {
        set_error("lm_fuse_spare_dev: bad arguments");
        return LM_ERR_INVALID;
    }
    if (nvox == 0) return LM_OK;
    LM_DEVICE(e);
    e->prof.begin(e->stream, e->prof.kind_id("fuse_labels"), 0, (double)nvox * 3);
    const hipError_t err = fuse_labels(res_l_dev, res_r_dev, (uint8_t)spare, nvox, e->stream);
    e->prof.end(e->stream);
    if (err != hipSuccess) {
        set_error("fuse_labels failed: %s", hipGetErrorString(err));
        return LM_ERR_DEVICE;
    }
    return LM_OK;
}

int lm_fuse_dev(lm_engine* e, uint8_t* res_l_dev, const uint8_t* res_r_dev, size_t nvox, int* spare_out) {
    if (!e || !res_l_dev || !res_r_dev) return LM_ERR_INVALID;
    int mx = 0;
    LM_TRY(lm_label_max_dev(e, res_l_dev, nvox, &mx));
    const int spare = (mx + 1) & 0xff;  // res_l.max() + 1 in uint8 arithmetic (mask.py:228)
    LM_TRY(lm_fuse_spare_dev(e, res_l_dev, res_r_dev, nvox, spare));
    if (spare_out) *spare_out = spare;
    return LM_OK;
}

int lm_apply_dev(lm_engine* e, int slot, int fill_slot, const void* vol_dev, int dtype, int n, int h, int w, int batch_size,
                 int volume_postprocessing, uint8_t* out_dev) {
    if (!e || !vol_dev || !out_dev || n < 0 || h <= 0 || w <= 0) {
        set_error("lm_apply_dev: bad arguments");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    return apply_volume(e, slot, fill_slot, vol_dev, dtype, n, h, w, batch_size, volume_postprocessing, out_dev);
}

int lm_apply_host(lm_engine* e, int slot, int fill_slot, const void* vol_host, int dtype, int n, int h, int w, int batch_size,
                  int volume_postprocessing, uint8_t* out_host) {
    return lm_apply_host_ex(e, slot, fill_slot, vol_host, dtype, n, h, w, batch_size, volume_postprocessing, out_host, 0u);
}

int lm_apply_host_ex(lm_engine* e, int slot, int fill_slot, const void* vol_host, int dtype, int n, int h, int w, int batch_size,
                     int volume_postprocessing, uint8_t* out_host, unsigned flags) {
    if (!e || !vol_host || !out_host || n < 0 || h <= 0 || w <= 0 || (flags & ~(unsigned)LM_APPLY_OUT_SCRATCH)) {
        set_error("lm_apply_host: bad arguments");
        return LM_ERR_INVALID;
    }
    // LM_APPLY_OUT_SCRATCH: the output array's previous contents are of no value to the caller.  The helper thread then fills it
    // with zeros while the network runs and, at the end, only the slab of slices x rows that holds a label is copied back (a
    // strided copy) -- a lung mask is mostly background: 31 of 79 MB on the bench phantom.
    const bool scratch = (flags & LM_APPLY_OUT_SCRATCH) != 0;
    const int esz = dtype == LM_I16 ? 2 : ((dtype == LM_I32 || dtype == LM_F32) ? 4 : ((dtype == LM_I64 || dtype == LM_F64) ? 8 : 0));
    if (!esz) {
        set_error("lm_apply_host: unsupported dtype code %d", dtype);
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    const size_t nvox = (size_t)n * h * w, slice = (size_t)h * w;
    LM_TRY(e->app.vol.reserve(nvox * esz));
    LM_TRY(e->app.out.reserve(nvox));
    // mask.py:178-186 moves every batch to the device and back on its own; here the boundary is crossed once per volume, in two
    // pieces.  The head (two batches) goes in on the main stream, and while it is pre-processed and run through the network the
    // engine's helper thread (persistent: started on the first call, parked between calls) (a) copies the tail in on a second
    // stream and (b) faults in every page of the caller's output array WITHOUT changing it (each page's first byte is read and
    // written back), so that the copy back at the end does not take the page faults of a freshly allocated buffer.  The
    // caller's array is only ever overwritten by the final copy of a successful call; on any error it is left as it was.
    if (batch_size <= 0) batch_size = 20;
    // two lanes: the head is the main stream's first batch, the second lane's first batch follows as a piece of its own (`mid`)
    const bool two_lanes = e->n_streams > 1 && e->stream2 != nullptr;
    const int mid = (two_lanes && n > 2 * batch_size) ? 2 * batch_size : 0;
    const int head = mid ? batch_size : (two_lanes ? 2 : 1) * batch_size;
    bool split = n > head;
    if (split && !e->copy_stream) {
        if (hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->tail_ready, hipEventDisableTiming) != hipSuccess) {
            set_error("lm_apply_host: creating the copy stream failed");
            return LM_ERR_DEVICE;
        }
    }
    const bool out_pinned = host_range_is_pinned(out_host, nvox);
    const bool threaded = e->helper.start();  // false: no thread could be created -- one copy on the main stream, no page touching
    if (!threaded) split = false;
    if (split && mid && !e->mid_ready && hipEventCreateWithFlags(&e->mid_ready, hipEventDisableTiming) != hipSuccess) {
        set_error("lm_apply_host: creating the copy events failed");
        return LM_ERR_DEVICE;
    }
    const int tail0 = (split && mid) ? mid : head;  // first slice of the tail piece
    e->mid_enqueued.store(0, std::memory_order_release);
    static const bool timing = [] { const char* v = getenv("LM_HOST_TIMING"); return v && v[0] == '1'; }();  // stderr breakdown of this call
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    double t_tail = 0, t_touch = 0;
    e->tail_enqueued.store(0, std::memory_order_release);
    if (threaded)
        e->helper.run([&, split] {
            hipError_t terr = hipSuccess;
            if (split) {
                terr = hipSetDevice(e->device);
                // (a pageable source makes this call return only when the data has been staged: that is why it lives on this thread)
                if (terr == hipSuccess && tail0 > head) {  // lane 1's first batch, in front of the tail
                    terr = hipMemcpyAsync(reinterpret_cast<char*>(e->app.vol.p) + (size_t)head * slice * esz, reinterpret_cast<const char*>(vol_host) + (size_t)head * slice * esz,
                                          (size_t)(tail0 - head) * slice * esz, hipMemcpyHostToDevice, e->copy_stream);
                    if (terr == hipSuccess) terr = hipEventRecord(e->mid_ready, e->copy_stream);
                    e->mid_enqueued.store(terr == hipSuccess ? 1 : -1, std::memory_order_release);
                }
                if (terr == hipSuccess)
                    terr = hipMemcpyAsync(reinterpret_cast<char*>(e->app.vol.p) + (size_t)tail0 * slice * esz, reinterpret_cast<const char*>(vol_host) + (size_t)tail0 * slice * esz,
                                          (size_t)(n - tail0) * slice * esz, hipMemcpyHostToDevice, e->copy_stream);
                if (terr == hipSuccess) terr = hipEventRecord(e->tail_ready, e->copy_stream);
            }
            if (terr != hipSuccess) e->mid_enqueued.store(-1, std::memory_order_release);
            e->tail_enqueued.store(terr == hipSuccess ? 1 : -1, std::memory_order_release);
            t_tail = ms_since(t_start);
            // (Pinning the caller's array here with hipHostRegister makes the copy back 0.4 ms faster -- tools/host_pin_probe.py -- but an
            // array from the malloc heap shares its first and last page with other objects and with the runtime's own cache of pinned
            // user ranges; one whole-suite run aborted inside the next model load after such a registration, so the pages are touched.)
            if (scratch) {
                memset(out_host, 0, nvox);
            } else if (!out_pinned) {  // (a page-locked block -- lm_host_alloc, what LMInferer hands out -- has no faults to take)
                volatile uint8_t* o = out_host;
                for (size_t i = 0; i < nvox; i += 4096) o[i] = o[i];
                if (nvox) o[nvox - 1] = o[nvox - 1];
            }
            t_touch = ms_since(t_start);
        });
    hipError_t err = hipMemcpyAsync(e->app.vol.p, vol_host, (size_t)(split ? head : n) * slice * esz, hipMemcpyHostToDevice, e->stream);
    const double t_head = ms_since(t_start);
    int rc = LM_OK;
    if (err != hipSuccess) {
        set_error("lm_apply_host: host-to-device copy failed: %s", hipGetErrorString(err));
        rc = LM_ERR_DEVICE;
    } else {
        e->head_slices = split ? head : 0;
        e->mid_slices = split ? mid : 0;
        rc = apply_volume(e, slot, fill_slot, e->app.vol.p, dtype, n, h, w, batch_size, volume_postprocessing, e->app.out.as<uint8_t>());
        e->head_slices = 0;
        e->mid_slices = 0;
    }
    const double t_apply = ms_since(t_start);
    if (threaded) e->helper.wait();
    const double t_join = ms_since(t_start);
    if (rc != LM_OK) {
        // the tail copy may still be reading the caller's volume / writing the staging buffer: nothing of this call may be in
        // flight when it returns
        if (split && e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
        (void)hipStreamSynchronize(e->stream);
        return rc;
    }
    hipError_t back = hipSuccess;
    double t_ext = 0;
    int ext[4] = {0, n, 0, h};
    if (scratch && threaded) {
        // where the labels are: one pass over the result on the device (tens of microseconds), four ints back, then ONE strided copy
        // of rows [y0, y1) of slices [z0, z1) -- whole image rows, so every piece of the copy is (y1 - y0) * w contiguous bytes
        if (e->post.scalars.reserve(4096) != LM_OK) return LM_ERR_ALLOC;
        int* ext_dev = e->post.scalars.as<int>() + 64;
        back = launch_label_extent(e->app.out.as<uint8_t>(), n, h, w, ext_dev, e->stream);
        if (back == hipSuccess) back = hipMemcpyAsync(ext, ext_dev, sizeof ext, hipMemcpyDeviceToHost, e->stream);
        if (back == hipSuccess) back = hipStreamSynchronize(e->stream);
        t_ext = ms_since(t_start);
        if (back == hipSuccess && ext[1] > ext[0] && ext[3] > ext[2]) {
            const size_t off = ((size_t)ext[0] * h + ext[2]) * w;
            back = hipMemcpy2DAsync(out_host + off, slice, e->app.out.as<uint8_t>() + off, slice, (size_t)(ext[3] - ext[2]) * w, (size_t)(ext[1] - ext[0]),
                                    hipMemcpyDeviceToHost, e->stream);
        }
    } else {
        // (without the helper thread nothing was zero-filled: the whole volume is copied)
        back = hipMemcpyAsync(out_host, e->app.out.p, nvox, hipMemcpyDeviceToHost, e->stream);
    }
    if (back == hipSuccess) back = hipStreamSynchronize(e->stream);
    if (back != hipSuccess) {
        set_error("lm_apply_host: device-to-host copy failed: %s", hipGetErrorString(back));
        return LM_ERR_DEVICE;
    }
    if (timing)
        fprintf(stderr, "lm_apply_host: head H2D returned %.2f ms | helper: tail enqueued %.2f, pages %s %.2f | hot path returned %.2f | joined %.2f | extent known %.2f "
                "(slices %d..%d rows %d..%d) | D2H done %.2f ms\n", t_head, t_tail, scratch ? "zeroed" : "touched", t_touch, t_apply, t_join, t_ext, ext[0], ext[1], ext[2], ext[3],
                ms_since(t_start));
    return LM_OK;
}

// ---- volumes queued through one engine (include/lungmask_hip.h: lm_pipe_*) ---------------------------------------------------
static int pipe_init(lm_engine* e) {
    lm_engine::Pipe& q = e->pipe;
    if (q.up_stream) return LM_OK;
    // The two copy streams get the HIGHEST priority class: same-priority streams share a small pool of hardware queues (see
    // lm_engine_create), and a copy that lands on the queue of a forward lane holds that lane's kernels behind its barrier packet
    // for the length of the copy -- measured: +2.9 ms per volume with ordinary streams (profiles/r06e_async_probe.log).  The streams
    // carry nothing but copies, so the priority preempts nobody.
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&q.up_stream, hipStreamNonBlocking, greatest) != hipSuccess ||
        hipStreamCreateWithPriority(&q.out_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
        set_error("lm_pipe: creating the copy streams failed");
        return LM_ERR_DEVICE;
    }
    for (int k = 0; k < 2; ++k)
        if (hipEventCreateWithFlags(&q.uploaded[k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&q.done[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q.copied[k], hipEventDisableTiming) != hipSuccess) {
            set_error("lm_pipe: creating the events failed");
            return LM_ERR_DEVICE;
        }
    return LM_OK;
}

int lm_pipe_upload(lm_engine* e, int k, const void* vol_host, size_t bytes) {
    if (!e || (k != 0 && k != 1) || (!vol_host && bytes)) {
        set_error("lm_pipe_upload: bad arguments");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    LM_TRY(pipe_init(e));
    lm_engine::Pipe& q = e->pipe;
    LM_TRY(q.vol[k].reserve(bytes ? bytes : 16));
    // A pageable source goes through a page-locked staging block of this buffer FIRST (a plain memcpy on this thread, which is why
    // the call has a host thread of its own), then ONE truly asynchronous copy: handing the runtime a pageable pointer makes it stage
    // the data itself, piece by piece, inside calls that the hot path's kernel launches on the other thread then queue behind.
    const void* src = vol_host;
    if (bytes && !host_range_is_pinned(vol_host, bytes)) {
        if (q.staged[k]) LM_HIP(hipEventSynchronize(q.uploaded[k]));  // the previous copy out of this staging block
        if (q.stage[k].reserve(bytes) == LM_OK) {
            memcpy(q.stage[k].p, vol_host, bytes);
            src = q.stage[k].p;
            q.staged[k] = true;
        }  // (no page-locked memory left: the runtime stages it)
    }
    if (bytes) LM_HIP(hipMemcpyAsync(q.vol[k].p, src, bytes, hipMemcpyHostToDevice, q.up_stream));
    LM_HIP(hipEventRecord(q.uploaded[k], q.up_stream));
    return LM_OK;
}

int lm_pipe_apply(lm_engine* e, int k, int slot, int fill_slot, int dtype, int n, int h, int w, int batch_size, int volume_postprocessing) {
    if (!e || (k != 0 && k != 1) || n < 0 || h <= 0 || w <= 0) {
        set_error("lm_pipe_apply: bad arguments");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    LM_TRY(pipe_init(e));
    lm_engine::Pipe& q = e->pipe;
    const int esz = dtype == LM_I16 ? 2 : ((dtype == LM_I32 || dtype == LM_F32) ? 4 : ((dtype == LM_I64 || dtype == LM_F64) ? 8 : 0));
    const size_t nvox = (size_t)n * h * w;
    if (!esz || q.vol[k].cap < nvox * esz) {
        set_error("lm_pipe_apply: buffer %d does not hold a volume of this size and dtype (lm_pipe_upload first)", k);
        return LM_ERR_INVALID;
    }
    LM_TRY(q.out[k].reserve(nvox ? nvox : 16));
    // the hot path reads vol[k] behind its copy-in and overwrites out[k] behind the copy-back of the volume that used it last
    LM_HIP(hipStreamWaitEvent(e->stream, q.uploaded[k], 0));
    if (q.copied_valid[k]) LM_HIP(hipStreamWaitEvent(e->stream, q.copied[k], 0));
    LM_TRY(apply_volume(e, slot, fill_slot, q.vol[k].p, dtype, n, h, w, batch_size, volume_postprocessing, q.out[k].as<uint8_t>()));
    LM_HIP(hipEventRecord(q.done[k], e->stream));
    return LM_OK;
}

int lm_pipe_download(lm_engine* e, int k, uint8_t* out_host, size_t bytes) {
    if (!e || (k != 0 && k != 1) || (!out_host && bytes) || !e->pipe.out_stream || e->pipe.out[k].cap < bytes) {
        set_error("lm_pipe_download: bad arguments (lm_pipe_apply first)");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    lm_engine::Pipe& q = e->pipe;
    LM_HIP(hipStreamWaitEvent(q.out_stream, q.done[k], 0));
    if (bytes) LM_HIP(hipMemcpyAsync(out_host, q.out[k].p, bytes, hipMemcpyDeviceToHost, q.out_stream));
    LM_HIP(hipEventRecord(q.copied[k], q.out_stream));
    q.copied_valid[k] = true;
    return LM_OK;
}

int lm_pipe_wait(lm_engine* e, int k) {
    if (!e || (k != 0 && k != 1) || !e->pipe.out_stream) {
        set_error("lm_pipe_wait: bad arguments");
        return LM_ERR_INVALID;
    }
    LM_DEVICE(e);
    if (e->pipe.copied_valid[k]) LM_HIP(hipEventSynchronize(e->pipe.copied[k]));
    return LM_OK;
}

int lm_profile_enable(lm_engine* e, int on) {
    if (!e) return LM_ERR_INVALID;
    e->prof.on = on != 0;
    e->prof.per_layer = on == 2 || on == 4;
    e->prof.dominant_only = on == 3;
    e->prof.timeline = on == 4;
    return LM_OK;
}
int lm_profile_timeline(lm_engine* e, lm_launch_span* out, int cap) {
    if (!e) return LM_ERR_INVALID;
    e->prof.collect();
    const int n = (int)e->prof.spans.size();
    for (int i = 0; i < n && i < cap && out; ++i) out[i] = e->prof.spans[i];
    return n;
}
int lm_profile_reset(lm_engine* e) {
    if (!e) return LM_ERR_INVALID;
    e->prof.reset();
    return LM_OK;
}
int lm_profile_read(lm_engine* e, lm_kernel_stat* out, int cap) {
    if (!e) return LM_ERR_INVALID;
    e->prof.collect();
    int n = 0;
    for (auto& kv : e->prof.acc) {
        if (n < cap && out) out[n] = kv.second;
        ++n;
    }
    return n;
}

}  // extern "C"
