// extern "C" surface of liblungmask_hip.so (include/lungmask_hip.h).
#include "engine.h"

using namespace lm;

extern "C" {

const char* lm_last_error(void) { return get_error(); }
const char* lm_version(void) { return "lungmask_hip 0.1 (gfx950)"; }
int lm_is_gpu_build(void) {
#ifdef LM_EMU_BUILD
    return 0;
#else
    return 1;
#endif
}

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
    *out = e;
    return LM_OK;
}

void lm_engine_destroy(lm_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    e->prof.release();
    for (auto& m : e->models) m.release();
    e->nn.t1.release();
    e->nn.t2.release();
    e->nn.t3.release();
    for (int i = 0; i < 4; ++i) {
        e->nn.cat[i].release();
        e->nn.pool[i].release();
    }
    (void)hipStreamDestroy(e->stream);
    delete e;
}

int lm_engine_sync(lm_engine* e) {
    if (!e) return LM_ERR_INVALID;
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
    return model_load(e, slot, tensors, n_tensors);
}
int lm_model_classes(lm_engine* e, int slot) {
    if (!e || slot < 0 || slot >= 4 || !e->models[slot].loaded) return LM_ERR_NOMODEL;
    return e->models[slot].n_classes;
}

int lm_forward_dev(lm_engine* e, int slot, const float* x_dev, int b, int h, int w, uint8_t* labels_dev, float* logp_dev) {
    if (!e || !x_dev) return LM_ERR_INVALID;
    return forward(e, slot, x_dev, b, h, w, labels_dev, logp_dev);
}

int lm_profile_enable(lm_engine* e, int on) {
    if (!e) return LM_ERR_INVALID;
    e->prof.on = on != 0;
    return LM_OK;
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
