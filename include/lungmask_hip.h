/* lungmask_hip.h -- C ABI of liblungmask_hip.so, the MI355X (gfx950) engine that
 * replaces the hot path of JoHof/lungmask: `LMInferer.apply()` ->
 * `LMInferer._inference()` (reference lungmask/mask.py:141-232) and everything
 * it calls in lungmask/resunet.py and lungmask/utils.py.
 *
 * The reference has no FFI seam (it is pure Python over torch/scipy/skimage);
 * the entry points below are what a ctypes binding inside the reference's
 * `mask.py` / `utils.py` would call -- one per reference call site, cited on
 * each declaration.  INTEGRATION.md shows that binding.
 *
 * Conventions
 *   - plain C: pointers + sizes only; no torch / numpy types.
 *   - every function returns 0 on success or a negative lm_status; the message
 *     is available from lm_last_error() (thread-local).
 *   - "dev" pointers are HBM addresses (hipMalloc / torch.cuda tensors /
 *     lm_dev_alloc); "host" pointers are ordinary memory.
 *   - an lm_engine owns one device, one HIP stream and its workspaces; it is not
 *     thread-safe, distinct engines are independent.
 *   - volumes are C-contiguous [n][h][w]; labels are uint8.
 */
#ifndef LUNGMASK_HIP_H
#define LUNGMASK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lm_engine lm_engine;

typedef enum lm_status {
    LM_OK = 0,
    LM_ERR_INVALID = -1,  /* bad argument / shape / state            */
    LM_ERR_DEVICE = -2,   /* HIP runtime error (message has details) */
    LM_ERR_NOMODEL = -3,  /* model slot empty                        */
    LM_ERR_ALLOC = -4
} lm_status;

/* dtype codes for input volumes (numpy mode of mask.py:153-155 accepts any dtype) */
enum { LM_I16 = 0, LM_I32 = 1, LM_F32 = 2, LM_F64 = 3, LM_U8 = 4, LM_U16 = 5, LM_I64 = 6 };

/* One named tensor of a torch state_dict (fp32, C-contiguous, host memory). */
typedef struct lm_tensor {
    const char* name; /* e.g. "down_path.0.block.0.weight" */
    const float* data;
    int64_t numel;
} lm_tensor;

/* ---- lifetime ---------------------------------------------------------------- */
const char* lm_last_error(void);
const char* lm_version(void);
/* 1 if the library was built for the GPU (always, for the shipped library). */
int lm_is_gpu_build(void);
/* mask.py:118-134 (device selection) -> explicit device ordinal. */
int lm_engine_create(lm_engine** out, int device_id);
void lm_engine_destroy(lm_engine* e);
int lm_engine_sync(lm_engine* e);

/* The HIP stream (hipStream_t, as void*) every stage call of this engine enqueues on.  A host that owns other GPU work or
 * collectives (lungmask_amd/pipeline.py: torch.distributed over RCCL) orders them against the engine's kernels by enqueuing on
 * / waiting for this stream instead of synchronising the device.  NULL for a NULL engine. */
void* lm_engine_stream(lm_engine* e);

/* ---- one rank per engine: RCCL communicator behind the C ABI --------------------------------------
 * The reference has no multi-GPU form; the slice-sharded pipeline (lungmask_amd/pipeline.py) needs exactly one
 * collective -- an equal-size all-gather of device buffers (label shards, slab-protocol planes/tables, the result) -- and with
 * these entry points it needs nothing else from the host's GPU stack.  RCCL is bound at run time (dlopen; torch's copy when
 * one is already mapped); a world of one initialised WITHOUT an id involves no library at all (its all-gather is a copy), with
 * an id it is a real RCCL communicator of one rank.
 *   rank 0: lm_dist_unique_id(id);  -> hand the 128 bytes to every rank by any side channel (a TCP store, a file, MPI)
 *   every rank: lm_dist_init(e, rank, world, id);   ...   lm_dist_all_gather(e, send, recv, bytes);   ...   lm_dist_destroy(e);
 * lm_dist_all_gather: recv_dev holds world * bytes, rank r's contribution at r * bytes (send_dev may be that very slot: in
 * place).  It is ENQUEUED on the engine's stream (lm_engine_stream): ordered after the engine's earlier kernels and before its
 * later ones, the host does not wait. */
#define LM_DIST_ID_BYTES 128
int lm_dist_unique_id(uint8_t* id_out /* [LM_DIST_ID_BYTES] */);
int lm_dist_init(lm_engine* e, int rank, int world, const uint8_t* id /* [LM_DIST_ID_BYTES]; NULL allowed when world == 1 */);
int lm_dist_rank(lm_engine* e);  /* < 0 without a communicator */
int lm_dist_world(lm_engine* e); /* < 0 without a communicator */
int lm_dist_all_gather(lm_engine* e, const void* send_dev, void* recv_dev, size_t bytes);
int lm_dist_destroy(lm_engine* e);

/* ---- device memory helpers (so a binding needs no other GPU runtime) ---------- */
int lm_dev_alloc(lm_engine* e, void** dev_ptr, size_t bytes);
int lm_dev_free(lm_engine* e, void* dev_ptr);
int lm_copy_h2d(lm_engine* e, void* dev_dst, const void* host_src, size_t bytes);
int lm_copy_d2h(lm_engine* e, void* host_dst, const void* dev_src, size_t bytes);
/* Page-locked host memory for volumes / label arrays that cross the boundary (lm_apply_host): the device copies straight into and
 * out of it at link speed, and a result block that is used again has no page faults to take.  The block belongs to the caller --
 * the engine keeps no record of it and lm_engine_destroy does not free it; lm_host_free accepts e == NULL. */
int lm_host_alloc(lm_engine* e, void** host_ptr, size_t bytes);
int lm_host_free(lm_engine* e, void* host_ptr);

/* ---- model (mask.py:38-68 get_model) ------------------------------------------ */
/* Loads a U-Net state_dict into `slot` (0..3).  n_classes is taken from
 * "last.bias" exactly as mask.py:56 does; the always-present-but-unused
 * residual_* tensors and num_batches_tracked are accepted and ignored. */
int lm_model_load(lm_engine* e, int slot, const lm_tensor* tensors, int n_tensors);
int lm_model_classes(lm_engine* e, int slot);

/* Arithmetic of the convolutions: 1 (default) = split-f16 3-product on v_mfma_f32_32x32x16_f16
 * (values carried as hi/lo f16 pairs, fp32 accumulate; ~2^-22 relative, i.e. fp32-class:
 * measured max |log-prob error| vs the reference <= 1.7e-4); 0 = exact fp32 matrix ops
 * (v_mfma_f32_32x32x2_f32, 16x lower peak). */
int lm_set_precision(lm_engine* e, int mode);
/* The arithmetic the NEXT forward of `slot` will use: 1 split-f16, 0 exact fp32.  The split form stores activations as
 * f16 pairs; its kernels watch the range (|v| >= 2^15 or non-finite), and a model that ever trips the guard is re-run --
 * in the same call -- and pinned to the exact-fp32 kernels (the reference computes in fp32, resunet.py:58-70, and has no
 * such range limit).  lm_forward_dev / lm_forward_batches_dev / lm_apply_* therefore return only after the forward has
 * finished. */
int lm_model_precision(lm_engine* e, int slot);
/* Accuracy guard.  lm_model_load on a split-f16 engine (and lm_set_precision(e, 1) for models loaded before it) runs TWO deterministic
 * probe slices (256 x 256: a phantom-like image and uniform noise, both in the network's [0, 1] input range) through the split-f16 and the exact-fp32 kernels of the
 * model and pins it to the exact-fp32 kernels -- with a notice on stderr -- when max |delta log-prob| exceeds the limit (environment
 * LM_ACC_GUARD, default 5e-4: half of the 1e-3 of the reference's fp32 result the engine is held to; "0" disables the probe).  A
 * checkpoint with a logit range or weight tails beyond what the split arithmetic resolves therefore cannot silently sit outside the
 * tolerance.  Between the two there are middle tiers: a model above the limit is probed again with the 3x3 convs split along K so that
 * no fp32 accumulator chain runs over more than 4608, then 2304, then 1152 products (the chain's roundings are where the split
 * arithmetic's error comes from; parts are work items of their own, added in a fixed order; +1 % / +4 % / +12 % forward time instead
 * of the exact kernels' 4x) and runs on the first form that is within the limit (lm_model_precision still says 1: split-f16;
 * lm_model_chain_limit says which).
 * *err_out = max |delta log-prob| of the probe of the form the model runs on (< 0: no probe was run -- guard off, or the f16 range
 * guard tripped on the probe and pinned the model first); returns 0 fast split-f16 form, 2 a split-K form, 1 pinned to the
 * exact-fp32 kernels by the probe, < 0 on error.  LM_H3_KSPLIT_K=<products> puts every model on that split-K form (A/B and test hook). */
int lm_model_probe_error(lm_engine* e, int slot, float* err_out);
/* products per accumulator chain of the model's 3x3 convs: 0 = not split (the fast form), else the limit the accuracy guard chose. */
int lm_model_chain_limit(lm_engine* e, int slot);

/* ---- network forward (mask.py:178-186: model(mbt) + torch.max(pred,1)[1]) ------ */
/* x_dev: f32 [b][h][w] (h, w multiples of 16).  labels_dev: u8 [b][h][w] or NULL.
 * logp_dev: f32 [b][C][h][w] log-softmax exactly like UNet.forward's return
 * (resunet.py:70), or NULL. */
int lm_forward_dev(lm_engine* e, int slot, const float* x_dev, int b, int h, int w,
                   uint8_t* labels_dev, float* logp_dev);

/* ---- pre-processing (utils.py:32-111 preprocess / simple_bodymask / crop_and_resize,
 *      mask.py:166-168 clip + (x+1024)/1624) -------------------------------------- */
/* vol_dev: [n][h][w] of `dtype` (LM_I16, LM_I32, LM_I64, LM_F32 or LM_F64).  Outputs (all dev):
 *   bbox_dev   int32 [n][4]  body bounding box (r0,c0,r1,c1), utils.py:102-106
 *   x_f32_dev  f32 [n][oh][ow] normalised network input        (or NULL)
 *   x_i16_dev  i16 [n][oh][ow] == utils.preprocess()[0]        (or NULL; integer volumes only)
 *   bmask_dev  u8  [n][h][w]   == utils.simple_bodymask(slice) (or NULL; test seam) */
int lm_preprocess_dev(lm_engine* e, const void* vol_dev, int dtype, int n, int h, int w, int oh, int ow,
                      int32_t* bbox_dev, float* x_f32_dev, int16_t* x_i16_dev, uint8_t* bmask_dev);

/* ---- mask un-crop (utils.py:114-129 reshape_mask, mask.py:196-202) --------------- */
/* mask_dev u8 [n][mh][mw], bbox_dev int32 [n][4] -> out_dev u8 [n][h][w]. */
int lm_reshape_mask_dev(lm_engine* e, const uint8_t* mask_dev, const int32_t* bbox_dev, int n, int mh, int mw,
                        int h, int w, uint8_t* out_dev);

/* ---- orientation (mask.py:156-164 sitk.DICOMOrient(image, "LPS") and its undo at :204-208) ---- */
/* Axis permutation / flip as an index transform: out[i0][i1][i2] = in[base + i0*s0 + i1*s1 + i2*s2]
 * (strides and base in ELEMENTS of elem_size = 1|2|4|8 bytes; strides may be negative; in/out must not
 * overlap).  The host side derives (s, base) from the image direction cosines (lungmask_amd/volume_io.py). */
int lm_reorient_dev(lm_engine* e, const void* in_dev, void* out_dev, int elem_size, int n0, int n1, int n2,
                    int64_t s0, int64_t s1, int64_t s2, int64_t base);

/* ---- volume post-processing (utils.py:272-358 postprocessing incl. :361-404 bbox_3D /
 *      keep_largest_connected_component and the hole filler of :344-352) -------------- */
/* lab_dev u8 [n][h][w], processed IN PLACE.  spare: label values that are merged into
 * neighbours and dropped (fusion), may be NULL; skip_below: utils.py default 3. */
int lm_postprocess_dev(lm_engine* e, uint8_t* lab_dev, int n, int h, int w, const int* spare, int n_spare,
                       int skip_below);
/* ---- the two helpers of postprocessing as seams of their own (utils.py:361-387 bbox_3D, :390-404
 *      keep_largest_connected_component; the reference's tests call bbox_3D directly, tests/test_utils.py:58-63) ---- */
/* mask_dev u8 [n][h][w] (non-zero = set).  bbox_out (HOST, 6 ints) = [zmin, zmax, ymin, ymax, xmin, xmax]: first / last set index
 * per axis, grown by `margin`, clipped to the volume, maxima exclusive (utils.py:376-383).  A mask without a set voxel has no
 * box -- the reference raises IndexError at utils.py:377 -- and all six come back as -1.  Returns after the result is known. */
int lm_bbox3d_dev(lm_engine* e, const uint8_t* mask_dev, int n, int h, int w, int margin, int32_t bbox_out[6]);
/* mask_dev u8 [n][h][w], IN PLACE -> 1 on the voxels of the largest region of skimage.measure.label(mask) (full connectivity:
 * 26 neighbours, 8 when n == 1; voxels of different non-zero values are different regions), 0 elsewhere.  Equal areas: the
 * region whose first voxel (raster order) comes last -- what `np.argsort(resizes)[-1]` of utils.py:402 yields while numpy's sort
 * is its stable insertion sort (<= 16 regions); beyond that the reference itself leaves the tie to numpy's quicksort.
 * *area_out (HOST, may be NULL) = that region's voxel count; 0 = no region at all (the reference raises IndexError at
 * utils.py:402), mask unchanged. */
int lm_keep_largest_dev(lm_engine* e, uint8_t* mask_dev, int n, int h, int w, int64_t* area_out);

/* What the last lm_postprocess_dev saw: info[0]=regions, [1]=boundary voxels shipped to the
 * host, [2]=regions processed by the merge loop, [3]=regions merged, [4]=host replay in us. */
/* ---- the same post-processing with the volume's slices spread over `world` ranks (multi-GPU pipeline) ----
 * Every rank holds a contiguous slab lab_slab_dev u8 [n][h][w] = slices [z0, z0+n) of a volume of n_total slices and
 * runs the voxel passes on its slab only; the slabs are tied together by six small exchanges (four with LM_SLAB_GRAPH=1, the region-graph form of the second labelling) the CALLER performs
 * (torch.distributed all_gather over RCCL; csrc/slab_engine.hip describes each).  Protocol, identical on every rank:
 *     lm_slab_begin(...);
 *     do { len = lm_slab_pending(e);                       // int32 words this rank contributes
 *          all-gather len -> lens[world];  stride = max(lens);
 *          lm_slab_emit(e, mine_dev);                       // device buffer of >= len words
 *          all-gather mine_dev (padded to stride) -> gathered_dev [world][stride];
 *     } while (lm_slab_step(e, gathered_dev, stride, lens) == 0);   // 1 = finished: the slab holds the result
 * Requires n >= 1 on every rank.  Result == lm_postprocess_dev on the gathered volume, bit for bit. */
int lm_slab_begin(lm_engine* e, uint8_t* lab_slab_dev, int n, int h, int w, int rank, int world, int z0, int n_total,
                  const int* spare, int n_spare, int skip_below);
int64_t lm_slab_pending(lm_engine* e);
/* 1 when the pending length is the same on every rank by construction (the face-plane exchanges): the caller may skip
 * the all-gather of the lengths for this round. */
int lm_slab_pending_uniform(lm_engine* e);
int lm_slab_emit(lm_engine* e, int32_t* dst_dev);
int lm_slab_step(lm_engine* e, const int32_t* gathered_dev, int64_t stride, const int64_t* lens);

int lm_postprocess_info(lm_engine* e, int64_t info[5]);

/* ---- label fusion (mask.py:228-230): res_l <- fuse(res_l, res_r); returns the spare label -- */
int lm_fuse_dev(lm_engine* e, uint8_t* res_l_dev, const uint8_t* res_r_dev, size_t nvox, int* spare_out);
/* The two halves of lm_fuse_dev for a volume whose slices are spread over several ranks (lungmask_amd/pipeline.py): `spare =
 * res_l.max() + 1` (mask.py:228) is a maximum over the WHOLE volume, so every rank reports the maximum of its slab
 * (lm_label_max_dev), the caller combines them (one small all-gather) and every rank fuses its slab with the agreed value
 * (lm_fuse_spare_dev == mask.py:229-230). */
int lm_label_max_dev(lm_engine* e, const uint8_t* lab_dev, size_t nvox, int* max_out);
int lm_fuse_spare_dev(lm_engine* e, uint8_t* res_l_dev, const uint8_t* res_r_dev, size_t nvox, int spare);

/* ---- the whole hot path: LMInferer.apply on a numpy volume (mask.py:212-232) ---------- */
/* slot: model; fill_slot: fill model for the fused LTRCLobes_R231 mode or -1.
 * vol: [n][h][w] of `dtype` (LM_I16/LM_I32/LM_I64/LM_F32/LM_F64); out: u8 [n][h][w].
 * batch_size, volume_postprocessing: the LMInferer constructor arguments (mask.py:72-82).
 * _dev: both buffers already in HBM, nothing leaves the device.  _host: does the H2D / D2H. */
int lm_apply_dev(lm_engine* e, int slot, int fill_slot, const void* vol_dev, int dtype, int n, int h, int w,
                 int batch_size, int volume_postprocessing, uint8_t* out_dev);
int lm_apply_host(lm_engine* e, int slot, int fill_slot, const void* vol_host, int dtype, int n, int h, int w,
                  int batch_size, int volume_postprocessing, uint8_t* out_host);

/* lm_apply_host with flags.  LM_APPLY_OUT_SCRATCH: the previous contents of out_host are of no value to the caller (a result
 * array the binding allocated itself, mask.py:210) -- the engine may write it before the call is known to succeed: a helper thread
 * fills it with zeros while the network runs, and only the slab of slices x image rows that carries a label is copied back (one
 * strided device-to-host copy).  Without the flag (== lm_apply_host) out_host is only written by the final copy of a successful
 * call.  The labels are the same either way. */
#define LM_APPLY_OUT_SCRATCH 1u
int lm_apply_host_ex(lm_engine* e, int slot, int fill_slot, const void* vol_host, int dtype, int n, int h, int w,
                     int batch_size, int volume_postprocessing, uint8_t* out_host, unsigned flags);

/* ---- volumes QUEUED through one engine ("multi-volume queueing"; the reference's apply is one blocking call
 *      per volume, mask.py:212-232) -------------------------------------------------------------------------------------------
 * lm_apply_host crosses the host boundary inside the call: copy-in before the first kernel, copy-back behind the last.  With a
 * stream of volumes both can run BESIDE the hot path of the neighbouring volumes.  The engine holds two resident input buffers and
 * two result buffers (k = 0, 1); volume i uses k = i % 2:
 *     lm_pipe_upload(e, k, vol_host, bytes)      copy-in on a stream of its own (returns when a pageable source has been staged:
 *                                                call it from a thread of its own, ahead of the volume's turn);
 *     lm_pipe_apply(e, k, slot, ...)             == lm_apply_dev on buffer k: waits ON THE DEVICE for the copy-in of k and for the
 *                                                copy-back of the volume that used result buffer k before (two volumes earlier);
 *     lm_pipe_download(e, k, out_host, bytes)    enqueues the copy-back on a third stream behind the hot path, returns at once;
 *     lm_pipe_wait(e, k)                         blocks until that copy-back has arrived.
 * The caller's part of the protocol: lm_pipe_upload(k) of volume i + 2 only after lm_pipe_apply(k) of volume i has returned, and
 * out_host stays alive until lm_pipe_wait.  lm_pipe_upload may run on another thread than the other three (it touches nothing of
 * theirs); lungmask_amd.LMInferer.apply_async is the binding.  Labels == lm_apply_host. */
int lm_pipe_upload(lm_engine* e, int k, const void* vol_host, size_t bytes);
int lm_pipe_apply(lm_engine* e, int k, int slot, int fill_slot, int dtype, int n, int h, int w, int batch_size, int volume_postprocessing);
int lm_pipe_download(lm_engine* e, int k, uint8_t* out_host, size_t bytes);
int lm_pipe_wait(lm_engine* e, int k);

/* The batch loop of mask.py:173-187 in one call: n slices in batches of batch_size.  With two forward
 * lanes (default; lm_set_streams(e, 1) disables) consecutive batches alternate between two HIP streams and
 * workspaces so that one batch's kernel tails are filled by the next batch's work; results are identical. */
int lm_forward_batches_dev(lm_engine* e, int slot, const float* x_dev, int n, int h, int w, int batch_size,
                           uint8_t* labels_dev);
int lm_set_streams(lm_engine* e, int n);
/* Producer/consumer fusions of the split-f16 forward (default 11 = bits 0 and 3; A/B and test hook).
 * bit 0: down_path.0's first conv (Cin = 1, resunet.py:93-95) computed inside the loader of its second conv -- the stand-alone
 *        kernel's operation order: bit-identical results;
 * bit 1: reserved (the decoder's bilinear x2, resunet.py:131-133, inside the loader of the block's first conv: ruled out by
 *        measurement, DESIGN.md 3.6 -- setting it changes nothing);
 * bit 2: split-K of the 16 x 16 / 32 x 32 decoder 1x1 convs (parts added in a fixed order: deterministic, last bits differ from
 *        the single chain); measured slower than the single chain, hence off by default;
 * bit 3: the head (last 1x1 conv + log-softmax + argmax, resunet.py:69-70, mask.py:184-186) inside the last conv's epilogue, on the
 *        conv's fp32 results instead of the stored 22-bit hi/lo tensor: log-probabilities differ in the last bits, labels on
 *        near-tie pixels only. */
int lm_set_fusion(lm_engine* e, int mask);

/* Per-kernel timing of the launches since the last reset (HIP events on the launching stream).
 * lm_profile_enable(e, on): 0 off; 1 every kernel; 2 every kernel, one entry per conv layer shape;
 * 3 the dominant kernel (conv3x3) only -- a third of the events, for timed regions (the events of mode 1 cost ~1 %).
 * lm_profile_read returns the number of distinct kernel kinds and fills up to `cap` entries. */
typedef struct lm_kernel_stat {
    char name[48];
    int64_t launches;
    double total_ms;
    double flops; /* algorithmic FLOPs of those launches (0 for non-GEMM kernels) */
    double bytes; /* algorithmic HBM bytes of those launches */
} lm_kernel_stat;
int lm_profile_enable(lm_engine* e, int on);
int lm_profile_reset(lm_engine* e);
int lm_profile_read(lm_engine* e, lm_kernel_stat* out, int cap);
/* Timeline of the launches since the last reset (lm_profile_enable(e, 4): per-layer names + start/end of every launch in
 * milliseconds since the first recorded launch, with the forward lane it ran on): what runs beside what when two lanes are on. */
typedef struct lm_launch_span {
    char name[48];
    int lane;
    double start_ms, end_ms;
} lm_launch_span;
int lm_profile_timeline(lm_engine* e, lm_launch_span* out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* LUNGMASK_HIP_H */
