// Kernel-side parameter blocks and launchers of the U-Net forward pass
// (replaces torch's conv2d / relu / batch_norm / avg_pool2d / upsample / cat /
// log_softmax / max as called from lungmask/resunet.py:58-70, mask.py:183-186).
#pragma once
#include "lm_platform.h"

namespace lm {

// Activations are NHWC fp32: element (b,y,x,c) of a tensor with `cstride`
// channels per pixel lives at ((b*H + y)*W + x)*cstride + coff + c.  `coff`
// lets a producer write straight into one half of a concat buffer
// (resunet.py:147 torch.cat([up, bridge], 1) is therefore never materialised).
struct ConvParams {
    const float* in;
    int in_cstride, in_coff;
    const float* w;     // packed [taps][Cin][Cout]
    const float* bias;  // [Cout]
    const float* bn_s;  // [Cout] gamma/sqrt(var+eps)   (nullptr: no ReLU/BN epilogue)
    const float* bn_t;  // [Cout] beta - mean*s
    float* out;
    int out_cstride, out_coff;
    float* pool;  // optional avg_pool2d(2) of the output (nullptr: none)
    int pool_cstride, pool_coff;
    int B, H, W, Cin, Cout;
};

struct FirstConvParams {
    const float* in;  // [B,H,W] single channel
    const float* w;   // [9][64]
    const float* bias;
    const float* bn_s;
    const float* bn_t;
    float* out;
    int out_cstride, out_coff;
    int B, H, W;
    unsigned* range_flag = nullptr;  // split-f16 form only: see ConvParamsH3::range_flag
};

struct UpsampleParams {
    const float* in;  // [B,h,w,C] dense
    float* out;       // [B,2h,2w,out_cstride] at out_coff
    int out_cstride, out_coff;
    int B, h, w, C;
};

struct HeadParams {
    const float* in;  // [B,H,W,64] dense
    const float* w;   // [C][64]
    const float* bias;
    uint8_t* labels;  // [B,H,W]            (nullptr: skip)
    float* logp;      // [B,C,H,W] log-softmax (nullptr: skip)
    int B, H, W, C;
};

// Split-f16 variant (nn_kernels_h3.hip): tensors are "split NHWC" byte buffers (4 bytes per element:
// groups of 8 channels = 8 hi halves + 8 lo halves, v = hi + lo); strides/offsets are still counted in channels.
struct ConvParamsH3 {
    const char* in;
    int in_cstride, in_coff;
    const char* w;  // packed [taps][Cout][Cin/8 groups][hi8|lo8] of w * 2^k (k per layer, so that lo stays a normal f16)
    float acc_scale;  // 2^-k: applied to the accumulator in the epilogue
    const float* bias;
    const float* bn_s;
    const float* bn_t;
    char* out;
    int out_cstride, out_coff;
    char* pool;
    int pool_cstride, pool_coff;
    const char* zeros;  // >= 16 zero bytes in device memory (source of out-of-image halo pixels)
    int B, H, W, Cin, Cout;
    // Fused head (last conv of the decoder, Cout == 64): when head_labels is set the conv output is NOT stored; the
    // 1x1 head conv + argmax (resunet.py:69, mask.py:184-186) -- and, with head_logp, the log-softmax (resunet.py:70) -- run in the
    // epilogue on the conv's fp32 results (launch_head_h3's summation order; that kernel reads the stored 22-bit tensor instead and
    // may therefore differ from this path on near-tie pixels).
    const float* head_w = nullptr;   // [C][64]
    const float* head_b = nullptr;   // [C]
    uint8_t* head_labels = nullptr;  // [B][H][W]
    float* head_logp = nullptr;      // [B][C][H][W] or nullptr
    int head_C = 0;
    // Border correction of a consumer of a deferred-shift ("r-form") tensor: [16][Cout] floats indexed by the pixel's border mask
    // (1 top row, 2 bottom row, 4 left column, 8 right column), SUBTRACTED from the bias for pixels on the image border; the
    // interior part is folded into `bias` by the host.  nullptr: the input tensor carries no shift.  3x3 only.
    const float* border_corr = nullptr;
    // f16 range guard: hi = f16(v) overflows beyond 65504 and nothing downstream would notice.  Every producer of a split
    // tensor ORs 1 into this device word when a value it writes is not below kF16Guard in magnitude (or is not finite);
    // the engine checks the word after the forward and re-runs the model on the exact-fp32 kernels (nn_engine.hip).
    unsigned* range_flag = nullptr;
    // the output tensor is far larger than the caches (the 256 x 256 and 128 x 128 levels of a 20-slice batch): its stores are
    // issued non-temporal, so that they do not evict the weights and halo rows the kernel re-reads
    int stream_out = 0;
    // Fused first layer (down_path.0.block.0-2, Cin = 1; resunet.py:93-95): when fc_x is set this conv's INPUT tensor -- 64 channels
    // in the deferred-shift form -- is never read from (or written to) memory: the loader of the persistent kernel computes it, 16
    // channels at a time, from the network's 1-channel input with the arithmetic of first_conv_h3_kernel (same operation
    // order: bit-identical tensors).  `in` is ignored.  Needs Cin == 64, W % 32 == 0 and a deferred BatchNorm shift.
    const float* fc_x = nullptr;  // [B][H][W] f32
    const float* fc_c = nullptr;  // 704 floats: w[9][64] | bias[64] | bn_s[64]
    // Split-K (1x1 form on the persistent kernel only): the 16 x 16 / 32 x 32 decoder 1x1 convs have 80 / 160 work items of 32 / 16
    // sequential stages on 256 compute units.  With ksplit = S > 1 every item is cut into S items over Cin / S input channels each
    // whose raw fp32 accumulators go to kpart[S][B][H][W][Cout]; launch_conv1x1_h3 then runs the reduction (parts added in index
    // order, then scale, bias, split: a fixed order -- deterministic, not the unsplit kernel's single chain).
    int ksplit = 1;
    float* kpart = nullptr;
};
// the K split launch_conv1x1_h3 can use for this shape (1: none) -- the engine sizes kpart with it
int conv1x1_h3_ksplit(const ConvParamsH3& p);
// the K split of a 3x3 launch whose accumulator chains may not run over more than max_k products (the accuracy guard's "precise"
// tier; 1: none): set ConvParamsH3::ksplit to it and kpart to ksplit * B * H * W * Cout floats
int conv3x3_h3_ksplit(const ConvParamsH3& p, int max_k);
// whether launch_conv3x3_h3 can take the first layer into its loader for this shape (else run launch_first_conv_h3 first)
bool conv3x3_h3_can_fuse_first(const ConvParamsH3& p);
// LM_H3_FOLD_SCALE = 1: the BatchNorm SCALE s = 2^e * m (|m| in [1, 2), per channel) of a Conv -> ReLU -> BN layer leaves the
// epilogue: the exact power of two is folded into the layer's OWN packed weight row and bias (relu(x) 2^e == relu(x 2^e)), so the
// layer stores r' = relu(acc + b) * 2^e[co] -- every channel at the magnitude BatchNorm gives it, within a factor of two -- and every
// consumer multiplies by w[co][ci] * m[ci], the way it already carries the shift (ConvLayer::bias_h3); average pool, bilinear
// upsample and concat are per-channel linear, the sign of s is irrelevant.  One multiply per output less in every 3x3 epilogue and in
// the first-conv producer, two LDS reads less per channel group; results differ from the unfolded form in the last bits only.
#ifndef LM_H3_FOLD_SCALE
#define LM_H3_FOLD_SCALE 1  // (0: the scale applied in the producing epilogue, as in rounds 1-4 -- the A/B arm of profiles/r05a_*)
#endif
// 0: round 5's form -- ONE power of two per layer (below the median |s|), s / 2^E in the consumers (A/B arm of profiles/r06a_*)
#ifndef LM_H3_FOLD_PER_CHANNEL
#define LM_H3_FOLD_PER_CHANNEL 1
#endif
constexpr float kF16Guard = 32768.f;  // 2^15: a factor 2 below the largest finite half
// whether launch_conv3x3_h3 can take the fused head for this shape (else run launch_head_h3 on the stored output)
bool conv3x3_h3_can_fuse_head(const ConvParamsH3& p);
hipError_t launch_conv3x3_h3(const ConvParamsH3& p, hipStream_t stream);
hipError_t launch_conv1x1_h3(const ConvParamsH3& p, hipStream_t stream);
hipError_t launch_first_conv_h3(const FirstConvParams& p, hipStream_t stream);  // out: split tensor
hipError_t launch_upsample2x_h3(const UpsampleParams& p, hipStream_t stream);   // in/out: split tensors
hipError_t launch_head_h3(const HeadParams& p, hipStream_t stream);             // in: split tensor

// All launchers enqueue on `stream` and return hipGetLastError().
hipError_t launch_conv3x3(const ConvParams& p, hipStream_t stream);
hipError_t launch_conv1x1(const ConvParams& p, hipStream_t stream);
hipError_t launch_first_conv(const FirstConvParams& p, hipStream_t stream);
hipError_t launch_upsample2x(const UpsampleParams& p, hipStream_t stream);
hipError_t launch_head(const HeadParams& p, hipStream_t stream);

constexpr int kMaxClasses = 8;

}  // namespace lm
