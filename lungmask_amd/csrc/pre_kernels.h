// Parameter blocks / launchers of the pre-processing kernels (pre_kernels.hip).
#pragma once
#include "../../include/lungmask_hip.h"
#include "lm_platform.h"

namespace lm {

struct BodyMaskParams {
    const void* vol;  // [N][H][W] of `dtype`
    int dtype;        // LM_I16 / LM_I32 / LM_I64
    int N, H, W;
    int* bbox;       // [N][4] = (r0, c0, r1, c1) half-open, utils.py:102-106
    uint8_t* bmask;  // optional [N][H][W] full-resolution body mask (test seam), else nullptr
};

struct ResampleParams {
    const void* vol;
    int dtype;
    int N, H, W;
    const int* bbox;
    int OH, OW;        // 256 x 256 (mask.py:166)
    int16_t* out_i16;  // optional [N][OH][OW]: utils.preprocess output
    float* out_f32;    // optional [N][OH][OW]: normalised network input (mask.py:167-168,178-181)
};

struct ReshapeParams {
    const uint8_t* mask;  // [N][MH][MW]
    const int* bbox;      // [N][4]
    uint8_t* out;         // [N][H][W]
    int N, MH, MW, H, W;
};

// Axis permutation / flip of a volume (sitk.DICOMOrient at mask.py:156-164,204-208 as an index transform):
// out[i0][i1][i2] = in[base + i0*s0 + i1*s1 + i2*s2], element strides (may be negative).
struct ReorientParams {
    const void* in;
    void* out;
    int elem;  // bytes per element: 1, 2, 4 or 8
    int n0, n1, n2;
    long long s0, s1, s2, base;
};

hipError_t launch_reorient(const ReorientParams& p, hipStream_t stream);
hipError_t launch_bodymask_bbox(const BodyMaskParams& p, hipStream_t stream);
hipError_t launch_resample_norm(const ResampleParams& p, hipStream_t stream);
hipError_t launch_reshape_mask(const ReshapeParams& p, hipStream_t stream);
// Extent of the non-zero labels of vol [N][H][W] along the two slow axes: ext[4] = {z0, z1, y0, y1} half-open (z0 >= z1: the
// volume is all zero).  lm_apply_host copies only that slab back to a zero-filled result array (a lung mask is mostly background).
hipError_t launch_label_extent(const uint8_t* vol, int N, int H, int W, int* ext_dev, hipStream_t stream);

}  // namespace lm
