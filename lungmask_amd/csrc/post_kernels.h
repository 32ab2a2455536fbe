// Launchers of the volume post-processing kernels (post_kernels.hip): 3-D
// connected-component labelling and the voxel-level parts of
// lungmask/utils.py:272-358 (postprocessing), :390-404
// (keep_largest_connected_component), fill_voids.fill and mask.py:228-230 (fusion).
#pragma once
#include "lm_platform.h"

namespace lm {

struct Dims {
    int N, H, W;
    __host__ __device__ size_t nvox() const { return (size_t)N * H * W; }
};

// A class of boundary voxels: `count` voxels of region `atom` whose 6-neighbourhood holds exactly the distinct other
// regions nb[] (descending, zero padded).  Identical voxels are merged per workgroup before they leave the device
// (a 300-slice volume has ~2x10^5 boundary voxels but only ~10^3-10^4 distinct classes).
struct BoundaryRec {
    int atom;
    int nb[6];
    int count;
};

// ---- connected components (union-find on an int32 parent volume; root = first voxel in raster order)
// lab: u8 volume; foreground = non-zero; voxels are connected when adjacent AND equal.
// flatten = false: the forest is left as the hooking built it (roots are roots: parent[v] == v); ccl_rank(flat = false) then walks
// the chains itself, which spares a read-modify-write pass over the parent volume when nothing else needs flat parents.
hipError_t ccl_label(const uint8_t* lab, int* parent, Dims d, bool conn26, hipStream_t s, bool flatten = true);
// Dense ids in raster order of each component's first voxel (== skimage.measure.label numbering).
// blockcnt: scratch of >= nblocks(nvox)+1 ints.  total_dev receives the number of components.
size_t rank_blocks(size_t nvox);
hipError_t ccl_rank(const int* parent, int* rank, int* ids, int* blockcnt, int* total_dev, size_t nvox, hipStream_t s, bool flat = true);
hipError_t region_stats(const int* ids, const uint8_t* lab, int* area, uint8_t* labval, size_t nvox, hipStream_t s, int cap = 0x7fffffff);
hipError_t boundary_records(const int* ids, Dims d, BoundaryRec* recs, unsigned* count_dev, unsigned cap, hipStream_t s);
hipError_t apply_lut(const int* ids, const uint8_t* lut, uint8_t* out, size_t nvox, hipStream_t s);

// ---- the second labelling on the region graph (post_engine.hip: postprocess, N > 1)
// region_stats + box[id][6] = {zmin, ymin, xmin, zmax, ymax, xmax} of every region with id <= cap (box: 6 * (cap + 1) ints, preset here);
// area == nullptr: the boxes only (area / labval untouched)
hipError_t region_stats_box(const int* ids, const uint8_t* lab, int* area, uint8_t* labval, int* box, Dims d, hipStream_t s, int cap);
// (smaller id << 32 | larger id) of regions with voxels that touch diagonally (26- but not 6-adjacent); duplicates possible;
// *count_dev is raised for every pair, stored or not (cap)
hipError_t diag_pairs(const uint8_t* lab, const int* ids, Dims d, unsigned long long* pairs, unsigned* count_dev, unsigned cap, hipStream_t s);

// ---- per-label largest component + hole filling
// area_by_root[root] = component size (array of nvox ints, zeroed here); best[256] u64 = max over the
// components of each label value of (area << 32 | root)  (largest area, highest root index on ties).
hipError_t component_max(const int* parent, const uint8_t* lab, int* area_by_root, unsigned long long* best, size_t nvox, hipStream_t s);
// bg[v] = (parent[v] != keep_root)
hipError_t complement_of_component(const int* parent, int keep_root, uint8_t* bg, size_t nvox, hipStream_t s);
// flags[root] = 1 for every bg component touching a face of the volume (flags: nvox ints, zeroed here).
hipError_t flag_face_components(const int* bgparent, int* flags, Dims d, hipStream_t s);
// N == 1 path (utils.py:344-350, area_closing(area_threshold=64)): flags[root] = 1 when the 2-D
// 4-connected background component has area >= threshold.
hipError_t flag_large_components(const int* bgparent, int* flags, int threshold, size_t nvox, hipStream_t s);
// out[v] = label where v is in the kept component or in an unflagged background component.
hipError_t fill_write(const int* parent, int keep_root, const int* bgparent, const int* flags, uint8_t label, uint8_t* out, size_t nvox,
                      hipStream_t s);

// Hole filling confined to a box.  A background voxel of the kept component's complement can only be a hole INSIDE the
// component's bounding box: with the box grown by one voxel (clipped at the volume) a background component is "outside" exactly
// when it reaches a face of the box -- the grown shell lies outside the bounding box, hence is all background and connected to
// the volume's faces; where the box was clipped its face IS a face of the volume.  So the 6-connected background labelling, the
// face flags and the final write run on the box only (a lobe's box is a fraction of a 300 x 512 x 512 volume).
struct Box {
    int z0, y0, x0;  // origin in the volume
    Dims d;          // extent
};
// bbox[label] = {zmin, ymin, xmin, zmax, ymax, xmax} (ints; mins preset to INT_MAX, maxs to -1 by the caller) of the voxels whose
// root is keep_root[lab[v]] (keep_root: 256 ints, -1 for absent labels)
// keep_root[label] = root of the component `best` (component_max) names for the label, -1 for label 0 / absent labels; bbox preset
// for component_bboxes -- on the device, so that the areas and the boxes come back to the host in ONE round trip
hipError_t keep_roots_init(const unsigned long long* best, int* keep_root, int* bbox, hipStream_t s);
hipError_t component_bboxes(const int* parent, const uint8_t* lab, const int* keep_root, int* bbox, Dims d, hipStream_t s);
// bg[compact index in the box] = (parent[v] != keep_root)
hipError_t complement_of_component_box(const int* parent, int keep_root, Dims d, Box box, uint8_t* bg, hipStream_t s);
// out[v] = label for the voxels of the box that are in the kept component or in an unflagged background component of the
// box-local labelling bgparent/flags
hipError_t fill_write_box(const int* parent, int keep_root, const int* bgparent, const int* flags, uint8_t label, uint8_t* out, Dims d, Box box,
                          hipStream_t s);
// complement_of_component_box / fill_write_box with "in the kept component of `label`" == keeplut[ids[v]] == label
hipError_t complement_of_lut_box(const int* ids, const uint8_t* keeplut, uint8_t label, Dims d, Box box, uint8_t* bg, hipStream_t s);
hipError_t fill_write_lut_box(const int* ids, const uint8_t* keeplut, const int* bgparent, const int* flags, uint8_t label, uint8_t* out, Dims d, Box box,
                              hipStream_t s);

// ---- slab-sharded post-processing (slab_engine.hip): the same passes on ONE rank's slices, plus the few
//      planes/tables that tie the slabs together.  "atom" = component of the slab-local labelling (dense id).
// Halo neighbours in boundary records are tagged: id | HALO_LO (atom of the previous rank's last slice) or
// id | HALO_HI (atom of the next rank's first slice).
constexpr int HALO_LO = 1 << 29, HALO_HI = 1 << 30, HALO_MASK = HALO_LO | HALO_HI;
// halo_lo / halo_hi: int32 [H][W] atom ids of the adjacent slices of the neighbouring slabs (or nullptr).
hipError_t boundary_records_halo(const int* ids, Dims d, const int* halo_lo, const int* halo_hi, BoundaryRec* recs, unsigned* count_dev,
                                 unsigned cap, hipStream_t s);
// first[id] = voxel_base + (index of the first voxel of atom id) -- the key regions are numbered by.
hipError_t atom_first(const int* parent, const int* rank, int* first, int voxel_base, size_t nvox, hipStream_t s);
// flags[id] = 1 for atoms with a voxel on a lateral face, on z == 0 (if zlo_face) or on z == N-1 (if zhi_face).
hipError_t atom_face_flags(const int* ids, Dims d, bool zlo_face, bool zhi_face, int* flags, hipStream_t s);
// Pairs (atom of plane a, atom of plane b) of voxels that are neighbours across the slab face (straight across, or
// the 9 of 26-connectivity) and carry the same label (lab_* == nullptr: any two non-zero ids).  Runs are deduplicated.
hipError_t face_edges(const int* ids_a, const int* lab_a, const int* ids_b, const int* lab_b, int H, int W, bool conn26, int* edges,
                      unsigned* count_dev, unsigned cap, hipStream_t s);
hipError_t widen_u8(const uint8_t* in, int* out, size_t n, hipStream_t s);
// bg[v] = (keeplut[ids[v]] != label)
hipError_t lut_complement(const int* ids, const uint8_t* keeplut, uint8_t label, uint8_t* bg, size_t nvox, hipStream_t s);
// out[v] = label where keeplut[ids2[v]] == label or holelut[ids3[v]] != 0
hipError_t fill_write_lut(const int* ids2, const uint8_t* keeplut, const int* ids3, const uint8_t* holelut, uint8_t label, uint8_t* out,
                          size_t nvox, hipStream_t s);

// ---- stand-alone seams of utils.bbox_3D (utils.py:361-387) and keep_largest_connected_component (utils.py:390-404)
// box_dev[6] = {zmin, ymin, xmin, zmax, ymax, xmax} over the non-zero voxels of mask (INT_MAX / -1 when there is none)
hipError_t mask_bbox(const uint8_t* mask, int* box_dev, Dims d, hipStream_t s);
// out[v] = (parent[v] == keep_root)
hipError_t component_mask(const int* parent, int keep_root, uint8_t* out, size_t nvox, hipStream_t s);

// ---- fusion (mask.py:228-230)
hipError_t volume_max(const uint8_t* a, unsigned* max_dev, size_t nvox, hipStream_t s);
hipError_t fuse_labels(uint8_t* res_l, const uint8_t* res_r, uint8_t spare, size_t nvox, hipStream_t s);

}  // namespace lm
