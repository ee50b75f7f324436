// Internal declarations shared by the HIP translation units of libfyrox_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fyx {

constexpr int kCUs = 256;  // MI355X

struct LbsTuning {
    int block = 256;         // threads per workgroup: 256 | 512 | 1024
    int blocks_per_cu = 8;   // persistent grid = kCUs * blocks_per_cu (capped by the work)
    int prefetch = 0;        // software-pipeline each wave (loads of unit i+1 before math of unit i)
    int exact = 1;           // 1: reference operation order, unfused; 0: FMA
    int nt = 1;              // non-temporal streaming loads/stores
};

struct LbsArgs {
    const float* pos;        // 3N packed xyz
    const float* nrm;        // 3N or null
    const float* tan;        // 4N or null
    const float* wgt;        // 4N
    const uint32_t* idx;     // N (4 x u8 packed little-endian: idx0 in the low byte)
    const float* palette;    // n_instances * n_bones * 16, column-major mat4
    float* out_pos;          // n_instances * 3N or null
    float* out_nrm;
    float* out_tan;
    uint32_t n_verts;
    uint32_t n_bones;
    uint32_t n_instances;
};

// Launch the skinning kernel.  Returns hipSuccess or the launch error.
hipError_t launch_lbs(const LbsArgs& a, const LbsTuning& t, hipStream_t stream);

// AoS -> SoA de-interleave (device to device).  off_* in bytes, -1 = absent.
hipError_t launch_deinterleave(const uint8_t* d_aos, uint32_t n_verts, uint32_t stride,
                               int off_pos, int off_nrm, int off_tan, int off_wgt, int off_idx,
                               float* d_pos, float* d_nrm, float* d_tan, float* d_wgt,
                               uint32_t* d_idx, hipStream_t stream);

// max over all 4N index bytes -> *d_out (single uint32).
hipError_t launch_max_bone_index(const uint32_t* d_idx, uint32_t n_verts, uint32_t* d_out,
                                 hipStream_t stream);

// skinned-position AABB. d_partials: scratch of 6 * n_blocks floats (n_blocks returned by
// aabb_partial_blocks()).  Result in d_out[6] = {min xyz, max xyz}.
uint32_t aabb_partial_blocks(uint32_t n_verts);
hipError_t launch_skinned_aabb(const LbsArgs& a, float* d_partials, float* d_out,
                               hipStream_t stream);
// min/max over an already skinned packed xyz stream of n points.
hipError_t launch_points_aabb(const float* d_xyz, uint64_t n_points, float* d_partials,
                              float* d_out, hipStream_t stream);

// out[i] = a[i] * b[i] (mat4, nalgebra operation order)
hipError_t launch_palette(const float* d_global, const float* d_inv_bind, uint32_t n,
                          float* d_out, hipStream_t stream);

// calibration: read 48*units bytes from d_src, write 32*units bytes to d_dst
hipError_t launch_stream_copy(const float* d_src, float* d_dst, uint32_t units, int blocks_per_cu,
                              hipStream_t stream);

}  // namespace fyx
