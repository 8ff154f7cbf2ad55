/* sp3d_tuning.h - measurement-only entry points of libsp3d.so (NOT part of the drop-in ABI
 * in include/sp3d.h).  Used by tools/ab_variants.py for within-process A/B of kernel variants. */
#ifndef SP3D_TUNING_H
#define SP3D_TUNING_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* variant: bits[1:0] voxels in flight (first NHWC kernel); bit 2: no XCD-affine tile map; bit 3: pipelined
 * per-wave kernel; bit 4: one wave per workgroup; bit 5: 4x4x4 voxel bricks per wave (workgroup = z-stack of bricks);
 * bit 6: with bit 5, every brick its own workgroup; bit 10: with bit 5, a workgroup barrier per view (view-synchronous
 * workgroups); bits 11-13: n * 20 KB of unused LDS per brick workgroup (caps the workgroups resident on a CU); bits 17-20: log2(tiles per XCD chunk)+1; bit 21: plain chunk
 * sweep; bit 24: channels-last output.  Library default: 120 (channels-last result), 56 (planar, Z % 32 == 0), else 24. */
int sp3d_unproject_fwd_variant(const float *const *hm_views, int Jp, const float *cam, const float *centers,
                               const uint8_t *valid, float *cubes, float *grids, int B, int V, int J, int h, int w,
                               int X, int Y, int Z, const float *grid_size, int W_in, int H_in, int variant,
                               void *stream);
/* per-wave s_memtime timeline of the pipelined kernel (18 uint64 per wave: start, after P1(0), after each
 * view, ..., [17] = number of cameras seeing the lane-0 voxel); NULL switches it off (tools/wave_timeline.py) */
int sp3d_debug_set_timeline(void *dev_buffer);
/* one-thread kernel: *slot = the chip-wide 100 MHz clock.  Two around a kernel inside a captured graph time it in the step. */
int sp3d_debug_stamp(uint64_t *slot, void *stream);
#ifdef __cplusplus
}
#endif
#endif
