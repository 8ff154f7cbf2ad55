// sp3d_unproject_patch.h - launcher of the LDS-staged dense-grid unprojection kernel (sp3d_unproject_patch.hip), called by
// launch_nhwc() in sp3d_unproject.hip.  Internal to the library (not part of include/sp3d.h).
#pragma once
#include "sp3d_device.h"

namespace sp3d {

// io: bit 0 = packed heat-maps are bf16, bit 1 = cubes are bf16.  Packed (channels-last, 16 floats per pixel) heat-maps only.
int launch_patch(const Views &v, int Jp, const float *cam, const float *centers, const uint8_t *valid, float *cubes,
                 float *grids, const Geom &g, bool out_cl, int io, hipStream_t s);

} // namespace sp3d
