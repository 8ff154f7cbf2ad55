// sp3d_device.h - device-side math shared by the gfx950 unprojection kernels.
//
// Arithmetic contract: every operation below is the fp32 operation the
// reference performs, in the reference's order; the only fused multiply-adds are the ones
// torch's K=3 `mm` performs (rigid transform, crop affine) and ATen's bilinear chain, and
// they are written as explicit fmaf().  The translation unit is compiled with
// -ffp-contract=off so the compiler adds none of its own; division is IEEE (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).  Result: bit-identical to oracle/sp3d_oracle.c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sp3d.h"

namespace sp3d {

struct Views {
    const float *p[SP3D_MAX_VIEWS];
};
struct ViewsMut {
    float *p[SP3D_MAX_VIEWS];
};

struct Geom {
    int B, V, J, h, w, X, Y, Z;
    int N;          // X*Y*Z
    int YZ;         // Y*Z
    int W_in, H_in; // network input size (cfg.NETWORK.IMAGE_SIZE)
    float Lx, Ly, Lz;
};

// torch.linspace(-L/2, L/2, n)[i] in fp32 (project_layer.py:28-30; ATen CPU kernel form)
__device__ __forceinline__ float linspace_at(float L, int n, int i)
{
    const float start = -(L / 2.0f), end = L / 2.0f;
    if (n == 1) return start;
    const float step = (end - start) / (float)(n - 1);
    return (i < n / 2) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(n - 1 - i), end);
}

__device__ __forceinline__ float clamp_nan(float v, float lo, float hi)
{
    // torch.clamp: NaN propagates
    return (v != v) ? v : (v < lo ? lo : (v > hi ? hi : v));
}

// Voxel centre -> sample position in heat-map pixels for one view.
// project_layer.py:76-90 (project_pose = cameras.py:27-55) + ATen unnormalize(align_corners).
// `cm` is wave-uniform (one sample, one view): the compiler keeps it in SGPRs via s_load.
__device__ __forceinline__ bool sample_pos(const float *__restrict__ cm, float x, float y, float z, int w, int h,
                                           float W_in, float H_in, float &ix, float &iy)
{
    const float dx = x - cm[SP3D_CAM_T + 0], dy = y - cm[SP3D_CAM_T + 1], dz = z - cm[SP3D_CAM_T + 2];
    const float xc = fmaf(cm[2], dz, fmaf(cm[1], dy, cm[0] * dx));
    const float yc = fmaf(cm[5], dz, fmaf(cm[4], dy, cm[3] * dx));
    const float zc = fmaf(cm[8], dz, fmaf(cm[7], dy, cm[6] * dx));
    const float den = zc + 1e-5f;
    const float y0 = xc / den, y1 = yc / den;
    float r2 = y0 * y0 + y1 * y1;
    r2 = r2 > 1e10f ? 1e10f : r2;
    const float r4 = r2 * r2, r6 = r4 * r2;
    const float radial = 1.0f + ((cm[SP3D_CAM_K] * r2 + cm[SP3D_CAM_K + 1] * r4) + cm[SP3D_CAM_K + 2] * r6);
    const float tan = cm[SP3D_CAM_P] * y1 + cm[SP3D_CAM_P + 1] * y0;
    const float corr = radial + 2.0f * tan;
    const float u0 = y0 * corr + cm[SP3D_CAM_P + 1] * r2;
    const float u1 = y1 * corr + cm[SP3D_CAM_P] * r2;
    float px = cm[SP3D_CAM_F] * u0 + cm[SP3D_CAM_C];
    float py = cm[SP3D_CAM_F + 1] * u1 + cm[SP3D_CAM_C + 1];
    const float W0 = cm[SP3D_CAM_W0], H0 = cm[SP3D_CAM_H0];
    const bool bound = (px >= 0.0f) && (py >= 0.0f) && (px < W0) && (py < H0);
    const float mx = W0 > H0 ? W0 : H0;
    px = clamp_nan(px, -1.0f, mx);
    py = clamp_nan(py, -1.0f, mx);
    float qx = fmaf(cm[SP3D_CAM_A + 2], 1.0f, fmaf(cm[SP3D_CAM_A + 1], py, cm[SP3D_CAM_A + 0] * px));
    const float qy = fmaf(cm[SP3D_CAM_A + 5], 1.0f, fmaf(cm[SP3D_CAM_A + 4], py, cm[SP3D_CAM_A + 3] * px));
    if (cm[SP3D_CAM_FLIP] != 0.0f) qx = W_in - qx;
    const float ux = qx * (float)w / W_in;
    const float uy = qy * (float)h / H_in;
    float gx = ux / (float)(w - 1) * 2.0f - 1.0f;
    float gy = uy / (float)(h - 1) * 2.0f - 1.0f;
    gx = clamp_nan(gx, -1.1f, 1.1f);
    gy = clamp_nan(gy, -1.1f, 1.1f);
    ix = (gx + 1.0f) * ((float)(w - 1) / 2.0f);
    iy = (gy + 1.0f) * ((float)(h - 1) / 2.0f);
    return bound;
}

// bilinear weights, ATen CPU form: w = ix - floor(ix), e = 1 - w, ...
struct Bilin {
    int x0, y0;
    float wnw, wne, wsw, wse;
};
__device__ __forceinline__ Bilin bilin(float ix, float iy)
{
    Bilin b;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float wx = ix - fx0, ex = 1.0f - wx, ny = iy - fy0, sy = 1.0f - ny;
    b.wnw = sy * ex; b.wne = sy * wx; b.wsw = ny * ex; b.wse = ny * wx;
    b.x0 = (int)fx0; b.y0 = (int)fy0;
    return b;
}

// output of the view fusion for one channel (project_layer.py:96-99)
__device__ __forceinline__ float fuse(float acc, float den)
{
    float o = acc / den;
    if (o != o) o = 0.0f;
    return o < 0.0f ? 0.0f : (o > 1.0f ? 1.0f : o);
}

// blockIdx -> logical tile so that each XCD (observed: block b runs on XCD b % 8) walks a
// CONTIGUOUS range of (sample, voxel-tile) work and its 4 MiB L2 sees one sample's
// heat-maps instead of all of them.  Speed only: any placement gives the same results.
__device__ __forceinline__ int xcd_remap(int bid, int total)
{
    const int per = (total + 7) >> 3;
    return (bid & 7) * per + (bid >> 3);
}

} // namespace sp3d
