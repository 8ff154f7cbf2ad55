// sp3d_synth.hip - the synthetic-root branch of the self-supervised root net as two kernels
// (SURVEY.md §8 f3/f4).  Reference: /root/reference/lib/models/cuboid_proposal_net_soft.py
//   :168-203  per-root Python loop with .item() and searchsorted windows -> 3D Gaussian target volume
//   :205-227  cameras.project_pose_batch (lib/utils/cameras.py:58-108: NO r^2 clamp) + crop affine,
//             per-view / per-sample loops rendering sigma=3 Gaussians, sum over roots, clip
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sp3d.h"

namespace sp3d {

// target[b,x,y,z] = clip(max_r [ |g - mu_r| <= 3 sigma per axis ] * exp(-|g - mu_r|^2 / (2 sigma^2)), 0, 1)
__global__ __launch_bounds__(256) void gaussian_target_kernel(const float *__restrict__ roots, int R,
                                                             const float *__restrict__ gx,
                                                             const float *__restrict__ gy,
                                                             const float *__restrict__ gz, int X, int Y, int Z,
                                                             float sigma, float *__restrict__ target)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int N = X * Y * Z;
    if (n >= N) return;
    const int ix = n / (Y * Z), r0 = n - ix * (Y * Z), iy = r0 / Z, iz = r0 - iy * Z;
    const float x = gx[ix], y = gy[iy], z = gz[iz];
    const float lim = 3.0f * sigma, inv = 1.0f / (2.0f * sigma * sigma);
    float m = 0.0f;
    for (int r = 0; r < R; ++r) {
        const float *mu = roots + ((size_t)b * R + r) * 3;
        const float dx = x - mu[0], dy = y - mu[1], dz = z - mu[2];
        if (fabsf(dx) <= lim && fabsf(dy) <= lim && fabsf(dz) <= lim) {
            const float v = expf(-((dx * dx + dy * dy) + dz * dz) * inv);
            m = fmaxf(m, v);
        }
    }
    target[(size_t)b * N + n] = fminf(fmaxf(m, 0.0f), 1.0f);
}

// heat-map of the projected roots for one (view, sample): out[v,b,0,y,x] = clip(sum_r exp(-((x-qx)/3)^2/2 - ((y-qy)/3)^2/2))
// cam: (B,V,32) table of include/sp3d.h (its affine A = meta['trans']); stride = image px per heat-map px
__global__ __launch_bounds__(256) void render_roots_kernel(const float *__restrict__ roots, int R,
                                                          const float *__restrict__ cam, int B, int V, int h, int w,
                                                          float stride, float *__restrict__ out)
{
    __shared__ float sq[2 * SP3D_MAX_TOPK];
    const int v = blockIdx.z, b = blockIdx.y;
    const float *cm = cam + ((size_t)b * V + v) * SP3D_CAM_STRIDE;
    if (threadIdx.x < R) {
        const float *X = roots + ((size_t)b * R + threadIdx.x) * 3;
        const float dx = X[0] - cm[SP3D_CAM_T], dy = X[1] - cm[SP3D_CAM_T + 1], dz = X[2] - cm[SP3D_CAM_T + 2];
        const float xc = cm[0] * dx + cm[1] * dy + cm[2] * dz;
        const float yc = cm[3] * dx + cm[4] * dy + cm[5] * dz;
        const float zc = cm[6] * dx + cm[7] * dy + cm[8] * dz;
        const float y0 = xc / (zc + 1e-5f), y1 = yc / (zc + 1e-5f);
        const float r2 = y0 * y0 + y1 * y1;                                         // cameras.py:80 (no clamp)
        const float radial = 1.0f + cm[SP3D_CAM_K] * r2 + cm[SP3D_CAM_K + 1] * r2 * r2 + cm[SP3D_CAM_K + 2] * r2 * r2 * r2;
        const float tan = cm[SP3D_CAM_P] * y1 + cm[SP3D_CAM_P + 1] * y0;
        const float u0 = y0 * (radial + 2.0f * tan) + cm[SP3D_CAM_P + 1] * r2;
        const float u1 = y1 * (radial + 2.0f * tan) + cm[SP3D_CAM_P] * r2;
        const float px = cm[SP3D_CAM_F] * u0 + cm[SP3D_CAM_C], py = cm[SP3D_CAM_F + 1] * u1 + cm[SP3D_CAM_C + 1];
        sq[2 * threadIdx.x] = (cm[SP3D_CAM_A] * px + cm[SP3D_CAM_A + 1] * py + cm[SP3D_CAM_A + 2]) / stride;
        sq[2 * threadIdx.x + 1] = (cm[SP3D_CAM_A + 3] * px + cm[SP3D_CAM_A + 4] * py + cm[SP3D_CAM_A + 5]) / stride;
    }
    __syncthreads();
    float *o = out + ((size_t)v * B + b) * h * w;
    for (int pidx = blockIdx.x * 256 + threadIdx.x; pidx < h * w; pidx += gridDim.x * 256) {
        const float yy = (float)(pidx / w), xx = (float)(pidx % w);
        float s = 0.0f;
        for (int r = 0; r < R; ++r) {
            const float ex = (xx - sq[2 * r]) / 3.0f, ey = (yy - sq[2 * r + 1]) / 3.0f;
            s += expf(-(ex * ex) / 2.0f - (ey * ey) / 2.0f);
        }
        o[pidx] = fminf(fmaxf(s, 0.0f), 1.0f);
    }
}

} // namespace sp3d

using namespace sp3d;

extern "C" int sp3d_gaussian_target_3d(const float *roots, int B, int R, const float *gx, const float *gy,
                                       const float *gz, int X, int Y, int Z, float sigma, float *target, void *stream)
{
    if (B <= 0 || R <= 0 || X <= 0 || Y <= 0 || Z <= 0 || !(sigma > 0.0f)) return SP3D_EINVAL;
    if (!roots || !gx || !gy || !gz || !target) return SP3D_ENULL;
    const int64_t N = (int64_t)X * Y * Z;
    if (N > 0x7ffffffe) return SP3D_ERANGE;
    hipLaunchKernelGGL(gaussian_target_kernel, dim3((unsigned)((N + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, roots,
                       R, gx, gy, gz, X, Y, Z, sigma, target);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_render_root_heatmaps(const float *roots, int B, int R, const float *cam, int V, int h, int w,
                                         float stride, float *out, void *stream)
{
    if (B <= 0 || R <= 0 || R > SP3D_MAX_TOPK || V <= 0 || V > SP3D_MAX_VIEWS || h <= 0 || w <= 0 || !(stride > 0.0f))
        return SP3D_EINVAL;
    if (!roots || !cam || !out) return SP3D_ENULL;
    const int blocks = (h * w + 255) / 256 < 64 ? (h * w + 255) / 256 : 64;
    hipLaunchKernelGGL(render_roots_kernel, dim3(blocks, B, V), dim3(256), 0, (hipStream_t)stream, roots, R, cam, B, V, h, w,
                       stride, out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

// ------------------------------------------------------------------------------------------
// Differentiable joint heat-map rendering of the self-supervised pose loss (reference:
// lib/models/multi_person_posenet_ssv.py:409-465): for every (view, sample) the projected joints of the P predicted
// people are rendered as sigma-3 Gaussians, summed over people and clipped,
//     out[n, j, y, x] = clip( sum_p exp(-((x - kx)/s)^2/2 - ((y - ky)/s)^2/2), 0, 1 ),     kps (N, P, J, 2) heat-map px
// (the reference builds a (P, J, h, w) temporary per (view, sample) in Python loops).  Backward: torch.clip passes the
// gradient where the sum is <= 1; d/dkx of a Gaussian is g * (x - kx) / s^2 - one workgroup per (n, j) recomputes the
// sum per pixel and reduces the 2P partial derivatives over the image.
// ------------------------------------------------------------------------------------------
namespace sp3d {

constexpr int RJ_MAXP = 16;

__global__ __launch_bounds__(256) void render_joints_fwd_kernel(const float *__restrict__ kps, const int *__restrict__ count,
                                                               int P, int J, int h, int w, float sigma,
                                                               float *__restrict__ out)
{
    __shared__ float sk[2 * RJ_MAXP];
    const int n = blockIdx.z, j = blockIdx.y;
    const int np = count ? min(count[n], P) : P;
    if (threadIdx.x < np) {
        const float *k = kps + (((size_t)n * P + threadIdx.x) * J + j) * 2;
        sk[2 * threadIdx.x] = k[0]; sk[2 * threadIdx.x + 1] = k[1];
    }
    __syncthreads();
    float *o = out + ((size_t)n * J + j) * h * w;
    for (int pidx = blockIdx.x * 256 + threadIdx.x; pidx < h * w; pidx += gridDim.x * 256) {
        const float yy = (float)(pidx / w), xx = (float)(pidx % w);
        float s = 0.0f;
        for (int p = 0; p < np; ++p) {
            const float ex = (xx - sk[2 * p]) / sigma, ey = (yy - sk[2 * p + 1]) / sigma;
            s += expf(-(ex * ex) / 2.0f - (ey * ey) / 2.0f);
        }
        o[pidx] = fminf(fmaxf(s, 0.0f), 1.0f);
    }
}

__global__ __launch_bounds__(256) void render_joints_bwd_kernel(const float *__restrict__ kps, const int *__restrict__ count,
                                                               const float *__restrict__ gout, int P, int J, int h, int w,
                                                               float sigma, float *__restrict__ gkps)
{
    __shared__ float sk[2 * RJ_MAXP];
    __shared__ float red[4][2 * RJ_MAXP];
    const int n = blockIdx.y, j = blockIdx.x;
    const int np = count ? min(count[n], P) : P;
    if (threadIdx.x < np) {
        const float *k = kps + (((size_t)n * P + threadIdx.x) * J + j) * 2;
        sk[2 * threadIdx.x] = k[0]; sk[2 * threadIdx.x + 1] = k[1];
    }
    __syncthreads();
    float acc[2 * RJ_MAXP];
#pragma unroll
    for (int q = 0; q < 2 * RJ_MAXP; ++q) acc[q] = 0.0f;
    const float *g = gout + ((size_t)n * J + j) * h * w;
    const float inv2 = 1.0f / (sigma * sigma);
    for (int pidx = threadIdx.x; pidx < h * w; pidx += 256) {
        const float yy = (float)(pidx / w), xx = (float)(pidx % w);
        float gp[RJ_MAXP];
        float s = 0.0f;
#pragma unroll
        for (int p = 0; p < RJ_MAXP; ++p) {
            gp[p] = 0.0f;
            if (p < np) {
                const float ex = (xx - sk[2 * p]) / sigma, ey = (yy - sk[2 * p + 1]) / sigma;
                gp[p] = expf(-(ex * ex) / 2.0f - (ey * ey) / 2.0f);
                s += gp[p];
            }
        }
        const float go = (s <= 1.0f) ? g[pidx] : 0.0f;            // clip(., 0, 1) backward; the sum is never negative
#pragma unroll
        for (int p = 0; p < RJ_MAXP; ++p)
            if (p < np) {
                const float c = go * gp[p] * inv2;
                acc[2 * p] += c * (xx - sk[2 * p]);
                acc[2 * p + 1] += c * (yy - sk[2 * p + 1]);
            }
    }
    // wave reduction, then across the four waves
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 2 * RJ_MAXP; ++q) {
        float v = acc[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) red[wv][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 2 * P) {
        const int p = threadIdx.x >> 1, c = threadIdx.x & 1;
        float v = 0.0f;
        if (p < np) v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        gkps[(((size_t)n * P + p) * J + j) * 2 + c] = v;
    }
}

} // namespace sp3d

extern "C" int sp3d_render_joints_fwd(const float *kps, const int *count, int N, int P, int J, int h, int w, float sigma,
                                      float *out, void *stream)
{
    if (N <= 0 || P <= 0 || P > sp3d::RJ_MAXP || J <= 0 || h <= 0 || w <= 0 || !(sigma > 0.0f)) return SP3D_EINVAL;
    if (!kps || !out) return SP3D_ENULL;
    if (N > 65535 || J > 65535) return SP3D_ERANGE;
    const int blocks = (h * w + 255) / 256 < 32 ? (h * w + 255) / 256 : 32;
    hipLaunchKernelGGL(sp3d::render_joints_fwd_kernel, dim3(blocks, J, N), dim3(256), 0, (hipStream_t)stream, kps, count, P, J, h,
                       w, sigma, out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_render_joints_bwd(const float *kps, const int *count, const float *grad_out, int N, int P, int J, int h,
                                      int w, float sigma, float *grad_kps, void *stream)
{
    if (N <= 0 || P <= 0 || P > sp3d::RJ_MAXP || J <= 0 || h <= 0 || w <= 0 || !(sigma > 0.0f)) return SP3D_EINVAL;
    if (!kps || !grad_out || !grad_kps) return SP3D_ENULL;
    if (N > 65535) return SP3D_ERANGE;
    hipLaunchKernelGGL(sp3d::render_joints_bwd_kernel, dim3(J, N), dim3(256), 0, (hipStream_t)stream, kps, count, grad_out, P, J,
                       h, w, sigma, grad_kps);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}
