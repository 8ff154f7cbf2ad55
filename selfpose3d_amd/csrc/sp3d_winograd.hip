// sp3d_winograd.hip - input / output transforms of Winograd F(2x2x2, 3x3x3) for the low-resolution 3x3x3 convolutions
// of the V2V nets in inference (reference: the `Res3DBlock`s of lib/models/v2v_net.py:23-45 at 1/4 resolution).
//
// At 1/4 resolution (20x20x5 voxels, 128 channels, batch 4) a 3x3x3 convolution is a GEMM with only 8 000 rows;
// MIOpen's implicit-GEMM kernels reach 60 TFLOP/s there (112 us per layer).  Winograd needs 64 multiplies per 8 outputs
// instead of 216, as 64 independent (tiles x C) x (C x O) products - one batched fp32 GEMM (rocBLAS through torch.bmm,
// 36 us) - between two memory-bound transforms, which are these kernels:
//   wino_input : channels-last activations (B,X,Y,Z,C) -> V[64][tiles][C],  V = B^T d B  along each axis
//   wino_output: M[64][tiles][O] -> channels-last (B,X,Y,Z,O), y = A^T m A, fused with the layer's epilogue
//                (shift [+ residual] [+ ReLU], the modes of sp3d_channel_shift_act)
// lane = channel (coalesced 4-byte accesses across the channel-contiguous layouts), one thread = one (tile, channel).
// The weight transform U = G g G^T is done once per plan on the host side of the binding (torch).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sp3d.h"

namespace sp3d {

__device__ __forceinline__ void bt4(float &d0, float &d1, float &d2, float &d3)
{
    const float t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3;
    d0 = t0; d1 = t1; d2 = t2; d3 = t3;
}

__global__ __launch_bounds__(256) void wino_input_kernel(const float *__restrict__ x, float *__restrict__ V, int B, int X,
                                                        int Y, int Z, int C, int TX, int TY, int TZ)
{
    const int64_t T = (int64_t)B * TX * TY * TZ;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= T * C) return;
    const int c = (int)(gid % C);
    int64_t t = gid / C;
    const int tz = (int)(t % TZ); int64_t r = t / TZ;
    const int ty = (int)(r % TY); r /= TY;
    const int tx = (int)(r % TX);
    const int b = (int)(r / TX);
    float d[4][4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int xi = 2 * tx - 1 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yj = 2 * ty - 1 + j;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int zk = 2 * tz - 1 + k;
                const bool in = xi >= 0 && xi < X && yj >= 0 && yj < Y && zk >= 0 && zk < Z;
                d[i][j][k] = in ? x[((((int64_t)b * X + xi) * Y + yj) * Z + zk) * C + c] : 0.0f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) bt4(d[i][j][0], d[i][j][1], d[i][j][2], d[i][j][3]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) bt4(d[i][0][k], d[i][1][k], d[i][2][k], d[i][3][k]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) bt4(d[0][j][k], d[1][j][k], d[2][j][k], d[3][j][k]);
    const int64_t plane = T * C;
    float *v = V + t * C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[(int64_t)((i * 4 + j) * 4 + k) * plane] = d[i][j][k];
}

template <int MODE>
__global__ __launch_bounds__(256) void wino_output_kernel(const float *__restrict__ M, float *__restrict__ y,
                                                         const float *__restrict__ shift, const float *__restrict__ res,
                                                         int B, int X, int Y, int Z, int O, int TX, int TY, int TZ)
{
    const int64_t T = (int64_t)B * TX * TY * TZ;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= T * O) return;
    const int o = (int)(gid % O);
    int64_t t = gid / O;
    const int tz = (int)(t % TZ); int64_t r = t / TZ;
    const int ty = (int)(r % TY); r /= TY;
    const int tx = (int)(r % TX);
    const int b = (int)(r / TX);
    const int64_t plane = T * O;
    const float *m = M + t * O + o;
    // A^T along x while loading: two rows of (4 x 4)
    float a[2][4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float m0 = m[(int64_t)((0 * 4 + j) * 4 + k) * plane], m1 = m[(int64_t)((1 * 4 + j) * 4 + k) * plane];
            const float m2 = m[(int64_t)((2 * 4 + j) * 4 + k) * plane], m3 = m[(int64_t)((3 * 4 + j) * 4 + k) * plane];
            a[0][j][k] = (m0 + m1) + m2;
            a[1][j][k] = (m1 - m2) - m3;
        }
    const float sh = shift[o];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float bq[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bq[0][k] = (a[i][0][k] + a[i][1][k]) + a[i][2][k];
            bq[1][k] = (a[i][1][k] - a[i][2][k]) - a[i][3][k];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float c0 = (bq[j][0] + bq[j][1]) + bq[j][2], c1 = (bq[j][1] - bq[j][2]) - bq[j][3];
            const int xo = 2 * tx + i, yo = 2 * ty + j;
            if (xo >= X || yo >= Y) continue;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int zo = 2 * tz + k;
                if (zo >= Z) continue;
                const int64_t idx = ((((int64_t)b * X + xo) * Y + yo) * Z + zo) * O + o;
                float v = (k == 0 ? c0 : c1) + sh;
                if (MODE == 2) v += res[idx];
                if (MODE >= 1) v = fmaxf(v, 0.0f);
                if (MODE == 3) v += res[idx];
                y[idx] = v;
            }
        }
    }
}

} // namespace sp3d

using namespace sp3d;

extern "C" int sp3d_wino_input(const float *x, float *V, int B, int X, int Y, int Z, int C, void *stream)
{
    if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || C <= 0) return SP3D_EINVAL;
    if (!x || !V) return SP3D_ENULL;
    const int TX = (X + 1) / 2, TY = (Y + 1) / 2, TZ = (Z + 1) / 2;
    const int64_t n = (int64_t)B * TX * TY * TZ * C;
    if ((n + 255) / 256 > 0x7fffffff) return SP3D_ERANGE;
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, V, B, X, Y, Z,
                       C, TX, TY, TZ);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_wino_output(const float *M, float *y, const float *shift, const float *residual, int mode, int B, int X,
                                int Y, int Z, int O, void *stream)
{
    if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || O <= 0 || mode < 0 || mode > 3) return SP3D_EINVAL;
    if (!M || !y || !shift || (mode >= 2 && !residual)) return SP3D_ENULL;
    const int TX = (X + 1) / 2, TY = (Y + 1) / 2, TZ = (Z + 1) / 2;
    const int64_t n = (int64_t)B * TX * TY * TZ * O;
    if ((n + 255) / 256 > 0x7fffffff) return SP3D_ERANGE;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define SP3D_WO(MODE_) hipLaunchKernelGGL((wino_output_kernel<MODE_>), grid, block, 0, s, M, y, shift, residual, B, X, Y, Z, O, TX, TY, TZ)
    switch (mode) { case 0: SP3D_WO(0); break; case 1: SP3D_WO(1); break; case 2: SP3D_WO(2); break; default: SP3D_WO(3); }
#undef SP3D_WO
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}
