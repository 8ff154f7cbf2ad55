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
    // residual values first (clamped addresses, no predicate): inside the bounds branches below each was a load -> wait ->
    // store round trip of its own
    float rv[2][2][2];
    if (MODE >= 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int xo = min(2 * tx + i, X - 1), yo = min(2 * ty + j, Y - 1), zo = min(2 * tz + k, Z - 1);
                    rv[i][j][k] = res[((((int64_t)b * X + xo) * Y + yo) * Z + zo) * O + o];
                }
    }
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
                if (MODE == 2) v += rv[i][j][k];
                if (MODE >= 1) v = fmaxf(v, 0.0f);
                if (MODE == 3) v += rv[i][j][k];
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

// ------------------------------------------------------------------------------------------
// Fused Winograd F(2x2x2, 3x3x3) for the FULL-resolution 3x3x3 layers (C = 16 or 32 -> O = 32), where the
// transformed tensor of the three-launch form above would be 524 MB.  One wave = a block of 4x4x2 tiles (8x8x4
// outputs x 32 channels); nothing but x, U and y touches memory:
//   per chunk of 8 input channels: stage the 10x10x6 input region in LDS (20.2 KB, 16-byte aligned rows);
//   per transform point: each lane builds its A operand on the fly - V[p][tile][c] is a signed sum of 8 region
//     voxels (B^T has two non-zeros per row) - B = U[p][c][o] streams from L2, and v_mfma_f32_32x32x2_f32
//     (tiles x outputs, K = 8 channels) accumulates;
//   the inverse transform is linear: along x it is folded into the MFMA accumulation, along y,z it is applied to
//     the accumulators on the VALU; nothing transformed is ever stored.
// ------------------------------------------------------------------------------------------
#ifndef SP3D_W16_ABLATE
#define SP3D_W16_ABLATE 0      // measurement builds only (tools/diag_w16.py): 1 no MFMA, 2 no weight loads, 4 no split, 8 no LDS reads
#endif

namespace sp3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef SP3D_WF_TIMELINE
__device__ unsigned long long *g_wf_tl = nullptr;
#define WF_STAMP(slot) do { __builtin_amdgcn_sched_barrier(0); if (tl && lane == 0) tl[slot] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WF_STAMP(slot) do { } while (0)
#endif

constexpr int WF_RX = 10, WF_RY = 10, WF_RZ = 6;
constexpr int WF_VS = 8;                       // floats per staged voxel (one chunk of 8 input channels)
// Row pitch 84 floats: rows stay 16-byte aligned, so a lane fetches its 4 channels of a voxel with ONE ds_read_b128 (and the
// staging writes are ds_write_b128).  With stride-2 tiles the 16 lanes of a b128 lane group can spread over only 32 of
// the 64 banks whatever the pitch (2*tty*ROW + 2*ttz*PLANE is a multiple of 8 floats): 2-way conflicts, 8 LDS cycles per
// instruction, 16 instructions per (j,k) step = 128 cycles - against 64 ds_read_b32 x >= 4 cycles with the former odd
// pitch (profiles/r01_pmc_wino_fused.json: 74 % of the LDS cycles were conflicts).
constexpr int WF_ROW = WF_RX * WF_VS + 4;
constexpr int WF_LDS = WF_RY * WF_RZ * WF_ROW; // 5 040 floats = 20 160 B -> 8 waves per CU (161 280 of 163 840 B)

// B^T rows as (first tap +, second tap, sign of second): d0-d2, d1+d2, d2-d1, d1-d3
__device__ constexpr int wf_ta(int r) { return r == 0 ? 0 : (r == 1 ? 1 : (r == 2 ? 2 : 1)); }
__device__ constexpr int wf_tb(int r) { return r == 0 ? 2 : (r == 1 ? 2 : (r == 2 ? 1 : 3)); }
__device__ constexpr float wf_sb(int r) { return r == 1 ? 1.0f : -1.0f; }
// A^T = [[1,1,1,0],[0,1,-1,-1]]
__device__ constexpr float wf_at(int a, int r) { return a == 0 ? (r < 3 ? 1.0f : 0.0f) : (r == 0 ? 0.0f : (r == 1 ? 1.0f : -1.0f)); }

template <int C, int MODE>
__global__ __launch_bounds__(64) void wino_fused_kernel(const float *__restrict__ x, const float *__restrict__ U,
                                                       float *__restrict__ y, const float *__restrict__ shift,
                                                       const float *__restrict__ res, int B, int X, int Y, int Z, int NBX,
                                                       int NBY, int NBZ)
{
    constexpr int O = 32;
    __shared__ __attribute__((aligned(16))) float region[WF_LDS];
    const int lane = threadIdx.x, t = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int bz = bid % NBZ; bid /= NBZ;
    const int by = bid % NBY; bid /= NBY;
    const int bx = bid % NBX;
    const int b = bid / NBX;
    const int ttx = t & 3, tty = (t >> 2) & 3, ttz = t >> 4;
    const int ox0 = bx * 8, oy0 = by * 8, oz0 = bz * 4;                 // first output voxel of the block
    // region address of (vx,vy,vz,c) = (vz*RY + vy)*ROW + vx*VS + c; this tile's patch origin.  MFMA k-slot h of step kk
    // carries channel kk + 4*h of the chunk, so a lane's four channels (kk = 0..3) are 16 contiguous bytes
    const float *rb = region + ((2 * ttz) * WF_RY + 2 * tty) * WF_ROW + (2 * ttx) * WF_VS + 4 * h;

    f32x16 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[a][v] = 0.0f;
#ifdef SP3D_WF_TIMELINE
    unsigned long long *tl = g_wf_tl ? g_wf_tl + (size_t)blockIdx.x * 80 : nullptr;
#endif
    WF_STAMP(0);

#pragma unroll 1
    for (int cc = 0; cc < C / 8; ++cc) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // stage 600 voxels x 8 channels = 1200 float4, 5 loads in flight per lane
#pragma unroll 1
        for (int i0 = 0; i0 < WF_RX * WF_RY * WF_RZ * 2; i0 += 64 * 5) {
            float4 d[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int idx = i0 + u * 64 + lane;
                const int v = idx >> 1, half = idx & 1;
                const int vx = v % WF_RX, vy = (v / WF_RX) % WF_RY, vz = v / (WF_RX * WF_RY);
                const int gx = ox0 - 1 + vx, gy = oy0 - 1 + vy, gz = oz0 - 1 + vz;
                d[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (idx < WF_RX * WF_RY * WF_RZ * 2 && gx >= 0 && gx < X && gy >= 0 && gy < Y && gz >= 0 && gz < Z)
                    d[u] = *reinterpret_cast<const float4 *>(x + ((((int64_t)b * X + gx) * Y + gy) * Z + gz) * C + cc * 8 + half * 4);
            }
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int idx = i0 + u * 64 + lane;
                if (idx < WF_RX * WF_RY * WF_RZ * 2) {
                    const int v = idx >> 1, half = idx & 1;
                    const int vx = v % WF_RX, vy = (v / WF_RX) % WF_RY, vz = v / (WF_RX * WF_RY);
                    *reinterpret_cast<float4 *>(region + (vz * WF_RY + vy) * WF_ROW + vx * WF_VS + half * 4) = d[u];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float *ub = U + ((int64_t)(cc * 8 + 4 * h)) * O + t;         // + p*C*O + kk*O
        // (y,z) index of the transform point: run-time loop; x index: unrolled.  The inverse transform along x is
        // folded into the MFMA accumulation (two accumulators, A^T = [1,1,1,0] / [0,1,-1,-1] as +-A operands), the one
        // along y,z is applied once per (j,k) on the VALU with wave-uniform coefficients.
        // one (j,k) step; bcur = its 16 B operands (fetched one step ahead: they come from L2, ~700 cycles away),
        // bnxt receives those of step jk_next
        auto step = [&](int jk, int jk_next, const float (&bcur)[16], float (&bnxt)[16]) {
            const int j = jk >> 2, k = jk & 3;
            const int ya = (j == 0) ? 0 : ((j == 2) ? 2 : 1), yb = (j == 3) ? 3 : ((j == 2) ? 1 : 2);
            const int za = (k == 0) ? 0 : ((k == 2) ? 2 : 1), zb = (k == 3) ? 3 : ((k == 2) ? 1 : 2);
            const float sy = (j == 1) ? 1.0f : -1.0f, sz = (k == 1) ? 1.0f : -1.0f;
            const float *r00 = rb + (za * WF_RY + ya) * WF_ROW, *r10 = rb + (za * WF_RY + yb) * WF_ROW;
            const float *r01 = rb + (zb * WF_RY + ya) * WF_ROW, *r11 = rb + (zb * WF_RY + yb) * WF_ROW;
            const float *un = ub + (int64_t)jk_next * C * O;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) bnxt[i * 4 + kk] = un[(int64_t)(i * 16) * C * O + kk * O];
            float g[4][4];                                                 // [x tap][kk]: y,z transform done
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                const float4 v00 = *reinterpret_cast<const float4 *>(r00 + xi * WF_VS), v10 = *reinterpret_cast<const float4 *>(r10 + xi * WF_VS);
                const float4 v01 = *reinterpret_cast<const float4 *>(r01 + xi * WF_VS), v11 = *reinterpret_cast<const float4 *>(r11 + xi * WF_VS);
                g[xi][0] = fmaf(sz, fmaf(sy, v11.x, v01.x), fmaf(sy, v10.x, v00.x));
                g[xi][1] = fmaf(sz, fmaf(sy, v11.y, v01.y), fmaf(sy, v10.y, v00.y));
                g[xi][2] = fmaf(sz, fmaf(sy, v11.z, v01.z), fmaf(sy, v10.z, v00.z));
                g[xi][3] = fmaf(sz, fmaf(sy, v11.w, v01.w), fmaf(sy, v10.w, v00.w));
            }
            if (cc == 0) WF_STAMP(8 + 4 * jk);                             // operands ready
            // wave-uniform y,z coefficients of A^T for the four (b,c) output positions
            float cyz[4];
#pragma unroll
            for (int bc = 0; bc < 4; ++bc) {
                const int bb = bc >> 1, c2 = bc & 1;
                const float cy = bb == 0 ? (j < 3 ? 1.0f : 0.0f) : (j == 0 ? 0.0f : (j == 1 ? 1.0f : -1.0f));
                const float cz = c2 == 0 ? (k < 3 ? 1.0f : 0.0f) : (k == 0 ? 0.0f : (k == 1 ? 1.0f : -1.0f));
                cyz[bc] = cy * cz;
            }
            f32x16 M0, M1;
#pragma unroll
            for (int v = 0; v < 16; ++v) { M0[v] = 0.0f; M1[v] = 0.0f; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float av = (i == 1) ? g[1][kk] + g[2][kk] : g[wf_ta(i)][kk] - g[wf_tb(i)][kk];
                    const float bv = bcur[i * 4 + kk];
                    if (i < 3) M0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, M0, 0, 0, 0);
                    if (i > 0) M1 = __builtin_amdgcn_mfma_f32_32x32x2f32(i == 1 ? av : -av, bv, M1, 0, 0, 0);
                }
            if (cc == 0) WF_STAMP(9 + 4 * jk);                             // MFMAs issued
#pragma unroll
            for (int bc = 0; bc < 4; ++bc) {
                const float coef = cyz[bc];
                if (coef != 0.0f) {
                    f32x16 cv;
#pragma unroll
                    for (int v = 0; v < 16; ++v) cv[v] = coef;
                    acc[bc] = __builtin_elementwise_fma(M0, cv, acc[bc]);
                    acc[4 + bc] = __builtin_elementwise_fma(M1, cv, acc[4 + bc]);
                }
            }
            if (cc == 0) WF_STAMP(10 + 4 * jk);                            // accumulated
        };
        float b0[16], b1[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) b0[i * 4 + kk] = ub[(int64_t)(i * 16) * C * O + kk * O];
#pragma unroll 1
        for (int jk = 0; jk < 16; jk += 2) {
            step(jk, jk + 1, b0, b1);
            step(jk + 1, (jk + 2) & 15, b1, b0);
        }
    }

    WF_STAMP(5);                                                           // all chunks done
    const float sh = shift[t];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        // residual values of this output position first, 16 loads in flight (clamped addresses, no predicate): inside the
        // bounds branch they came out as load -> s_waitcnt vmcnt(0) -> store chains, one memory round trip per element
        float rv[16];
        if (MODE >= 2) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = 8 * (v >> 2) + (v & 3) + 4 * h;
                const int rx = row & 3, ry = (row >> 2) & 3, rz = row >> 4;
                const int xo = min(ox0 + 2 * rx + (a >> 2), X - 1), yo = min(oy0 + 2 * ry + ((a >> 1) & 1), Y - 1);
                const int zo = min(oz0 + 2 * rz + (a & 1), Z - 1);
                rv[v] = res[((((int64_t)b * X + xo) * Y + yo) * Z + zo) * O + t];
            }
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = 8 * (v >> 2) + (v & 3) + 4 * h;              // tile of this accumulator element
            const int rx = row & 3, ry = (row >> 2) & 3, rz = row >> 4;
            const int xo = ox0 + 2 * rx + (a >> 2), yo = oy0 + 2 * ry + ((a >> 1) & 1), zo = oz0 + 2 * rz + (a & 1);
            if (xo < X && yo < Y && zo < Z) {
                const int64_t idx = ((((int64_t)b * X + xo) * Y + yo) * Z + zo) * O + t;
                float val = acc[a][v] + sh;
                if (MODE == 2) val += rv[v];
                if (MODE >= 1) val = fmaxf(val, 0.0f);
                if (MODE == 3) val += rv[v];
                y[idx] = val;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// The same kernel with the products on the bf16 matrix pipe at fp32 accuracy.  v_mfma_f32_32x32x2_f32 runs at 1/16 of
// the bf16 rate and (profiles/r02_pmc_wino_fused.json) keeps the matrix pipe 75 % busy at a throttled clock.
// Every fp32 operand is split into three bf16 pieces a = hi + mid + lo (8+8+8 mantissa bits: exact), the six products
// whose weight is >= 2^-16 relative - hh, hm, mh, hl, lh, mm - are formed exactly by v_mfma_f32_32x32x16_bf16
// (bf16 x bf16 fits fp32) and accumulated in fp32; the dropped terms are < 2^-24 relative, i.e. below one fp32 ulp of
// the product.  K = 16 of one MFMA = 4 channels x 2 (A piece, B piece) pairs per lane half:
//     {hi,hi} x {bm,bh}   +   {mid,mid} x {bm,bh}   +   {lo,hi} x {bh,bl}
// so 8 channels cost 3 MFMAs of 8 passes instead of 4 of 16: 2.7x fewer matrix cycles; the weights are split once on
// the host (U3: per (point, chunk, lane half, output) one 24-byte record [bm(4ch) bh(4ch) bl(4ch)]).
// ------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef unsigned u32x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct WfB { u32x4 mh; u32x2 l; };              // [bm01 bm23 bh01 bh23] [bl01 bl23]

template <int C, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_fused3_kernel(const float *__restrict__ x, const unsigned *__restrict__ U3,
                                                        float *__restrict__ y, const float *__restrict__ shift,
                                                        const float *__restrict__ res, int B, int X, int Y, int Z, int NBX,
                                                        int NBY, int NBZ)
{
    constexpr int O = 32, NCH = C / 8;
    __shared__ __attribute__((aligned(16))) float region[WF_LDS];
    const int lane = threadIdx.x, t = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int bz = bid % NBZ; bid /= NBZ;
    const int by = bid % NBY; bid /= NBY;
    const int bx = bid % NBX;
    const int b = bid / NBX;
    const int ttx = t & 3, tty = (t >> 2) & 3, ttz = t >> 4;
    const int ox0 = bx * 8, oy0 = by * 8, oz0 = bz * 4;
    const float *rb = region + ((2 * ttz) * WF_RY + 2 * tty) * WF_ROW + (2 * ttx) * WF_VS + 4 * h;

    f32x16 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[a][v] = 0.0f;

    // staging constants of this lane: region voxel (x = u >> 1, y = 5 * (u & 1) + sr, z = sz_), channel half sh_
    const int sh_ = lane & 1, sz_ = (lane >> 1) % WF_RZ, sr = (lane >> 1) / WF_RZ;       // lanes 60..63 idle
    const int sgz = oz0 - 1 + sz_, sgy0 = oy0 - 1 + sr, sgy1 = sgy0 + 5;
    const bool sg_inz = sgz >= 0 && sgz < Z, sg_iny0 = sgy0 >= 0 && sgy0 < Y, sg_iny1 = sgy1 >= 0 && sgy1 < Y;
    const int64_t sg_base = ((((int64_t)b * X + (ox0 - 1)) * Y + sgy0) * Z + sgz) * C + sh_ * 4;
    const int sg_lds = (sz_ * WF_RY + sr) * WF_ROW + sh_ * 4;

#pragma unroll 1
    for (int cc = 0; cc < NCH; ++cc) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // stage 600 voxels x 8 channels = 1200 float4, 5 loads in flight per lane (10 or all 19 at once measured no
        // faster: the accumulators start to spill).  Lane = (half, z, column-of-5): the (x,y) column of batch element u
        // is (u >> 1, 5 * (u & 1) + r), so every address is a per-lane constant plus a compile-time offset - the former
        // idx -> (x,y,z) divisions cost 60 VALU instructions per element, a fifth of the kernel's instruction count
        {
            const bool on = lane < 60;
            const int64_t gbase = sg_base + cc * 8;                           // + (x * Y + y5) * Z * C per element
#pragma unroll 1
            for (int u0 = 0; u0 < 20; u0 += 5) {
                float4 d[5];
#pragma unroll
                for (int uu = 0; uu < 5; ++uu) {
                    const int u = u0 + uu;
                    const int gx = ox0 - 1 + (u >> 1);
                    const bool iny = (u & 1) ? sg_iny1 : sg_iny0;
                    d[uu] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (on && sg_inz && iny && gx >= 0 && gx < X)
                        d[uu] = *reinterpret_cast<const float4 *>(x + gbase + ((int64_t)(u >> 1) * Y + 5 * (u & 1)) * Z * C);
                }
#pragma unroll
                for (int uu = 0; uu < 5; ++uu) {
                    const int u = u0 + uu;
                    if (on) *reinterpret_cast<float4 *>(region + sg_lds + 5 * (u & 1) * WF_ROW + (u >> 1) * WF_VS) = d[uu];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // record of (point p = i*16 + jk, chunk cc, lane half h, output t): 6 dwords
        const unsigned *ub = U3 + (((int64_t)cc * 2 + h) * 32 + t) * 6;
        auto load_b = [&](int i, int jk) {
            const unsigned *r = ub + (int64_t)(i * 16 + jk) * NCH * 2 * 32 * 6;
            WfB w;
#if SP3D_W16_ABLATE & 2
            (void)r;
            w.mh = u32x4{0x3f803f80u + (unsigned)jk, 0x3f803f80u, 0x3f803f80u + (unsigned)i, 0x3f803f80u};
            w.l = u32x2{0x3f803f80u, 0x3f803f80u};
#else
            w.mh = *reinterpret_cast<const u32x4_a8 *>(r);
            w.l = *reinterpret_cast<const u32x2_a8 *>(r + 4);
#endif
            return w;
        };
        // B of x points 0,1 arrives one step ahead; B of x points 2,3 is fetched at the top of its own step (it is
        // needed ~600 cycles later): 36 registers of weights in flight instead of 48
        auto step = [&](int jk, int jk_next, const WfB (&bcur)[2], WfB (&bnxt)[2]) {
            const int j = jk >> 2, k = jk & 3;
            const int ya = (j == 0) ? 0 : ((j == 2) ? 2 : 1), yb = (j == 3) ? 3 : ((j == 2) ? 1 : 2);
            const int za = (k == 0) ? 0 : ((k == 2) ? 2 : 1), zb = (k == 3) ? 3 : ((k == 2) ? 1 : 2);
            const float sy = (j == 1) ? 1.0f : -1.0f, sz = (k == 1) ? 1.0f : -1.0f;
            const float *r00 = rb + (za * WF_RY + ya) * WF_ROW, *r10 = rb + (za * WF_RY + yb) * WF_ROW;
            const float *r01 = rb + (zb * WF_RY + ya) * WF_ROW, *r11 = rb + (zb * WF_RY + yb) * WF_ROW;
            WfB blate[2];
            blate[0] = load_b(2, jk);
            blate[1] = load_b(3, jk);
            float g[4][4];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
#if SP3D_W16_ABLATE & 8
                const float f0 = __int_as_float(0x3f800000 + jk + xi), f1 = __int_as_float(0x3f900000 + lane);
                const float4 v00 = make_float4(f0, f1, f0, f1), v10 = make_float4(f1, f0, f1, f0), v01 = v00, v11 = v10;
                (void)r00; (void)r10; (void)r01; (void)r11;
#else
                const float4 v00 = *reinterpret_cast<const float4 *>(r00 + xi * WF_VS), v10 = *reinterpret_cast<const float4 *>(r10 + xi * WF_VS);
                const float4 v01 = *reinterpret_cast<const float4 *>(r01 + xi * WF_VS), v11 = *reinterpret_cast<const float4 *>(r11 + xi * WF_VS);
#endif
                g[xi][0] = fmaf(sz, fmaf(sy, v11.x, v01.x), fmaf(sy, v10.x, v00.x));
                g[xi][1] = fmaf(sz, fmaf(sy, v11.y, v01.y), fmaf(sy, v10.y, v00.y));
                g[xi][2] = fmaf(sz, fmaf(sy, v11.z, v01.z), fmaf(sy, v10.z, v00.z));
                g[xi][3] = fmaf(sz, fmaf(sy, v11.w, v01.w), fmaf(sy, v10.w, v00.w));
            }
            float cyz[4];
#pragma unroll
            for (int bc = 0; bc < 4; ++bc) {
                const int bb = bc >> 1, c2 = bc & 1;
                const float cy = bb == 0 ? (j < 3 ? 1.0f : 0.0f) : (j == 0 ? 0.0f : (j == 1 ? 1.0f : -1.0f));
                const float cz = c2 == 0 ? (k < 3 ? 1.0f : 0.0f) : (k == 0 ? 0.0f : (k == 1 ? 1.0f : -1.0f));
                cyz[bc] = cy * cz;
            }
            f32x16 M0, M1;
#pragma unroll
            for (int v = 0; v < 16; ++v) { M0[v] = 0.0f; M1[v] = 0.0f; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // x transform: d0-d2, d1+d2, d2-d1, d1-d3; point 3 enters M1 with a minus sign: negate it here
                float av[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    av[kk] = (i == 1) ? g[1][kk] + g[2][kk] : ((i == 3) ? g[3][kk] - g[1][kk] : g[wf_ta(i)][kk] - g[wf_tb(i)][kk]);
                // a = hi + mid + lo, each a bf16 (exact: 24 mantissa bits)
                const unsigned hi01 = pack_bf16(av[0], av[1]), hi23 = pack_bf16(av[2], av[3]);
#if SP3D_W16_ABLATE & 4
                const unsigned mid01 = hi01, mid23 = hi23, lo01 = hi01, lo23 = hi23;
#else
                const float r0 = av[0] - bf16_lo(hi01), r1 = av[1] - bf16_hi(hi01), r2 = av[2] - bf16_lo(hi23), r3 = av[3] - bf16_hi(hi23);
                const unsigned mid01 = pack_bf16(r0, r1), mid23 = pack_bf16(r2, r3);
                const unsigned lo01 = pack_bf16(r0 - bf16_lo(mid01), r1 - bf16_hi(mid01));
                const unsigned lo23 = pack_bf16(r2 - bf16_lo(mid23), r3 - bf16_hi(mid23));
#endif
                const u32x4 Qhh = {hi01, hi23, hi01, hi23}, Qmm = {mid01, mid23, mid01, mid23}, Qlh = {lo01, lo23, hi01, hi23};
                const WfB &w = (i < 2) ? bcur[i] : blate[i - 2];
                const u32x4 Bmh = w.mh;
                const u32x4 Bhl = {w.mh.z, w.mh.w, w.l.x, w.l.y};
#if SP3D_W16_ABLATE & 1
                {   // no matrix instructions: keep every operand alive with one integer op each
                    const unsigned z = (Qhh.x ^ Bmh.x) + (Qmm.y ^ Bmh.z) + (Qlh.x ^ Bhl.w) + (Qlh.z ^ Bhl.y);
                    M0[0] += __uint_as_float(z & 0x3fffffffu);
                    M1[1] += __uint_as_float((z >> 1) & 0x3fffffffu);
                    if (i == 1) {
                        bnxt[0] = load_b(0, jk_next);
                        bnxt[1] = load_b(1, jk_next);
                    }
                    continue;
                }
#endif
                if (i < 3) {
                    M0 = mfma_bf16(Qhh, Bmh, M0);
                    M0 = mfma_bf16(Qmm, Bmh, M0);
                    M0 = mfma_bf16(Qlh, Bhl, M0);
                }
                if (i == 1 || i == 3) {
                    M1 = mfma_bf16(Qhh, Bmh, M1);
                    M1 = mfma_bf16(Qmm, Bmh, M1);
                    M1 = mfma_bf16(Qlh, Bhl, M1);
                }
                if (i == 2) {                                      // M1 -= A.B: flip the sign bits of the weights
                    const u32x4 nmh = Bmh ^ 0x80008000u, nhl = Bhl ^ 0x80008000u;
                    M1 = mfma_bf16(Qhh, nmh, M1);
                    M1 = mfma_bf16(Qmm, nmh, M1);
                    M1 = mfma_bf16(Qlh, nhl, M1);
                }
                if (i == 1) {                                      // next step's early weights, mid-step
                    bnxt[0] = load_b(0, jk_next);
                    bnxt[1] = load_b(1, jk_next);
                }
            }
#pragma unroll
            for (int bc = 0; bc < 4; ++bc) {
                const float coef = cyz[bc];
                if (coef != 0.0f) {
                    f32x16 cv;
#pragma unroll
                    for (int v = 0; v < 16; ++v) cv[v] = coef;
                    acc[bc] = __builtin_elementwise_fma(M0, cv, acc[bc]);
                    acc[4 + bc] = __builtin_elementwise_fma(M1, cv, acc[4 + bc]);
                }
            }
        };
        WfB b0[2], b1[2];
        b0[0] = load_b(0, 0);
        b0[1] = load_b(1, 0);
#pragma unroll 1
        for (int jk = 0; jk < 16; jk += 2) {
            step(jk, jk + 1, b0, b1);
            step(jk + 1, (jk + 2) & 15, b1, b0);
        }
    }

    // accumulator element (a, v) of lane (t, h) is output voxel (ox0 + dx, oy0 + 2h + dy, oz0 + dz), channel t, with
    // dx = 2 (v & 3) + (a >> 2), dy = 4 ((v >> 2) & 1) + ((a >> 1) & 1), dz = 2 (v >> 3) + (a & 1) known at compile time:
    // one per-lane base address, wave-uniform offsets and bounds (only the y bound depends on the lane)
    const float sh = shift[t];
    const int yl = oy0 + 2 * h;
    const int64_t obase = ((((int64_t)b * X + ox0) * Y + yl) * Z + oz0) * O + t;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        float rv[16];                                 // residual values first (see wino_fused_kernel): clamped, unpredicated
        if (MODE >= 2) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int dx = 2 * (v & 3) + (a >> 2), dy = 4 * ((v >> 2) & 1) + ((a >> 1) & 1), dz = 2 * (v >> 3) + (a & 1);
                const int xo = min(ox0 + dx, X - 1), yo = min(yl + dy, Y - 1), zo = min(oz0 + dz, Z - 1);
                rv[v] = res[((((int64_t)b * X + xo) * Y + yo) * Z + zo) * O + t];
            }
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int dx = 2 * (v & 3) + (a >> 2), dy = 4 * ((v >> 2) & 1) + ((a >> 1) & 1), dz = 2 * (v >> 3) + (a & 1);
            if (ox0 + dx < X && oz0 + dz < Z && yl + dy < Y) {
                const int64_t idx = obase + (((int64_t)dx * Y + dy) * Z + dz) * O;
                float val = acc[a][v] + sh;
                if (MODE == 2) val += rv[v];
                if (MODE >= 1) val = fmaxf(val, 0.0f);
                if (MODE == 3) val += rv[v];
                y[idx] = val;
            }
        }
    }
}

// Fused Winograd F(2x2x2,3x3x3) for the HALF-resolution layers (C = 32 | 64 -> O = 64 on 40x40x10): the three-launch form
// (input transform, 64 batched GEMMs, output transform) moves the 131 MB transformed tensor four times (115 us per
// layer, HBM/MALL-bound); here nothing transformed leaves the CU.  Same scheme as wino_fused3_kernel - regions staged in
// LDS, operands built on the fly, exact three-piece bf16 splits, x fold on the matrix pipe, y,z fold on the VALU - on
// v_mfma_f32_16x16x32_bf16: a block is 4x4x1 tiles (8x8x2 outputs) x 64 output channels, chunks of 16 input channels
// (lane = tile x 4-channel group), and the NW waves of a workgroup share the staged region, each owning 64/NW outputs,
// so that the per-lane operand work (transforms + splits) is amortised over 16*NBW outputs.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int W16_RX = 10, W16_RY = 10, W16_RZ = 4, W16_VS = 16;
constexpr int W16_ROW = W16_RX * W16_VS + 4;
constexpr int W16_LDS = W16_RY * W16_RZ * W16_ROW;               // 6 560 floats = 26 240 B

__device__ __forceinline__ f32x4 mfma16_bf16(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int C, int MODE, int NBW, int KS>
__global__ __launch_bounds__(64 * (4 / NBW) * KS) __attribute__((amdgpu_waves_per_eu(NBW <= 2 ? 2 : 1))) void wino_fused16_kernel(const float *__restrict__ x,
                                                                          const unsigned *__restrict__ U3,
                                                                          float *__restrict__ y, const float *__restrict__ shift,
                                                                          const float *__restrict__ res, int B, int X, int Y,
                                                                          int Z, int NBX, int NBY, int NBZ)
{
    // workgroup = KS channel groups x (4 / NBW) output groups of one wave each: channel group kg owns the 16-channel
    // chunks kg, kg + KS, ... (its own staged region), output group ow owns outputs 16*NBW*ow ...; the KS partial sums
    // meet in LDS at the end.  More waves per block without repeating the operand work: the grid of a half-resolution
    // layer is only 500 blocks.
    constexpr int O = 64, NCH = C / 16, NOW = 4 / NBW, NTG = 64 * NOW;
    static_assert(NCH % KS == 0, "chunks must divide evenly over the channel groups");
    __shared__ __attribute__((aligned(16))) float lds[KS * W16_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = wave / NOW, ow = wave % NOW, tg = tid - kg * NTG;          // thread index inside the channel group
    float *region = lds + kg * W16_LDS;
    const int tl = lane & 15, q = lane >> 4;
    int bid = blockIdx.x;
    const int bz = bid % NBZ; bid /= NBZ;
    const int by = bid % NBY; bid /= NBY;
    const int bx = bid % NBX;
    const int b = bid / NBX;
    const int ttx = tl & 3, tty = tl >> 2;
    const int ox0 = bx * 8, oy0 = by * 8, oz0 = bz * 2;
    const float *rb = region + (2 * tty) * W16_ROW + (2 * ttx) * W16_VS + 4 * q;

    f32x4 acc[8][NBW];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int n = 0; n < NBW; ++n) acc[a][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#ifdef SP3D_W16_SETPRIO
    __builtin_amdgcn_s_setprio(SP3D_W16_SETPRIO);       // experiment (round 5): wave priority of the matrix-instruction waves
#endif

    // region of one 16-channel chunk: 400 voxels x 4 float4; every thread of the channel group owns PER of them.  The
    // loads of the group's next chunk are issued before the 16 steps of the current one and parked in registers (a
    // dependent load->store loop costs as much as the steps themselves: ~2 us of memory latency per iteration)
    constexpr int NV4 = W16_RX * W16_RY * W16_RZ * 4, PER = (NV4 + NTG - 1) / NTG;
    int goff[PER], loff[PER];                      // global offset (floats, chunk 0; -1: padding) and LDS offset (-1: none)
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int idx = tg + u * NTG;
        const int v = idx >> 2, part = idx & 3;
        const int vx = v % W16_RX, vy = (v / W16_RX) % W16_RY, vz = v / (W16_RX * W16_RY);
        const int gx = ox0 - 1 + vx, gy = oy0 - 1 + vy, gz = oz0 - 1 + vz;
        const bool in = idx < NV4 && gx >= 0 && gx < X && gy >= 0 && gy < Y && gz >= 0 && gz < Z;
        goff[u] = in ? (int)(((((int64_t)gx) * Y + gy) * Z + gz) * C + part * 4) : -1;
        loff[u] = idx < NV4 ? (vz * W16_RY + vy) * W16_ROW + vx * W16_VS + part * 4 : -1;
    }
    const float *xb = x + (int64_t)b * X * Y * Z * C;
    float4 pre[PER];
    auto fetch = [&](int cc) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            pre[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (goff[u] >= 0) pre[u] = *reinterpret_cast<const float4 *>(xb + goff[u] + cc * 16);
        }
    };
    fetch(kg);
#pragma unroll 1
    for (int cc = kg; cc < NCH; cc += KS) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (loff[u] >= 0) *reinterpret_cast<float4 *>(region + loff[u]) = pre[u];
        __syncthreads();
        if (cc + KS < NCH) fetch(cc + KS);
        // record of (point p, chunk cc, channel group q, output o): 6 dwords [mid(4ch) hi(4ch) lo(4ch)]
        const unsigned *ub = U3 + ((((int64_t)cc * 4 + q) * O) + ow * NBW * 16 + tl) * 6;
        auto load_b = [&](int i, int jk, int n) {
            const unsigned *r = ub + (int64_t)(i * 16 + jk) * NCH * 4 * O * 6 + n * 16 * 6;
            WfB w;
            w.mh = *reinterpret_cast<const u32x4_a8 *>(r);
            w.l = *reinterpret_cast<const u32x2_a8 *>(r + 4);
            return w;
        };
#pragma unroll 1
        for (int jk = 0; jk < 16; ++jk) {
            const int j = jk >> 2, k = jk & 3;
            // this step's weights: in flight while the operands are built (~1000 cycles)
            WfB bw[NBW][4];
#pragma unroll
            for (int n = 0; n < NBW; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#if SP3D_W16_ABLATE & 2
                    bw[n][i].mh = u32x4{0x3f803f80u + (unsigned)(jk + n), 0x3f803f80u, 0x3f803f80u + (unsigned)i, 0x3f803f80u};
                    bw[n][i].l = u32x2{0x3f803f80u, 0x3f803f80u};
#else
                    bw[n][i] = load_b(i, jk, n);
#endif
                }
            const int ya = (j == 0) ? 0 : ((j == 2) ? 2 : 1), yb = (j == 3) ? 3 : ((j == 2) ? 1 : 2);
            const int za = (k == 0) ? 0 : ((k == 2) ? 2 : 1), zb = (k == 3) ? 3 : ((k == 2) ? 1 : 2);
            const float sy = (j == 1) ? 1.0f : -1.0f, sz = (k == 1) ? 1.0f : -1.0f;
            const float *r00 = rb + (za * W16_RY + ya) * W16_ROW, *r10 = rb + (za * W16_RY + yb) * W16_ROW;
            const float *r01 = rb + (zb * W16_RY + ya) * W16_ROW, *r11 = rb + (zb * W16_RY + yb) * W16_ROW;
            float g[4][4];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
#if SP3D_W16_ABLATE & 8
                const float f0 = __int_as_float(0x3f800000 + jk + xi), f1 = __int_as_float(0x3f900000 + lane);
                const float4 v00 = make_float4(f0, f1, f0, f1), v10 = make_float4(f1, f0, f1, f0), v01 = v00, v11 = v10;
                (void)r00; (void)r10; (void)r01; (void)r11;
#else
                const float4 v00 = *reinterpret_cast<const float4 *>(r00 + xi * W16_VS), v10 = *reinterpret_cast<const float4 *>(r10 + xi * W16_VS);
                const float4 v01 = *reinterpret_cast<const float4 *>(r01 + xi * W16_VS), v11 = *reinterpret_cast<const float4 *>(r11 + xi * W16_VS);
#endif
                g[xi][0] = fmaf(sz, fmaf(sy, v11.x, v01.x), fmaf(sy, v10.x, v00.x));
                g[xi][1] = fmaf(sz, fmaf(sy, v11.y, v01.y), fmaf(sy, v10.y, v00.y));
                g[xi][2] = fmaf(sz, fmaf(sy, v11.z, v01.z), fmaf(sy, v10.z, v00.z));
                g[xi][3] = fmaf(sz, fmaf(sy, v11.w, v01.w), fmaf(sy, v10.w, v00.w));
            }
            float cyz[4];
#pragma unroll
            for (int bc = 0; bc < 4; ++bc) {
                const int bb = bc >> 1, c2 = bc & 1;
                const float cy = bb == 0 ? (j < 3 ? 1.0f : 0.0f) : (j == 0 ? 0.0f : (j == 1 ? 1.0f : -1.0f));
                const float cz = c2 == 0 ? (k < 3 ? 1.0f : 0.0f) : (k == 0 ? 0.0f : (k == 1 ? 1.0f : -1.0f));
                cyz[bc] = cy * cz;
            }
            f32x4 M0[NBW], M1[NBW];
#pragma unroll
            for (int n = 0; n < NBW; ++n) { M0[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; M1[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float av[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    av[kk] = (i == 1) ? g[1][kk] + g[2][kk] : ((i == 3) ? g[3][kk] - g[1][kk] : g[wf_ta(i)][kk] - g[wf_tb(i)][kk]);
                const unsigned hi01 = pack_bf16(av[0], av[1]), hi23 = pack_bf16(av[2], av[3]);
#if SP3D_W16_ABLATE & 4
                const unsigned mid01 = hi01, mid23 = hi23, lo01 = hi01, lo23 = hi23;
#else
                const float r0 = av[0] - bf16_lo(hi01), r1 = av[1] - bf16_hi(hi01), r2 = av[2] - bf16_lo(hi23), r3 = av[3] - bf16_hi(hi23);
                const unsigned mid01 = pack_bf16(r0, r1), mid23 = pack_bf16(r2, r3);
                const unsigned lo01 = pack_bf16(r0 - bf16_lo(mid01), r1 - bf16_hi(mid01));
                const unsigned lo23 = pack_bf16(r2 - bf16_lo(mid23), r3 - bf16_hi(mid23));
#endif
                const u32x4 Qhh = {hi01, hi23, hi01, hi23}, Qmm = {mid01, mid23, mid01, mid23}, Qlh = {lo01, lo23, hi01, hi23};
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    const WfB &w = bw[n][i];
                    const u32x4 Bmh = w.mh;
                    const u32x4 Bhl = {w.mh.z, w.mh.w, w.l.x, w.l.y};
#if SP3D_W16_ABLATE & 1
                    {   // no matrix instructions: keep every operand alive with one integer op each
                        const unsigned z = (Qhh.x ^ Bmh.x) + (Qmm.y ^ Bmh.z) + (Qlh.x ^ Bhl.w) + (Qlh.z ^ Bhl.y);
                        M0[n].x += __uint_as_float(z & 0x3fffffffu);
                        M1[n].y += __uint_as_float((z >> 1) & 0x3fffffffu);
                        continue;
                    }
#endif
                    if (i < 3) {
                        M0[n] = mfma16_bf16(Qhh, Bmh, M0[n]);
                        M0[n] = mfma16_bf16(Qmm, Bmh, M0[n]);
                        M0[n] = mfma16_bf16(Qlh, Bhl, M0[n]);
                    }
                    if (i == 1 || i == 3) {
                        M1[n] = mfma16_bf16(Qhh, Bmh, M1[n]);
                        M1[n] = mfma16_bf16(Qmm, Bmh, M1[n]);
                        M1[n] = mfma16_bf16(Qlh, Bhl, M1[n]);
                    }
                    if (i == 2) {
                        const u32x4 nmh = Bmh ^ 0x80008000u, nhl = Bhl ^ 0x80008000u;
                        M1[n] = mfma16_bf16(Qhh, nmh, M1[n]);
                        M1[n] = mfma16_bf16(Qmm, nmh, M1[n]);
                        M1[n] = mfma16_bf16(Qlh, nhl, M1[n]);
                    }
                }
            }
            // the coefficients are 0 or +-1 and wave-uniform; multiplying by the zeros costs less than branching or
            // selecting around them on accumulators this small
#pragma unroll
            for (int n = 0; n < NBW; ++n)
#pragma unroll
                for (int bc = 0; bc < 4; ++bc) {
#if defined(SP3D_NO_PK) || defined(SP3D_W16_SCALAR_ACC)
                    for (int e = 0; e < 4; ++e) {     // one v_fma_f32 per component instead of v_pk_fma_f32 pairs
                        acc[bc][n][e] = fmaf(M0[n][e], cyz[bc], acc[bc][n][e]);
                        acc[4 + bc][n][e] = fmaf(M1[n][e], cyz[bc], acc[4 + bc][n][e]);
                    }
#else
                    const f32x4 cv = {cyz[bc], cyz[bc], cyz[bc], cyz[bc]};
                    acc[bc][n] = __builtin_elementwise_fma(M0[n], cv, acc[bc][n]);
                    acc[4 + bc][n] = __builtin_elementwise_fma(M1[n], cv, acc[4 + bc][n]);
#endif
                }
        }
    }

    if (KS > 1) {                                   // partial sums of channel groups 1.. -> group 0, through LDS
        __syncthreads();                            // every region has been read for the last time
        f32x4 *red = reinterpret_cast<f32x4 *>(lds);
        if (kg > 0) {
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int n = 0; n < NBW; ++n) red[(((kg - 1) * NOW + ow) * 8 * NBW + a * NBW + n) * 64 + lane] = acc[a][n];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g2 = 1; g2 < KS; ++g2)
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int n = 0; n < NBW; ++n) acc[a][n] += red[(((g2 - 1) * NOW + ow) * 8 * NBW + a * NBW + n) * 64 + lane];
    }

    // D of the 16x16 MFMA: lane (col = lane & 15, group = lane >> 4) holds tiles 4*group + v, output 16*nb + col
    if (ox0 + 8 <= X && oy0 + 8 <= Y && oz0 + 2 <= Z) {
        // interior block (every block of a 40x40x10 or 32^3 grid): one per-lane base address, compile-time offsets, no bounds
        // branch per element (64 of them cost ~750 instructions per lane, and kept the residual loads apart)
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
            const int o = (ow * NBW + n) * 16 + tl;
            const float sh = shift[o];
            const int64_t obase = ((((int64_t)b * X + ox0) * Y + oy0) * Z + oz0) * O + o;
            int off[4];                                                    // tile 4 q + v: (rx, ry) = (v, q)
#pragma unroll
            for (int v = 0; v < 4; ++v) off[v] = ((2 * v * Y + 2 * q) * Z) * O;
            float rv[8][4];
            if (MODE >= 2) {
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        rv[a][v] = res[obase + off[v] + (((a >> 2) * Y + ((a >> 1) & 1)) * Z + (a & 1)) * O];
            }
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float val = acc[a][n][v] + sh;
                    if (MODE == 2) val += rv[a][v];
                    if (MODE >= 1) val = fmaxf(val, 0.0f);
                    if (MODE == 3) val += rv[a][v];
                    y[obase + off[v] + (((a >> 2) * Y + ((a >> 1) & 1)) * Z + (a & 1)) * O] = val;
                }
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
        const int o = (ow * NBW + n) * 16 + tl;
        const float sh = shift[o];
        // residual values first, all 32 loads of this output group in flight at once: with the load inside the bounds
        // branch below the compiler emitted branch -> load -> s_waitcnt vmcnt(0) -> store per element, 64 dependent round
        // trips per lane (MODE 2 cost 8.7 us more than MODE 1 at (4,64,40,40,10)).  The address of an element outside the
        // volume is clamped into it (never stored), so the loads need no predicate.
        float rv[8][4];
        if (MODE >= 2) {
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int tile = 4 * q + v;
                    const int rx = tile & 3, ry = tile >> 2;
                    const int xo = min(ox0 + 2 * rx + (a >> 2), X - 1), yo = min(oy0 + 2 * ry + ((a >> 1) & 1), Y - 1);
                    const int zo = min(oz0 + (a & 1), Z - 1);
                    rv[a][v] = res[((((int64_t)b * X + xo) * Y + yo) * Z + zo) * O + o];
                }
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int tile = 4 * q + v;
                const int rx = tile & 3, ry = tile >> 2;
                const int xo = ox0 + 2 * rx + (a >> 2), yo = oy0 + 2 * ry + ((a >> 1) & 1), zo = oz0 + (a & 1);
                if (xo < X && yo < Y && zo < Z) {
                    const int64_t idx = ((((int64_t)b * X + xo) * Y + yo) * Z + zo) * O + o;
                    float val = acc[a][n][v] + sh;
                    if (MODE == 2) val += rv[a][v];
                    if (MODE >= 1) val = fmaxf(val, 0.0f);
                    if (MODE == 3) val += rv[a][v];
                    y[idx] = val;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// Direct 3x3x3 convolution on the bf16 matrix pipe at fp32 accuracy (implicit GEMM, no Winograd).
// With exact three-piece splits a multiply costs 3 bf16 MFMA slots, and the bf16 pipe is 16x the fp32 one: the 2.25x
// multiplication saving of (x-folded) Winograd no longer pays for its operand transforms - the fused Winograd kernels
// above are VALU-bound (183 VALU instructions per 18 MFMAs, every step re-transforms and re-splits its operands).  Here
// an input value is split ONCE when its region is staged (pieces stored in LDS: 24 B per voxel and 4-channel group,
// [lo hi mid]), the A operand of tap (dx,dy,dz) is the staged record of the shifted voxel - read with a compile-time LDS
// offset, no VALU at all - and the accumulators ARE the outputs.  Per (tap, 32-voxel block): 3 LDS reads + 3 MFMAs
//     {lo,hi} x {bh,bl}  +  {hi,mid} x {bh,bh}  +  {hi,mid} x {bm,bm}      (A quads are consecutive registers)
// Block = 8x8x2 outputs (4 MFMA row blocks) x NB*32 output channels per wave, 8-channel chunks, region 10x10x4 voxels
// = 19.2 KB of LDS.  Weights: per (tap, chunk, lane half, output) one 24-byte record [mid hi lo] (conv_weights_split).
// ------------------------------------------------------------------------------------------
// Workgroup = 4 consumer waves + 4 producer waves (one of each per SIMD), persistent over output blocks of 16x8x4 voxels.
//   consumers (wave = z layer, 4 row blocks of 16x2 voxels x 32 outputs): per tap 8 ds_read_b128 + 3 global_load_dwordx4
//     + 12 matrix instructions, nothing else; operands one tap ahead, weights three taps ahead (across chunk boundaries);
//   producers: fetch the 18x10x6 region of the NEXT (block, 8-channel chunk) item, split it, write it to the other LDS
//     buffer.  They are separate waves because vmcnt is in-order: a consumer that had the region loads of the next chunk
//     in flight waited for them at its next weight wait (58 us of 207), and between workgroups nobody covered the first
//     chunk's latency.  One barrier per item.
// LDS per buffer: one plane per 4-channel group (lane half); a voxel is the two A operands ready to use, 8 dwords
// [lo hi | hi mid]; rows of 18 voxels (144 dwords), z planes of 10 rows + 4 dwords: with lane = (y = t & 7, z = t >> 3) the
// 16 lanes of a ds_read_b128 group sit on 16 distinct 4-bank slots (searched over row / plane pitches).
// A consumer wave owns 4 consecutive x of the block and ALL its (y,z): the operand of region voxel x' serves the taps
// dx = x' - x of every x it owns, so a (dy,dz) step reads 6 operands for 36 matrix instructions instead of 12 - the LDS
// (85 of its 128 B/clk with one operand read per tap and row block) was what the matrix pipe and the loader waves waited on.
constexpr int CD_BX = 16, CD_BY = 8, CD_BZ = 4;
constexpr int CD_RX = CD_BX + 2, CD_RY = CD_BY + 2, CD_RZ = CD_BZ + 2, CD_VOX = 8, CD_ROW = CD_RX * CD_VOX, CD_ZP = CD_RY * CD_ROW + 4;
constexpr int CD_PLANE = CD_RZ * CD_ZP;                                // 8 664 dwords
constexpr int CD_BUF = 2 * CD_PLANE;                                   // 17 328 dwords = 69 312 B per buffer
constexpr int CD_NV4 = CD_RX * CD_RY * CD_RZ * 2;                      // float4 per item: 2 160
constexpr int CD_PROD = 4;                                             // producer waves: one per SIMD
constexpr int CD_PER = (CD_NV4 + 64 * CD_PROD - 1) / (64 * CD_PROD);   // 9 float4 per producer lane
// buffer stride in dwords (17 408 = 69 632 B: the LDS layout the bank-conflict search was done for; CD_BUF rounded up to 68 x 256)
constexpr int CD_BUFS = ((CD_BUF / 4 + 63) / 64) * 256;
#ifndef SP3D_WG_ABLATE
#define SP3D_WG_ABLATE 0
#endif
struct CdRec { u32x4 hl, hh, mm; };                                    // weight record: B operands {bh,bl} {bh,bh} {bm,bm}


#ifdef SP3D_CD_TIMELINE
__device__ unsigned long long *g_cd_tl = nullptr;      // [wave 6][item 64][4] s_memtime stamps of workgroup 0
#define CD_STAMP(slot) do { __builtin_amdgcn_sched_barrier(0); if (g_cd_tl && blockIdx.x == 0 && lane == 0 && item < 64) { g_cd_tl[(wave * 64 + item) * 4 + (slot)] = __builtin_readcyclecounter(); \
    /* the constant 100 MHz counter next to the first and the latest stamp of wave 0: cycles per microsecond = the clock the kernel ran at */ \
    if (wave == 0 && (slot) == 0 && item == 0) { g_cd_tl[(7 * 64 + 62) * 4 + 0] = wall_clock64(); g_cd_tl[(7 * 64 + 62) * 4 + 1] = __builtin_readcyclecounter(); } \
    if (wave == 0 && (slot) == 3) { g_cd_tl[(7 * 64 + 63) * 4 + 0] = wall_clock64(); g_cd_tl[(7 * 64 + 63) * 4 + 1] = __builtin_readcyclecounter(); } } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CD_STAMP(slot) do { (void)item; } while (0)
#endif

// a = hi + mid + lo, each a bf16 (exact: 24 mantissa bits), as the two A operands {lo,hi} and {hi,mid} of 4 channels
__device__ __forceinline__ void split3(const float4 a, u32x4 &q0, u32x4 &q1)
{
    const unsigned hi01 = pack_bf16(a.x, a.y), hi23 = pack_bf16(a.z, a.w);
    // a non-finite input has a non-finite hi piece and (inf - inf | NaN - NaN) = NaN as its residual: v_med3_f32(r, 0, r)
    // is r for every number and 0 for NaN (one instruction per value), so mid = lo = 0 and the value travels in the hi
    // piece alone.  The outputs that come out non-finite are then EXACTLY those of an fp32 convolution; their kind is
    // the convolution's or NaN (inf * w is formed from the weight's three pieces, whose signs differ):
    // tests/test_gpu_parity.py::test_direct_conv3_split_kernel_nonfinite_inputs.
    const float r0 = __builtin_amdgcn_fmed3f(a.x - bf16_lo(hi01), 0.0f, a.x - bf16_lo(hi01));
    const float r1 = __builtin_amdgcn_fmed3f(a.y - bf16_hi(hi01), 0.0f, a.y - bf16_hi(hi01));
    const float r2 = __builtin_amdgcn_fmed3f(a.z - bf16_lo(hi23), 0.0f, a.z - bf16_lo(hi23));
    const float r3 = __builtin_amdgcn_fmed3f(a.w - bf16_hi(hi23), 0.0f, a.w - bf16_hi(hi23));
    const unsigned mid01 = pack_bf16(r0, r1), mid23 = pack_bf16(r2, r3);
    const unsigned lo01 = pack_bf16(r0 - bf16_lo(mid01), r1 - bf16_hi(mid01));
    const unsigned lo23 = pack_bf16(r2 - bf16_lo(mid23), r3 - bf16_hi(mid23));
    q0 = u32x4{lo01, lo23, hi01, hi23};
    q1 = u32x4{hi01, hi23, mid01, mid23};
}

template <int C, int MODE>
__global__ __launch_bounds__(64 * (4 + CD_PROD)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv3_split_kernel(const float *__restrict__ x, const unsigned *__restrict__ W3, float *__restrict__ y,
                        const float *__restrict__ shift, const float *__restrict__ res, int B, int X, int Y, int Z, int NBX,
                        int NBY, int NBZ, int nblocks)
{
    constexpr int O = 32, NCH = C / 8;
    // operand roles: a split result wants (voxel, 4 consecutive channels) per lane = weights as the A operand (rows), an
    // fp32-only result wants (channel, 16 voxels) per lane = full 128-byte rows per store instruction (the transposed form's
    // 32-byte pieces cost 8-10 k cycles per block against 3-6 k)
    extern __shared__ __attribute__((aligned(16))) unsigned cd_lds[];      // 2 x CD_BUFS dwords + 4 x 1024 floats of scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, t = lane & 31, h = lane >> 5;
    const int my_blocks = ((int)blockIdx.x < nblocks) ? (nblocks - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int n_items = my_blocks * NCH;
    auto decode = [&](int item, int &b, int &ox0, int &oy0, int &oz0) {
        int bid = (int)blockIdx.x + (item / NCH) * (int)gridDim.x;
        const int bz = bid % NBZ; bid /= NBZ;
        const int by = bid % NBY; bid /= NBY;
        const int bx = bid % NBX;
        b = bid / NBX;
        ox0 = bx * CD_BX; oy0 = by * CD_BY; oz0 = bz * CD_BZ;
    };

    if (wave >= 4) {
        // ---------------- producers: item k -> buffer k & 1 ----------------
        const int pt = tid - 256;                                          // 0 .. 64*CD_PROD-1
        // fp32 input.  VALU issue on a SIMD goes to the older wave first: without priority the (younger) producers got the
        // slots the consumer's matrix stream left over - 28 cycles per instruction, 14 k cycles per item against the
        // consumers' 12 k
        __builtin_amdgcn_s_setprio(2);
        // the producers get one issue slot per consumer matrix instruction (324 per item): everything that does not depend
        // on the item is computed once - LDS offset, offset inside the sample, region coordinates
        float4 d[CD_PER];
        int lo_[CD_PER], rel[CD_PER], vxyz[CD_PER];
#pragma unroll
        for (int u = 0; u < CD_PER; ++u) {
            const int idx = pt + 64 * CD_PROD * u;
            const int v = idx >> 1, half = idx & 1;
            const int vx = v % CD_RX, vy = (v / CD_RX) % CD_RY, vz = v / (CD_RX * CD_RY);
            lo_[u] = idx < CD_NV4 ? half * CD_PLANE + vz * CD_ZP + vy * CD_ROW + vx * CD_VOX : -1;
            rel[u] = ((vx * Y + vy) * Z + vz) * C + half * 4;
            vxyz[u] = idx < CD_NV4 ? (vx | (vy << 8) | (vz << 16)) : 0x00ffffff;      // 255: never in range
        }
        auto issue = [&](int k) {                                          // loads of item k: in flight until iteration k
            int b, ox0, oy0, oz0;
            decode(k, b, ox0, oy0, oz0);
            // element offset of region voxel (0,0,0), chunk k % NCH; may be negative at the volume border (never read there)
            const int64_t vox0 = (((int64_t)b * X + (ox0 - 1)) * Y + (oy0 - 1)) * Z + (oz0 - 1);
            const float *xb = x + vox0 * C + (k % NCH) * 8;
            // voxel (vx,vy,vz) is inside the volume iff vx in [xlo, xhi) ...: wave-uniform bounds
            const int xlo = 1 - ox0, xhi = X + 1 - ox0, ylo = 1 - oy0, yhi = Y + 1 - oy0, zlo = 1 - oz0, zhi = Z + 1 - oz0;
#pragma unroll
            for (int u = 0; u < CD_PER; ++u) {
                const int vx = vxyz[u] & 255, vy = (vxyz[u] >> 8) & 255, vz = vxyz[u] >> 16;
                const bool in = vx >= xlo && vx < xhi && vy >= ylo && vy < yhi && vz >= zlo && vz < zhi;
                d[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (in) d[u] = *reinterpret_cast<const float4 *>(xb + rel[u]);
            }
        };
        if (n_items > 0) issue(0);
        for (int k = 0; k <= n_items; ++k) {
            if (k < n_items) {
                { const int item = k; CD_STAMP(0); }
                unsigned *buf = cd_lds + (k & 1) * CD_BUFS;
#ifdef SP3D_CD_TIMELINE
                __builtin_amdgcn_s_waitcnt(0);            // vmcnt(0) lgkmcnt(0): separates the load wait from the split
                { const int item = k; CD_STAMP(3); }      // (slot 3 is overwritten after the barrier for consumers only)
#endif
#pragma unroll
                for (int u = 0; u < CD_PER; ++u) {
                    u32x4 q0, q1;
                    split3(d[u], q0, q1);
                    if (lo_[u] >= 0) {
                        unsigned *p = buf + lo_[u];
                        *reinterpret_cast<u32x4 *>(p) = q0;
                        *reinterpret_cast<u32x4 *>(p + 4) = q1;
                    }
                }
                { const int item = k; CD_STAMP(1); }
                if (k + 1 < n_items) issue(k + 1);                         // one item ahead: its latency hides behind the barrier
            }
            if (k < n_items) { const int item = k; CD_STAMP(2); }
            __syncthreads();
        }
        return;
    }

    // ---------------- consumers: item k - 1 from buffer (k - 1) & 1 ----------------
    // lane (t, h): output voxels (x = 4 wave + i, y = t & 7, z = t >> 3), i = 0..3 (one accumulator each); operand j = 0..5
    // of step (dy,dz) is region voxel (4 wave + j, y + dy, z + dz), channel group h
    const int a_off = h * CD_PLANE + (t >> 3) * CD_ZP + (t & 7) * CD_ROW + 4 * wave * CD_VOX;
    // weight record of (tap, chunk, lane half h, output t): 12 dwords
    const unsigned *wl = W3 + ((int64_t)h * O + t) * 12;
    struct W3Rec { CdRec d[3]; };                      // the three dx taps of one (dy,dz) step
    auto load_w = [&](int q) {                         // q = flattened (item, step) index; weights depend on (chunk, step)
        const int cc = (q / 9) % NCH, st = q % 9;
        const unsigned *r = wl + ((int64_t)(3 * st) * NCH + cc) * 2 * O * 12;
        W3Rec w;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
#if SP3D_W16_ABLATE & 2
            // measurement build: no weight loads (what would LDS-broadcast weights be worth at most?)
            (void)r; w.d[dx].hl = u32x4{0x3f803f80u + (unsigned)q, 0x3f803f80u, 0x3f803f80u + (unsigned)dx, 0x3f803f80u};
            w.d[dx].hh = w.d[dx].hl ^ 0x00010001u; w.d[dx].mm = w.d[dx].hl ^ 0x00020002u;
#else
            w.d[dx].hl = *reinterpret_cast<const u32x4 *>(r + (int64_t)dx * NCH * 2 * O * 12);
            w.d[dx].hh = *reinterpret_cast<const u32x4 *>(r + (int64_t)dx * NCH * 2 * O * 12 + 4);
            w.d[dx].mm = *reinterpret_cast<const u32x4 *>(r + (int64_t)dx * NCH * 2 * O * 12 + 8);
#endif
        }
        return w;
    };
    struct Opnd { u32x4 lh[6], hm[6]; };
    f32x16 acc[4];
    W3Rec w0, w1;                                      // weights of flattened step q, q+1
    if (n_items > 0) w0 = load_w(0);
    __syncthreads();                                   // item 0 staged
    for (int k = 1; k <= n_items; ++k) {
        const int item = k - 1, cc = item % NCH;
        const unsigned *ab = cd_lds + (item & 1) * CD_BUFS + a_off;
        if (cc == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[m][v] = 0.0f;
        }
        auto load_a = [&](int st, Opnd &a) {
            const int dz = st / 3, dy = st % 3;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const unsigned *p = ab + dz * CD_ZP + dy * CD_ROW + j * CD_VOX;
                a.lh[j] = *reinterpret_cast<const u32x4 *>(p);
                a.hm[j] = *reinterpret_cast<const u32x4 *>(p + 4);
            }
        };
        Opnd a0, a1;
        load_a(0, a0);
        const int q0 = item * 9;
        CD_STAMP(0);
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            __builtin_amdgcn_sched_barrier(0);
            w1 = load_w(min(q0 + st + 1, n_items * 9 - 1));             // unconditional: a branch here costs the register renaming
            if (st + 1 < 9) load_a(st + 1, a1);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mfma_bf16(a0.lh[i + dx], w0.d[dx].hl, acc[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mfma_bf16(a0.hm[i + dx], w0.d[dx].hh, acc[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mfma_bf16(a0.hm[i + dx], w0.d[dx].mm, acc[i]);
            }
            // issue order: one load between matrix instructions (a wave blocked on LDS issue cannot issue its matrix
            // instructions either: tools/conv3_timeline.py)
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = a1;
            w0 = w1;
        }
        CD_STAMP(1);
        if (cc == NCH - 1) {
            // D of the 32x32 MFMA: lane (col = t, h) holds rows m = 8 (v >> 2) + (v & 3) + 4 h, v = 0..15, of every accumulator;
            // row m of accumulator i is voxel (x = 4 wave + i, y = m & 7, z = m >> 3).  Each accumulator goes through the wave's
            // 4 KB of LDS scratch [voxel][channel] and comes back as (voxel, 4 channels) per lane: shift, residual, ReLU,
            // the fp32 result as float4 (128 B per voxel over 8 lanes) and / or the split operands for the next layer
            int b, ox0, oy0, oz0;
            decode(item, b, ox0, oy0, oz0);
            if (ox0 + CD_BX <= X && oy0 + CD_BY <= Y && oz0 + CD_BZ <= Z) {
                // fp32 result only, interior block: straight from the accumulators, lane (t, h) owns channel t of voxels
                // (x = 4 wave + i, y = 4 h + (v & 3), z = v >> 2); 128-byte rows per store, no LDS round trip (3 k cycles
                // against 6-9 k for the transposed form below)
                const float sh = shift[t];
                const int64_t obase = ((((int64_t)b * X + ox0 + 4 * wave) * Y + oy0 + 4 * h) * Z + oz0) * O + t;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int64_t idx = obase + (((int64_t)i * Y + (v & 3)) * Z + (v >> 2)) * O;
                        float val = acc[i][v] + sh;
                        if (MODE == 2) val += res[idx];
                        if (MODE >= 1) val = fmaxf(val, 0.0f);
                        if (MODE == 3) val += res[idx];
#if SP3D_W16_ABLATE & 16
                        if (val == 123.456f)
#endif
                        y[idx] = val;
                    }
            } else {
            float *scr = reinterpret_cast<float *>(cd_lds + 2 * CD_BUFS) + wave * 1024;
            const int g = lane & 7;                                          // channel group of this lane on the way out
            const float4 sh4 = *reinterpret_cast<const float4 *>(shift + 4 * g);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
                for (int v = 0; v < 16; ++v) scr[(8 * (v >> 2) + (v & 3) + 4 * h) * 32 + t] = acc[mb][v];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = (lane >> 3) + 8 * r;
                    float4 a = *reinterpret_cast<const float4 *>(scr + m * 32 + 4 * g);
                    const int xo = ox0 + 4 * wave + mb, yo = oy0 + (m & 7), zo = oz0 + (m >> 3);
                    if (xo < X && yo < Y && zo < Z) {
                        const int64_t vox = (((int64_t)b * X + xo) * Y + yo) * Z + zo;
                        a.x += sh4.x; a.y += sh4.y; a.z += sh4.z; a.w += sh4.w;
                        float4 rr = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if (MODE >= 2) rr = *reinterpret_cast<const float4 *>(res + vox * O + 4 * g);
                        if (MODE == 2) { a.x += rr.x; a.y += rr.y; a.z += rr.z; a.w += rr.w; }
                        if (MODE >= 1) { a.x = fmaxf(a.x, 0.0f); a.y = fmaxf(a.y, 0.0f); a.z = fmaxf(a.z, 0.0f); a.w = fmaxf(a.w, 0.0f); }
                        if (MODE == 3) { a.x += rr.x; a.y += rr.y; a.z += rr.z; a.w += rr.w; }
#if SP3D_W16_ABLATE & 16
                        if (a.x == 123.456f)
#endif
                        *reinterpret_cast<float4 *>(y + vox * O + 4 * g) = a;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            }
        }
        CD_STAMP(2);
        __syncthreads();
        CD_STAMP(3);
    }
}


} // namespace sp3d

extern "C" int sp3d_wino_fused(const float *x, const float *U, float *y, const float *shift, const float *residual, int mode,
                               int B, int X, int Y, int Z, int C, int O, void *stream)
{
    using namespace sp3d;
    if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || mode < 0 || mode > 3) return SP3D_EINVAL;
    if (!x || !U || !y || !shift || (mode >= 2 && !residual)) return SP3D_ENULL;
    if (O != 32 || (C != 16 && C != 32)) return SP3D_EUNSUPPORTED;
    const int NBX = (X + 7) / 8, NBY = (Y + 7) / 8, NBZ = (Z + 3) / 4;
    const int64_t blocks = (int64_t)B * NBX * NBY * NBZ;
    if (blocks > 0x7fffffff) return SP3D_ERANGE;
    const dim3 grid((unsigned)blocks), block(64);
    hipStream_t s = (hipStream_t)stream;
#define SP3D_WF(C_, M_) hipLaunchKernelGGL((wino_fused_kernel<C_, M_>), grid, block, 0, s, x, U, y, shift, residual, B, X, Y, Z, NBX, NBY, NBZ)
#define SP3D_WFM(C_) switch (mode) { case 0: SP3D_WF(C_, 0); break; case 1: SP3D_WF(C_, 1); break; case 2: SP3D_WF(C_, 2); break; default: SP3D_WF(C_, 3); }
    if (C == 16) { SP3D_WFM(16) } else { SP3D_WFM(32) }
#undef SP3D_WFM
#undef SP3D_WF
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_debug_wino_fused_timeline(void *dev_buffer)
{
#ifdef SP3D_WF_TIMELINE
    unsigned long long *p = (unsigned long long *)dev_buffer;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sp3d::g_wf_tl), &p, sizeof(p));
#else
    (void)dev_buffer;
    return SP3D_EUNSUPPORTED;
#endif
}

extern "C" int sp3d_wino_fused_split(const float *x, const void *U3, float *y, const float *shift, const float *residual, int mode,
                                     int B, int X, int Y, int Z, int C, int O, void *stream)
{
    using namespace sp3d;
    if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || mode < 0 || mode > 3) return SP3D_EINVAL;
    if (!x || !U3 || !y || !shift || (mode >= 2 && !residual)) return SP3D_ENULL;
    if (O != 32 || (C != 16 && C != 32) || (reinterpret_cast<uintptr_t>(U3) & 7)) return SP3D_EUNSUPPORTED;
    const int NBX = (X + 7) / 8, NBY = (Y + 7) / 8, NBZ = (Z + 3) / 4;
    const int64_t blocks = (int64_t)B * NBX * NBY * NBZ;
    if (blocks > 0x7fffffff) return SP3D_ERANGE;
    const dim3 grid((unsigned)blocks), block(64);
    hipStream_t s = (hipStream_t)stream;
    const unsigned *u3 = reinterpret_cast<const unsigned *>(U3);
#define SP3D_WF(C_, M_) hipLaunchKernelGGL((wino_fused3_kernel<C_, M_>), grid, block, 0, s, x, u3, y, shift, residual, B, X, Y, Z, NBX, NBY, NBZ)
#define SP3D_WFM(C_) switch (mode) { case 0: SP3D_WF(C_, 0); break; case 1: SP3D_WF(C_, 1); break; case 2: SP3D_WF(C_, 2); break; default: SP3D_WF(C_, 3); }
    if (C == 16) { SP3D_WFM(16) } else { SP3D_WFM(32) }
#undef SP3D_WFM
#undef SP3D_WF
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_wino_fused_split64(const float *x, const void *U3, float *y, const float *shift, const float *residual,
                                       int mode, int B, int X, int Y, int Z, int C, int O, void *stream)
{
    using namespace sp3d;
    if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || mode < 0 || mode > 3) return SP3D_EINVAL;
    if (!x || !U3 || !y || !shift || (mode >= 2 && !residual)) return SP3D_ENULL;
    if (O != 64 || (C != 32 && C != 64) || (reinterpret_cast<uintptr_t>(U3) & 7)) return SP3D_EUNSUPPORTED;
    const int NBX = (X + 7) / 8, NBY = (Y + 7) / 8, NBZ = (Z + 1) / 2;
    const int64_t blocks = (int64_t)B * NBX * NBY * NBZ;
    if (blocks > 0x7fffffff) return SP3D_ERANGE;
    // waves per block: two output groups x two input-channel groups of one wave each (2 waves per SIMD resident): best of the
    // {1,2,4} x {1,2} configurations measured at (4,64,40,40,10) / (4,32,40,40,10) / (8,64,32,32,32) in round 2; the others
    // are no longer instantiated
    constexpr int nbw = 2, ks = 2;
    hipStream_t s = (hipStream_t)stream;
    const unsigned *u3 = reinterpret_cast<const unsigned *>(U3);
    const dim3 grid((unsigned)blocks), block(64 * (4 / nbw) * ks);
#define SP3D_WF(C_, M_) hipLaunchKernelGGL((wino_fused16_kernel<C_, M_, 2, 2>), grid, block, 0, s, x, u3, y, shift, residual, B, X, Y, Z, NBX, NBY, NBZ)
#define SP3D_WFM(C_) switch (mode) { case 0: SP3D_WF(C_, 0); break; case 1: SP3D_WF(C_, 1); break; case 2: SP3D_WF(C_, 2); break; default: SP3D_WF(C_, 3); }
    if (C == 32) { SP3D_WFM(32) } else { SP3D_WFM(64) }
#undef SP3D_WFM
#undef SP3D_WF
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_conv3_split(const float *x, const void *W3, float *y, const float *shift, const float *residual, int mode,
                                int B, int X, int Y, int Z, int C, int O, void *stream)
{
    using namespace sp3d;
    if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || mode < 0 || mode > 3) return SP3D_EINVAL;
    if (!x || !W3 || !y || !shift || (mode >= 2 && !residual)) return SP3D_ENULL;
    if (O != 32 || (C != 16 && C != 32) || (reinterpret_cast<uintptr_t>(W3) & 15)) return SP3D_EUNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(y) & 15) return SP3D_EUNSUPPORTED;
    if ((int64_t)X * Y * Z * C * 2 > 0x7fffffff) return SP3D_ERANGE;
    const int NBX = (X + CD_BX - 1) / CD_BX, NBY = (Y + CD_BY - 1) / CD_BY, NBZ = (Z + CD_BZ - 1) / CD_BZ;
    const int64_t blocks = (int64_t)B * NBX * NBY * NBZ;
    if (blocks > 0x7fffffff) return SP3D_ERANGE;
    static int cu_count[64] = {0};                     // per device, queried once (hipGetDeviceProperties is slow)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (cu_count[dev] == 0) {
            int n = 0;
            cu_count[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        }
        cus = cu_count[dev];
    }
    // persistent workgroups, one per CU (158 KB of LDS each), an equal number of blocks each
    const int rounds = (int)((blocks + cus - 1) / cus);
    const int nwg = (int)((blocks + rounds - 1) / rounds);
    const dim3 grid((unsigned)nwg), block(64 * (4 + CD_PROD));
    const size_t lds = (size_t)2 * CD_BUFS * sizeof(unsigned) + 4 * 1024 * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const unsigned *w3 = reinterpret_cast<const unsigned *>(W3);
    // the attribute is per device: remember it per device (a process that drives several GPUs launches on each)
    int cd_dev = 0;
    { const hipError_t ed = hipGetDevice(&cd_dev); if (ed != hipSuccess) return (int)ed; }
    if (cd_dev < 0 || cd_dev >= 64) cd_dev = 63;
#define SP3D_CD(C_, M_) { static bool attr_dev[64] = {}; bool &attr = attr_dev[cd_dev]; if (!attr || cd_dev == 63) { hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3_split_kernel<C_, M_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (ea != hipSuccess) return (int)ea; attr = true; } \
    hipLaunchKernelGGL((conv3_split_kernel<C_, M_>), grid, block, lds, s, x, w3, y, shift, residual, B, X, Y, Z, NBX, NBY, NBZ, (int)blocks); }
#define SP3D_CDI(C_) switch (mode) { case 0: SP3D_CD(C_, 0); break; case 1: SP3D_CD(C_, 1); break; case 2: SP3D_CD(C_, 2); break; default: SP3D_CD(C_, 3); }
    if (C == 16) { SP3D_CDI(16) } else { SP3D_CDI(32) }
#undef SP3D_CDI
#undef SP3D_CD
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_debug_conv3_timeline(void *dev_buffer)
{
#ifdef SP3D_CD_TIMELINE
    unsigned long long *p = (unsigned long long *)dev_buffer;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sp3d::g_cd_tl), &p, sizeof(p));
#else
    (void)dev_buffer;
    return SP3D_EUNSUPPORTED;
#endif
}
