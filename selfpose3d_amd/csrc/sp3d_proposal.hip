// sp3d_proposal.hip - the two small reductions either side of the unprojection:
//   * 3D NMS + top-k + index->mm   (core/proposal.py:28-48, cuboid_proposal_net.py:42-52)
//   * soft-argmax over a fine cube  (pose_regression_net.py:19-28)
// Reference: /root/reference/lib/core/proposal.py, lib/models/cuboid_proposal_net.py,
// lib/models/pose_regression_net.py.  The reference issues max_pool3d + eq + mul + topk +
// 6 index ops (and softmax + mul + sum) as separate library kernels with (B,N) temporaries.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/sp3d.h"

namespace sp3d {

constexpr int NMS_THREADS = 256;
constexpr int NMS_PER_THREAD = 2;
constexpr int NMS_CHUNK = NMS_THREADS * NMS_PER_THREAD;

struct Cand {
    float v;
    int i;
};

// total order used for top-k: larger value first, then LOWER flat index (torch leaves ties
// unspecified - SURVEY.md App. D-6; the oracle applies the same rule).
__device__ __forceinline__ bool better(const Cand &a, const Cand &b)
{
    return (a.v > b.v) || (a.v == b.v && a.i < b.i);
}

__device__ __forceinline__ Cand wave_best(Cand c)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Cand o;
        o.v = __shfl_xor(c.v, off);
        o.i = __shfl_xor(c.i, off);
        if (better(o, c)) c = o;
    }
    return c;
}

// block-wide arg-best; result valid in every thread.  sv/si: LDS scratch of NMS_THREADS/64 entries
__device__ __forceinline__ Cand block_best(Cand c, float *sv, int *si)
{
    c = wave_best(c);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) { sv[wave] = c.v; si[wave] = c.i; }
    __syncthreads();
    Cand r;
    r.v = sv[0]; r.i = si[0];
#pragma unroll
    for (int w = 1; w < NMS_THREADS / 64; ++w) {
        Cand o;
        o.v = sv[w]; o.i = si[w];
        if (better(o, r)) r = o;
    }
    return r;
}

// nms value of voxel n: (x == max_pool3d(x,3,1,1)) * x     (core/proposal.py:28-32)
__device__ __forceinline__ float nms_value(const float *__restrict__ c, int X, int Y, int Z, int n)
{
    const int YZ = Y * Z;
    const int x = n / YZ, r = n - x * YZ, y = r / Z, z = r - y * Z;
    float m = -INFINITY;
    for (int dx = -1; dx <= 1; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= X) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= Y) continue;
            const float *row = c + ((size_t)xx * Y + yy) * Z;
            for (int dz = -1; dz <= 1; ++dz) {
                const int zz = z + dz;
                if (zz < 0 || zz >= Z) continue;
                const float v = row[zz];
                m = (v > m || v != v) ? v : m;
            }
        }
    }
    const float self = c[n];
    const float keep = (self == m) ? 1.0f : 0.0f;
    return keep * self;
}

// stage 1: every workgroup owns NMS_CHUNK voxels of one sample and emits its local top-k
__global__ __launch_bounds__(NMS_THREADS) void nms_chunk_topk_kernel(const float *__restrict__ cubes, int X, int Y,
                                                                    int Z, int k, Cand *__restrict__ ws)
{
    __shared__ float sv[NMS_THREADS / 64];
    __shared__ int si[NMS_THREADS / 64];
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int N = X * Y * Z;
    const float *c = cubes + (size_t)b * N;
    Cand mine[NMS_PER_THREAD];
#pragma unroll
    for (int e = 0; e < NMS_PER_THREAD; ++e) {
        const int n = chunk * NMS_CHUNK + e * NMS_THREADS + threadIdx.x;
        mine[e].i = n < N ? n : 0x7fffffff;
        mine[e].v = n < N ? nms_value(c, X, Y, Z, n) : -INFINITY;
    }
    Cand *out = ws + ((size_t)b * nchunks + chunk) * k;
    for (int t = 0; t < k; ++t) {
        Cand best;
        best.v = -INFINITY; best.i = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < NMS_PER_THREAD; ++e)
            if (better(mine[e], best)) best = mine[e];
        const Cand win = block_best(best, sv, si);
#pragma unroll
        for (int e = 0; e < NMS_PER_THREAD; ++e)
            if (mine[e].i == win.i) { mine[e].v = -INFINITY; mine[e].i = 0x7fffffff; }
        if (threadIdx.x == 0) out[t] = win;
    }
}

// stage 2: one workgroup per sample merges nchunks*k candidates, unravels and converts to mm.
// Every thread keeps its strided share of the candidates in registers (<= MERGE_PER_THREAD), so the k
// selection rounds touch no memory; larger candidate sets fall back to the in-memory loop.
constexpr int MERGE_PER_THREAD = 16;

__global__ __launch_bounds__(NMS_THREADS) void nms_merge_kernel(Cand *__restrict__ ws, int ncand, int X, int Y, int Z,
                                                               int k, float Lx, float Ly, float Lz, float cx, float cy,
                                                               float cz, float *__restrict__ vals,
                                                               int64_t *__restrict__ idx, float *__restrict__ locs,
                                                               float *__restrict__ gcent, float threshold)
{
    __shared__ float sv[NMS_THREADS / 64];
    __shared__ int si[NMS_THREADS / 64];
    const int b = blockIdx.x;
    Cand *cand = ws + (size_t)b * ncand;
    const int YZ = Y * Z;
    const bool inreg = ncand <= NMS_THREADS * MERGE_PER_THREAD;
    Cand mine[MERGE_PER_THREAD];
    if (inreg) {
#pragma unroll
        for (int e = 0; e < MERGE_PER_THREAD; ++e) {
            const int i = e * NMS_THREADS + threadIdx.x;
            if (i < ncand) mine[e] = cand[i];
            else { mine[e].v = -INFINITY; mine[e].i = 0x7fffffff; }
        }
    }
    for (int t = 0; t < k; ++t) {
        Cand best;
        best.v = -INFINITY; best.i = 0x7fffffff;
        if (inreg) {
#pragma unroll
            for (int e = 0; e < MERGE_PER_THREAD; ++e)
                if (better(mine[e], best)) best = mine[e];
        } else {
            for (int e = threadIdx.x; e < ncand; e += NMS_THREADS) {
                const Cand c = cand[e];
                if (better(c, best)) best = c;
            }
        }
        const Cand win = block_best(best, sv, si);
        if (inreg) {
#pragma unroll
            for (int e = 0; e < MERGE_PER_THREAD; ++e)
                if (mine[e].i == win.i) { mine[e].v = -INFINITY; mine[e].i = 0x7fffffff; }
        } else {
            for (int e = threadIdx.x; e < ncand; e += NMS_THREADS)
                if (cand[e].i == win.i) { cand[e].v = -INFINITY; cand[e].i = 0x7fffffff; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const int n = win.i == 0x7fffffff ? 0 : win.i;
            const int ix = n / YZ, iy = (n % YZ) / Z, iz = n % Z;          // core/proposal.py:21-23
            vals[(size_t)b * k + t] = win.i == 0x7fffffff ? 0.0f : win.v;
            int64_t *ip = idx + ((size_t)b * k + t) * 3;
            ip[0] = ix; ip[1] = iy; ip[2] = iz;
            if (locs) {                                                      // cuboid_proposal_net.py:47-51
                float *lp = locs + ((size_t)b * k + t) * 3;
                lp[0] = ((float)ix / (float)(X - 1) * Lx + cx) - Lx / 2.0f;
                lp[1] = ((float)iy / (float)(Y - 1) * Ly + cy) - Ly / 2.0f;
                lp[2] = ((float)iz / (float)(Z - 1) * Lz + cz) - Lz / 2.0f;
                if (gcent) {   // ProposalLayer.forward in eval (cuboid_proposal_net.py:62-81): [x,y,z, (score>thr)-1, score]
                    float *gp = gcent + ((size_t)b * k + t) * 5;
                    const float v = win.i == 0x7fffffff ? 0.0f : win.v;
                    gp[0] = lp[0]; gp[1] = lp[1]; gp[2] = lp[2];
                    gp[3] = (v > threshold ? 1.0f : 0.0f) - 1.0f;
                    gp[4] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// soft-argmax: one workgroup per (sample, joint); pass 1 max, pass 2 exp-sum and moments
// ------------------------------------------------------------------------------------------
constexpr int SA_THREADS = 1024;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// GRID = true: voxel centres are regenerated from (centre, size, bins) exactly as the unprojection
// kernel writes them (same fp32 linspace form), so `grids` (12 B/voxel) is never materialised.
struct GridSpec {
    int X, Y, Z;
    float Lx, Ly, Lz, sx, sy, sz;   // box size and fp32 linspace steps (host-computed)
};

__device__ __forceinline__ float lin_at(float L, float step, int n, int i)
{
    const float start = -(L / 2.0f), end = L / 2.0f;
    if (n == 1) return start;
    return (i < n / 2) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(n - 1 - i), end);
}

template <bool GRID>
__global__ __launch_bounds__(SA_THREADS) void soft_argmax_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ grids,
                                                                const float *__restrict__ centers, GridSpec gs,
                                                                float *__restrict__ out, int J, int64_t N, float beta)
{
    __shared__ float red[4][SA_THREADS / 64];
    const int j = blockIdx.x, b = blockIdx.y;
    const float *xv = x + ((size_t)b * J + j) * N;
    const float *gv = GRID ? nullptr : grids + (size_t)b * N * 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float m = -INFINITY;
    for (int64_t n = threadIdx.x; n < N; n += SA_THREADS) m = fmaxf(m, beta * xv[n]);
    m = wave_max(m);
    if (lane == 0) red[0][wave] = m;
    __syncthreads();
    m = red[0][0];
#pragma unroll
    for (int w = 1; w < SA_THREADS / 64; ++w) m = fmaxf(m, red[0][w]);
    __syncthreads();
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (GRID) { cx = centers[3 * b]; cy = centers[3 * b + 1]; cz = centers[3 * b + 2]; }
    const int YZ = gs.Y * gs.Z;
    float s = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int64_t n = threadIdx.x; n < N; n += SA_THREADS) {
        const float e = __expf(beta * xv[n] - m);
        float g0, g1, g2;
        if (GRID) {
            const int nn = (int)n;
            const int ix = nn / YZ, r = nn - ix * YZ, iy = r / gs.Z, iz = r - iy * gs.Z;
            g0 = lin_at(gs.Lx, gs.sx, gs.X, ix) + cx;
            g1 = lin_at(gs.Ly, gs.sy, gs.Y, iy) + cy;
            g2 = lin_at(gs.Lz, gs.sz, gs.Z, iz) + cz;
        } else {
            g0 = gv[3 * n + 0]; g1 = gv[3 * n + 1]; g2 = gv[3 * n + 2];
        }
        s += e;
        a0 = fmaf(e, g0, a0);
        a1 = fmaf(e, g1, a1);
        a2 = fmaf(e, g2, a2);
    }
    s = wave_sum(s); a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { red[0][wave] = s; red[1][wave] = a0; red[2][wave] = a1; red[3][wave] = a2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float S = 0.f, A0 = 0.f, A1 = 0.f, A2 = 0.f;
        for (int w = 0; w < SA_THREADS / 64; ++w) { S += red[0][w]; A0 += red[1][w]; A1 += red[2][w]; A2 += red[3][w]; }
        float *o = out + ((size_t)b * J + j) * 3;
        o[0] = A0 / S; o[1] = A1 / S; o[2] = A2 / S;
    }
}

} // namespace sp3d

using namespace sp3d;

extern "C" int64_t sp3d_nms_topk_workspace_bytes(int B, int X, int Y, int Z, int k)
{
    if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || k <= 0) return 0;
    const int64_t N = (int64_t)X * Y * Z;
    const int64_t nchunks = (N + NMS_CHUNK - 1) / NMS_CHUNK;
    return (int64_t)B * nchunks * k * (int64_t)sizeof(Cand);
}

extern "C" int sp3d_nms_proposals(const float *root_cubes, int B, int X, int Y, int Z, int k, const float *grid_size,
                                  const float *grid_center, float threshold, float *vals, int64_t *idx, float *locs,
                                  float *grid_centers, void *workspace, void *stream)
{
    if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || k <= 0 || k > SP3D_MAX_TOPK) return SP3D_EINVAL;
    if (!root_cubes || !vals || !idx || !workspace) return SP3D_ENULL;
    if (locs && (!grid_size || !grid_center)) return SP3D_ENULL;
    if (grid_centers && !locs) return SP3D_ENULL;
    const int64_t N = (int64_t)X * Y * Z;
    if (N > 0x7ffffffe) return SP3D_ERANGE;
    const int nchunks = (int)((N + NMS_CHUNK - 1) / NMS_CHUNK);
    hipStream_t s = (hipStream_t)stream;
    Cand *ws = (Cand *)workspace;
    hipLaunchKernelGGL(nms_chunk_topk_kernel, dim3(nchunks, B), dim3(NMS_THREADS), 0, s, root_cubes, X, Y, Z, k, ws);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const float L[3] = {locs ? grid_size[0] : 0.f, locs ? grid_size[1] : 0.f, locs ? grid_size[2] : 0.f};
    const float C[3] = {locs ? grid_center[0] : 0.f, locs ? grid_center[1] : 0.f, locs ? grid_center[2] : 0.f};
    hipLaunchKernelGGL(nms_merge_kernel, dim3(B), dim3(NMS_THREADS), 0, s, ws, nchunks * k, X, Y, Z, k, L[0], L[1],
                       L[2], C[0], C[1], C[2], vals, idx, locs, grid_centers, threshold);
    e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_nms_topk(const float *root_cubes, int B, int X, int Y, int Z, int k, const float *grid_size,
                             const float *grid_center, float *vals, int64_t *idx, float *locs, void *workspace,
                             void *stream)
{
    return sp3d_nms_proposals(root_cubes, B, X, Y, Z, k, grid_size, grid_center, 0.0f, vals, idx, locs, nullptr, workspace,
                              stream);
}

extern "C" int sp3d_soft_argmax(const float *x, const float *grids, float *out, int Bv, int J, int64_t N, float beta,
                                void *stream)
{
    if (Bv <= 0 || J <= 0 || N <= 0) return SP3D_EINVAL;
    if (!x || !grids || !out) return SP3D_ENULL;
    GridSpec gs = {};
    hipLaunchKernelGGL(soft_argmax_kernel<false>, dim3(J, Bv), dim3(SA_THREADS), 0, (hipStream_t)stream, x, grids,
                       (const float *)nullptr, gs, out, J, N, beta);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_soft_argmax_grid(const float *x, const float *centers, const float *grid_size, int X, int Y, int Z,
                                     float *out, int Bv, int J, float beta, void *stream)
{
    if (Bv <= 0 || J <= 0 || X <= 0 || Y <= 0 || Z <= 0) return SP3D_EINVAL;
    if (!x || !centers || !grid_size || !out) return SP3D_ENULL;
    const int64_t N = (int64_t)X * Y * Z;
    if (N > 0x7ffffffe) return SP3D_ERANGE;
    GridSpec gs;
    gs.X = X; gs.Y = Y; gs.Z = Z; gs.Lx = grid_size[0]; gs.Ly = grid_size[1]; gs.Lz = grid_size[2];
    const int n[3] = {X, Y, Z};
    float st[3];
    for (int a = 0; a < 3; ++a) {
        volatile float start = -(grid_size[a] / 2.0f), end = grid_size[a] / 2.0f;
        volatile float diff = end - start;
        st[a] = n[a] > 1 ? diff / (float)(n[a] - 1) : 0.0f;
    }
    gs.sx = st[0]; gs.sy = st[1]; gs.sz = st[2];
    hipLaunchKernelGGL(soft_argmax_kernel<true>, dim3(J, Bv), dim3(SA_THREADS), 0, (hipStream_t)stream, x,
                       (const float *)nullptr, centers, gs, out, J, N, beta);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}
